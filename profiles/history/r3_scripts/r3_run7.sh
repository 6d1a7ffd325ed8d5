#!/bin/bash
# usage: r3_run7.sh <tag>   -- the whole GPU suite, the driver-flag bench line, a kernel trace of the same window
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 120 python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-8m --no-extra --cpu-seconds 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_density']['avg_us']); print([(k['name'], round(k['avg_us_hip_events'],1)) for k in d['kernels'][:8]])"
timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 100 2>&1 | tail -1; echo "generic Jacobi:"; SPH_JACOBI_GENERIC=1 SPH_TIME_WARMUP=5 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 2>&1 | tail -1; echo "OpJacobiU:"; SPH_TIME_WARMUP=5 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
SPH_TIME_WARMUP=5 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 20 > $OUT/kt.log 2>&1; echo "kt rc=$?"
python - <<PY
import csv,glob,collections,statistics
f=glob.glob("$OUT/kt/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)): d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:9]:
    ref=sorted(v)[int(0.9*(len(v)-1))]; w=[x for x in v if x>0.25*ref]
    print(f"{k:62s} n={len(v):5d} working={len(w):5d} med={statistics.median(w):7.1f} total_ms={sum(v)/1e3:8.2f}")
PY
