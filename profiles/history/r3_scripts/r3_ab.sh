#!/bin/bash
# usage: r3_ab.sh <variant .so basename>   -- product library vs a variant library (csrc/<name>): driver window, steady window, kernel medians
cd $GRAFT_REPO_ROOT
V=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab
mkdir -p $OUT
for rep in 1 2; do
for lib in libsph_hip.so $V; do
  echo "== $lib: driver window (5 warm-up, 20 steps), steady window (20 warm-up, 100 steps), configs[2] 1.8M"
  SPH_HIP_LIBRARY=$lib SPH_TIME_WARMUP=5 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 2>&1 | tail -1
  SPH_HIP_LIBRARY=$lib timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 100 2>&1 | tail -1
done
done
for lib in libsph_hip.so $V; do
  SPH_HIP_LIBRARY=$lib timeout -k 5 200 python scripts/gpu_time.py ${AB_EXTRA:-dam_break_8m} 30 2>&1 | tail -1
done
cd /tmp; export TMPDIR=/tmp
for lib in libsph_hip.so $V; do
SPH_HIP_LIBRARY=$lib SPH_TIME_WARMUP=5 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$lib -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 20 > $OUT/kt_$lib.log 2>&1; echo "kt rc=$?"
python - <<PY
import csv,glob,collections,statistics
f=glob.glob("$OUT/kt_$lib/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)): d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:7]:
    ref=sorted(v)[int(0.9*(len(v)-1))]; w=[x for x in v if x>0.25*ref]
    print(f"{k:62s} n={len(v):5d} working={len(w):5d} med={statistics.median(w):7.1f} total_ms={sum(v)/1e3:8.2f}")
PY
done
