#!/bin/bash
# usage: r3_kt.sh [workload] [steps] [warmup]  -- kernel trace of a window: per-kernel medians of the launches that did work + gap summary
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_now
rm -rf $OUT
SPH_TIME_WARMUP=${3:-5} timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py ${1:-dam_break_1m} ${2:-20} > $OUT.log 2>&1; echo "rc=$?"; tail -1 $OUT.log
python - <<PY
import csv,glob,collections,statistics
f=glob.glob("$OUT/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))))
d=collections.defaultdict(list)
for s,e,n in rows: d[n[:70]].append((e-s)/1e3)
tot=sum(sum(v) for v in d.values())
print("kernel time total %.2f ms"%(tot/1e3))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:14]:
    ref=sorted(v)[int(0.9*(len(v)-1))]; w=[x for x in v if x>0.25*ref]
    print(f"{k:72s} n={len(v):5d} working={len(w):5d} med={statistics.median(w):7.1f} total_ms={sum(v)/1e3:8.2f} ({100*sum(v)/tot:.1f}%)")
PY
