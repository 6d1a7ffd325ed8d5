#!/bin/bash
# usage: kt_slab.sh <outdir> [k] -- rocprofv3 kernel trace of a k-rank loopback group stepping dam_break_1m (scripts/gpu_slab_group.py)
OUT=$1; K=${2:-2}; STEPS=40
cd /tmp; export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$OUT
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_slab_group.py $K $STEPS > $GRAFT_REPO_ROOT/$OUT/kt.log 2>&1
echo "rc=$?"; tail -1 $GRAFT_REPO_ROOT/$OUT/kt.log
python - <<PY
import csv, collections, statistics, glob
f = glob.glob("$GRAFT_REPO_ROOT/$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("k_sweep<", "").replace("<MathUniform, false>", "").replace("<MathUniform>", "").replace(", false>", "").replace(", true>", "[build]")
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:28]:
    print(f"{n[:44]:44s} n={len(v):6d} median={statistics.median(v):7.1f} us  sum/step={sum(v)/$STEPS:8.1f} us")
for v in d.values(): tot += sum(v)
print("GPU busy per group step (us):", tot / $STEPS)
PY
