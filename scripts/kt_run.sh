#!/bin/bash
# usage: kt_run.sh <outdir> [workload] [steps]  -- rocprofv3 kernel trace of scripts/gpu_time.py, prints per-kernel medians
OUT=$1; WL=${2:-dam_break_1m}; STEPS=${3:-30}
cd /tmp; export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$OUT
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $WL $STEPS > $GRAFT_REPO_ROOT/$OUT/kt.log 2>&1
echo "rc=$?"; tail -1 $GRAFT_REPO_ROOT/$OUT/kt.log
python - <<PY
import csv, collections, statistics, glob
f = glob.glob("$GRAFT_REPO_ROOT/$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("k_sweep<", "").replace("<MathUniform, false>", "").replace("<MathUniform, true>", "<dist>").replace("<MathUniform>", "").replace(", false>", "").replace(", true>", "[build]")
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    real = [x for x in v if x > 3.0] or v
    print(f"{n[:44]:44s} n={len(v):5d} median={statistics.median(real):7.1f} us  sum/step={sum(v)/($STEPS+20):7.1f} us")
    tot += sum(v)
print("GPU busy per step (us):", tot / ($STEPS + 20))
PY
