#!/bin/bash
# usage: kt_forced.sh <outdir> -- rocprofv3 kernel trace of scripts/gpu_forced_slab_time.py (one rank, slab driver, RCCL)
OUT=$1; STEPS=60
cd /tmp; export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$OUT
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_forced_slab_time.py $STEPS > $GRAFT_REPO_ROOT/$OUT/kt.log 2>&1
echo "rc=$?"; grep "forced slab" $GRAFT_REPO_ROOT/$OUT/kt.log
python - <<PY
import csv, collections, statistics, glob
f = glob.glob("$GRAFT_REPO_ROOT/$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("k_sweep<", "").replace("<MathUniform, false>", "").replace("<MathUniform>", "").replace(", false>", "").replace(", true>", "[build]")
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
S = $STEPS + 20
tot = 0
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:30]:
    print(f"{n[:48]:48s} n/step={len(v)/S:6.1f} median={statistics.median(v):7.1f} us  sum/step={sum(v)/S:8.1f} us")
for v in d.values(): tot += sum(v)
print("GPU busy per step (us):", tot / S)
PY
