#!/bin/bash
# rocprofv3 kernel trace of configs[4]'s scene + EmptyAngle level estimation: where the level estimation's time goes at 4M
OUT=$GRAFT_REPO_ROOT/$1; shift; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
env "$@" timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_run_workload.py ratio_stress_4m 5 6 level_estimation_method=EmptyAngle > $OUT/kt.log 2>&1
echo rc=$?; tail -1 $OUT/kt.log
python - <<PY
import csv, glob, collections, re
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    m = re.search(r"(Op\w+|k_\w+|fillBuffer)", r["Kernel_Name"]); nm = m.group(1) if m else r["Kernel_Name"][:30]
    if "k_sweep<" in r["Kernel_Name"] and ", true>" in r["Kernel_Name"]: nm += "[build]"
    acc[nm][0] += 1; acc[nm][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for nm, (k, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{nm:32s} launches/step {k/11:7.1f}  us/step {us/11:9.1f}  avg us {us/k:8.1f}")
PY
