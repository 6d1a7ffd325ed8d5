#!/bin/bash
# rocprofv3 kernel trace of configs[1] + EmptyAngle: durations of the propagation launches (sweep form and queue form) and the gaps
# usage: scripts/kt_level2.sh <outdir under gpurun_out> [env assignments ...]
OUT=$GRAFT_REPO_ROOT/$1; shift; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
env "$@" timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_run_workload.py dam_break_1m 20 6 level_estimation_method=EmptyAngle maximum_surface_distance=0.2 particle_radius_fine=0.0005 particle_radius_base=0.002 > $OUT/kt.log 2>&1
echo rc=$?; tail -1 $OUT/kt.log
python - <<PY
import csv, glob, numpy as np
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def is_prop(r): return "OpLevelPropagate" in r["Kernel_Name"] or "k_level_frontier" in r["Kernel_Name"]
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if is_prop(r)])
print("propagate launches", len(d), "median us", np.median(d), "p10", np.percentile(d, 10), "p90", np.percentile(d, 90), "max", d.max(), "sum ms", d.sum() / 1e3)
pr = [r for r in rows if is_prop(r)]
gaps = np.array([(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(pr[:-1], pr[1:])])
print("start-to-start between consecutive propagate launches: median us", np.median(gaps + d[:-1]), "gap median", np.median(gaps), "p90", np.percentile(gaps, 90))
PY
