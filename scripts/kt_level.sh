#!/bin/bash
# rocprofv3 kernel trace of the level-estimation workload: duration distribution of the propagation sweeps
OUT=$GRAFT_REPO_ROOT/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_level_time.py dam_break_1m 3 > $OUT/kt.log 2>&1
echo rc=$?
python - <<PY
import csv, glob, numpy as np
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "OpLevelPropagate" in r["Kernel_Name"]])
print("propagate launches", len(d), "median us", np.median(d), "p10", np.percentile(d, 10), "p90", np.percentile(d, 90), "max", d.max())
idx = [k for k, r in enumerate(rows) if "OpLevelPropagate" in r["Kernel_Name"]]
gaps = np.array([(int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 for a, b in zip(idx[:-1], idx[1:]) if b == a + 1])
print("gap between consecutive propagate launches: median us", np.median(gaps), "p90", np.percentile(gaps, 90))
PY
