#!/bin/bash
# Round-4 profile set (through gpurun): scripts/profile_round4.sh gpurun_out/r4d
#   profile_round.sh's set (bench lines, rocprofv3 --kernel-trace --stats of the driver-flag command, FETCH_SIZE / WRITE_SIZE in their
#   own --pmc passes) + the SQ counters of the sweeps (instruction counts per wave: what bench.py's `valu_issue` is priced with) +
#   the dispatch-bracket calibration under a kernel trace + the other configs (profile_configs.sh)
# then: python scripts/summarize_profile.py gpurun_out/r4d profiles/r4d ; python scripts/pmc_table.py gpurun_out/r4d/sq > profiles/r4_sq_counters.txt
R=$1
bash $GRAFT_REPO_ROOT/scripts/profile_round.sh $R
bash $GRAFT_REPO_ROOT/scripts/pmc_run2.sh $R/sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH"
OUT=$GRAFT_REPO_ROOT/$R
cd /tmp; export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/calib -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_event_calibration.py > $OUT/calib.log 2>&1; echo "calibration rc=$?"
grep "spin" $OUT/calib.log
python - <<PY
import csv, glob, statistics
f = glob.glob("$OUT/calib/**/*kernel_trace.csv", recursive=True)
if f:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f[0])) if "k_spin" in r["Kernel_Name"]]
    # five groups of 50 launches (+ warm-up launches): medians per group in launch order
    print("rocprofv3 durations of the spin launches, medians per 50:", [round(statistics.median(d[k:k + 50]), 2) for k in range(0, len(d) - 49, 50)])
PY
bash $GRAFT_REPO_ROOT/scripts/profile_configs.sh $R
