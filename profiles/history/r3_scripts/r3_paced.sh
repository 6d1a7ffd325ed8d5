#!/bin/bash
# paced solves vs the predicted queue: driver window, steady window, settings of the lead and of the unpaced head
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" SPH_TIME_WARMUP=5 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 2>&1 | tail -1; env "$@" timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 100 2>&1 | tail -1; }
for rep in 1 2; do
run SPH_PACED=0
run SPH_PACED=1 SPH_PACE_LEAD=1 SPH_PACE_PRED=1
run SPH_PACED=1 SPH_PACE_LEAD=2 SPH_PACE_PRED=1
run SPH_PACED=1 SPH_PACE_LEAD=1 SPH_PACE_PRED=0
run SPH_PACED=1 SPH_PACE_LEAD=1 SPH_PACE_PRED=2
run SPH_PACED=1 SPH_PACE_LEAD=3 SPH_PACE_PRED=0
done
timeout -k 5 200 python scripts/gpu_iters.py dam_break_1m 30 2>&1 | tail -2
timeout -k 5 900 python -m pytest tests -m gpu -x -q -k "parity or config or golden" 2>&1 | tail -5
