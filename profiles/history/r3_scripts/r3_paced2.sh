#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 5 200 python scripts/gpu_time.py $WL $STEPS 2>&1 | tail -1; }
for WL in dam_break_64k dam_break_1m_adaptive dam_break_8m; do
STEPS=60; [ $WL = dam_break_8m ] && STEPS=20
for rep in 1 2; do
run SPH_PACED=0
run SPH_PACED=1
done
done
WL=dam_break_64k; STEPS=200
run SPH_PACED=1 SPH_PACE_LEAD=1 SPH_PACE_PRED=0
run SPH_PACED=1 SPH_PACE_LEAD=4 SPH_PACE_PRED=1
run SPH_PACED=1 SPH_PACE_LEAD=8 SPH_PACE_PRED=1
run SPH_PACED=0
timeout -k 5 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
