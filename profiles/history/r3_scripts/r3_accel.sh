#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" SPH_TIME_WARMUP=5 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 2>&1 | tail -1; env "$@" timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 100 2>&1 | tail -1; }
for rep in 1 2; do
run SPH_ACCEL_GENERIC=1
run SPH_X=1
done
WL=dam_break_8m
for e in SPH_ACCEL_GENERIC=1 SPH_X=1; do env $e timeout -k 5 200 python scripts/gpu_time.py dam_break_8m 20 2>&1 | tail -1; done
timeout -k 5 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
