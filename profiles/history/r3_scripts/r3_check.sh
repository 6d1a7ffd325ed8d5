#!/bin/bash
# quick regression + timing: sort / parity subset, then the two windows
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests -m gpu -x -q -k "${1:-sort or parity or chain or abi}" 2>&1 | tail -4
for rep in 1 2; do
SPH_TIME_WARMUP=5 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 2>&1 | tail -1
timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 100 2>&1 | tail -1
done
