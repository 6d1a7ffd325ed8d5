#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests -m gpu -x -q -k "level and (slab or group or rank or multiprocess)" 2>&1 | grep -E "passed|failed|rror" | tail -5
for k in 2 3; do
SPH_SLAB_LEVEL_PLAIN=1 timeout -k 5 300 python scripts/gpu_slab_level_time.py $k 20 2>&1 | tail -2
timeout -k 5 300 python scripts/gpu_slab_level_time.py $k 20 2>&1 | tail -2
done
