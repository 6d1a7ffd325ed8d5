#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests -m gpu -x -q -k "level" 2>&1 | grep -E "passed|failed|rror" | tail -5
OV="dict(level_estimation_method='EmptyAngle', maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002)"
for r in 1 2; do
SPH_LEVEL_MARK_NOW=1 timeout -k 5 200 python scripts/gpu_time.py dam_break_1m 20 "$OV" 2>&1 | tail -1
timeout -k 5 200 python scripts/gpu_time.py dam_break_1m 20 "$OV" 2>&1 | tail -1
done
