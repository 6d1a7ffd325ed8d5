#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 120 python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout -k 5 90 ./scripts/ubench/bin/jacobi_lab_check 1024 0.15 2 > $OUT/jacobi_lab_check.md 2>&1; echo "lab check rc=$?"; cat $OUT/jacobi_lab_check.md

timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "level or dedicated" > $OUT/pytest_level.log 2>&1; echo "pytest level rc=$?"; tail -5 $OUT/pytest_level.log
LV="dict(level_estimation_method='EmptyAngle', maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002)"
timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 "$LV" 2>&1 | tail -1
SPH_LEVEL_GENERIC=1 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 "$LV" 2>&1 | tail -1
timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 2>&1 | tail -1
timeout -k 5 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench_window" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
