#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 120 python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout -k 5 90 ./scripts/ubench/bin/valu_issue > $OUT/valu_issue.md 2>&1; echo "valu_issue rc=$?"; cat $OUT/valu_issue.md
timeout -k 5 90 ./scripts/ubench/bin/jacobi_lab 1024 0.15 50 > $OUT/jacobi_lab_j15.md 2>&1; echo "lab rc=$?"; cat $OUT/jacobi_lab_j15.md
timeout -k 5 90 ./scripts/ubench/bin/jacobi_lab 1024 0.0 50 > $OUT/jacobi_lab_j0.md 2>&1; cat $OUT/jacobi_lab_j0.md
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "level or one_launch" > $OUT/pytest_level.log 2>&1; echo "pytest level rc=$?"; tail -5 $OUT/pytest_level.log
LV="dict(level_estimation_method='EmptyAngle', maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002)"
timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 "$LV" 2>&1 | tail -1
SPH_LEVEL_LAUNCHES=1 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 "$LV" 2>&1 | tail -1
timeout -k 5 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench_window or full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
cat gpurun_out/bench_window_parity.txt
