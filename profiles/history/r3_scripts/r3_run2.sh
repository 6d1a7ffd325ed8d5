#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 120 ./scripts/ubench/bin/valu_issue > $OUT/valu_issue.md 2>&1; echo "valu_issue rc=$?"; cat $OUT/valu_issue.md
timeout -k 5 120 ./scripts/ubench/bin/jacobi_lab 1024 0.15 50 > $OUT/jacobi_lab_j15.md 2>&1; echo "lab rc=$?"; cat $OUT/jacobi_lab_j15.md
timeout -k 5 120 ./scripts/ubench/bin/jacobi_lab 1024 0.0 50 > $OUT/jacobi_lab_j0.md 2>&1; cat $OUT/jacobi_lab_j0.md
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench_window or full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
cat gpurun_out/bench_window_parity.txt
