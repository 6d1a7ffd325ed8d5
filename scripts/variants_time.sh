#!/bin/bash
# time prebuilt library variants (gpurun_variants/*.so, built locally): 3 runs of gpu_time.py each, plus the k_solver_final median
cd $GRAFT_REPO_ROOT
cp adaptive_sph_amd/csrc/libsph_hip.so /tmp/keep.so
for r in 1 2 3; do
for f in gpurun_variants/*.so; do
  cp $f adaptive_sph_amd/csrc/libsph_hip.so
  echo "$(basename $f): $(python scripts/gpu_time.py dam_break_1m 100 | tail -1)"
done; done
for f in gpurun_variants/*.so; do
  cp $f adaptive_sph_amd/csrc/libsph_hip.so
  echo "=== $(basename $f)"; scripts/kt_run.sh gpurun_out/variants_tmp | grep -E "k_solver_final"
done
cp /tmp/keep.so adaptive_sph_amd/csrc/libsph_hip.so
