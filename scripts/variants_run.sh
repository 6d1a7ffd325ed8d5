#!/bin/bash
# time prebuilt library variants (gpurun_variants/*.so, built locally) in one GPU call
cd $GRAFT_REPO_ROOT
cp adaptive_sph_amd/csrc/libsph_hip.so /tmp/keep.so
for f in gpurun_variants/*.so; do
  echo "=== $(basename $f)"
  cp $f adaptive_sph_amd/csrc/libsph_hip.so
  scripts/kt_run.sh gpurun_out/variants_tmp | grep -E "^Op(Jacobi|PressureAccel|Source|NonPressure|AiiConst) "
  python scripts/gpu_time.py dam_break_1m 60 | tail -1
done
cp /tmp/keep.so adaptive_sph_amd/csrc/libsph_hip.so
