#!/bin/bash
# (lab build) the level-set propagation on a side stream that OWNS a few CUs of every XCD (hipExtStreamCreateWithCUMask): does it run
# under the step's sweeps then?  configs[1] + EmptyAngle, 40 steps after 20; and what masking the main stream off those CUs costs the headline
cd $GRAFT_REPO_ROOT
export SPH_HIP_LIBRARY=libsph_lab.so
LV="dict(level_estimation_method='EmptyAngle', maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002)"
for E in "" "SPH_SIDE_CUS=1" "SPH_SIDE_CUS=2" "SPH_SIDE_CUS=4" "SPH_SIDE_CUS=1 SPH_MAIN_EXCLUDE=1" "SPH_SIDE_CUS=2 SPH_MAIN_EXCLUDE=1" "SPH_SIDE_CUS=4 SPH_MAIN_EXCLUDE=1" "SPH_LEVEL_SERIAL=1"; do
  echo "== level estimation on, ${E:-defaults}"; env $E python scripts/gpu_time.py dam_break_1m 40 "$LV" 2>&1 | grep -v amdgpu | tail -1
done
for E in "" "SPH_SIDE_CUS=2 SPH_MAIN_EXCLUDE=1" "SPH_SIDE_CUS=4 SPH_MAIN_EXCLUDE=1"; do
  echo "== headline (no level estimation), ${E:-defaults}"; env $E SPH_TIME_WARMUP=5 python scripts/gpu_time.py dam_break_1m 20 2>&1 | grep -v amdgpu | tail -1
done
