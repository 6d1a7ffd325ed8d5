#!/bin/bash
# usage (through gpurun): scripts/ablate.sh "<hipcc flags variant 1>" "<variant 2>" ...
# rebuilds libsph_hip.so with each set of extra flags ON THE GPU BOX and prints the sweep medians + ms/step
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  echo "=== variant: $V"
  SPH_EXTRA_HIPCC_FLAGS="$V" python -c "from adaptive_sph_amd import build; build.build_hip(force=True)" 2>&1 | grep -E "error" | head -3
  scripts/kt_run.sh gpurun_out/ablate_tmp | grep -E "^Op|rc="
  python scripts/gpu_time.py dam_break_1m 100 | tail -1
done
python -c "from adaptive_sph_amd import build; build.build_hip(force=True)"
