#!/bin/bash
# A/B of the headline window (bench.py's timed region only) between library builds in adaptive_sph_amd/csrc: scripts/ab_head.sh libA.so libB.so ...
for i in 1 2 3; do
for lib in "$@"; do
  echo -n "$lib "; SPH_HIP_LIBRARY=$lib python bench.py --steps 20 --warmup 5 --no-8m --no-extra --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"
done; done
