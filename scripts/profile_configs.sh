#!/bin/bash
# usage (through gpurun): scripts/profile_configs.sh gpurun_out/r2a
#   for each of the other BASELINE configs: rocprofv3 --kernel-trace --stats of scripts/gpu_time.py, then FETCH_SIZE and
#   WRITE_SIZE in their own --pmc passes.   then:  python scripts/summarize_profile.py gpurun_out/r2a/<cfg> profiles/r2a_<cfg>
OUT=$GRAFT_REPO_ROOT/$1
cd /tmp; export TMPDIR=/tmp
run() {   # name workload steps overrides
  local D=$OUT/$1; mkdir -p $D
  SPH_TIME_WARMUP=${5:-20} timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $2 $3 "$4" > $D/kt.log 2>&1; echo "$1 kernel trace rc=$? $(tail -1 $D/kt.log)"
  SPH_TIME_WARMUP=${5:-20} timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $2 6 "$4" > $D/pmc_fetch.log 2>&1; echo "$1 fetch rc=$?"
  SPH_TIME_WARMUP=${5:-20} timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $2 6 "$4" > $D/pmc_write.log 2>&1; echo "$1 write rc=$?"
}
run adaptive_4to1 dam_break_1m_adaptive 30 "dict()"
run dam_break_8m dam_break_8m 20 "dict()" 10
run ratio_stress_4m ratio_stress_4m 20 "dict()" 5
run level_1m dam_break_1m 20 "dict(level_estimation_method='EmptyAngle', maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002)"
