#!/bin/bash
cd $GRAFT_REPO_ROOT
SPH_HIP_TRACE=1 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 100 2>&1 | tail -4
SPH_HIP_TRACE=1 SPH_TIME_WARMUP=5 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/steady_kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 100 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/kt_gaps.py $GRAFT_REPO_ROOT/gpurun_out/steady_kt
