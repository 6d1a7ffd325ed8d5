#!/bin/bash
# usage (gpurun): scripts/slab_vs_plain_trace.sh gpurun_out/slabtrace -- kernel traces of the plain context and of ONE rank in forced slab mode
OUT=$GRAFT_REPO_ROOT/$1
WL=${2:-dam_break_1m}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
SPH_TIME_WARMUP=20 timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/plain -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $WL 40 > $OUT/plain.log 2>&1; echo "plain rc=$?"
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/slab -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_forced_slab_time.py 40 $WL > $OUT/slab.log 2>&1; echo "slab rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/gpu_time.py $WL 60 > $OUT/plain_time.txt 2>&1
python scripts/gpu_forced_slab_time.py 60 $WL > $OUT/slab_time.txt 2>&1
tail -n 1 $OUT/plain_time.txt; grep "forced slab" $OUT/slab_time.txt
python scripts/kt_step_timeline.py $OUT/plain k_cell_start > $OUT/plain_timeline.txt
python scripts/kt_step_timeline.py $OUT/slab k_cell_start > $OUT/slab_timeline.txt
python scripts/kt_step_timeline.py $OUT/slab k_cell_start --full > $OUT/slab_timeline_full.txt
python scripts/kt_step_timeline.py $OUT/plain k_cell_start --full > $OUT/plain_timeline_full.txt
rm -rf $OUT/plain $OUT/slab
