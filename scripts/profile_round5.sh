#!/bin/bash
# Round-5 profile set (through gpurun): scripts/profile_round5.sh gpurun_out/r5p
#   profile_round.sh's set (bench lines with the driver's flags and the defaults, rocprofv3 --kernel-trace --stats of the driver-flag
#   command, FETCH_SIZE / WRITE_SIZE in their own --pmc passes)
#   + the SQ counters of the sweeps (instruction counts per wave, wait / active cycles, occupancy: VERDICT r4 next 2d) in separate
#     short passes (scripts/pmc_run2.sh: some TA / TD counters hang rocprofv3 on this pool, every pass has its own timeout)
#   + ONE kernel trace of the timed window AND its instrumented repeat (scripts/kt_two_passes.py: do the line's dispatch timestamps
#     agree with rocprofv3's, does instrumenting the queue change a sweep?)
#   + the other configs (profile_configs.sh: configs[2], [3] -- whose traffic decides roofline.bound --, [4]'s scene, configs[1] + level estimation)
# then: python scripts/summarize_profile.py gpurun_out/r5p profiles/r5p ; python scripts/pmc_table.py gpurun_out/r5p/sq > profiles/r5_sq_counters.txt
R=$1
bash $GRAFT_REPO_ROOT/scripts/profile_round.sh $R
bash $GRAFT_REPO_ROOT/scripts/pmc_run2.sh $R/sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" \
     "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
     "SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ_LATENCY" \
     "TCC_HIT TCC_MISS TCC_REQ" "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_LOAD_WAVEFRONTS"
OUT=$GRAFT_REPO_ROOT/$R
cd /tmp; export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt2 -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-8m > $OUT/kt2_bench.json 2> $OUT/kt2.err; echo "two-pass trace rc=$?"
cd $GRAFT_REPO_ROOT; python scripts/kt_two_passes.py $R/kt2 $R/kt2_bench.json | tee $R/two_passes.txt; rm -rf $OUT/kt2
bash $GRAFT_REPO_ROOT/scripts/profile_configs.sh $R
