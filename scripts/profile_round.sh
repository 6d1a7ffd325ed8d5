#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/profile_round.sh gpurun_out/r2b
#   bench.py with the driver's flags (--steps 20 --warmup 5: the judged line) -> bench.json ; with its defaults -> bench_default.json ;
#   rocprofv3 --kernel-trace --stats of the driver-flag command -> kt/ ; HBM counters in their own passes (FETCH_SIZE, WRITE_SIZE)
#   over the same window -> pmc_fetch/, pmc_write/
# then:  python scripts/summarize_profile.py gpurun_out/r2b profiles/r2b
OUT=$GRAFT_REPO_ROOT/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench (driver flags) rc=$?"
python bench.py --no-cpu-baseline --no-extra > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench (defaults) rc=$?"
cd /tmp; export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-8m --profile-steps 0 > $OUT/kt.log 2>&1; echo "kernel trace rc=$?"
SPH_TIME_WARMUP=5 timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 20 > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
SPH_TIME_WARMUP=5 timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 20 > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
python $GRAFT_REPO_ROOT/scripts/kt_gaps.py $OUT/kt
head -c 500 $OUT/bench.json; echo
