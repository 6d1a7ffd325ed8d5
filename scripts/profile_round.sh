#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/profile_round.sh gpurun_out/r1b
#   bench.py (the judged line) -> bench.json ; rocprofv3 --kernel-trace --stats of the same command -> kt/ ;
#   HBM counters in their own passes (FETCH_SIZE, WRITE_SIZE) -> pmc_fetch/, pmc_write/
# then:  python scripts/summarize_profile.py gpurun_out/r1b profiles/r1b
OUT=$GRAFT_REPO_ROOT/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp; export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --profile-steps 0 > $OUT/kt.log 2>&1; echo "kernel trace rc=$?"
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 10 > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 10 > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
head -c 600 $OUT/bench.json; echo
