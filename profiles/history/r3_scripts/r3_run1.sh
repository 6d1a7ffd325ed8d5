#!/bin/bash
# round 3, first GPU call: the suite (new full-size bars), the VALU issue ubench, profiles of the two multi-resolution configs
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
./scripts/ubench/bin/valu_issue > $OUT/valu_issue.md 2>&1; echo "valu_issue rc=$?"; cat $OUT/valu_issue.md
timeout -k 5 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
cat gpurun_out/bench_window_parity.txt
cd /tmp; export TMPDIR=/tmp
run() {   # name workload steps overrides warmup
  local D=$OUT/$1; mkdir -p $D
  SPH_TIME_WARMUP=${5:-20} timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $2 $3 "$4" > $D/kt.log 2>&1; echo "$1 kernel trace rc=$? $(tail -1 $D/kt.log)"
  SPH_TIME_WARMUP=${5:-20} timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $2 6 "$4" > $D/pmc_fetch.log 2>&1; echo "$1 fetch rc=$?"
  SPH_TIME_WARMUP=${5:-20} timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $2 6 "$4" > $D/pmc_write.log 2>&1; echo "$1 write rc=$?"
}
run adaptive_4to1 dam_break_1m_adaptive 30 "dict()"
run ratio_stress_4m ratio_stress_4m 20 "dict()" 5
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-8m > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; head -c 600 $OUT/bench.json; echo
