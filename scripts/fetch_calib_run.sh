#!/bin/bash
# usage (on the GPU box): bash scripts/fetch_calib_run.sh <outdir under gpurun_out>   -- two rocprofv3 --pmc passes of scripts/ubench/fetch_calib
OUT=$GRAFT_REPO_ROOT/$1
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/scripts/ubench/fetch_calib
timeout -k 5 300 $BIN > $OUT/bytes.csv 2> $OUT/run.err
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- $BIN > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- $BIN > $OUT/write.log 2>&1; echo "write rc=$?"
python $GRAFT_REPO_ROOT/scripts/fetch_calib_table.py $OUT > $OUT/table.md; cat $OUT/table.md
