#!/bin/bash
# like pmc_run.sh but with a short timeout per pass (some TA/TD counters hang rocprofv3 on this pool)
OUT=$1; shift
mkdir -p $GRAFT_REPO_ROOT/$OUT; cd /tmp; export TMPDIR=/tmp
k=0
for PASS in "$@"; do
  k=$((k+1))
  timeout -k 5 100 rocprofv3 --pmc $PASS --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$k -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 6 > $GRAFT_REPO_ROOT/$OUT/p$k.log 2>&1
  echo "pass $k ($PASS) rc=$?"
done
