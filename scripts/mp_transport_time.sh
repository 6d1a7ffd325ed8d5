#!/bin/bash
# usage (gpurun): scripts/mp_transport_time.sh gpurun_out/r4_transports.md  -- the transports that can put several ranks on one GPU, timed
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1
echo "| workload | transport | driver window ms/step | iterations | exchanges/step | host waits/step | settled ms/step | iterations |" > $OUT
echo "|---|---|---|---|---|---|---|---|" >> $OUT
port=29700
for W in 2 4; do
  for T in shm ipc; do
    port=$((port+1))
    SPH_TRANSPORT=$T timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$W --master-addr 127.0.0.1 --master-port $port scripts/mp_transport_time.py dam_break_1m 2>/dev/null | grep "^|" >> $OUT
  done
done
cat $OUT
