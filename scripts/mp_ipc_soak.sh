#!/bin/bash
# Soak of the push transport's fused rounds (round 6): tests/mp_slab_check.py with 2, 3 and 4 real processes on the box's GPU, many steps,
# bit for bit against the loopback group.   usage (through gpurun): bash scripts/mp_ipc_soak.sh [steps=150]
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 SPH_TRANSPORT=ipc MP_SOAK=1 MP_STEPS=${1:-150}
for W in 2 3 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$W --master-addr 127.0.0.1 --master-port $((29700 + W)) tests/mp_slab_check.py 2>&1 | grep -E "MP_CHECK|Error|error|assert" | cut -c1-300 | head -5
  echo "world $W rc=${PIPESTATUS[0]}"
done
