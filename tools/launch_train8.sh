#!/bin/bash
# Training-step scaling curve on one 8-GPU MI355X node (BASELINE.json configs[3]: DDP over RCCL / xGMI).
# One process per GPU; the launcher is the FIRST program that runs (nothing has touched the GPU before torch.distributed.run
# spawns the ranks).  Prints one JSON line per N; scaling efficiency is for the reader to compute from the values.
#   tools/launch_train8.sh [steps] [warmup]
set -e
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
STEPS=${1:-10}
WARMUP=${2:-3}
for N in 1 2 4 8; do
    if [ "$N" = 1 ]; then
        python3 bench.py --gpus 1 --train --steps "$STEPS" --warmup "$WARMUP" --no-profile
    else
        python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
            bench.py --gpus "$N" --train --steps "$STEPS" --warmup "$WARMUP" --no-profile
    fi
done
