#!/bin/bash
# Two ranks on ONE GPU over gloo (the only multi-rank GPU run a 1-GPU box allows): `bench.py --train` with the 16-bit parameter
# gather and with the fp32 gather (MVLDM_TRAIN_GATHER16=0).  The losses and the gradient norm of the two must agree to the
# digits printed (the packs every rank computes with are bit-identical, tests/test_dist_gloo.py); the timing is NOT a scaling
# number (two ranks time-slice one device).
set -u
mkdir -p gpurun_out
export MVLDM_BENCH_SHARE_GPU=1 MVLDM_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for g in ${GATHER_MODES:-1 0}; do
  MVLDM_TRAIN_GATHER16=$g timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$g \
    bench.py --train --gpus 2 --steps 3 --warmup 2 --scenes 2 > gpurun_out/r06_train_2rank_g16_$g$TAG.log 2>&1
  echo "gather16=$g exit $?"
  grep '^{' gpurun_out/r06_train_2rank_g16_$g$TAG.log | tail -1 > gpurun_out/r06_train_2rank_g16_$g$TAG.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_train_2rank_g16_$g$TAG.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "loss_first_last", "grad_norm")}, d.get("comm"))
except Exception as e:
    print("no json:", e)
PY
done
