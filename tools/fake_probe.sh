#!/bin/bash
# EXPERIMENT: MVLDM_IGEMM_FAKE bits: 1 = A pieces out of range (zeros, no L2 traffic), 2 = same for W,
# 4 = no global stores / residual loads in the epilogue, 8 = no epilogue at all.
# The knob only exists in a library built with -DMVLDM_EXPERIMENTS: this script builds one, probes, and restores
# the product build.
MVLDM_EXPERIMENTS=1 python -m mv_ldm_amd._build --force > /dev/null
for f in "$@"; do
  MVLDM_IGEMM_FAKE=$f python tools/igemm_sweep.py --scenes 32 --out gpurun_out/fake$f.json > /dev/null 2>&1
done
python -m mv_ldm_amd._build --force > /dev/null
