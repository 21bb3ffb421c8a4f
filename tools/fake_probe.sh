#!/bin/bash
# EXPERIMENT: how fast is the main loop when the operand pieces never touch L2 (range-check zeros)?
for f in 0 3; do
  MVLDM_IGEMM_FAKE=$f python tools/igemm_sweep.py --scenes 32 --out gpurun_out/fake$f.json > /dev/null 2>&1
done
