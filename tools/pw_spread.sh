#!/bin/bash
# experiment builds of tile 13's spread issue (tile bit 14): where the P = 8 / 9 LDS-DMA pieces of a step go -- pieces behind the barrier (sub-step 3) /
# in sub-steps 0 / 1 of the next step, the rest in its sub-step 2.  Builds libmvldm_hip_exp_sp<a><b><c>.so; run HERE, then on the GPU
#   for v in 333 423 243 033 522 900; do python tools/linear_tiles.py 64 bf16 13,16397,13,16397 --lib libmvldm_hip_exp_sp$v.so; done
set -e
cd "$(dirname "$0")/.."
python -m mv_ldm_amd._build > /dev/null
C=mv_ldm_amd/csrc
OBJS=$(ls $C/*.o | grep -v linear_pw.o)
for v in ${VARIANTS:-333 423 243 033 522 900}; do
    a=${v:0:1}; b=${v:1:1}; c=${v:2:1}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DPW_SP3=$a -DPW_SP0=$b -DPW_SP1=$c -x hip -c $C/linear_pw.hip -o /tmp/linear_pw_sp$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmvldm_hip_exp_sp$v.so $OBJS /tmp/linear_pw_sp$v.o
    echo built $C/libmvldm_hip_exp_sp$v.so
done
