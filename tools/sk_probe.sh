#!/bin/bash
# builds libmvldm_hip_exp_sk.so (product objects + skinny.hip with the experiment knobs) -- run HERE (cross-compile), then
# gpurun -- 'for f in 0 1 2 3 4 8 16; do MVLDM_SK_FAKE=$f python tools/skinny_bench.py --lib-suffix _sk --only "L3 conv3x3 1280" --cfgs 1,21; done'
set -e
cd "$(dirname "$0")/.."
python -m mv_ldm_amd._build > /dev/null
C=mv_ldm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMVLDM_EXPERIMENTS $SK_EXTRA -x hip -c $C/skinny.hip -o /tmp/skinny_exp.o
OBJS=$(ls $C/*.o | grep -v skinny.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmvldm_hip_exp_sk.so $OBJS /tmp/skinny_exp.o
echo built $C/libmvldm_hip_exp_sk.so
