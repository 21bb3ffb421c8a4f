#!/bin/bash
# builds libmvldm_hip_exp_rs.so (product objects + linear_rs.hip with the experiment knobs) -- run HERE (cross-compile), then
# gpurun -- 'for f in 0 1 2 3 4 7 8 15; do python tools/rs_probe.py $f; done'
set -e
cd "$(dirname "$0")/.."
python -m mv_ldm_amd._build > /dev/null
C=mv_ldm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMVLDM_EXPERIMENTS $RS_EXTRA -x hip -c $C/linear_rs.hip -o /tmp/linear_rs_exp.o
OBJS=$(ls $C/*.o | grep -v linear_rs.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmvldm_hip_exp_rs${RS_SUFFIX}.so $OBJS /tmp/linear_rs_exp.o
echo built $C/libmvldm_hip_exp_rs${RS_SUFFIX}.so
