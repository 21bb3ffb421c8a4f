#!/bin/bash
# builds libmvldm_hip_exp.so (product objects + linear_pp.hip with the experiment knob) -- run HERE (cross-compile), then
# gpurun -- 'for f in 0 1 2 3 7; do python tools/lpp_probe.py $f; done'
set -e
cd "$(dirname "$0")/.."
python -m mv_ldm_amd._build > /dev/null
C=mv_ldm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMVLDM_EXPERIMENTS $LPP_EXTRA -x hip -c $C/linear_pp.hip -o /tmp/linear_pp_exp.o
OBJS=$(ls $C/*.o | grep -v linear_pp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmvldm_hip_exp.so $OBJS /tmp/linear_pp_exp.o
echo built $C/libmvldm_hip_exp.so
