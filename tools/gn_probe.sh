#!/bin/bash
# builds libmvldm_hip_exp_gn.so (product objects + norm.hip with the MVLDM_GN_SPAN / MVLDM_GN_NTHR experiment knobs) -- run HERE, then
# gpurun -- 'for s in 0 40 80; do for t in 1024 512 320; do MVLDM_GN_SPAN=$s MVLDM_GN_NTHR=$t python3 tools/gn_time.py 64 --lib-suffix _gn; done; done'
set -e
cd "$(dirname "$0")/.."
python -m mv_ldm_amd._build > /dev/null
C=mv_ldm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMVLDM_EXPERIMENTS $GN_EXTRA -x hip -c $C/norm.hip -o /tmp/norm_exp.o
OBJS=$(ls $C/*.o | grep -v "/norm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmvldm_hip_exp_gn.so $OBJS /tmp/norm_exp.o
echo built $C/libmvldm_hip_exp_gn.so
