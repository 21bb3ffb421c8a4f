#!/bin/bash
# builds libmvldm_hip_exp_pwt.so (product objects + linear_pw.hip with -DMVLDM_PW_TRACE: s_memtime stamps of one wave) -- run HERE, then
# gpurun -- 'python tools/pw_trace.py L2.qkv'
set -e
cd "$(dirname "$0")/.."
python -m mv_ldm_amd._build > /dev/null
C=mv_ldm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMVLDM_PW_TRACE -x hip -c $C/linear_pw.hip -o /tmp/linear_pw_trace.o
OBJS=$(ls $C/*.o | grep -v linear_pw.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmvldm_hip_exp_pwt.so $OBJS /tmp/linear_pw_trace.o
echo built $C/libmvldm_hip_exp_pwt.so
