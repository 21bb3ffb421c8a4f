#!/bin/bash
# builds libmvldm_hip_exp.so (product objects + linear_pw.hip with the experiment knobs) -- run HERE (cross-compile), then
# gpurun -- 'for f in 0 1 2 3 4 7; do python tools/pw_probe.py $f; done'      (PW_EXTRA=-DMVLDM_EXPERIMENTS_NOGELU: GELU -> identity)
set -e
cd "$(dirname "$0")/.."
python -m mv_ldm_amd._build > /dev/null
C=mv_ldm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMVLDM_EXPERIMENTS $PW_EXTRA -x hip -c $C/linear_pw.hip -o /tmp/linear_pw_exp.o
OBJS=$(ls $C/*.o | grep -v linear_pw.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmvldm_hip_exp${PW_SUFFIX}.so $OBJS /tmp/linear_pw_exp.o
echo built $C/libmvldm_hip_exp${PW_SUFFIX}.so
