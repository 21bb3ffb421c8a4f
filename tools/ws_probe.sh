#!/bin/bash
# builds libmvldm_hip_exp${WS_SUFFIX}.so (product objects + linear_ws.hip with the experiment knobs) -- run HERE (cross-compile), then
# gpurun -- 'for f in 0 1 4 5; do python tools/ws_probe.py $f; done'      (WS_EXTRA=-DMVLDM_EXPERIMENTS_NOGELU: GELU -> identity)
set -e
cd "$(dirname "$0")/.."
python -m mv_ldm_amd._build > /dev/null
C=mv_ldm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMVLDM_EXPERIMENTS $WS_EXTRA -x hip -c $C/linear_ws.hip -o /tmp/linear_ws_exp.o
OBJS=$(ls $C/*.o | grep -v linear_ws.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmvldm_hip_exp${WS_SUFFIX}.so $OBJS /tmp/linear_ws_exp.o
echo built $C/libmvldm_hip_exp${WS_SUFFIX}.so
