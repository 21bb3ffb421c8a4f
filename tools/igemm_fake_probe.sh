#!/bin/bash
# builds libmvldm_hip_exp.so = the product objects + igemm.hip compiled -DMVLDM_EXPERIMENTS (the MVLDM_IGEMM_FAKE knob) -- run HERE
# (cross-compile, ~6 min), then:  gpurun -- 'for f in 0 4 8 3 11; do python tools/igemm_fake_probe.py $f; done'
set -e
cd "$(dirname "$0")/.."
python -m mv_ldm_amd._build > /dev/null
C=mv_ldm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMVLDM_EXPERIMENTS -x hip -c $C/igemm.hip -o /tmp/igemm_exp.o
OBJS=$(ls $C/*.o | grep -v "/igemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmvldm_hip_exp.so $OBJS /tmp/igemm_exp.o
echo built $C/libmvldm_hip_exp.so
