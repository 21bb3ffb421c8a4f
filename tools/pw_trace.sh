#!/bin/bash
# builds libmvldm_hip_exp_pwt.so (product objects + linear_pw.hip with -DMVLDM_PW_TRACE: s_memtime stamps of one wave) and
# libmvldm_hip_exp_pwc.so (the same + -DMVLDM_PW_FAKE_COALESCE: WRONG results, the epilogue's stores / residual loads addressed as whole
# 128-byte row segments) -- run HERE, then    gpurun -- 'python tools/pw_trace.py L2.qkv; python tools/pw_trace.py L2.qkv 8 0 pwc'
set -e
cd "$(dirname "$0")/.."
python -m mv_ldm_amd._build > /dev/null
C=mv_ldm_amd/csrc
OBJS=$(ls $C/*.o | grep -v linear_pw.o)
for v in "pwt:" "pwc:-DMVLDM_PW_FAKE_COALESCE"; do
    suf=${v%%:*}; extra=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMVLDM_PW_TRACE $extra -x hip -c $C/linear_pw.hip -o /tmp/linear_pw_$suf.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmvldm_hip_exp_$suf.so $OBJS /tmp/linear_pw_$suf.o
    echo built $C/libmvldm_hip_exp_$suf.so
done
