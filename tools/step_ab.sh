#!/bin/bash
# same-box A/B of two builds of the library on the headline workload (short bench line: value + DDIM step), alternating, 2 rounds
#   tools/step_ab.sh mv_ldm_amd/csrc/libmvldm_hip_exp_burst.so mv_ldm_amd/csrc/libmvldm_hip.so
old=$1; new=$2
cp mv_ldm_amd/csrc/libmvldm_hip.so /tmp/lib_keep.so
cp "$old" /tmp/lib_old.so; cp "$new" /tmp/lib_new.so
for rep in 1 2; do
  for v in old new; do
    cp /tmp/lib_$v.so mv_ldm_amd/csrc/libmvldm_hip.so
    timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-small-batch --no-parity --no-train-line --no-full-walk --no-alt-dtype --no-dropin --no-other-configs 2>/dev/null \
      | grep '^{' | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$v', d['value'], 'views/s', d['ms_per_step'], 'ms/sample', d.get('ddim_step_ms'), d.get('roofline', {}).get('frac'))"
  done
done
cp /tmp/lib_keep.so mv_ldm_amd/csrc/libmvldm_hip.so
