#!/bin/bash
# same-box A/B of two library builds: small-batch steps (b = 1, 4, 16), the 64-scene step and the training line: tools/small_train_ab.sh old.so new.so
old=$1; new=$2
cp mv_ldm_amd/csrc/libmvldm_hip.so /tmp/lib_keep.so
cp "$old" /tmp/lib_old.so; cp "$new" /tmp/lib_new.so
for v in old new old new; do
  cp /tmp/lib_$v.so mv_ldm_amd/csrc/libmvldm_hip.so
  timeout 700 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-full-walk --no-alt-dtype --no-dropin --no-other-configs 2>/dev/null \
    | grep '^{' | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$v', d['value'], d.get('ddim_step_ms'), {k: v['ddim_step_ms'] for k, v in d['small_batch'].items()}, 'train', d.get('training', {}).get('value'))"
done
cp /tmp/lib_keep.so mv_ldm_amd/csrc/libmvldm_hip.so
