#!/bin/bash
# same-box A/B of two library builds, per op of the 64-scene plan (bench.py --op-table): tools/optable_ab.sh old.so new.so
old=$1; new=$2
cp mv_ldm_amd/csrc/libmvldm_hip.so /tmp/lib_keep.so
cp "$old" /tmp/lib_old.so; cp "$new" /tmp/lib_new.so
for v in old new; do
  cp /tmp/lib_$v.so mv_ldm_amd/csrc/libmvldm_hip.so
  timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-batch --no-parity --no-train-line --no-full-walk --no-alt-dtype --no-dropin --no-other-configs \
      --op-table gpurun_out/optable_ab_$v.json > /dev/null 2>&1
done
cp /tmp/lib_keep.so mv_ldm_amd/csrc/libmvldm_hip.so
python3 tools/optable_ab.py gpurun_out/optable_ab_old.json gpurun_out/optable_ab_new.json
