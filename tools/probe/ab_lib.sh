#!/bin/bash
# A/B two builds of libmvldm_hip.so on ONE box (box-to-box clocks differ by a few per cent):
#   tools/probe/ab_lib.sh old.so new.so python tools/attn_one.py 64 5 32
old=$1; new=$2; shift 2
cp mv_ldm_amd/csrc/libmvldm_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do
  for v in "$old" "$new"; do
    cp "$v" mv_ldm_amd/csrc/libmvldm_hip.so
    echo -n "$(basename $v): "; "$@" | tail -1
  done
done
cp /tmp/lib_keep.so mv_ldm_amd/csrc/libmvldm_hip.so
