#!/bin/bash
# Per-kernel time of steady-state training steps: rocprofv3 kernel trace of `bench.py --train`, cut into optimizer steps (tools/train_trace_cut.py).
set -u
mkdir -p gpurun_out/train_stats
export TMPDIR=/tmp
timeout 700 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/train_stats -o ts -- python3 bench.py --train --steps 6 --warmup 2 --no-profile > gpurun_out/train_stats.log 2>&1
echo "rc $?"
f=$(find gpurun_out/train_stats -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python3 tools/train_trace_cut.py "$f" gpurun_out/r06_train_kernel_stats.csv
grep '^{' gpurun_out/train_stats.log | tail -1 | cut -c1-300
rm -rf gpurun_out/train_stats
