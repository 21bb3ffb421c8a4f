#!/bin/bash
# rocprofv3 kernel stats of the training bench (tune cache primed by an untraced run first: the traced run times no candidates)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MVLDM_TUNE_CACHE=/tmp/train_tune_cache.json
O=gpurun_out/r04prof; mkdir -p $O
python3 bench.py --train --steps 2 --warmup 1 --no-profile --no-parity > $O/train_pre.json 2> $O/train_pre.err
rm -rf /tmp/trstats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trstats -o p -- python3 bench.py --train --steps 8 --warmup 2 --no-profile --no-parity > $O/train_traced.json 2> $O/train_traced.err
cp $(find /tmp/trstats -name p_kernel_stats.csv | head -1) $O/r04_rocprofv3_stats_train.csv
head -40 $O/r04_rocprofv3_stats_train.csv
