#!/bin/bash
# HBM-side traffic of one optimizer step of the training bench (step 5 of tools/r04_profiles.sh on its own)
cd "$(dirname "$0")/.."
# (MVLDM_TRAIN_PREFETCH=0: with the next window's VAE encode running under the backward its kernels would land inside the plan's dispatch range)
export TMPDIR=/tmp MVLDM_TUNE_CACHE=/tmp/train_tune_cache.json MVLDM_TRAIN_PREFETCH=0
O=gpurun_out/${MVLDM_ROUND_DIR:-r05prof}; mkdir -p $O
P=/tmp/r04pmc; rm -rf $P
python3 bench.py --train --steps 2 --warmup 1 --no-profile --no-parity > $O/train_pre.json 2> $O/train_pre.err     # fills the tune cache: the traced runs time no candidates
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/tf -o p -- python3 bench.py --train --steps 2 --warmup 1 --no-profile --no-parity > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/tw -o p -- python3 bench.py --train --steps 2 --warmup 1 --no-profile --no-parity > /dev/null 2>&1
python3 tools/pmc_train_traffic.py $(find $P/tf -name p_counter_collection.csv | head -1) $(find $P/tw -name p_counter_collection.csv | head -1) profiles/pmc_traffic.json train_bf16_b4_res256 "${MVLDM_ROUND_NOTE:-round 5} (tools/pmc_train.sh), bench.py --train --steps 2 --warmup 1" | tee $O/pmc_train_traffic.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
