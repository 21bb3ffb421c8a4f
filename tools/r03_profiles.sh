#!/bin/bash
# the round's committed evidence, one GPU call: the default bench line (+ op table), rocprofv3 per-kernel stats of the same command,
# SQ counters per kernel (MFMA / VALU busy, LDS conflicts), HBM-side traffic (two separate PMC passes), the training bench.
# Writes under gpurun_out/; copy the summaries into profiles/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --steps 6 --warmup 2 --op-table $R/gpurun_out/r03_optable_b64.json > $R/gpurun_out/r03_bench.json 2> $R/gpurun_out/r03_bench.err
python3 bench.py --train --steps 8 --warmup 2 > $R/gpurun_out/r03_train_bench_n1.json 2>> $R/gpurun_out/r03_bench.err
bash $R/tools/pmc_kernels.sh r03_unet python3 bench.py --unet-pass-only > $R/gpurun_out/r03_mfma_util.txt 2>&1
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/prof_r03
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python3 bench.py --unet-pass-only > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python3 bench.py --unet-pass-only > /dev/null 2>&1
F=$(find $R/gpurun_out/pmc_fetch -name p_counter_collection.csv | head -1); W=$(find $R/gpurun_out/pmc_write -name p_counter_collection.csv | head -1)
cp $R/profiles/pmc_traffic.json $R/gpurun_out/r03_pmc_traffic.json
python3 tools/pmc_traffic.py $F $W $R/gpurun_out/r03_pmc_traffic.json 192 bf16_b64_res256 "round 3 (tools/r03_profiles.sh), the last 192 igemm dispatches of bench.py --unet-pass-only" > $R/gpurun_out/r03_pmc_traffic.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03 -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-small-batch --no-parity --no-train-line --no-alt-dtype --no-full-walk > $R/gpurun_out/r03_prof_bench.log 2>&1
S=$(find $R/gpurun_out/prof_r03 -name "bench_kernel_stats.csv" | head -1)
cp $S $R/gpurun_out/r03_kernel_stats.csv
# attention counters (d = 40 level-0 3-D block and d = 64 SD self-attention at 64 scenes), the weight-gradient forms
bash $R/tools/pmc_kernels.sh r03_attn40 python3 tools/attn_one.py 40 8 64 > /dev/null 2>&1
bash $R/tools/pmc_kernels.sh r03_attn64 python3 tools/attn_one.py 64 5 64 > /dev/null 2>&1
python3 - <<'PY'
import json, os
R = os.environ["GRAFT_REPO_ROOT"]
out = {}
for tag, what in (("r03_attn40", "level-0 3-D attention: 8 heads x 40, 64 scenes (5120 + 4096 keys)"), ("r03_attn64", "SD self-attention shape: 5 heads x 64, 64 scenes")):
    d = json.load(open(f"{R}/gpurun_out/pmc_{tag}.json"))
    out[what] = {k: v for k, v in d.items() if "attention" in k}
json.dump(out, open(f"{R}/gpurun_out/r03_attention_pmc.json", "w"), indent=1)
PY
python3 tools/attn_bench.py 64 > $R/gpurun_out/r03_attn_bench.txt 2>&1
# the raw traces are large: keep the summaries only
rm -rf $R/gpurun_out/pmc_r03_attn40 $R/gpurun_out/pmc_r03_attn64
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/prof_r03 $R/gpurun_out/pmc_r03_unet
head -8 $R/gpurun_out/r03_kernel_stats.csv | cut -c1-200
tail -c 400 $R/gpurun_out/r03_bench.json
