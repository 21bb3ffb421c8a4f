#!/bin/bash
# the round's committed evidence, one GPU call: SQ counters per kernel (MFMA / VALU busy, LDS conflicts), HBM-side traffic,
# rocprofv3 per-kernel stats.  Writes under gpurun_out/; copy the summaries into profiles/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_kernels.sh r02_unet python3 bench.py --unet-pass-only > $R/gpurun_out/r02_mfma_util.txt 2>&1
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/prof_r02
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python3 bench.py --unet-pass-only > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python3 bench.py --unet-pass-only > /dev/null 2>&1
F=$(find $R/gpurun_out/pmc_fetch -name p_counter_collection.csv | head -1); W=$(find $R/gpurun_out/pmc_write -name p_counter_collection.csv | head -1)
python3 tools/pmc_traffic.py $F $W $R/gpurun_out/r02_pmc_traffic.json 192 > $R/gpurun_out/r02_pmc_traffic.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-small-batch --no-parity --no-train-line > $R/gpurun_out/r02_prof_bench.log 2>&1
S=$(find $R/gpurun_out/prof_r02 -name "bench_kernel_stats.csv" | head -1)
cp $S $R/gpurun_out/r02_kernel_stats.csv
# the raw traces are large: keep the summaries only
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/prof_r02 $R/gpurun_out/pmc_r02_unet
head -12 $R/gpurun_out/r02_kernel_stats.csv | cut -c1-200
