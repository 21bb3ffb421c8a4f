#!/bin/bash
# round-6 evidence in one call on the GPU box:  bash tools/r06_profiles.sh
#   0. the full default bench line (writes the round's tune cache, which makes every traced process below trial-free)
#   1. rocprofv3 kernel trace + stats of one eager UNet + DDIM pass at 64 scenes (the population `roofline` is quoted on)
#   2. 20 graph replays of the ONE-scene step under the tracer (tile 15 on) + the per-op table of that step
#   3. MFMA / VALU busy per kernel (PMC pass of its own)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MVLDM_TUNE_CACHE=$PWD/gpurun_out/r06_tune_cache.json
O=gpurun_out/r06prof; mkdir -p $O
rm -f $MVLDM_TUNE_CACHE
timeout 900 python3 bench.py --op-table gpurun_out/r06_optable_b64.json > gpurun_out/r06_bench.json 2> $O/bench.err
tail -c 600 gpurun_out/r06_bench.json; echo
cp $MVLDM_TUNE_CACHE /tmp/tune_cache_before.json
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b64 -o p -- python3 bench.py --unet-pass-only > $O/b64.log 2>&1
MVLDM_OP_TABLE=$PWD/gpurun_out/r06_optable_b1.json timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -o p -- python3 tools/step_trace.py 1 20 > $O/b1.log 2>&1
timeout 700 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc -o p -- python3 bench.py --unet-pass-only > $O/pmc.log 2>&1
python3 tools/profile_tables.py $O 06
cmp -s /tmp/tune_cache_before.json $MVLDM_TUNE_CACHE && echo "tune cache unchanged: no problem was timed in the traced runs" | tee $O/trial_free.txt
for d in b64 b1 pmc; do
  cp $O/$d/p_kernel_stats.csv $O/${d}_kernel_stats.csv 2>/dev/null
  rm -rf $O/$d
done
cp profiles/r06_kernel_stats.csv profiles/r06_mfma_util.json profiles/r06_b1_timeline.json profiles/r06_roofline_check.json $O/ 2>/dev/null
# 4. HBM-side traffic per kernel family (two PMC passes of their own), trial-free: the last 192 igemm dispatches = the eager UNet pass
P=/tmp/r06pmc; rm -rf $P
timeout 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/f -o p -- python3 bench.py --unet-pass-only > /dev/null 2>&1
timeout 700 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/w -o p -- python3 bench.py --unet-pass-only > /dev/null 2>&1
python3 tools/pmc_traffic.py $(find $P/f -name p_counter_collection.csv | head -1) $(find $P/w -name p_counter_collection.csv | head -1) profiles/pmc_traffic.json 192 bf16_b64_res256 "round 6 (tools/r06_profiles.sh), the last 192 igemm dispatches of bench.py --unet-pass-only, plans from the round's tune cache (no trials)" | tee $O/pmc_traffic.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
