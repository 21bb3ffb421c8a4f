#!/bin/bash
# round-4 profiler evidence, trial-free: the tune cache of an earlier bench run (profiles/r04_tune_cache.json) is loaded, so the traced
# processes record their plans without timing one candidate tile.  Run on the GPU box:  bash tools/r04_profiles.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MVLDM_TUNE_CACHE=$PWD/profiles/r04_tune_cache.json
O=gpurun_out/r04prof; mkdir -p $O
cp profiles/r04_tune_cache.json /tmp/tune_cache_before.json
# 1. kernel trace + stats of one eager UNet + DDIM pass at 64 scenes (the population `roofline` is quoted on)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b64 -o p -- python3 bench.py --unet-pass-only > $O/b64.log 2>&1
# 2. kernel trace of 20 graph replays at one scene: busy time against wall time
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -o p -- python3 tools/step_trace.py 1 20 > $O/b1.log 2>&1
# 3. MFMA / VALU busy per kernel (PMC pass on its own: no trace domains beside --kernel-trace)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc -o p -- python3 bench.py --unet-pass-only > $O/pmc.log 2>&1
python3 tools/profile_tables.py $O 04
cmp -s /tmp/tune_cache_before.json profiles/r04_tune_cache.json && echo "tune cache unchanged: no problem was timed in the traced runs" | tee $O/trial_free.txt
# keep the small summaries, drop the raw traces (gpurun_out/ is capped at 64 MiB)
for d in b64 b1 pmc; do
  cp $O/$d/p_kernel_stats.csv $O/${d}_kernel_stats.csv 2>/dev/null
  rm -rf $O/$d
done
cp profiles/r04_kernel_stats.csv profiles/r04_mfma_util.json profiles/r04_b1_timeline.json $O/ 2>/dev/null
# 4. HBM-side traffic per kernel family (two PMC passes of their own), trial-free: the last 192 igemm dispatches = the eager UNet pass
P=/tmp/r04pmc; rm -rf $P
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/f -o p -- python3 bench.py --unet-pass-only > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/w -o p -- python3 bench.py --unet-pass-only > /dev/null 2>&1
python3 tools/pmc_traffic.py $(find $P/f -name p_counter_collection.csv | head -1) $(find $P/w -name p_counter_collection.csv | head -1) profiles/pmc_traffic.json 192 bf16_b64_res256 "round 4 (tools/r04_profiles.sh), the last 192 igemm dispatches of bench.py --unet-pass-only, plans from profiles/r04_tune_cache.json (no trials)" | tee $O/pmc_traffic.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
# 5. HBM-side traffic of one optimizer step of the training bench (same two counters, own passes)
rm -rf $P
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/tf -o p -- python3 bench.py --train --steps 2 --warmup 1 --no-profile --no-parity > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/tw -o p -- python3 bench.py --train --steps 2 --warmup 1 --no-profile --no-parity > /dev/null 2>&1
python3 tools/pmc_train_traffic.py $(find $P/tf -name p_counter_collection.csv | head -1) $(find $P/tw -name p_counter_collection.csv | head -1) profiles/pmc_traffic.json train_bf16_b4_res256 "round 4 (tools/r04_profiles.sh step 5), bench.py --train --steps 2 --warmup 1" | tee $O/pmc_train_traffic.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
