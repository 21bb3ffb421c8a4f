"""replay the DDIM-step graph of the B-scene sampler N times (valid inputs) -- the program to put under rocprofv3 --kernel-trace
(per-kernel durations and the gaps between graph nodes).  With MVLDM_TUNE_CACHE=<file> from an earlier run the plans are recorded
without one trial launch, so every dispatch in the trace is a plan launch.
   python3 tools/step_trace.py [scenes=1] [replays=20]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import mv_ldm_amd
import mv_ldm_amd._lib as L
from mv_ldm_amd import plan as P
from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg
from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
from mv_ldm_amd.scheduler import DDIMScheduler
from mv_ldm_amd.vae import AutoencoderKL

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
L.load()
mv_ldm_amd.set_compute_dtype(torch.bfloat16)
with torch.device(dev):
    den = MultiViewUNet(MultiViewUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1", allow_random_init=True), 11, 4)
    vae = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1", allow_random_init=True)
bench.random_init_(den, 1234)
bench.random_init_(vae, 1235)
pipe = MVLDMPipeline(den, vae, DDIMScheduler(clip_sample=False), SamplerCfg(True, 3.0, 50))
pipe.set_timesteps(50)
batch = bench.synthetic_batch(B, 1, 4, 256, 1234, dev, scene_ids=list(range(B)))
n0 = len(P._TUNE_CACHE)
st = pipe.prepare(batch)
print(f"tune cache: {n0} entries loaded, {len(P._TUNE_CACHE) - n0} problems timed while recording", flush=True)
st["plan"].replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    st["plan"].replay()
torch.cuda.synchronize()
print(f"B={B}: {1e3 * (time.perf_counter() - t0) / N:.3f} ms per DDIM step over {N} replays ({len(st['plan'].meta)} plan ops)", flush=True)
if os.environ.get("MVLDM_OP_TABLE"):
    import json
    ms = st["plan"].profile(10)
    with open(os.environ["MVLDM_OP_TABLE"], "w") as f:
        json.dump([{"name": m.name, "kind": m.kind, "ms": t, "flops": m.flops, "bytes": m.bytes} for m, t in zip(st["plan"].meta, ms)], f, indent=0)
    print(f"per-op table ({sum(ms):.3f} ms eager sum) -> {os.environ['MVLDM_OP_TABLE']}", flush=True)
