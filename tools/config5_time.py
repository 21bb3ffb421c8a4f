"""configs[4] shape (BASELINE.json: 8-view 512x512 sampling, fp16, 50 DDIM steps) through `MVLDMPipeline.sample` on synthetic scenes:
views/s and ms per DDIM step.   python tools/config5_time.py [scenes=8] [samples=2]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import mv_ldm_amd
from mv_ldm_amd import _lib
from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg
from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
from mv_ldm_amd.scheduler import DDIMScheduler
from mv_ldm_amd.vae import AutoencoderKL

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
_lib.load()
mv_ldm_amd.set_compute_dtype(torch.float16)
with torch.device(dev):
    den = MultiViewUNet(MultiViewUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1", allow_random_init=True), 11, 4)
    vae = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1", allow_random_init=True)
bench.random_init_(den, 1234)
bench.random_init_(vae, 1235)
pipe = MVLDMPipeline(den, vae, DDIMScheduler(clip_sample=False), SamplerCfg(True, 3.0, 50))
pipe.set_timesteps(50)
batch = bench.synthetic_batch(B, 1, 8, 512, 1234, dev)
out = pipe.sample(batch)
torch.cuda.synchronize()
img = out[0] if isinstance(out, (tuple, list)) else out
assert torch.isfinite(img).all()
t0 = time.perf_counter()
for _ in range(K):
    pipe.sample(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(json.dumps({"workload": f"configs[4]: {B} scene(s) x (1 ctx + 8 tgt) @ 512x512, 50 DDIM steps, CFG 3.0, f16, VAE encode + decode",
                  "views_per_s": round(B * 8 / dt, 3), "sample_s": round(dt, 3), "approx_ms_per_ddim_step": round(dt / 50 * 1e3, 2),
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
