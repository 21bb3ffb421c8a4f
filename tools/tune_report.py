"""which tile the plan-time selection froze for every large igemm problem of the bench plan.  python tools/tune_report.py [scenes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, mv_ldm_amd
from mv_ldm_amd import plan as P
from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg
from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
from mv_ldm_amd.scheduler import DDIMScheduler
from mv_ldm_amd.vae import AutoencoderKL
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.set_grad_enabled(False)
mv_ldm_amd.set_compute_dtype(torch.bfloat16)
with torch.device("cuda"):
    den = MultiViewUNet(MultiViewUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1"), 11, 4)
    vae = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1")
bench.random_init_(den, 1234); bench.random_init_(vae, 1235)
pipe = MVLDMPipeline(den, vae, DDIMScheduler(clip_sample=False), SamplerCfg(True, 3.0, 50))
pipe.set_timesteps(50)
pipe.prepare(bench.synthetic_batch(b, 1, 4, 256, 1, torch.device("cuda")))
for key, tile in sorted(P._TUNE_CACHE.items(), key=lambda kv: (kv[0][7], -kv[0][0] * kv[0][3] * kv[0][4])):
    n, h, w, ho, wo, c0, c1, ks = key[:8]
    if h == 256 or h == 128 or h == 64: continue
    print(f"k{ks} {n}x{h}x{w} c{c0}+{c1} -> {key[11]}  epi{key[14]}  tile {tile}")
