"""soak of the headline workload: N full samples (50 DDIM steps x 64 scenes, VAE encode + decode) of the SAME batch and noise; every sample must be
bit-identical to the first (the plan is a replayed hipGraph of fixed kernels: any difference is a race).  python tools/soak.py [samples=20] [scenes=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mv_ldm_amd
from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg
from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
from mv_ldm_amd.scheduler import DDIMScheduler
from mv_ldm_amd.vae import AutoencoderKL

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda")
torch.set_grad_enabled(False)
mv_ldm_amd.set_compute_dtype(torch.bfloat16)
with torch.device(dev):
    den = MultiViewUNet(MultiViewUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1", allow_random_init=True), 11, 4)
    vae = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1", allow_random_init=True)
bench.random_init_(den, 1234)
bench.random_init_(vae, 1235)
pipe = MVLDMPipeline(den, vae, DDIMScheduler(clip_sample=False), SamplerCfg(True, 3.0, 50))
pipe.set_timesteps(50)
batch = bench.synthetic_batch(scenes, 1, 4, 256, 99, dev)
x_T = torch.randn((scenes, 4, 4, 32, 32), generator=torch.Generator().manual_seed(5))
noise = torch.randn((scenes, 4, 32, 32), generator=torch.Generator().manual_seed(6))
img0, x0 = pipe.sample(batch, x_T=x_T, encode_noise=noise)
assert torch.isfinite(img0).all()
bad, t0 = 0, time.perf_counter()
for i in range(n):
    img, x = pipe.sample(batch, x_T=x_T, encode_noise=noise)
    if not (torch.equal(img, img0) and torch.equal(x, x0)):
        bad += 1
        print(f"sample {i}: differs from the first (max abs diff of the latents {(x.float() - x0.float()).abs().max().item():.3e})", flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"soak: {n} samples of {scenes} scenes ({n * scenes * 4} views, {n * 50} DDIM steps) in {dt:.1f} s = {n * scenes * 4 / dt:.2f} views/s; {bad} differing from the first")
sys.exit(1 if bad else 0)
