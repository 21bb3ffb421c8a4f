"""Probe: does running the DDIM loop as TWO half-batch plans on two HIP streams beat one full-batch plan?  (GPU)
One workgroup per CU (the 8-wave conv / Linear tiles own the LDS), so every kernel ends with a partially filled last round and
an exposed epilogue; a second, independent stream of kernels can fill those holes.
   python tools/two_stream_probe.py [scenes=64] [ddim_steps=50]
Prints ms per DDIM step (per `scenes` scenes) for: one plan of B scenes; two plans of B/2 on one stream; two plans of B/2 on two
streams; with MVLDM_PROBE_THREE=1 also three plans of B/3 (rounded) on three streams."""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
_lib.load()
mv_ldm_amd.set_compute_dtype(torch.bfloat16)
with torch.device(dev):
    den = MultiViewUNet(MultiViewUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1", allow_random_init=True), 11, 4)
    vae = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1", allow_random_init=True)
bench.random_init_(den, 1234)
bench.random_init_(vae, 1235)


def make(b, seed):
    pipe = MVLDMPipeline(den, vae, DDIMScheduler(clip_sample=False), SamplerCfg(True, 3.0, N))
    pipe.set_timesteps(N)
    batch = bench.synthetic_batch(b, 1, 4, 256, seed, dev)
    return pipe, batch


def loop(states, streams, reps=2):
    best = 1e9
    for _ in range(reps):
        for pipe, batch in states:
            pipe.prepare(batch)                       # valid inputs again (clocks depend on the data)
        torch.cuda.synchronize()
        sts = [pipe.prepare(batch) for pipe, batch in states]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            for st, s in zip(sts, streams):
                with torch.cuda.stream(s):
                    st["plan"].replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / N * 1e3)
    return best


one = [make(B, 1)]
cur = torch.cuda.current_stream()
t_one = loop(one, [cur])
print(f"one plan of {B} scenes:                 {t_one:8.2f} ms per DDIM step", flush=True)
halves = [make(B // 2, 1), make(B - B // 2, 2)]
t_seq = loop(halves, [cur, cur])
print(f"two plans of {B // 2}, one stream:           {t_seq:8.2f} ms", flush=True)
s = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
t_two = loop(halves, s[:2])
print(f"two plans of {B // 2}, two streams:          {t_two:8.2f} ms   ({t_one / t_two:.3f}x of one plan)", flush=True)
if os.environ.get("MVLDM_PROBE_THREE") == "1":
    k = B // 3
    thirds = [make(k, 1), make(k, 2), make(B - 2 * k, 3)]
    t_three = loop(thirds, s)
    print(f"three plans of ~{k}, three streams:     {t_three:8.2f} ms   ({t_one / t_three:.3f}x of one plan)", flush=True)
