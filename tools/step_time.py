"""ms per DDIM step (graph replay, valid inputs) of the B-scene sampler with a chosen build of the library (GPU).
   python tools/step_time.py [scenes=64] [library file name in mv_ldm_amd/csrc, default libmvldm_hip.so] [reps=3]
Used for same-box A/B of experiment builds (e.g. MVLDM_IGEMM_FAKE=16 with libmvldm_hip_exp.so: streaming stores)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mv_ldm_amd._lib as L

if len(sys.argv) > 2:
    L.LIB_PATH = L.LIB_PATH.with_name(sys.argv[2])
import bench
import mv_ldm_amd
from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg
from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
from mv_ldm_amd.scheduler import DDIMScheduler
from mv_ldm_amd.vae import AutoencoderKL

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
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
batch = bench.synthetic_batch(B, 1, 4, 256, 1, dev)
best = []
for _ in range(reps):
    st = pipe.prepare(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        st["plan"].replay()
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / 50 * 1e3)
print(f"{os.path.basename(str(L.LIB_PATH))} FAKE={os.environ.get('MVLDM_IGEMM_FAKE', '0')} scenes={B}: " + " ".join(f"{b:.2f}" for b in best) + " ms per DDIM step", flush=True)
