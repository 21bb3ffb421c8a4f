"""per-op profile of the VAE decode / encode plans (GPU).  python tools/vae_profile.py [n_images]"""
import json, os, re, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mv_ldm_amd
from mv_ldm_amd.vae import AutoencoderKL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.set_grad_enabled(False)
mv_ldm_amd.set_compute_dtype(torch.bfloat16)
with torch.device("cuda"):
    vae = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1")
for p in vae.parameters():
    p.data.normal_(0, 0.02)
for kind, shape in (("decode", (n, 4, 32, 32)), ("encode", (n // 4, 3, 256, 256))):
    x = torch.randn(*shape, device="cuda")
    getattr(vae, kind)(x)
    st = vae._compile(kind, shape[0], shape[2], shape[3], torch.bfloat16)
    plan = st["plan"]
    plan.profile(1)
    ms = plan.profile(3)
    agg = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
    for m, t in zip(plan.meta, ms):
        k = (m.kind, re.sub(r"\d+", "#", m.name).split("/")[-1])
        a = agg[k]; a[0] += t; a[1] += m.flops; a[2] += m.bytes; a[3] += 1
    print(f"{kind}: {shape[0]} images, total {sum(ms):.2f} ms")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {str(k):44s} n={a[3]:3d} {a[0]:8.3f} ms  {a[1]/a[0]/1e9:7.0f} TF/s  {a[2]/a[0]/1e6:7.0f} GB/s")
    if len(sys.argv) > 2:
        json.dump([{"name": m.name, "kind": m.kind, "ms": t, "flops": m.flops, "bytes": m.bytes} for m, t in zip(plan.meta, ms)],
                  open(f"{sys.argv[2]}_{kind}.json", "w"))
