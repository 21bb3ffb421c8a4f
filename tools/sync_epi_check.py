"""debug aid: outputs of 16-bit implicit GEMMs on the register-prefetch loop (channel counts that are not multiples of 64) for a set of
epilogue configurations; run once with MVLDM_IGEMM_NOSTAGE=1 (per-element epilogue) and once without, then compare.
   python3 tools/sync_epi_check.py <out.pt> [compare-with.pt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
dt = torch.bfloat16
g = torch.Generator().manual_seed(3)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale)
out = {}
for name, n, h, cin, cout, k, kw in [
    ("conv3 32->64", 2, 16, 32, 64, 3, {}),
    ("conv3 32->64 bias+res", 2, 16, 32, 64, 3, dict(bias=True, res=True)),
    ("conv3 32->96 rowbias", 3, 8, 32, 96, 3, dict(bias=True, rowbias=True)),
    ("conv3 16->128 (conv_in like)", 2, 32, 16, 128, 3, dict(bias=True)),
    ("conv1 32->32 res", 2, 16, 32, 32, 1, dict(res=True)),
    ("linear 96->192 silu", 1, 20, 96, 192, 1, dict(bias=True, epi=1)),
    ("geglu 32->256", 1, 20, 32, 256, 1, dict(bias=True, epi=2, geglu=True)),
    ("conv3 32->64 stride2", 2, 16, 32, 64, 3, dict(stride=2)),
    ("conv3 32->64 splitk2", 2, 16, 32, 64, 3, dict(splitk=2, bias=True, res=True)),
    ("conv3 two-source 32+16->64", 2, 16, 32, 64, 3, dict(c1=16, bias=True)),
]:
    c1 = kw.get("c1", 0)
    x = rnd(n, h, h, cin).to(dt).cuda()
    x2 = rnd(n, h, h, c1).to(dt).cuda() if c1 else None
    w = rnd(cout, cin + c1, k, k, scale=1.0 / (k * (cin + c1) ** 0.5)).cuda()
    pw = ops.pack_weight(w if k == 3 else w[:, :, 0, 0], dt, geglu=kw.get("geglu", False), c_split=cin if c1 else None)
    stride = kw.get("stride", 1)
    ho = (h + 2 * (k // 2) - k) // stride + 1
    nd = cout // 2 if kw.get("geglu") else cout
    b = rnd(cout).cuda() * 0.1 if kw.get("bias") else None
    res = rnd(n, ho, ho, nd).to(dt).cuda() if kw.get("res") else None
    rb = rnd(n, nd).cuda() if kw.get("rowbias") else None
    for tile in (0, 1, 2, 3, 4, 5):
        try:
            y = ops.conv2d(x, pw, b, x2=x2, stride=stride, residual=res, row_bias=rb, epilogue=kw.get("epi", 0), tile=tile, splitk=kw.get("splitk", 1))
        except Exception as e:
            print(name, tile, "refused:", str(e)[:80]); continue
        out[f"{name}/t{tile}"] = y.float().cpu()
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for k_, v in out.items():
        e = float((v - ref[k_]).norm() / ref[k_].norm())
        print(f"{k_:44s} rel diff {e:.3e}" + ("   <-----" if e > 1e-2 else ""))
