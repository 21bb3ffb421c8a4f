"""debug aid: tile 18 against tile 2 over epilogue / geometry configurations on block-major shapes.  python3 tools/tile18_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
dt = torch.bfloat16
g = torch.Generator().manual_seed(3)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale)
for name, n, h, cin, cout, k, kw in [
    ("conv3 64->64", 2, 16, 64, 64, 3, {}),
    ("conv3 64->128 bias+res", 2, 16, 64, 128, 3, dict(bias=True, res=True)),
    ("conv3 64->128 rowbias", 3, 8, 64, 128, 3, dict(bias=True, rowbias=True)),
    ("conv3 128->128 stride2", 2, 16, 128, 128, 3, dict(stride=2, bias=True)),
    ("conv3 64->64 stride2 h=8", 3, 8, 64, 64, 3, dict(stride=2)),
    ("conv1 64->64 res", 2, 16, 64, 64, 1, dict(res=True)),
    ("linear 128->192 silu", 1, 20, 128, 192, 1, dict(bias=True, epi=1)),
    ("linear 64->64 rows=9", 1, 3, 64, 64, 1, dict(bias=True)),
    ("geglu 64->512", 1, 20, 64, 512, 1, dict(bias=True, epi=2, geglu=True)),
    ("conv3 64->64 splitk2", 2, 16, 64, 64, 3, dict(splitk=2, bias=True, res=True)),
    ("conv3 two-source 64+64->64", 2, 16, 64, 64, 3, dict(c1=64, bias=True)),
    ("conv1 two-source 128+64->128", 2, 8, 128, 128, 1, dict(c1=64, bias=True)),
    ("conv3 64->64 tiny 4x4", 3, 4, 64, 64, 3, dict(bias=True)),
    ("conv3 64->64 2x2", 3, 2, 64, 64, 3, dict(bias=True)),
    ("linear 64->64 f32 out", 1, 20, 64, 64, 1, dict(f32out=True)),
    ("linear 64->192 slice dst", 1, 20, 64, 64, 1, dict(slice=True)),
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
    ys = {}
    for tile in (2, 18):
        kwargs = dict(x2=x2, stride=stride, residual=res, row_bias=rb, epilogue=kw.get("epi", 0), tile=tile, splitk=kw.get("splitk", 1))
        if kw.get("f32out"):
            kwargs["out_dtype"] = torch.float32
        if kw.get("slice"):
            big = torch.zeros(n, ho, ho, 192, dtype=dt, device="cuda")
            kwargs["out"] = big[..., 64:128]
        try:
            y = ops.conv2d(x, pw, b, **kwargs)
        except Exception as e:
            print(name, tile, "refused:", str(e)[:100]); continue
        ys[tile] = y.float().cpu()
    if len(ys) == 2:
        e = float((ys[18] - ys[2]).norm() / ys[2].norm())
        print(f"{name:36s} rel diff {e:.3e}" + ("   <-----" if e > 1e-2 or e != e else ""))
