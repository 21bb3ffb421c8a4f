"""randomised parity sweep of the 8-wave conv tiles (10: spread issue since round 6; 7, 9, 17 and the rule beside it) against torch's fp32 conv on
the same bf16 / f16 operands: random image counts and map sizes (ragged row tiles), channel counts, 1x1 / 3x3, one or two sources, bias / per-image
row bias / residual / SiLU, each case run twice (bit-identical).  A case a tile refuses is counted, not failed.
python tools/conv_fuzz.py [cases=300] [seed=0]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mv_ldm_amd import ops, _lib as L

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(2)
bad = refused = ran = 0
worst = 0.0
for ci in range(cases):
    tile = rng.choice((10, 10, 10, 7, 9, 17, 0))
    dtype = rng.choice((torch.bfloat16, torch.bfloat16, torch.float16))
    ks = rng.choice((1, 3, 3))
    hw = rng.choice((8, 16, 24, 32, 12))
    n = rng.choice((1, 3, 9, 36, 37, 80, rng.randrange(1, 120)))
    c0 = rng.choice((64, 128, 320, 640, 192))
    dual = rng.random() < 0.25
    c1 = rng.choice((64, 320)) if dual else 0
    co = rng.choice((320, 640, 328, 64, 200, 1280))
    silu = rng.random() < 0.2
    use_rb = rng.random() < 0.4
    use_res = (not silu) and rng.random() < 0.4
    x = torch.randn(n, hw, hw, c0, device="cuda").to(dtype)
    x2 = torch.randn(n, hw, hw, c1, device="cuda").to(dtype) if dual else None
    w = torch.randn(co, c0 + c1, ks, ks, device="cuda") / ((c0 + c1) * ks * ks) ** 0.5
    b = torch.randn(co, device="cuda") if rng.random() < 0.8 else None
    rb = torch.randn(n, co, device="cuda") if use_rb else None
    res = torch.randn(n, hw, hw, co, device="cuda").to(dtype) if use_res else None
    try:
        pw = ops.pack_weight(w, dtype, c_split=c0 if dual else None)
        run = lambda: ops.conv2d(x, pw, b, x2=x2, row_bias=rb, residual=res, epilogue=L.EPI_SILU if silu else L.EPI_NONE, tile=tile, splitk=1)
        y = run()
        y2 = run()
    except L.MvldmError as e:
        refused += 1
        continue
    ran += 1
    xa = torch.cat([x, x2], -1) if dual else x
    ref = F.conv2d(xa.float().permute(0, 3, 1, 2), w.to(dtype).float(), None if b is None else b, padding=ks // 2).permute(0, 2, 3, 1)
    if rb is not None:
        ref = ref + rb[:, None, None, :]
    if silu:
        ref = F.silu(ref)
    if res is not None:
        ref = ref + res.float()
    err = ((y.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()
    tol = 1.2e-2 if dtype == torch.bfloat16 else 2.5e-3
    worst = max(worst, err / tol)
    same = torch.equal(y, y2)
    if not (err < tol and same and torch.isfinite(y.float()).all()):
        bad += 1
        print(f"FAIL case {ci}: tile {tile} {dtype} n {n} {hw}x{hw} c {c0}+{c1} -> {co} k{ks} silu {silu} row_bias {use_rb} res {use_res} bias {b is not None}: "
              f"rel err {err:.2e} (tol {tol:.1e}), repeat identical {same}", flush=True)
print(f"conv_fuzz: {ran} cases run, {refused} refused by the tile's applicability check, {bad} failed; worst error / tolerance {worst:.2f}")
sys.exit(1 if bad else 0)
