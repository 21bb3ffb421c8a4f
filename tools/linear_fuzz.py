"""randomised parity sweep of the persistent Linear tiles (13, 13 + L2 touch, 19) against a torch fp32 reference of the same op: random row counts
(ragged tiles, fewer rows than one tile, more than the persistent grid walks in one round), K / N over the shapes the kernels accept and the ragged ones
around them, every epilogue (none / GELU / GEGLU / residual / residual in place), one or two sources, bf16 and f16; each case also run twice
(bit-identical).  A case a tile refuses (MVLDM_ERR_ARG) is counted, not failed.  python tools/linear_fuzz.py [cases=300] [seed=0]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mv_ldm_amd import ops, _lib as L

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(1)
TILES = (13, 13, 13 | (1 << 13), 13 | (1 << 14), 19)        # 13: spread issue (default), bit 13: + L2 touch, bit 14: burst issue (A/B)
bad = refused = ran = 0
worst = 0.0
for ci in range(cases):
    tile = rng.choice(TILES)
    dtype = rng.choice((torch.bfloat16, torch.bfloat16, torch.float16))
    k = rng.choice((256, 320, 384, 512, 640, 768, 1280, 2560)) if (tile & 0xFFF) == 19 else rng.choice((320, 384, 448, 640, 1280, 1920, 2560))
    dual = rng.random() < 0.2
    k2 = rng.choice((320, 640)) if dual else 0
    epi = rng.choice((L.EPI_NONE, L.EPI_NONE, L.EPI_NONE, L.EPI_GEGLU, L.EPI_GEGLU, L.EPI_GELU))
    n = rng.choice((320, 640, 960, 1280, 1920, 2560, 200, 264, 72, 328, 1000))
    if epi == L.EPI_GEGLU:
        n = rng.choice((2560, 5120, 640, 1280, 10240, 128, 192))
    rows = rng.choice((1, 37, 255, 256, 257, 1000, 4096, 5000, 36864, 70001, rng.randrange(1, 150000)))
    res = epi == L.EPI_NONE and rng.random() < 0.5
    inplace = res and rng.random() < 0.3
    n_dst = n // 2 if epi == L.EPI_GEGLU else n
    x = torch.randn(rows, k, device="cuda").to(dtype)
    x2 = torch.randn(rows, k2, device="cuda").to(dtype) if dual else None
    w = torch.randn(n, k + k2, device="cuda") / (k + k2) ** 0.5
    b = torch.randn(n, device="cuda") if rng.random() < 0.8 else None
    r = torch.randn(rows, n_dst, device="cuda").to(dtype) if res else None
    try:
        pw = ops.pack_weight(w, dtype, geglu=epi == L.EPI_GEGLU, c_split=k if dual else None)

        def run():
            rr = r.clone() if inplace else r
            if dual:
                return ops.conv2d(x.view(rows, 1, 1, k), pw, b, x2=x2.view(rows, 1, 1, k2), residual=None if rr is None else rr.view(rows, 1, 1, -1),
                                  epilogue=epi, tile=tile, splitk=1, out=rr if inplace else None).view(rows, -1)
            return ops.linear(x, pw, b, residual=rr, epilogue=epi, tile=tile, splitk=1, out=rr if inplace else None)
        y = run()
        y2 = run()
    except L.MvldmError as e:          # the tile's applicability check (argument error with a message): counted, not failed
        refused += 1
        if refused <= 8:
            print(f"refused: tile {tile} rows {rows} k {k}+{k2} n {n} epi {epi} res {res}: {str(e)[:110]}", flush=True)
        continue
    ran += 1
    xa = torch.cat([x, x2], 1).float() if dual else x.float()
    ref = xa @ w.to(dtype).float().t()
    if b is not None:
        ref = ref + b
    if epi == L.EPI_GELU:
        ref = F.gelu(ref)
    elif epi == L.EPI_GEGLU:
        ref = ref[:, :n_dst] * F.gelu(ref[:, n_dst:])
    if res:
        ref = ref + r.float()
    err = ((y.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()
    tol = 1.2e-2 if dtype == torch.bfloat16 else 2.5e-3
    worst = max(worst, err / tol)
    same = torch.equal(y, y2)
    if not (err < tol and same and torch.isfinite(y).all()):
        bad += 1
        print(f"FAIL case {ci}: tile {tile} {dtype} rows {rows} k {k}+{k2} n {n} epi {epi} res {res} inplace {inplace} bias {b is not None}: "
              f"rel err {err:.2e} (tol {tol:.1e}), repeat identical {same}", flush=True)
print(f"linear_fuzz: {ran} cases run, {refused} refused by the tile's applicability check, {bad} failed; worst error / tolerance {worst:.2f}")
sys.exit(1 if bad else 0)
