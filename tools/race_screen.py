"""race screen of the persistent Linears (igemm tiles 12, 13, 14): the same launch repeated under load must give the bit-identical result
every time (the counted-vmcnt rings, the cross-tile prefetch, the asm loads / stores and the loader wave are exactly the kind of code
whose hazards show up as rare wrong tiles).  python tools/race_screen.py [iterations] [tile=12]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
TILE = int(sys.argv[2]) if len(sys.argv) > 2 else 12
torch.manual_seed(0)
bad = 0
CASES = (("L0.geglu", 589824, 320, 2560, 2, False, 0), ("L0.to_out", 589824, 320, 320, 0, True, 0),
         ("L1.qkv", 147456, 640, 1920, 0, False, 0), ("ragged", 70001, 384, 200, 1 if TILE == 12 else 0, TILE != 12, 0),
         ("dual", 147456, 320, 640, 0, False, 320))
if TILE == 14:      # K = 320, one source only
    CASES = (("L0.geglu", 589824, 320, 2560, 2, False, 0), ("L0.to_out", 589824, 320, 320, 0, True, 0), ("L0.qkv", 589824, 320, 960, 0, False, 0),
             ("ragged", 70001, 320, 640, 0, True, 0))
for name, rows, k, n, epi, res, dual in CASES:
    x = torch.randn(rows, k, device="cuda").to(torch.bfloat16)
    x2 = torch.randn(rows, dual, device="cuda").to(torch.bfloat16) if dual else None
    w = torch.randn(n, k + dual, device="cuda") / (k + dual) ** 0.5
    pw = ops.pack_weight(w, torch.bfloat16, geglu=epi == 2, c_split=k if dual else None)
    b = torch.randn(n, device="cuda")
    r = torch.randn(rows, n, device="cuda").to(torch.bfloat16) if res else None
    if dual:
        f = lambda: ops.conv2d(x.view(rows // 1024, 32, 32, k), pw, b, x2=x2.view(rows // 1024, 32, 32, dual), tile=TILE)
    else:
        f = lambda: ops.linear(x, pw, b, residual=r, epilogue=epi, tile=TILE, splitk=1)
    ref = f().clone()
    noise = torch.randn(64 << 20, device="cuda")          # a second stream keeps HBM / L2 busy with unrelated traffic
    side = torch.cuda.Stream()
    mism = 0
    for i in range(iters):
        if i % 4 == 0:
            with torch.cuda.stream(side):
                noise.mul_(1.0001)
        y = f()
        if not torch.equal(y, ref):
            mism += 1
    torch.cuda.synchronize()
    print(f"tile {TILE} {name}: {iters} runs, {mism} mismatching", flush=True)
    bad += mism
sys.exit(1 if bad else 0)
