"""Numerical study (CPU): error of Winograd F(2x2, 3x3) with bf16-rounded transformed operands and fp32 accumulation,
against the direct bf16 conv (bf16 inputs/weights, fp32 accumulate) and fp64, at UNet-like sizes.
python tools/winograd_study.py"""
import torch
import torch.nn.functional as F
torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
bf = lambda t: t.to(torch.bfloat16).to(torch.float64)


def winograd(x, w, round_fn):
    n, c, h, wd = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                        # [n, c, h/2, w/2, 4, 4]
    V = round_fn(torch.einsum("ij,ncxyjk,lk->ncxyil", BT, tiles, BT))   # input transform, rounded to the GEMM dtype
    U = round_fn(torch.einsum("ij,ocjk,lk->ocil", G, w, G))             # weight transform
    M = torch.einsum("ncxyil,ocil->noxyil", V, U)                       # 16 GEMMs over c (fp32-accumulate idealised as exact)
    Y = torch.einsum("ij,noxyjk,lk->noxyil", AT, M, AT)                 # [n, o, h/2, w/2, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], h, wd)


for c, h in ((320, 32), (640, 16), (1280, 8)):
    x = bf(torch.randn(2, c, h, h, dtype=torch.float64))
    w = bf(torch.randn(64, c, 3, 3, dtype=torch.float64) / (3 * c ** 0.5))
    ref = F.conv2d(x, w, padding=1)
    exact = winograd(x, w, lambda t: t)
    wb = winograd(x, w, bf)
    rel = lambda a: float((a - ref).norm() / ref.norm())
    out_round = float((bf(ref) - ref).norm() / ref.norm())
    print(f"C={c} @{h}: winograd fp64 {rel(exact):.1e} | transformed operands rounded to bf16 {rel(wb):.2e} | "
          f"(rounding the direct result to bf16 alone: {out_round:.2e})")
