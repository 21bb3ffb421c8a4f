"""one tile-15 problem, a few launches over rotating (cold) weight copies: the subject of tools/pmc_skinny.sh
   python3 tools/sk_one.py <conv4|conv8|lin144|lin576|phase4> <cfg> [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
shape, cfg = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dt = torch.bfloat16
S = {"conv4": (9, 4, 1280, 1280, 3), "conv8": (9, 8, 1280, 1280, 3), "lin144": (144, 1, 1280, 1280, 1), "lin576": (576, 1, 1280, 1280, 1),
     "conv4w": (9, 4, 2560, 1280, 3)}[shape]
ni, h, ci, co, k = S
x = torch.randn(ni, h, h, ci, device="cuda").to(dt)
pws = []
for c in range(n):
    w = torch.randn(co, ci, k, k, device="cuda") / (k * ci ** 0.5)
    pw = ops.pack_weight(w if k > 1 else w[:, :, 0, 0], dt)
    pw.skinny()
    pws.append(pw)
tile = cfg if cfg < 0 else (15 | (cfg << 8))
torch.cuda.synchronize()
for i in range(n):
    ops.conv2d(x, pws[i], None, tile=tile if cfg >= 0 else -cfg, splitk=1 if cfg >= 0 else 0)
torch.cuda.synchronize()
