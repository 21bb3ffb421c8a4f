"""run one transformer-block Linear with several tiles AND the vendor GEMM (torch -> hipBLASLt), a few launches each: the workload of
tools/gemm_diag.sh (kernel trace + PMC passes).   python3 tools/gemm_diag_run.py <shape> <tiles, e.g. 10,13,19> [scenes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
name, tiles = sys.argv[1], [int(t) for t in sys.argv[2].split(",")]
n = 9 * (int(sys.argv[3]) if len(sys.argv) > 3 else 64)
lvl = int(name[1]); hw, c = [(32, 320), (16, 640), (8, 1280)][lvl]
rows = n * hw * hw
k, nn, epi, res = {"geglu": (c, 8 * c, 2, False), "ff_out": (4 * c, c, 0, True), "qkv": (c, 3 * c, 0, False), "to_out": (c, c, 0, True)}[name[3:]]
x = torch.randn(rows, k, device="cuda").to(torch.bfloat16)
w = torch.randn(nn, k, device="cuda") / k ** 0.5
pw = ops.pack_weight(w, torch.bfloat16, geglu=epi == 2)
b = torch.randn(nn, device="cuda")
r = torch.randn(rows, nn, device="cuda").to(torch.bfloat16) if res else None
wt, bt = w.to(torch.bfloat16), b.to(torch.bfloat16)
for t in tiles:
    for _ in range(5):
        ops.linear(x, pw, b, residual=r, epilogue=epi, tile=t, splitk=1)
    torch.cuda.synchronize()
for _ in range(5):
    torch.nn.functional.linear(x, wt, bt)
torch.cuda.synchronize()
