"""one weight-bound implicit GEMM, a fixed (tile, split-K), 200 launches -- the program to put under rocprofv3 --kernel-trace --stats
   python3 tools/skinny_one.py <shape 0..5> <tile> <splitk>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
dt = torch.bfloat16
SHAPES = [(9, 4, 1280, 1280, 3, False), (9, 8, 1280, 1280, 3, False), (9, 8, 1280, 10240, 1, True), (9, 8, 5120, 1280, 1, False),
          (9, 8, 1280, 3840, 1, False), (9, 8, 1280, 1280, 1, False)]
ni, h, ci, co, k, geglu = SHAPES[int(sys.argv[1])]
tile, sk = int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(ni, h, h, ci, device="cuda").to(dt)
w = torch.randn(co, ci, k, k, device="cuda") / (k * ci ** 0.5)
pw = ops.pack_weight(w if k == 3 else w[:, :, 0, 0], dt, geglu=geglu)
for _ in range(200):
    ops.conv2d(x, pw, epilogue=2 if geglu else 0, tile=tile, splitk=sk)
torch.cuda.synchronize()
