"""time a few implicit-GEMM shapes with the default (rule-based) tile.  python tools/igemm_time.py [scenes] [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
flt = sys.argv[2] if len(sys.argv) > 2 else ""
n = 9 * scenes
SH = [("L0.geglu", n, 32, 320, 2560, 1, True), ("L1.geglu", n, 16, 640, 5120, 1, True), ("L2.geglu", n, 8, 1280, 10240, 1, True),
      ("L0.qkv", n, 32, 320, 960, 1, False), ("L0.conv", n, 32, 320, 320, 3, False), ("L1.conv", n, 16, 640, 640, 3, False)]
out = []
for name, ni, h, c, co, k, geglu in SH:
    if flt not in name:
        continue
    x = torch.randn(ni, h, h, c, device="cuda").to(torch.bfloat16)
    w = torch.randn(co, c, k, k, device="cuda") / (k * c ** 0.5)
    pw = ops.pack_weight(w if k == 3 else w[:, :, 0, 0], torch.bfloat16, geglu=geglu)
    b = torch.randn(co, device="cuda")
    f = lambda: ops.conv2d(x, pw, b, epilogue=2 if geglu else 0, splitk=1)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    out.append(f"{name} {us:.0f}us {2.0*ni*h*h*co*c*k*k/us/1e6:.0f}TF")
print(" | ".join(out))
