"""run ONE implicit-GEMM shape a few times (for rocprofv3 --pmc).  python3 tools/igemm_one.py <shape-idx> [tilecode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
code = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(os.environ.get('MVLDM_ONE_N', '36'))
SH = [(n, 32, 320, 0, 320, 3, False), (n, 8, 1280, 1280, 1280, 3, False), (n, 32, 320, 0, 2560, 1, True), (n, 16, 640, 0, 640, 3, False),
      (n, 32, 1280, 0, 320, 1, False), (n, 32, 320, 0, 320, 1, False), (n, 16, 640, 0, 5120, 1, True)]
ni, h, c0, c1, co, k, geglu = SH[idx]
dt = torch.bfloat16
x = torch.randn(ni, h, h, c0, device="cuda").to(dt)
x2 = torch.randn(ni, h, h, c1, device="cuda").to(dt) if c1 else None
w = torch.randn(co, c0 + c1, k, k, device="cuda") / (k * (c0 + c1) ** 0.5)
pw = ops.pack_weight(w if k == 3 else w[:, :, 0, 0], dt, geglu=geglu, c_split=c0 if c1 else None)
for _ in range(5):
    ops.conv2d(x, pw, x2=x2, epilogue=2 if geglu else 0, tile=code)
torch.cuda.synchronize()
