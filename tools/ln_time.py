"""LayerNorm GB/s at the UNet's shapes.  python tools/ln_time.py [scenes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
n = 9 * (int(sys.argv[1]) if len(sys.argv) > 1 else 64)
out = []
for hw, c in ((1024, 320), (256, 640), (64, 1280)):
    rows = n * hw
    x = torch.randn(rows, c, device="cuda").to(torch.bfloat16)
    g, b = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    f = lambda: ops.layernorm(x, g, b, 1e-5)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    out.append(f"c={c} rows={rows}: {us:.0f} us {4.0 * rows * c / us / 1e3:.0f} GB/s")
print(" | ".join(out))
