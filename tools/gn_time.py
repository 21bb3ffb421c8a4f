"""time GroupNorm(+SiLU) at the UNet's shapes.  python tools/gn_time.py [scenes]   (MVLDM_GN_TWOPASS=1: two-launch path)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import _lib as L
if '--lib-suffix' in sys.argv:      # EXPERIMENT library (tools/gn_probe.sh)
    L.LIB_PATH = L.LIB_PATH.with_name('libmvldm_hip_exp%s.so' % sys.argv[sys.argv.index('--lib-suffix') + 1])
from mv_ldm_amd import ops
scenes = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
n = 9 * scenes
out = []
for name, h, c0, c1 in (("L0.320", 32, 320, 0), ("L0.640+320", 32, 640, 320), ("L1.640", 16, 640, 0), ("L1.1280+640", 16, 1280, 640),
                        ("L2.1280", 8, 1280, 0), ("L2.1280+1280", 8, 1280, 1280), ("L3.1280", 4, 1280, 0)):
    x = torch.randn(n, h, h, c0, device="cuda").to(torch.bfloat16)
    x2 = torch.randn(n, h, h, c1, device="cuda").to(torch.bfloat16) if c1 else None
    g, b = torch.randn(c0 + c1, device="cuda"), torch.randn(c0 + c1, device="cuda")
    f = lambda: ops.groupnorm(x, g, b, 32, 1e-5, os.environ.get("GN_SILU", "1") != "0", x2=x2)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    out.append(f"{name} {us:.0f}us {2.0 * n * h * h * (c0 + c1) * 2 / us / 1e3:.0f}GB/s(rw)")
print(os.environ.get("MVLDM_GN_SPAN", "-"), os.environ.get("MVLDM_GN_NTHR", "-"), " | ".join(out))
