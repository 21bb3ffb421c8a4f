"""Weight-gradient kernel timing on the shapes of the training window plan (32 images = 2 micro-batches of 4 scenes x 4 views) (GPU).
   MVLDM_WGRAD_WIDE=0|1|2 python tools/wgrad_bench.py     (0: register-staged kernel only, 1: wide LDS-DMA form where it applies, 2: 3x3 only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops

dt = torch.bfloat16
n_img = 32
SHAPES = [  # name, ksize, h, c_in, n_out
    ("L0 conv 320->320", 3, 32, 320, 320), ("L0 conv 640->320", 3, 32, 640, 320), ("L0 conv 960->320", 3, 32, 960, 320),
    ("L1 conv 640->640", 3, 16, 640, 640), ("L1 conv 1920->640", 3, 16, 1920, 640), ("L2 conv 1280->1280", 3, 8, 1280, 1280),
    ("L2 conv 2560->1280", 3, 8, 2560, 1280), ("L3 conv 1280->1280", 3, 4, 1280, 1280),
    ("L0 to_out 320x320", 1, 32, 320, 320), ("L0 qkv 960x320", 1, 32, 320, 960), ("L0 geglu 2560x320", 1, 32, 320, 2560),
    ("L0 ff.out 320x1280", 1, 32, 1280, 320), ("L1 geglu 5120x640", 1, 16, 640, 5120), ("L2 geglu 10240x1280", 1, 8, 1280, 10240),
]
tot = 0.0
for name, k, h, c, n in SHAPES:
    x = torch.randn(n_img, h, h, c, device="cuda").to(dt)
    dy = torch.randn(n_img, h, h, n, device="cuda").to(dt)
    grad = torch.zeros(n, c, k, k, device="cuda") if k == 3 else torch.zeros(n, c, device="cuda")
    f = lambda: ops.conv_wgrad(x, dy, grad, ksize=k)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * n_img * h * h * n * c * k * k
    tot += us
    print(f"{name:24s} {us:8.1f} us {fl / us / 1e6:7.0f} TFLOP/s  checksum {float(grad.abs().mean()):.5f}", flush=True)
print(f"WIDE={os.environ.get('MVLDM_WGRAD_WIDE', '1')} sum {tot:.0f} us")
