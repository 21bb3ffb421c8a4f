"""Tuning tool (GPU): time the main implicit-GEMM shapes of one UNet pass under tile / ring-depth /
split-K / XCD-partition overrides.   python tools/igemm_sweep.py [--scenes 4] [--quick]"""
import argparse, itertools, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=4)
ap.add_argument("--quick", action="store_true")
ap.add_argument("--out", default=None)
args = ap.parse_args()
n = 9 * args.scenes
dt = torch.bfloat16
SHAPES = [  # name, n_img, h, cin, cin2, cout, ksize, geglu
    ("L0.conv3x3 320->320 @32", n, 32, 320, 0, 320, 3, False),
    ("L1.conv3x3 640->640 @16", n, 16, 640, 0, 640, 3, False),
    ("up1.conv1 2560->1280 @8", n, 8, 1280, 1280, 1280, 3, False),
    ("up0.conv1 2560->1280 @4", n, 4, 1280, 1280, 1280, 3, False),
    ("up3.conv1 960->320 @32", n, 32, 640, 320, 320, 3, False),
    ("L0.geglu 320->2560", n, 32, 320, 0, 2560, 1, True),
    ("L0.qkv 320->960", n, 32, 320, 0, 960, 1, False),
    ("L0.ff_out 1280->320", n, 32, 1280, 0, 320, 1, False),
    ("L2.geglu 1280->10240 @8", n, 8, 1280, 0, 10240, 1, True),
    ("L1.geglu 640->5120 @16", n, 16, 640, 0, 5120, 1, True),
    ("L1.ff_out 2560->640 @16", n, 16, 2560, 0, 640, 1, False),
    ("L2.conv3x3 1280->1280 @8", n, 8, 1280, 0, 1280, 3, False),
    ("up2.conv1 1920->640 @16", n, 16, 1280, 640, 640, 3, False),
    ("L0.to_out 320->320", n, 32, 320, 0, 320, 1, False),
]
if args.quick:
    SHAPES = SHAPES[:3]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


rows = []
for name, ni, h, c0, c1, co, k, geglu in SHAPES:
    x = torch.randn(ni, h, h, c0, device="cuda").to(dt)
    x2 = torch.randn(ni, h, h, c1, device="cuda").to(dt) if c1 else None
    w = torch.randn(co, c0 + c1, k, k, device="cuda") / (k * (c0 + c1) ** 0.5)
    pw = ops.pack_weight(w if k == 3 else w[:, :, 0, 0], dt, geglu=geglu, c_split=c0 if c1 else None)
    flops = 2.0 * ni * h * h * co * (c0 + c1) * k * k
    best = None
    combos = []
    for tile in (1, 2, 3, 6, 7, 8, 9, 10, 11):
        combos.append((tile, 0, 1, 0, 0))   # lean buffer-load loop (default for block-major K), ring depth st
    for tile in (2, 3):
        combos.append((tile, 0, 1, 0, 1))            # register-prefetch loop (bit 12)
    for px in (1, 2, 4, 8):
        combos.append((2, 0, 1, px, 0))
    for tile, stages, sk, px, sync in combos:
        code = tile | (px << 8) | ((sync & 1) << 12)
        try:
            us = timeit(lambda: ops.conv2d(x, pw, x2=x2, epilogue=2 if geglu else 0, tile=code, splitk=sk))
        except Exception as ex:
            us = float("nan")
        rows.append(dict(shape=name, tile=tile, stages=stages, splitk=sk, px=px, sync=sync, us=us, tflops=flops / us / 1e6))
        if best is None or us < best[0]:
            best = (us, tile, stages, sk, px, sync)
    auto = timeit(lambda: ops.conv2d(x, pw, x2=x2, epilogue=2 if geglu else 0))
    print(f"{name:28s} auto {auto:8.1f} us {flops/auto/1e6:7.1f} TF | best {best[0]:8.1f} us {flops/best[0]/1e6:7.1f} TF  tile={best[1]} stages={best[2]} splitk={best[3]} px={best[4]} sync={best[5]}", flush=True)
if args.out:
    json.dump(rows, open(args.out, "w"))
