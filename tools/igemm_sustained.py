"""sustained (seconds-long) timing of one implicit-GEMM shape per tile: bursts run at boost clocks, the DDIM loop does not.
python tools/igemm_sustained.py [scenes] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
n = 9 * scenes
for name, h, c0, c1, co, tiles in (("up1.conv1 2560->1280 @8", 8, 1280, 1280, 1280, (7, 8, 11)), ("up2.conv1 1920->640 @16", 16, 1280, 640, 640, (7, 10, 11)),
                                   ("L1.conv 640->640 @16", 16, 640, 0, 640, (7, 10, 11)), ("L0.conv 320->320 @32", 32, 320, 0, 320, (7, 10, 11))):
    x = torch.randn(n, h, h, c0, device="cuda").to(torch.bfloat16)
    x2 = torch.randn(n, h, h, c1, device="cuda").to(torch.bfloat16) if c1 else None
    w = torch.randn(co, c0 + c1, 3, 3, device="cuda") / (3 * (c0 + c1) ** 0.5)
    pw = ops.pack_weight(w, torch.bfloat16, c_split=c0 if c1 else None)
    flops = 2.0 * n * h * h * co * (c0 + c1) * 9
    res = []
    for tile in tiles:
        f = lambda: ops.conv2d(x, pw, x2=x2, tile=tile, splitk=1)
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter(); it = 0
        while time.perf_counter() - t0 < secs:
            for _ in range(20): f()
            torch.cuda.synchronize(); it += 20
        dt = time.perf_counter() - t0
        res.append(f"t{tile} {flops * it / dt / 1e12:.0f}")
    print(f"{name:28s} " + "  ".join(res), flush=True)
