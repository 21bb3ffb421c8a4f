"""3x3 convs of the 16x16 / 8x8 levels at B scenes: us per launch and TFLOP/s per tile candidate (7, 9, 10, 11 = pixel halo 256x128,
17 = pixel halo 256x320).   python3 tools/conv_tiles.py [scenes=64] [--json out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import _lib as L
if '--lib' in sys.argv:      # another build of the library (experiment builds: same-box A/B)
    L.LIB_PATH = L.LIB_PATH.with_name(sys.argv[sys.argv.index('--lib') + 1])
from mv_ldm_amd import ops
scenes = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
n = 9 * scenes
SH = [("L0 320->320 @32", 32, 320, 320), ("up3 960->320 @32", 32, 960, 320), ("L1 640->640 @16", 16, 640, 640), ("up2 1920->640 @16", 16, 1920, 640), ("up2 1280->640 @16", 16, 1280, 640), ("up2 960->640 @16", 16, 960, 640),
      ("L1 320->640 @16", 16, 320, 640), ("L2 1280->1280 @8", 8, 1280, 1280), ("up1 2560->1280 @8", 8, 2560, 1280), ("up1 1920->1280 @8", 8, 1920, 1280),
      ("L2 640->1280 @8", 8, 640, 1280), ("L3 1280->1280 @4", 4, 1280, 1280)]
rows = []
for name, h, c, co in SH:
    x = torch.randn(n, h, h, c, device="cuda").to(torch.bfloat16)
    w = torch.randn(co, c, 3, 3, device="cuda") / (3 * c ** 0.5)
    pw = ops.pack_weight(w, torch.bfloat16)
    b = torch.randn(co, device="cuda")
    res = {}
    y7 = ops.conv2d(x, pw, b, splitk=1, tile=7)
    for tile in (7, 9, 10, 11, 17):
        f = lambda: ops.conv2d(x, pw, b, splitk=1, tile=tile)
        assert torch.equal(f(), y7), (name, tile)      # same K order in every tile: bit-identical
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res[tile] = e0.elapsed_time(e1) * 100
    if "--miopen" in sys.argv:
        # yardstick: the vendor convolution (MIOpen through torch, channels-last bf16) on the same operands
        xc = x.permute(0, 3, 1, 2)                     # NCHW view of the NHWC tensor = channels_last
        wc = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        bc = b.to(torch.bfloat16)
        f = lambda: torch.nn.functional.conv2d(xc, wc, bc, padding=1)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res["miopen"] = e0.elapsed_time(e1) * 100
    fl = 2.0 * n * h * h * co * c * 9
    rows.append({"name": name, "us": {str(t): u for t, u in res.items()}, "tflops": {str(t): fl / u / 1e6 for t, u in res.items()}})
    print(f"{name:22s} " + "  ".join(f"{'t' + str(t) if isinstance(t, int) else t}: {u:7.1f}us {fl / u / 1e6:5.0f}TF" for t, u in res.items()), flush=True)
if "--json" in sys.argv:
    json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
