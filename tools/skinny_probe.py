"""weight-bound ("skinny") implicit GEMMs of the one-scene step: time per launch over tile x split-K, against the weight-stream floor.
   python3 tools/skinny_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
dt = torch.bfloat16
# (views, h, c_in, c_out, k, geglu): the 4x4 and 8x8 levels of one scene (9 view passes per DDIM step)
SHAPES = [("conv 4x4 1280->1280", 9, 4, 1280, 1280, 3, False), ("conv 8x8 1280->1280", 9, 8, 1280, 1280, 3, False),
          ("ff.geglu 8x8 1280->10240", 9, 8, 1280, 10240, 1, True), ("ff.out 8x8 5120->1280", 9, 8, 5120, 1280, 1, False),
          ("qkv 8x8 1280->3840", 9, 8, 1280, 3840, 1, False), ("to_out 8x8 1280->1280", 9, 8, 1280, 1280, 1, False)]
for name, ni, h, ci, co, k, geglu in SHAPES:
    x = torch.randn(ni, h, h, ci, device="cuda").to(dt)
    w = torch.randn(co, ci, k, k, device="cuda") / (k * ci ** 0.5)
    pw = ops.pack_weight(w if k == 3 else w[:, :, 0, 0], dt, geglu=geglu)
    wbytes = co * ci * k * k * 2
    ref = ops.conv2d(x, pw, epilogue=2 if geglu else 0).float()
    for tile in (16, 17, 18):
        for sk in (1, 4):
            y = ops.conv2d(x, pw, epilogue=2 if geglu else 0, tile=tile, splitk=sk).float()
            err = float((y - ref).norm() / ref.norm())
            assert err < 6e-3, (name, tile, sk, err)
    res = []
    for tile in (0, 2, 4, 16, 17, 18):
        for sk in (0, 1, 2, 4, 8, 16, 32):
            try:
                for _ in range(3):
                    ops.conv2d(x, pw, epilogue=2 if geglu else 0, tile=tile, splitk=sk)
            except Exception as e:
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.conv2d(x, pw, epilogue=2 if geglu else 0, tile=tile, splitk=sk)
            e1.record(); e1.synchronize()
            res.append((e0.elapsed_time(e1) / 30 * 1e3, tile, sk))
    res.sort()
    best = ", ".join(f"t{t}/s{s} {us:.1f}" for us, t, s in res[:5])
    rule = next(us for us, t, s in res if t == 0 and s == 0)
    per = {tt: min(((us, s) for us, t, s in res if t == tt), default=None) for tt in (2, 4, 16, 17, 18)}
    per = " ".join(f"t{tt}:{v[0]:.1f}/s{v[1]}" for tt, v in per.items() if v)
    print(f"{name:28s} M={ni*h*h:4d} W={wbytes/1e6:5.1f} MB floor {wbytes/5e12*1e6:5.1f} us | rule {rule:.1f} | best: {best} | {per}", flush=True)
