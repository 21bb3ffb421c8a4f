"""Tile 15 (csrc/skinny.hip) against the tiled kernels on the weight-bound launches of the one-scene step, COLD weights: every launch
of a timing loop reads another copy of the weight (copies total > 2 x the 256 MB Infinity Cache), as inside a DDIM step, where 1.85 GB
of weights pass between two uses of the same layer.
   python3 tools/skinny_bench.py [--json out.json] [--hot] [--views 9]"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mv_ldm_amd import _lib as L
_pre = [a for a in sys.argv if a.startswith('--lib-suffix')]
if '--lib-suffix' in sys.argv:
    L.LIB_PATH = L.LIB_PATH.with_name('libmvldm_hip_exp%s.so' % sys.argv[sys.argv.index('--lib-suffix') + 1])
from mv_ldm_amd import ops
from mv_ldm_amd.plan import Builder

ap = argparse.ArgumentParser()
ap.add_argument("--json", default=None)
ap.add_argument("--hot", action="store_true", help="one weight copy (Infinity-Cache resident)")
ap.add_argument("--views", type=int, default=9)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--only", default=None)
ap.add_argument("--lib-suffix", default=None, help="EXPERIMENT library libmvldm_hip_exp<suffix>.so (tools/sk_probe.sh): results are wrong by design")
ap.add_argument("--cfgs", default=None)
ap.add_argument("--no-tiled", action="store_true")
ap.add_argument("--set", default="low", choices=["low", "mid"], help="low: the 8x8 / 4x4-level launches (default); mid: the 32x32 / 16x16-level Linears (use --hot)")
args = ap.parse_args()
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
V = args.views
# name, images, h_in, c0, c1, c_out, ksize, stride, epilogue, kind
SHAPES = [
    ("L3 conv3x3 1280->1280", V, 4, 1280, 0, 1280, 3, 1, 0, "conv"),
    ("L3 conv3x3 2560->1280", V, 4, 2560, 0, 1280, 3, 1, 0, "conv"),
    ("L2 conv3x3 1280->1280", V, 8, 1280, 0, 1280, 3, 1, 0, "conv"),
    ("L2 conv3x3 2560->1280", V, 8, 2560, 0, 1280, 3, 1, 0, "conv"),
    ("L2 conv3x3 640->1280", V, 8, 640, 0, 1280, 3, 1, 0, "conv"),
    ("L2->L3 down s2 1280", V, 8, 1280, 0, 1280, 3, 2, 0, "conv"),
    ("L3 shortcut 1x1 2560->1280", V, 4, 1280, 1280, 1280, 1, 1, 0, "conv"),
    ("L2 shortcut 1x1 2560->1280", V, 8, 1280, 1280, 1280, 1, 1, 0, "conv"),
    ("L3 up phase 2x2 1280", V, 4, 1280, 0, 1280, 2, 1, 0, "phase"),
    ("L2 up phase 2x2 1280", V, 8, 1280, 0, 1280, 2, 1, 0, "phase"),
    ("L3 linear 1280->1280", V * 16, 1, 1280, 0, 1280, 1, 1, 0, "lin"),
    ("L3 qkv 1280->3840", V * 16, 1, 1280, 0, 3840, 1, 1, 0, "lin"),
    ("L3 ff.out 5120->1280", V * 16, 1, 5120, 0, 1280, 1, 1, 0, "lin"),
    ("L3 geglu 1280->10240", V * 16, 1, 1280, 0, 10240, 1, 1, 2, "lin"),
    ("L2 linear 1280->1280", V * 64, 1, 1280, 0, 1280, 1, 1, 0, "lin"),
    ("L2 qkv 1280->3840", V * 64, 1, 1280, 0, 3840, 1, 1, 0, "lin"),
    ("L2 ff.out 5120->1280", V * 64, 1, 5120, 0, 1280, 1, 1, 0, "lin"),
    ("L2 geglu 1280->10240", V * 64, 1, 1280, 0, 10240, 1, 1, 2, "lin"),
    ("temb linear_1 320->1280", V, 1, 320, 0, 1280, 1, 1, 1, "lin"),
    ("temb linear_2 1280->1280", V, 1, 1280, 0, 1280, 1, 1, 0, "lin"),
    ("time_emb_proj[all] 1280->20480", V, 1, 1280, 0, 20480, 1, 1, 0, "lin"),
]
MID = [
    ("L1 linear 640->640", V * 256, 1, 640, 0, 640, 1, 1, 0, "lin"),
    ("L1 qkv 640->1920", V * 256, 1, 640, 0, 1920, 1, 1, 0, "lin"),
    ("L1 ff.out 2560->640", V * 256, 1, 2560, 0, 640, 1, 1, 0, "lin"),
    ("L1 geglu 640->5120", V * 256, 1, 640, 0, 5120, 1, 1, 2, "lin"),
    ("L0 linear 320->320", V * 1024, 1, 320, 0, 320, 1, 1, 0, "lin"),
    ("L0 qkv 320->960", V * 1024, 1, 320, 0, 960, 1, 1, 0, "lin"),
    ("L0 ff.out 1280->320", V * 1024, 1, 1280, 0, 320, 1, 1, 0, "lin"),
    ("L0 geglu 320->2560", V * 1024, 1, 320, 0, 2560, 1, 1, 2, "lin"),
    ("L1 shortcut 1x1 1920->640", V, 16, 1280, 640, 640, 1, 1, 0, "conv"),
    ("L0 shortcut 1x1 960->320", V, 32, 640, 320, 320, 1, 1, 0, "conv"),
]
if args.set == "mid":
    SHAPES = MID


def timed(fn, n_iter):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n_iter):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n_iter * 1e3


out = []
for name, ni, h, c0, c1, co, k, stride, epi, kind in SHAPES:
    if args.only and args.only not in name:
        continue
    cin = c0 + c1
    wbytes = co * cin * k * k * 2
    copies = 1 if args.hot else max(2, int(640e6 // wbytes) + 1)
    x = torch.randn(ni, h, h, c0, device="cuda").to(dt)
    x2 = torch.randn(ni, h, h, c1, device="cuda").to(dt) if c1 else None
    b = torch.randn(co, device="cuda")
    pws = []
    for c in range(copies):
        if kind == "phase":
            w = torch.randn(co, cin, 2, 2, device="cuda") / (2 * cin ** 0.5)
        else:
            w = torch.randn(co, cin, k, k, device="cuda") / (k * cin ** 0.5)
        pw = ops.pack_weight(w if k > 1 else w[:, :, 0, 0], dt, geglu=epi == 2, c_split=c0 if c1 else None)
        pw.skinny()
        pws.append(pw)
        del w

    ho = h if kind == "phase" else (h + 2 * (k // 2) - k) // stride + 1
    dst = torch.zeros(ni, (2 * h if kind == "phase" else ho), (2 * h if kind == "phase" else ho), co // 2 if epi == 2 else co, dtype=dt, device="cuda")

    def make_plan(tile, sk):
        """`copies` launches, one per weight copy, as one captured plan: graph replay has no host work between the kernels"""
        bld = Builder("cuda", dt, record=True)
        for pw in (pws * (20 // len(pws) + 1) if len(pws) < 20 else pws):
            if kind == "phase":
                bld._phase_conv(x, pw, b, dst, 0, "p", tile=tile, splitk=sk)
            else:
                bld.conv(x, pw, b, x2=x2, stride=stride, epilogue=epi, tile=tile, splitk=sk, out=dst, name="c")
        plan = bld.finalize(autotune=False)
        plan.run()
        torch.cuda.synchronize()
        plan.capture()
        return plan

    def time_plan(plan, reps=5):
        for _ in range(2):
            plan.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            plan.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / (reps * len(plan)) * 1e3

    ref = make_plan(0, 0)
    ref.replay(); torch.cuda.synchronize()
    ref_out = dst.float().clone()
    res = []
    for tile in ((0,) if args.no_tiled else ((0, 1, 2, 3, 4, 5) if args.set == "low" else range(0, 15))):
        for sk in ((0,) if args.no_tiled else (0, 1, 2, 4, 8, 16, 32)):
            try:
                plan = make_plan(tile, sk)
            except Exception:
                continue
            res.append((time_plan(plan), tile, sk))
            del plan
    res.sort()
    sk_res = []
    for cfg in (range(0, 64) if not args.cfgs else [int(c) for c in args.cfgs.split(',')]):
        try:
            plan = make_plan(15 | (cfg << 8), 1)
        except Exception:
            continue
        err = float((dst.float() - ref_out).norm() / ref_out.norm())
        assert err < 8e-3 or args.lib_suffix, (name, cfg, err)
        sk_res.append((time_plan(plan), cfg))
        del plan
    rule = next(us for us, t, s in res if t == 0 and s == 0)
    best_old = res[0]
    best_sk = min(sk_res) if sk_res else (float("nan"), -1)
    floor = wbytes / 6.0e12 * 1e6
    line = {"name": name, "rows": ni * (h // stride) ** 2, "weight_mb": wbytes / 1e6, "floor_us": floor, "rule_us": rule,
            "best_tiled_us": best_old[0], "best_tiled": [best_old[1], best_old[2]],
            "skinny": {str(c): us for us, c in sk_res}, "best_skinny_us": best_sk[0], "best_skinny_cfg": best_sk[1], "copies": copies}
    out.append(line)
    print(f"{name:32s} M={line['rows']:4d} W={wbytes/1e6:6.1f} MB floor {floor:5.1f} | rule {rule:6.1f} best tiled {best_old[0]:6.1f} (t{best_old[1]}/s{best_old[2]}) | "
          f"skinny best {best_sk[0]:6.1f} (cfg {best_sk[1]}, {wbytes/best_sk[0]/1e6:5.2f} TB/s)  all: " + " ".join(f"c{c}:{us:.1f}" for us, c in sorted(sk_res, key=lambda r: r[1])),
          flush=True)
    del pws
    torch.cuda.empty_cache()
if args.json:
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1)
