"""time the transformer-block Linears of the UNet per tile candidate (incl. tile 12, the persistent pipelined kernel).
python tools/linear_tiles.py [scenes] [dtype] [tiles, e.g. 0,9,12]  -> one JSON line per shape"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import _lib as L
if '--lib' in sys.argv:      # another build of the library (experiment builds: same-box A/B)
    i = sys.argv.index('--lib'); L.LIB_PATH = L.LIB_PATH.with_name(sys.argv[i + 1]); del sys.argv[i:i + 2]
from mv_ldm_amd import ops

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
n = 9 * scenes
# (name, rows, K, N, epilogue, residual)
SH = []
for lvl, (hw, c) in enumerate([(32, 320), (16, 640), (8, 1280)]):
    rows = n * hw * hw
    SH += [(f"L{lvl}.geglu", rows, c, 8 * c, 2, False), (f"L{lvl}.ff_out", rows, 4 * c, c, 0, True),
           (f"L{lvl}.qkv", rows, c, 3 * c, 0, False), (f"L{lvl}.to_out", rows, c, c, 0, True)]
TILES = tuple(int(t) for t in sys.argv[3].split(",")) if len(sys.argv) > 3 else (0, 2, 3, 9, 10, 12, 13, 14)
for name, rows, k, nn, epi, res in SH:
    x = torch.randn(rows, k, device="cuda").to(dtype)
    w = torch.randn(nn, k, device="cuda") / k ** 0.5
    pw = ops.pack_weight(w, dtype, geglu=epi == 2)
    b = torch.randn(nn, device="cuda")
    r = torch.randn(rows, nn, device="cuda").to(dtype) if res else None
    rec = {"shape": name, "rows": rows, "K": k, "N": nn}
    ref = None
    for tile in TILES:
        try:
            f = lambda: ops.linear(x, pw, b, residual=r, epilogue=epi, tile=tile, splitk=1)
            y = f()
        except Exception as e:          # tile not applicable
            continue
        if ref is None:
            ref = y.float()
        else:
            err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
            rec[f"err{tile}"] = round(err, 4)      # (> 2e-2 = wrong tiles somewhere)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        if f"t{tile}" not in rec or us < rec[f"t{tile}"][0]:          # (a tile listed twice: the faster sample -- the first one of a shape meets a colder clock)
            rec[f"t{tile}"] = [round(us), round(2.0 * rows * k * nn / us / 1e6)]
    if "--blas" in sys.argv:
        # yardstick: the vendor GEMM (hipBLASLt through torch) on the same operands.  Plain GEMM + bias only -- no GEGLU product, no residual
        # add: its number is what the library needs for LESS work than the fused launch above does.
        wt = w.to(dtype)
        f = lambda: torch.nn.functional.linear(x, wt, b.to(dtype))
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        rec["hipblaslt_gemm_bias_only"] = [round(us), round(2.0 * rows * k * nn / us / 1e6)]
        if res or epi == 2:
            # the SAME op as the fused launch, the vendor way: its GEMM + bias, then the elementwise kernel(s) torch needs for the residual add / the
            # GEGLU product (in place where torch allows it)
            if res:
                g = lambda: torch.nn.functional.linear(x, wt, b.to(dtype)).add_(r)
            else:
                def g():
                    y = torch.nn.functional.linear(x, wt, b.to(dtype))
                    return y[:, :nn // 2] * torch.nn.functional.gelu(y[:, nn // 2:])
            g(); torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                g()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            rec["hipblaslt_plus_eltwise_same_op"] = [round(us), round(2.0 * rows * k * nn / us / 1e6)]
    print(json.dumps(rec), flush=True)
