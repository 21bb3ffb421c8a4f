"""EXPERIMENT (wrong results by design): time tile 14 (linear_ws.hip) with activation traffic / stores switched off.  Needs
mv_ldm_amd/csrc/libmvldm_hip_exp.so built by tools/ws_probe.sh.  python tools/ws_probe.py <fake bits: 1 no A traffic, 4 no stores> [lib suffix]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVLDM_WS_FAKE"] = sys.argv[1] if len(sys.argv) > 1 else "0"
import torch
import mv_ldm_amd._lib as L
L.LIB_PATH = L.LIB_PATH.with_name("libmvldm_hip_exp%s.so" % (sys.argv[2] if len(sys.argv) > 2 else ""))
from mv_ldm_amd import ops

n = 9 * 64
SH = [("L0.geglu", n * 1024, 2560, 2, 0), ("L0.qkv", n * 1024, 960, 0, 0), ("L0.to_out", n * 1024, 320, 0, 1), ("L0.proj", n * 1024, 320, 0, 0)]
out = []
for name, rows, nn, epi, res in SH:
    x = torch.randn(rows, 320, device="cuda").to(torch.bfloat16)
    w = torch.randn(nn, 320, device="cuda") / 320 ** 0.5
    pw = ops.pack_weight(w, torch.bfloat16, geglu=epi == 2)
    b = torch.randn(nn, device="cuda")
    r = torch.randn(rows, nn, device="cuda").to(torch.bfloat16) if res else None
    f = lambda: ops.linear(x, pw, b, residual=r, epilogue=epi, tile=14, splitk=1)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    out.append(f"{name} {us:.0f}us {2.0 * rows * 320 * nn / us / 1e6:.0f}TF")
print("fake", os.environ["MVLDM_WS_FAKE"], " | ".join(out), flush=True)
