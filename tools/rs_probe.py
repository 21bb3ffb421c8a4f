"""EXPERIMENT (wrong results by design): time tile 19 (linear_rs.hip) with operand traffic / stores / residual loads switched off, to
see what bounds it.  Needs mv_ldm_amd/csrc/libmvldm_hip_exp_rs.so = the library with linear_rs.hip compiled -DMVLDM_EXPERIMENTS
(tools/rs_probe.sh builds it).  python tools/rs_probe.py <fake bits: 1 no A traffic, 2 no W traffic, 4 no stores, 8 no residual> [lib suffix] [zero]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVLDM_RS_FAKE"] = sys.argv[1] if len(sys.argv) > 1 else "0"
import torch
import mv_ldm_amd._lib as L
L.LIB_PATH = L.LIB_PATH.with_name("libmvldm_hip_exp_rs%s.so" % (sys.argv[2] if len(sys.argv) > 2 else ""))
from mv_ldm_amd import ops
zero = len(sys.argv) > 3 and sys.argv[3] == "zero"

n = 9 * 64
SH = [("L1.geglu", n * 256, 640, 5120, 2, 0), ("L2.geglu", n * 64, 1280, 10240, 2, 0), ("L2.qkv", n * 64, 1280, 3840, 0, 0),
      ("L2.to_out", n * 64, 1280, 1280, 0, 1), ("L2.ff_out", n * 64, 5120, 1280, 0, 1), ("L2.qkv_x4rows", 4 * n * 64, 1280, 3840, 0, 0)]
out = []
for name, rows, k, nn, epi, res in SH:
    x = torch.randn(rows, k, device="cuda").to(torch.bfloat16)
    w = torch.randn(nn, k, device="cuda") / k ** 0.5
    if zero:
        x.zero_(); w.zero_()
    pw = ops.pack_weight(w, torch.bfloat16, geglu=epi == 2)
    b = torch.randn(nn, device="cuda")
    r = torch.randn(rows, nn, device="cuda").to(torch.bfloat16) if res else None
    f = lambda: ops.linear(x, pw, b, residual=r, epilogue=epi, tile=19, splitk=1)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    out.append(f"{name} {us:.0f}us {2.0 * rows * k * nn / us / 1e6:.0f}TF")
print("fake", os.environ["MVLDM_RS_FAKE"], "zero" if zero else "", " | ".join(out), flush=True)
