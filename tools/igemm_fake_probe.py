"""EXPERIMENT (wrong results by design): time the transformer-block Linears with parts of the implicit-GEMM kernel switched off, to
see what bounds them.  MVLDM_IGEMM_FAKE bits: 1 = activation pieces out of range (zeros, no L2 traffic), 2 = same for the weight,
4 = no global stores / residual loads in the epilogue, 8 = no epilogue at all.  Needs mv_ldm_amd/csrc/libmvldm_hip_exp.so =
the library with igemm.hip compiled -DMVLDM_EXPERIMENTS (tools/igemm_fake_probe.sh builds it here).
python tools/igemm_fake_probe.py <fake bits> [tiles, e.g. 10,9]  ->  one line per shape"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVLDM_IGEMM_FAKE"] = sys.argv[1] if len(sys.argv) > 1 else "0"
import torch
import mv_ldm_amd._lib as L
L.LIB_PATH = L.LIB_PATH.with_name("libmvldm_hip_exp.so")
from mv_ldm_amd import ops

tiles = tuple(int(t) for t in sys.argv[2].split(",")) if len(sys.argv) > 2 else (10, 9)
n = 9 * 64
SH = [("L0.qkv", n * 1024, 320, 960, 0, False), ("L0.to_out", n * 1024, 320, 320, 0, True), ("L0.ff_out", n * 1024, 1280, 320, 0, True),
      ("L0.geglu", n * 1024, 320, 2560, 2, False), ("L1.qkv", n * 256, 640, 1920, 0, False), ("L1.geglu", n * 256, 640, 5120, 2, False),
      ("L2.geglu", n * 64, 1280, 10240, 2, False)]
out = []
for name, rows, k, nn, epi, res in SH:
    x = torch.randn(rows, k, device="cuda").to(torch.bfloat16)
    w = torch.randn(nn, k, device="cuda") / k ** 0.5
    pw = ops.pack_weight(w, torch.bfloat16, geglu=epi == 2)
    b = torch.randn(nn, device="cuda")
    r = torch.randn(rows, nn, device="cuda").to(torch.bfloat16) if res else None
    for tile in tiles:
        if tile == 10 and epi == 2:
            continue
        f = lambda: ops.linear(x, pw, b, residual=r, epilogue=epi, tile=tile, splitk=1)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        out.append(f"{name}/t{tile} {us:.0f}us {2.0 * rows * k * nn / us / 1e6:.0f}TF")
print(f"fake={os.environ['MVLDM_IGEMM_FAKE']}: " + "  ".join(out), flush=True)
