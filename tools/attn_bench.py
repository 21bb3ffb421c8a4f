"""Attention kernel timing on the shapes of one fused CFG UNet pass (GPU).   python tools/attn_bench.py [scenes]
Set MVLDM_ATTN_QB=1 to force one query block per wave (A/B against the two-block form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = torch.bfloat16
SHAPES = [  # name, heads, d, q_lens
    ("L0 3-D      8 x 40", 8, 40, [5 * 1024] * b + [4 * 1024] * b),
    ("L0 per-view 8 x 40", 8, 40, [1024] * (9 * b)),
    ("L0 SD self  5 x 64", 5, 64, [1024] * (9 * b)),
    ("L1 3-D      8 x 80", 8, 80, [5 * 256] * b + [4 * 256] * b),
    ("L1 per-view 8 x 80", 8, 80, [256] * (9 * b)),
    ("L1 SD self 10 x 64", 10, 64, [256] * (9 * b)),
    ("L2 3-D     8 x 160", 8, 160, [5 * 64] * b + [4 * 64] * b),
    ("L2 SD self 20 x 64", 20, 64, [64] * (9 * b)),
]
tot = 0.0
for name, heads, d, lens in SHAPES:
    C = heads * d
    M = sum(lens)
    qkv = torch.randn(M, 3 * C, device="cuda").to(dt)
    seg = ops.make_segments(lens)
    fn = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], heads, d, seg, max(lens))
    out = fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 4.0 * sum(l * l for l in lens) * C
    tot += ms
    print(f"{name:22s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s   checksum {float(out.float().abs().mean()):.6f}", flush=True)
print(f"sum {tot:.3f} ms")
