"""run ONE attention shape a few times (for rocprofv3 --pmc / timing).  python3 tools/attn_one.py [d] [heads] [scenes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv_ldm_amd import ops
d = int(sys.argv[1]) if len(sys.argv) > 1 else 40
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scenes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tok = {40: 1024, 64: 1024, 80: 256, 160: 64}[d]
lens = [5 * tok] * scenes + [4 * tok] * scenes
n, C = sum(lens), heads * d
qkv = torch.randn(n, 3 * C, device="cuda").to(torch.bfloat16)
seg = ops.make_segments(lens)
f = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], heads, d, seg, max(lens))
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 5 * 1e3
flops = 4.0 * sum(l * l for l in lens) * heads * d
print(f"attn d={d} heads={heads} scenes={scenes}: {us:.1f} us  {flops/us/1e6:.1f} TF/s")
