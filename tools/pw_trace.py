"""EXPERIMENT: where one wave of tile 13 (linear_pw.hip) spends a K-step.  Needs libmvldm_hip_exp_pwt.so (tools/pw_trace.sh: linear_pw.hip
compiled -DMVLDM_PW_TRACE).  Per step four stamps of the shader clock: step start | in front of the counted wait (sub-steps 0-2 issued) |
behind the wait | behind the barrier.  python tools/pw_trace.py <shape> [block] [wave]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mv_ldm_amd._lib as L
L.LIB_PATH = L.LIB_PATH.with_name("libmvldm_hip_exp_%s.so" % (sys.argv[4] if len(sys.argv) > 4 else "pwt"))
from mv_ldm_amd import ops
name = sys.argv[1]
n = 9 * 64
lvl = int(name[1]); hw, c = [(32, 320), (16, 640), (8, 1280)][lvl]
rows = n * hw * hw
k, nn, epi, res = {"geglu": (c, 8 * c, 2, False), "ff_out": (4 * c, c, 0, True), "qkv": (c, 3 * c, 0, False), "to_out": (c, c, 0, True)}[name[3:]]
x = torch.randn(rows, k, device="cuda").to(torch.bfloat16)
w = torch.randn(nn, k, device="cuda") / k ** 0.5
pw = ops.pack_weight(w, torch.bfloat16, geglu=epi == 2)
b = torch.randn(nn, device="cuda")
r = torch.randn(rows, nn, device="cuda").to(torch.bfloat16) if res else None
for blk, wave in ((int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 0), (17, 5)):
    tr = torch.zeros(192, dtype=torch.int32, device="cuda")
    os.environ["MVLDM_PW_TRACE_PTR"] = hex(tr.data_ptr()); os.environ["MVLDM_PW_TRACE_BLK"] = str(blk); os.environ["MVLDM_PW_TRACE_WAVE"] = str(wave)
    for _ in range(3):
        ops.linear(x, pw, b, residual=r, epilogue=epi, tile=13, splitk=1)
    torch.cuda.synchronize()
    t = [v & 0xFFFFFFFF for v in tr.cpu().tolist()]
    t = [v for v in t if v]
    d = [(b - a) & 0xFFFFFFFF for a, b in zip(t, t[1:])]
    print(f"{name} block {blk} wave {wave}: {len(t)} stamps; deltas (cycles) in stamp order [start->prewait, wait, barrier, sub3+next start | E = epilogue stamp follows the last step]")
    print(" ".join(str(v) for v in d))
    big = [v for v in d if v > 6000]
    print("  epilogue-sized gaps (> 6000 cycles):", big, " median step (sum of 4):", sorted(sum(d[i:i + 4]) for i in range(4, min(len(d) - 4, 60), 4))[6] if len(d) > 40 else None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.linear(x, pw, b, residual=r, epilogue=epi, tile=13, splitk=1)
    e1.record(); torch.cuda.synchronize()
    print(f"  {e0.elapsed_time(e1) * 100:.0f} us per launch")
