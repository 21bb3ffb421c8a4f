import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVLDM_SK_TRACE"] = "1"
import torch
import mv_ldm_amd._lib as L
L.LIB_PATH = L.LIB_PATH.with_name("libmvldm_hip_exp_sk.so")
from mv_ldm_amd import ops
cfg = int(sys.argv[1]); kind = sys.argv[2]
dt = torch.bfloat16
if kind == "conv":
    x = torch.randn(9, 4, 4, 1280, device="cuda").to(dt); w = torch.randn(1280, 1280, 3, 3, device="cuda") / 100
else:
    x = torch.randn(144, 1, 1, 1280, device="cuda").to(dt); w = torch.randn(1280, 1280, device="cuda") / 30
pw = ops.pack_weight(w, dt)
for i in range(3):
    ops.conv2d(x, pw, None, tile=15 | (cfg << 8), splitk=1)
torch.cuda.synchronize()
os.environ["MVLDM_SK_TRACE_DUMP"] = "1"
ops.conv2d(x, pw, None, tile=15 | (cfg << 8), splitk=1)
torch.cuda.synchronize()
