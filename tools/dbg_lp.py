import sys, os
sys.path.insert(0, os.getcwd())
import torch, math
import torch.nn.functional as F
from mv_ldm_amd import ops
torch.manual_seed(0)
for rows, cin, cout, use_res in [(256, 320, 128, False), (256, 320, 128, True), (300, 320, 320, False), (300, 320, 320, True), (5000, 640, 640, True)]:
    x = torch.randn(rows, cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(cout, cin, device="cuda") / math.sqrt(cin)
    b = torch.randn(cout, device="cuda") * 0.1
    res = torch.randn(rows, cout, device="cuda").to(torch.bfloat16) if use_res else None
    pw = ops.pack_weight(w, torch.bfloat16)
    y = ops.linear(x, pw, b, residual=res, tile=12).float()
    ref = F.linear(x.float(), w.to(torch.bfloat16).float(), b) + (res.float() if use_res else 0)
    bad = ~torch.isfinite(y) | ((y - ref).abs() > 0.1)
    print(rows, cin, cout, use_res, "bad", int(bad.sum()), "nan", int((~torch.isfinite(y)).sum()))
    if bad.any():
        r = bad.any(1).nonzero().flatten(); c = bad.any(0).nonzero().flatten()
        print("  rows", r[:20].tolist(), "... n", len(r), " cols", c[:40].tolist(), "... n", len(c))
        i, j = bad.nonzero()[0].tolist()
        print("  first", i, j, y[i, j].item(), ref[i, j].item(), (y[i, j] - ref[i, j]).item(), res[i, j].item() if use_res else None)
