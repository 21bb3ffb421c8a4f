"""GPU parity of the TRAINING kernels, one by one through the C ABI, against torch.autograd on the CPU in fp64 of the
same op built from plain torch primitives (the ops the oracle's blocks are made of), from the SAME dtype-rounded inputs.
Tolerances (relative L2): f32 2e-5 (fp32 MFMA accumulation), f16 2e-3, bf16 1.2e-2 (inputs AND upstream gradients are
stored in the 16-bit type; weight gradients are fp32 sums of 16-bit products).  Optimizer arithmetic vs torch.optim.AdamW /
clip_grad_norm_ to fp32 round-off."""
import math

import pytest
import torch
import torch.nn.functional as F

GRAD_ENABLED = True       # tests/conftest.py::_grad_mode: torch references are differentiated here
pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.float16: 2e-3, torch.bfloat16: 1.2e-2}
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
IDS = ["f32", "bf16", "f16"]


@pytest.fixture(scope="module")
def ops():
    from mv_ldm_amd import ops as O
    from mv_ldm_amd import _lib as L
    L.load()
    return O


def G(seed):
    return torch.Generator().manual_seed(seed)


def rnd(shape, seed, dtype, scale=1.0):
    return (torch.randn(shape, generator=G(seed)) * scale).to(dtype).float()


def relerr(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def nhwc(t, dtype):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------ conv / linear
CONVS = [  # name, n, c_in, c_out, h, ksize, stride, upsample
    ("3x3", 3, 64, 128, 8, 3, 1, False), ("3x3_wide", 2, 128, 64, 12, 3, 1, False), ("1x1", 2, 64, 192, 8, 1, 1, False),
    ("3x3_s2", 2, 64, 64, 8, 3, 2, False), ("3x3_up", 2, 64, 64, 4, 3, 1, True), ("3x3_odd", 2, 24, 40, 6, 3, 1, False)]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", CONVS, ids=[c[0] for c in CONVS])
def test_conv_data_and_weight_gradients(ops, case, dtype):
    _, n, ci, co, h, ks, stride, up = case
    x, w = rnd((n, ci, h, h), 1, dtype), rnd((co, ci, ks, ks), 2, torch.float32, 1 / math.sqrt(ci * ks * ks))
    xd, wd = x.double().requires_grad_(), w.to(dtype).double().requires_grad_()
    xin = F.interpolate(xd, scale_factor=2, mode="nearest") if up else xd
    y = F.conv2d(xin, wd, None, stride, ks // 2)
    dy = rnd(tuple(y.shape), 3, dtype)
    gx, gw = torch.autograd.grad(y, (xd, wd), dy.double())
    dyd = nhwc(dy, dtype)
    # data gradient: the forward implicit GEMM on the transposed / flipped pack
    pw = ops.pack_weight_t(w.cuda().contiguous(), dtype)
    if stride == 2:
        dx = ops.conv2d(ops.zero_insert2x(dyd), pw)                       # zero insertion, then the stride-1 conv
    elif up:
        dx = ops.pool2x2_sum(ops.conv2d(dyd, pw))                         # gradient at the upsampled size, then 2x2 sums
    else:
        dx = ops.conv2d(dyd, pw)
    assert relerr(nchw(dx), gx) < TOL[dtype] * (2 if up else 1)
    # weight gradient (fp32, PyTorch layout), then accumulated a second time
    grad = torch.zeros(co, ci, ks, ks, device="cuda") if ks > 1 else torch.zeros(co, ci, device="cuda")
    ops.conv_wgrad(nhwc(x, dtype), dyd, grad, ksize=ks, stride=stride, upsample=up)
    gwr = gw if ks > 1 else gw.reshape(co, ci)
    assert relerr(grad, gwr) < TOL[dtype]
    ops.conv_wgrad(nhwc(x, dtype), dyd, grad, ksize=ks, stride=stride, upsample=up, accumulate=True)
    assert relerr(grad, 2 * gwr) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_skip_concat_conv_and_padded_conv_in_gradients(ops, dtype):
    """two-source (skip concat) conv: two data gradients from row subsets of the transposed pack, one weight gradient; and
    conv_in's 11 -> 16 channel padding dropped from the weight gradient; conv_out's 4 (-> 8) output channels"""
    n, c0, c1, co, h = 2, 64, 128, 64, 8
    a, b = rnd((n, c0, h, h), 4, dtype), rnd((n, c1, h, h), 5, dtype)
    w = rnd((co, c0 + c1, 3, 3), 6, torch.float32, 1 / math.sqrt(9 * (c0 + c1)))
    ad, bd, wd = a.double().requires_grad_(), b.double().requires_grad_(), w.to(dtype).double().requires_grad_()
    y = F.conv2d(torch.cat([ad, bd], 1), wd, None, 1, 1)
    dy = rnd(tuple(y.shape), 7, dtype)
    ga, gb, gw = torch.autograd.grad(y, (ad, bd, wd), dy.double())
    dyd = nhwc(dy, dtype)
    wc = w.cuda().contiguous()
    da = ops.conv2d(dyd, ops.pack_weight_t(wc, dtype, 0, c0))
    db = ops.conv2d(dyd, ops.pack_weight_t(wc, dtype, c0, c1))
    assert relerr(nchw(da), ga) < TOL[dtype] and relerr(nchw(db), gb) < TOL[dtype]
    grad = torch.zeros(co, c0 + c1, 3, 3, device="cuda")
    ops.conv_wgrad(nhwc(a, dtype), dyd, grad, ksize=3, x2=nhwc(b, dtype))
    assert relerr(grad, gw) < TOL[dtype]
    # conv_in: 11 real input channels in a 16-channel buffer
    x11 = rnd((n, 11, h, h), 8, dtype)
    w11 = rnd((64, 11, 3, 3), 9, torch.float32, 0.1)
    wd = w11.to(dtype).double().requires_grad_()
    y = F.conv2d(x11.double(), wd, None, 1, 1)
    dy = rnd(tuple(y.shape), 10, dtype)
    (gw,) = torch.autograd.grad(y, wd, dy.double())
    x16 = torch.zeros(n, h, h, 16, dtype=dtype, device="cuda")
    x16[..., :11] = nhwc(x11, dtype)
    grad = torch.zeros(64, 11, 3, 3, device="cuda")
    ops.conv_wgrad(x16, nhwc(dy, dtype), grad, ksize=3, c_in=11)
    assert relerr(grad, gw) < TOL[dtype]
    # conv_out: 4 output channels (gradient buffer padded to 8 columns)
    x = rnd((n, 64, h, h), 11, dtype)
    w4 = rnd((4, 64, 3, 3), 12, torch.float32, 0.05)
    xd, wd = x.double().requires_grad_(), w4.to(dtype).double().requires_grad_()
    y = F.conv2d(xd, wd, None, 1, 1)
    dy = rnd(tuple(y.shape), 13, dtype)
    gx, gw = torch.autograd.grad(y, (xd, wd), dy.double())
    dy8 = torch.zeros(n, h, h, 4 if dtype == torch.float32 else 8, dtype=dtype, device="cuda")      # padded to a 16-byte chunk
    dy8[..., :4] = nhwc(dy, dtype)
    dx = ops.conv2d(dy8, ops.pack_weight_t(w4.cuda().contiguous(), dtype))
    assert relerr(nchw(dx), gx) < TOL[dtype]
    grad = torch.zeros(4, 64, 3, 3, device="cuda")
    ops.conv_wgrad(nhwc(x, dtype), dy8, grad, ksize=3, n_out=4)
    assert relerr(grad, gw) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_linear_gradients_column_slices_and_bias_sums(ops, dtype):
    """a fused projection [rows, 3C]: weight gradients of column slices (fused QKV), total and per-image column sums"""
    rows, c, n3 = 300, 128, 3 * 64
    x, w = rnd((rows, c), 14, dtype), rnd((n3, c), 15, torch.float32, 1 / math.sqrt(c))
    xd, wd = x.double().requires_grad_(), w.to(dtype).double().requires_grad_()
    y = xd @ wd.t()
    dy = rnd((rows, n3), 16, dtype)
    gx, gw = torch.autograd.grad(y, (xd, wd), dy.double())
    dyd, xg = dy.to(dtype).cuda(), x.to(dtype).cuda()
    dx = ops.linear(dyd, ops.pack_weight_t(w.cuda().contiguous(), dtype))
    assert relerr(dx.float().cpu(), gx) < TOL[dtype]
    grad = torch.zeros(n3, c, device="cuda")
    for i in range(3):      # three 64-row slices, each from a column slice of dy
        ops.conv_wgrad(xg.view(rows, 1, 1, c), dyd[:, 64 * i:64 * (i + 1)], grad[64 * i:64 * (i + 1)], ksize=1)
    assert relerr(grad, gw) < TOL[dtype]
    tot = torch.full((n3,), 1.0, device="cuda")
    ops.colsum(dyd, tot, accumulate=True)
    assert relerr(tot, dy.double().sum(0) + 1.0) < max(TOL[dtype], 1e-5)
    per = torch.zeros(4, 256, device="cuda")                 # 4 images of 75 rows, destination row stride 256
    ops.colsum(dyd, per, rows_per_seg=75, per_seg=True, n=n3)
    assert relerr(per[:, :n3], dy.double().view(4, 75, n3).sum(1)) < max(TOL[dtype], 1e-5)


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("silu", [False, True], ids=["plain", "silu"])
def test_groupnorm_backward(ops, dtype, silu):
    n, c0, c1, h, groups = 3, 64, 32, 12, 32
    for dual in (False, True):
        a, b = rnd((n, c0, h, h), 17, dtype, 1.5), rnd((n, c1, h, h), 18, dtype)
        c = c0 + (c1 if dual else 0)
        gamma, beta = 1 + 0.2 * torch.randn(c, generator=G(19)), 0.1 * torch.randn(c, generator=G(20))
        ad, bd = a.double().requires_grad_(), b.double().requires_grad_()
        gd, btd = gamma.double().requires_grad_(), beta.double().requires_grad_()
        xin = torch.cat([ad, bd], 1) if dual else ad
        y = F.group_norm(xin, groups, gd, btd, 1e-5)
        y = F.silu(y) if silu else y
        dy = rnd(tuple(y.shape), 21, dtype)
        grads = torch.autograd.grad(y, (ad, bd, gd, btd) if dual else (ad, gd, btd), dy.double())
        xa, xb = nhwc(a, dtype), (nhwc(b, dtype) if dual else None)
        stats = torch.zeros(n, groups, 2, device="cuda")
        ops.groupnorm(xa, gamma.cuda(), beta.cuda(), groups, 1e-5, silu, x2=xb, stats_out=stats)
        dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        dx, dx2 = ops.groupnorm_bwd(xa, nhwc(dy, dtype), gamma.cuda(), beta.cuda(), stats, dg, db, groups, silu, x2=xb)
        tol = TOL[dtype] * 2
        assert relerr(nchw(dx), grads[0]) < tol, (dual, "dx")
        if dual:
            assert relerr(nchw(dx2), grads[1]) < tol
        assert relerr(dg, grads[-2]) < tol and relerr(db, grads[-1]) < tol, (dual, "params")
        # parameter gradients accumulate
        ops.groupnorm_bwd(xa, nhwc(dy, dtype), gamma.cuda(), beta.cuda(), stats, dg, db, groups, silu, x2=xb)
        assert relerr(dg, 2 * grads[-2]) < tol


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("c", [320, 640, 1280])
def test_layernorm_backward(ops, dtype, c):
    rows = 333
    x = rnd((rows, c), 22, dtype, 2.0)
    gamma, beta = 1 + 0.2 * torch.randn(c, generator=G(23)), 0.1 * torch.randn(c, generator=G(24))
    xd, gd, bd = x.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    y = F.layer_norm(xd, (c,), gd, bd, 1e-5)
    dy = rnd((rows, c), 25, dtype)
    gx, gg, gb = torch.autograd.grad(y, (xd, gd, bd), dy.double())
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    dx = ops.layernorm_bwd(x.to(dtype).cuda(), dy.to(dtype).cuda(), gamma.cuda(), dg, db)
    assert relerr(dx.float().cpu(), gx) < TOL[dtype] * 2
    assert relerr(dg, gg) < TOL[dtype] * 2 and relerr(db, gb) < TOL[dtype] * 2


# ------------------------------------------------------------------------------------------------ attention
ATTN = [("d40_3d", 8, 40, [(5 * 64, 5 * 64), (4 * 64, 4 * 64)]), ("d64_sd", 5, 64, [(256, 256)] * 3), ("d80", 8, 80, [(200, 200), (77, 77)]),
        ("d160", 8, 160, [(80, 80), (64, 64)]), ("d16_ragged", 2, 16, [(130, 70), (1, 1), (65, 129)])]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", ATTN, ids=[c[0] for c in ATTN])
def test_attention_backward(ops, case, dtype):
    """q/k/v as column slices of one fused [tokens, 3C] projection when the lengths allow it; segments of unequal length"""
    _, heads, d, segs = case
    C_ = heads * d
    q_lens, kv_lens = [s[0] for s in segs], [s[1] for s in segs]
    nq, nk = sum(q_lens), sum(kv_lens)
    q, k, v = rnd((nq, C_), 26, dtype), rnd((nk, C_), 27, dtype), rnd((nk, C_), 28, dtype)
    dout = rnd((nq, C_), 29, dtype)
    qd, kd, vd = (t.double().requires_grad_() for t in (q, k, v))
    outs, q0, k0 = [], 0, 0
    for ql, kl in segs:
        qq = qd[q0:q0 + ql].view(ql, heads, d).transpose(0, 1)
        kk = kd[k0:k0 + kl].view(kl, heads, d).transpose(0, 1)
        vv = vd[k0:k0 + kl].view(kl, heads, d).transpose(0, 1)
        p = torch.softmax(qq @ kk.transpose(1, 2) * d ** -0.5, dim=-1)
        outs.append((p @ vv).transpose(0, 1).reshape(ql, C_))
        q0, k0 = q0 + ql, k0 + kl
    ref = torch.cat(outs)
    gq, gk, gv = torch.autograd.grad(ref, (qd, kd, vd), dout.double())
    seg = ops.make_segments(q_lens, kv_lens)
    qg, kg, vg = q.to(dtype).cuda(), k.to(dtype).cuda(), v.to(dtype).cuda()
    lse = torch.zeros(heads, nq, device="cuda")
    out = ops.attention(qg, kg, vg, heads, d, seg, max(q_lens), lse=lse)
    assert relerr(out.float().cpu(), ref) < TOL[dtype] * 3
    # the saved statistic: log2-domain log-sum-exp of the scaled scores
    q0, k0 = 0, 0
    for ql, kl in segs[:1]:
        s = (q[:ql].view(ql, heads, d).transpose(0, 1).double() @ k[:kl].view(kl, heads, d).transpose(0, 1).double().transpose(1, 2)) * d ** -0.5
        assert (lse[:, :ql].double().cpu() - torch.logsumexp(s, -1) / math.log(2)).abs().max() < (1e-4 if dtype == torch.float32 else 2e-2)
    dq, dk, dv = ops.attention_bwd(qg, kg, vg, out, dout.to(dtype).cuda(), lse, heads, d, seg, max(q_lens), max(kv_lens))
    tol = 3e-5 if dtype == torch.float32 else TOL[dtype] * 3
    assert relerr(dq.float().cpu(), gq) < tol and relerr(dk.float().cpu(), gk) < tol and relerr(dv.float().cpu(), gv) < tol


# ------------------------------------------------------------------------------------------------ elementwise, loss
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_geglu_silu_resampling(ops, dtype):
    rows, D = 77, 256
    ag, dh = rnd((rows, 2 * D), 30, dtype), rnd((rows, D), 31, dtype)
    agd = ag.double().requires_grad_()
    h = agd[:, :D] * F.gelu(agd[:, D:])
    (gag,) = torch.autograd.grad(h, agd, dh.double())
    agg = ag.to(dtype).cuda()
    assert relerr(ops.geglu_fwd(agg).float().cpu(), h) < TOL[dtype] * 2
    assert relerr(ops.geglu_bwd(agg, dh.to(dtype).cuda()).float().cpu(), gag) < TOL[dtype] * 2
    x, dy = rnd((5, 1280), 32, dtype, 2.0), rnd((5, 1280), 33, dtype)
    xd = x.double().requires_grad_()
    (gx,) = torch.autograd.grad(F.silu(xd), xd, dy.double())
    assert relerr(ops.silu_bwd(x.to(dtype).cuda(), dy.to(dtype).cuda()).float().cpu(), gx) < TOL[dtype] * 2
    # silu_bwd with the pre-activation kept in fp32 (the time-embedding path)
    assert relerr(ops.train_eltwise(0, x.cuda(), dy.to(dtype).cuda(), torch.empty(5, 1280, dtype=dtype, device="cuda"), 1, 5 * 1280).float().cpu(), gx) < TOL[dtype] * 2
    du = rnd((2, 8, 8, 32), 34, dtype)
    assert relerr(ops.pool2x2_sum(du.to(dtype).cuda()).float().cpu(), du.double().view(2, 4, 2, 4, 2, 32).sum((2, 4))) < TOL[dtype]
    z = ops.zero_insert2x(du.to(dtype).cuda()).float().cpu()
    assert torch.equal(z[:, ::2, ::2], du) and float(z[:, 1::2].abs().max()) == 0 and float(z[:, :, 1::2].abs().max()) == 0
    a, b = rnd((100, 64), 35, dtype).to(dtype).cuda(), rnd((100, 64), 36, dtype).to(dtype).cuda()
    want = (a.float() + b.float()).to(dtype)
    ops.train_eltwise(1, b, None, a, 100, 64)
    assert torch.equal(a, want)


def test_add_noise_and_mse_loss(ops):
    """DDIMScheduler.add_noise (bit-exact vs the oracle scheduler) fused with the input scatter; mse_loss + gradient"""
    from mv_ldm_amd import _lib as L
    from oracle.scheduler import DDIMScheduler
    s = DDIMScheduler(clip_sample=False)
    n, c, h = 6, 4, 8
    g = G(37)
    x0, noise = torch.randn(n, c, h, h, generator=g), torch.randn(n, c, h, h, generator=g)
    t = torch.tensor([0, 10, 500, 999, 3, 77])
    ref = s.add_noise(x0, noise, t)
    ac = s.alphas_cumprod
    coef = torch.stack([ac[t] ** 0.5, (1 - ac[t]) ** 0.5], dim=1).float().contiguous().cuda()
    dst = torch.zeros(8, h, h, 16, dtype=torch.float32, device="cuda")
    rows = torch.tensor([7, 1, 2, 3, 4, 0], dtype=torch.int32, device="cuda")
    x0g, noiseg = x0.cuda(), noise.cuda()
    L.check(L.load().mvldm_add_noise(x0g.data_ptr(), noiseg.data_ptr(), coef.data_ptr(), dst.data_ptr(), n, c, h * h, 16, 0, L.F32,
                                     rows.data_ptr(), ops.stream()))
    assert torch.equal(dst[rows.long()][..., :4].cpu(), ref.permute(0, 2, 3, 1))
    # loss over the target images only
    n_img, n_tgt = 5, 3
    pred = torch.randn(n_img, h, h, c, generator=g)
    tgt_noise = torch.randn(n_tgt, c, h, h, generator=g)
    tgt_img = torch.tensor([1, 3, 4], dtype=torch.int32)
    pd = pred.double().requires_grad_()
    loss = F.mse_loss(pd[tgt_img.long()].permute(0, 3, 1, 2), tgt_noise.double())
    (gp,) = torch.autograd.grad(loss, pd)
    out = torch.zeros(1, device="cuda")
    dpred = torch.zeros(n_img, h, h, 8, dtype=torch.float32, device="cuda")
    ws = torch.zeros(256, dtype=torch.float64, device="cuda")
    predg, tng, tig = pred.cuda(), tgt_noise.cuda(), tgt_img.cuda()
    for acc in (0, 1):
        L.check(L.load().mvldm_mse_loss(predg.data_ptr(), tng.data_ptr(), tig.data_ptr(), n_tgt, h * h, c,
                                        out.data_ptr(), acc, 0.5, dpred.data_ptr(), 8, L.F32, 0.5, ws.data_ptr(), ops.stream()))
    assert abs(float(out) - float(loss.detach())) < 1e-6 * float(loss.detach())               # two half-weighted micro-batches
    assert relerr(dpred[..., :4], 0.5 * gp) < 1e-6 and float(dpred[..., 4:].abs().max()) == 0 and float(dpred[0].abs().max()) == 0


def test_adamw_and_gradient_clipping_match_torch(ops):
    """mvldm_adamw_step + mvldm_grad_norm against torch.optim.AdamW + clip_grad_norm_(0.1) (Lightning's
    gradient_clip_val, src/main.py:131): (a) the released schedule (lr 2e-5, LinearLR warm-up from 5e-4 over 200 steps,
    baseline.yaml:62-73) -- there the update (~1e-8) is below fp32 resolution of O(1) weights, so the weights themselves are
    compared; (b) lr 1e-2 without warm-up, where the UPDATE is compared"""
    from mv_ldm_amd.train import linear_lr_factor
    n = 10007
    for lr0, warm in ((2e-5, True), (1e-2, False)):
        g = G(38)
        p0 = torch.randn(n, generator=g)
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([ref], lr=lr0)
        sch = torch.optim.lr_scheduler.LinearLR(opt, start_factor=5e-4, total_iters=200) if warm else None
        p, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        for step in range(1, 7):
            grad = torch.randn(n, generator=g) * (0.01 if step % 2 else 1e-4)
            ref.grad = grad.clone()
            total = torch.nn.utils.clip_grad_norm_([ref], 0.1)
            lr = sch.get_last_lr()[0] if warm else lr0
            if warm:
                assert abs(lr - lr0 * linear_lr_factor(step - 1, start_factor=5e-4, total_iters=200)) < 1e-12 * lr0
            opt.step()
            if warm:
                sch.step()
            gg = grad.cuda()
            norm = ops.grad_norm(gg, 0.1)
            assert abs(float(norm[0]) - float(total)) < 1e-5 * float(total)
            assert abs(float(norm[1]) - min(1.0, 0.1 / (float(total) + 1e-6))) < 1e-5
            ops.adamw_step(p, gg, m, v, lr, step=step, clip=norm)
            assert relerr(p, ref.data) < 1e-6
        if not warm:
            assert relerr(p - p0.cuda(), ref.data - p0) < 2e-5


# ------------------------------------------------------------------------------------------------ production shapes
# The cases above are small (<= 192 channels, <= 320 tokens); these are the shapes of the released model at the configs[3]
# micro-batch (16 images): the wgrad split-K at M = 16 384 / K = 2880, the dual-source K = 23 040 conv of the up path, the
# K = 11 520 conv at 4x4 whose 180 K-steps are split, and attention backward over 4096-5120-token segments.  fp64 references
# (F.conv2d / explicit softmax + autograd on the host) from the same dtype-rounded inputs; same tolerances as above.
BIG_DTYPES, BIG_IDS = [torch.float32, torch.bfloat16], ["f32", "bf16"]
BIG_CONVS = [  # name, n, c0, c1, c_out, h
    ("320to320_at32_x16", 16, 320, 0, 320, 32), ("2560to1280_dual_at8_x16", 16, 1280, 1280, 1280, 8), ("1280_at4_x16", 16, 1280, 0, 1280, 4)]


@pytest.mark.parametrize("dtype", BIG_DTYPES, ids=BIG_IDS)
@pytest.mark.parametrize("case", BIG_CONVS, ids=[c[0] for c in BIG_CONVS])
def test_conv_gradients_at_production_shapes(ops, case, dtype):
    _, n, c0, c1, co, h = case
    ci = c0 + c1
    torch.set_num_threads(min(64, __import__("os").cpu_count() or 8))
    x, w = rnd((n, ci, h, h), 101, dtype), rnd((co, ci, 3, 3), 102, torch.float32, 1 / math.sqrt(ci * 9))
    xd, wd = x.double().requires_grad_(), w.to(dtype).double().requires_grad_()
    y = F.conv2d(xd, wd, None, 1, 1)
    dy = rnd(tuple(y.shape), 103, dtype)
    gx, gw = torch.autograd.grad(y, (xd, wd), dy.double())
    dyd = nhwc(dy, dtype)
    wc = w.cuda().contiguous()
    xa = nhwc(x[:, :c0], dtype)
    xb = nhwc(x[:, c0:], dtype) if c1 else None
    # data gradient(s): one transposed pack per source of the skip concat
    da = ops.conv2d(dyd, ops.pack_weight_t(wc, dtype, 0, c0))
    assert relerr(nchw(da), gx[:, :c0]) < TOL[dtype]
    if c1:
        db = ops.conv2d(dyd, ops.pack_weight_t(wc, dtype, c0, c1))
        assert relerr(nchw(db), gx[:, c0:]) < TOL[dtype]
    # weight gradient: split over the pixel range, slabs reduced in a fixed order; then accumulated
    grad = torch.zeros(co, ci, 3, 3, device="cuda")
    ops.conv_wgrad(xa, dyd, grad, ksize=3, x2=xb)
    assert relerr(grad, gw) < TOL[dtype]
    first = grad.clone()
    ops.conv_wgrad(xa, dyd, grad, ksize=3, x2=xb, accumulate=True)
    assert relerr(grad, 2 * gw) < TOL[dtype]
    again = torch.zeros_like(grad)
    ops.conv_wgrad(xa, dyd, again, ksize=3, x2=xb)
    assert torch.equal(again, first)            # deterministic reduction order


WIDE_CASES = [  # name, ksize, n_img, h, c_in, n_out: ragged n / c tiles, several splits, one split, borders of every tap
    ("conv320_at32", 3, 4, 32, 320, 320), ("conv_192to160_at16", 3, 3, 16, 192, 160), ("conv_64to480_at8", 3, 5, 8, 64, 480),
    ("conv1280_at4", 3, 2, 4, 1280, 1280), ("linear_1280to320", 1, 2, 32, 1280, 320), ("linear_72to200", 1, 1, 16, 72, 200)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
def test_wide_weight_gradient_form(ops, case, dtype):
    """the wide LDS-DMA weight-gradient kernel ([320 n] x [128 c] tile, form 2) against fp64 autograd and against the
    register-staged kernel (form 1): same products, same fp32 accumulation per split -- only the split count differs"""
    _, k, n, h, ci, co = case
    x, w = rnd((n, ci, h, h), 201, dtype), rnd((co, ci, k, k), 202, torch.float32, 1 / math.sqrt(ci * k * k))
    wd = w.to(dtype).double().requires_grad_()
    y = F.conv2d(x.double(), wd, None, 1, k // 2)
    dy = rnd(tuple(y.shape), 203, dtype)
    gw, = torch.autograd.grad(y, (wd,), dy.double())
    xa, dyd = nhwc(x, dtype), nhwc(dy, dtype)
    shape = (co, ci, k, k) if k == 3 else (co, ci)
    g2 = torch.zeros(shape, device="cuda")
    ops.conv_wgrad(xa, dyd, g2, ksize=k, form=2)
    assert relerr(g2.view_as(gw), gw) < TOL[dtype]
    g1 = torch.zeros(shape, device="cuda")
    ops.conv_wgrad(xa, dyd, g1, ksize=k, form=1)
    assert relerr(g2, g1.double().cpu()) < 1e-5
    ops.conv_wgrad(xa, dyd, g2, ksize=k, form=2, accumulate=True)
    assert relerr(g2.view_as(gw), 2 * gw) < TOL[dtype]
    # what the wide form does not take is refused when asked for explicitly, never silently replaced
    with pytest.raises(RuntimeError, match="wide form"):
        if k == 3:
            ops.conv_wgrad(xa, dyd, torch.zeros(shape, device="cuda"), ksize=3, stride=2, form=2)          # a downsampler
        else:
            ops.conv_wgrad(xa, dyd, torch.zeros(co, 2 * ci, device="cuda"), ksize=1, x2=xa, form=2)       # a skip-concat shortcut


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("form", [1, 2], ids=["small", "wide"])
def test_single_split_linear_weight_gradient_is_written_in_place(ops, dtype, form):
    """a Linear weight gradient that needs ONE split and is to be written (accumulate = 0: the first write of a store-first window plan)
    skips the slab and the reduce launch -- the kernel writes `grad` itself (wgrad.hip wgrad_direct).  Ragged in both directions
    (n_out, c_in not multiples of the tiles), into a poisoned buffer (every element must be written), bit-identical to the slab +
    reduce path (accumulate = 1 onto zeros), and the accumulating call still accumulates."""
    rows, c, n = 100, 328, 456
    x, w = rnd((rows, c), 301, dtype), rnd((n, c), 302, torch.float32, 1 / math.sqrt(c))
    wd = w.to(dtype).double().requires_grad_()
    y = x.double() @ wd.t()
    dy = rnd((rows, n), 303, dtype)
    gw, = torch.autograd.grad(y, (wd,), dy.double())
    xg, dyd = x.to(dtype).cuda().view(rows, 1, 1, c), dy.to(dtype).cuda()
    direct = torch.full((n, c), 7.0, device="cuda")
    ops.conv_wgrad(xg, dyd, direct, ksize=1, form=form)                       # rows < 256: one split
    assert relerr(direct, gw) < TOL[dtype]
    via_reduce = torch.zeros(n, c, device="cuda")
    ops.conv_wgrad(xg, dyd, via_reduce, ksize=1, form=form, accumulate=True)
    assert torch.equal(direct, via_reduce)
    ops.conv_wgrad(xg, dyd, direct, ksize=1, form=form, accumulate=True)
    assert relerr(direct, 2 * gw) < TOL[dtype]
    # a column slice of a fused projection's gradient (dy_ld > n_out) and a padded input (c_in < channels): the padded case keeps the reduce
    half = torch.full((n // 2, c), 7.0, device="cuda")
    ops.conv_wgrad(xg, dyd[:, :n // 2], half, ksize=1, form=form)
    assert relerr(half, gw[:n // 2]) < TOL[dtype]
    if form == 1:
        cut = torch.full((n, c - 8), 7.0, device="cuda")
        ops.conv_wgrad(xg, dyd, cut, ksize=1, c_in=c - 8, form=form)
        assert relerr(cut, gw[:, :c - 8]) < TOL[dtype]


BIG_ATTN = [("d40_3d_5120_4096", 8, 40, [5120, 4096]), ("d64_sd_1024x16", 5, 64, [1024] * 16)]


@pytest.mark.parametrize("dtype", BIG_DTYPES, ids=BIG_IDS)
@pytest.mark.parametrize("case", BIG_ATTN, ids=[c[0] for c in BIG_ATTN])
def test_attention_backward_at_production_shapes(ops, case, dtype):
    """level-0 3-D attention of a conditional (5 views) + an unconditional (4 views) scene, and the SD self-attention of 16
    images, on a fused [tokens, 3C] projection; reference per (segment, head) in fp64 so that one 5120 x 5120 score matrix
    is alive at a time"""
    _, heads, d, lens = case
    C_ = heads * d
    n = sum(lens)
    torch.set_num_threads(min(64, __import__("os").cpu_count() or 8))
    qkv = rnd((n, 3 * C_), 111, dtype)
    dout = rnd((n, C_), 112, dtype)
    ref = torch.empty(n, C_, dtype=torch.float64)
    gqkv = torch.empty(n, 3 * C_, dtype=torch.float64)
    r0 = 0
    for L_ in lens:
        for hd in range(heads):
            cs = [slice(j * C_ + hd * d, j * C_ + (hd + 1) * d) for j in range(3)]
            q, k, v = (qkv[r0:r0 + L_, c].double().requires_grad_() for c in cs)
            o = torch.softmax(q @ k.t() * d ** -0.5, dim=-1) @ v
            gs = torch.autograd.grad(o, (q, k, v), dout[r0:r0 + L_, hd * d:(hd + 1) * d].double())
            ref[r0:r0 + L_, hd * d:(hd + 1) * d] = o.detach()
            for c, g_ in zip(cs, gs):
                gqkv[r0:r0 + L_, c] = g_
        r0 += L_
    seg = ops.make_segments(lens, lens)
    x = qkv.to(dtype).cuda()
    q, k, v = x[:, :C_], x[:, C_:2 * C_], x[:, 2 * C_:]
    lse = torch.zeros(heads, n, device="cuda")
    out = ops.attention(q, k, v, heads, d, seg, max(lens), lse=lse)
    assert relerr(out.float().cpu(), ref) < TOL[dtype] * 3
    dqkv = torch.zeros(n, 3 * C_, dtype=dtype, device="cuda")
    ops.attention_bwd(q, k, v, out, dout.to(dtype).cuda(), lse, heads, d, seg, max(lens), max(lens), dqkv=dqkv)
    tol = 3e-5 if dtype == torch.float32 else TOL[dtype] * 3
    for j, nm in enumerate("qkv"):
        assert relerr(dqkv[:, j * C_:(j + 1) * C_].float().cpu(), gqkv[:, j * C_:(j + 1) * C_]) < tol, nm
