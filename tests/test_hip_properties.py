"""GPU property tests at BASELINE.json's full sizes (configs[1]: SD-2.1 topology + 9 multi-view blocks, 256x256,
1 context + 4 target views, 50 DDIM steps) where the CPU oracle would take minutes to hours.

The properties are size-independent and exact:
  * scaling an operand by a power of two scales an implicit-GEMM / attention output by exactly that factor
    (every product and every fp32 partial sum scales exactly; so does the final 16-bit rounding);
  * the result for an image does not depend on where the image sits in the batch (a different workgroup /
    tile / XCD computes it, with the same K order);
  * the whole sampler is deterministic (no float atomics on the data path), and permuting the scenes of a
    batch permutes the generated latents bit for bit.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
GRAD_ENABLED = False      # tests/conftest.py::_grad_mode: no autograd graphs in this module


@pytest.fixture(scope="module")
def ops():
    from mv_ldm_amd import _lib, ops as O
    _lib.load()
    return O


def _randn(shape, seed, dtype, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(dtype)


# one launch per tile family at the sizes of a 32-scene UNet pass (9 images per scene)
FULL_IGEMM = [
    # name, n_img, h, c0, c1, n_out, ksize, geglu
    ("L0 conv3x3 320->320 (256x320 tile)", 288, 32, 320, 0, 320, 3, False),
    ("up3 conv1 640+320->320 dual source", 288, 32, 640, 320, 320, 3, False),
    ("L1 conv3x3 640->640 (256x128 tile)", 288, 16, 640, 0, 640, 3, False),
    ("L2 conv3x3 1280->1280", 288, 8, 1280, 0, 1280, 3, False),
    ("L0 GEGLU 320->2560 (256x256 tile)", 288, 32, 320, 0, 2560, 1, True),
    ("L0 QKV 320->960", 288, 32, 320, 0, 960, 1, False),
    ("L3 conv3x3 1280->1280 @4x4 (split-K)", 288, 4, 1280, 0, 1280, 3, False),
]


@pytest.mark.parametrize("case", FULL_IGEMM, ids=[c[0] for c in FULL_IGEMM])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_igemm_full_size_scaling_and_batch_position(ops, case, dtype):
    name, n, h, c0, c1, co, k, geglu = case
    x = _randn((n, h, h, c0), 1, dtype)
    x2 = _randn((n, h, h, c1), 2, dtype) if c1 else None
    w = _randn((co, c0 + c1, k, k), 3, torch.float32, 1.0 / math.sqrt((c0 + c1) * k * k))
    pw = ops.pack_weight(w if k == 3 else w[:, :, 0, 0], dtype, geglu=geglu, c_split=c0 if c1 else None)
    epi = 2 if geglu else 0
    y = ops.conv2d(x, pw, x2=x2, epilogue=epi)
    assert torch.isfinite(y.float()).all()
    # the same launch twice: bit-identical
    assert torch.equal(y, ops.conv2d(x, pw, x2=x2, epilogue=epi))
    if not geglu:   # (GELU is not homogeneous)
        y4 = ops.conv2d(x * 4, pw, x2=None if x2 is None else x2 * 4, epilogue=epi)
        normal = y.float().abs() >= 2.0 ** -12     # (an f16 result in the subnormal range keeps fewer bits than its 4x)
        assert torch.equal(y4.float()[normal], (y.float() * 4)[normal]) and float(normal.float().mean()) > 0.99, name
    # reverse the image order: every image is now computed by a different workgroup
    idx = torch.arange(n - 1, -1, -1, device="cuda")
    yr = ops.conv2d(x[idx].contiguous(), pw, x2=None if x2 is None else x2[idx].contiguous(), epilogue=epi)
    assert torch.equal(yr, y[idx]), name
    # a 2-image launch (small tiles, other split-K) agrees to rounding with the 288-image one
    ys = ops.conv2d(x[:2].contiguous(), pw, x2=None if x2 is None else x2[:2].contiguous(), epilogue=epi)
    err = (ys.float() - y[:2].float()).norm() / y[:2].float().norm()
    assert err < (4e-3 if dtype == torch.bfloat16 else 5e-4), (name, float(err))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("heads,d,seq,nseg", [(5, 64, 5120, 16), (5, 64, 4096, 16), (10, 64, 256, 160), (8, 40, 5120, 8)],
                         ids=["3d_V5_32x32", "3d_V4_32x32", "self_16x16", "mv_d40"])
def test_attention_full_size_properties(ops, dtype, heads, d, seq, nseg):
    C = heads * d
    q, k, v = (_randn((nseg * seq, C), s, dtype) for s in (4, 5, 6))
    seg = ops.make_segments([seq] * nseg)
    o = ops.attention(q, k, v, heads, d, seg, seq)
    assert torch.isfinite(o.float()).all()
    assert torch.equal(o, ops.attention(q, k, v, heads, d, seg, seq))                       # deterministic
    o2, normal = ops.attention(q, k, v * 2, heads, d, seg, seq).float(), o.float().abs() >= 2.0 ** -12
    assert torch.equal(o2[normal], (o.float() * 2)[normal])                                # exactly linear in V
    # rows of softmax(QK^T) sum to one: V = const  =>  O = const (to the rounding of the normalisation)
    ones = torch.ones_like(v)
    o1 = ops.attention(q, k, ones, heads, d, seg, seq).float()
    assert (o1 - 1).abs().max() < (2e-2 if dtype == torch.bfloat16 else 2e-3)
    # segments are independent: reversing their order reverses the outputs bit for bit
    perm = torch.arange(nseg - 1, -1, -1, device="cuda").repeat_interleave(seq) * seq + torch.arange(seq, device="cuda").repeat(nseg)
    orv = ops.attention(q[perm].contiguous(), k[perm].contiguous(), v[perm].contiguous(), heads, d, seg, seq)
    assert torch.equal(orv, o[perm])


def test_groupnorm_layernorm_full_size_invariances(ops):
    dtype = torch.bfloat16
    x = _randn((288, 32, 32, 320), 7, dtype)
    g, b = _randn((320,), 8, torch.float32), _randn((320,), 9, torch.float32)
    y = ops.groupnorm(x, g, b, 32, 1e-5, True)
    assert torch.equal(y, ops.groupnorm(x, g, b, 32, 1e-5, True))             # deterministic (fp64 partials, fixed order)
    idx = torch.arange(287, -1, -1, device="cuda")
    assert torch.equal(ops.groupnorm(x[idx].contiguous(), g, b, 32, 1e-5, True), y[idx])
    # statistics of the normalised tensor (no affine, no SiLU): zero mean / unit variance per (image, group)
    one, zero = torch.ones(320, device="cuda"), torch.zeros(320, device="cuda")
    z = ops.groupnorm(x, one, zero, 32, 1e-5, False).float().view(288, 1024, 32, 10)
    assert z.mean(dim=(1, 3)).abs().max() < 2e-3
    assert (z.var(dim=(1, 3), unbiased=False) - 1).abs().max() < 1e-2
    t = x.view(-1, 320)
    ln = ops.layernorm(t, one, zero, 1e-5).float()
    assert ln.mean(dim=1).abs().max() < 2e-2 and (ln.var(dim=1, unbiased=False) - 1).abs().max() < 5e-2
    assert torch.equal(ops.layernorm(t, g, b, 1e-5), ops.layernorm(t, g, b, 1e-5))


@pytest.fixture(scope="module")
def full_pipeline():
    import bench
    import mv_ldm_amd
    from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg
    from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
    from mv_ldm_amd.scheduler import DDIMScheduler
    from mv_ldm_amd.vae import AutoencoderKL
    mv_ldm_amd.set_compute_dtype(torch.bfloat16)
    with torch.device("cuda"):
        den = MultiViewUNet(MultiViewUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1"), 11, 4)
        vae = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1")
    bench.random_init_(den, 1234)
    bench.random_init_(vae, 1235)
    pipe = MVLDMPipeline(den, vae, DDIMScheduler(clip_sample=False), SamplerCfg(True, 3.0, 50))
    pipe.set_timesteps(50)
    return pipe, bench


def test_full_config_sampler_is_deterministic_and_scene_equivariant(full_pipeline):
    """configs[1] exactly (full UNet, 256x256, 1+4 views, 50 DDIM steps, CFG 3.0), 3 scenes"""
    pipe, bench = full_pipeline
    b = 3
    batch = bench.synthetic_batch(b, 1, 4, 256, 77, torch.device("cuda"))
    x_T = torch.randn((b, 4, 4, 32, 32), generator=torch.Generator().manual_seed(5))
    noise = torch.randn((b, 4, 32, 32), generator=torch.Generator().manual_seed(6))   # one context view per scene
    img, x0 = pipe.sample(batch, x_T=x_T, encode_noise=noise)
    assert img.shape == (b, 4, 3, 256, 256) and torch.isfinite(img).all() and torch.isfinite(x0).all()
    assert float(x0.std()) > 1e-3
    img2, x0b = pipe.sample(batch, x_T=x_T, encode_noise=noise)
    assert torch.equal(x0, x0b) and torch.equal(img, img2)
    perm = [2, 0, 1]
    pb = {k: {kk: vv[perm].contiguous() for kk, vv in v.items()} for k, v in batch.items()}
    imgp, x0p = pipe.sample(pb, x_T=x_T[perm], encode_noise=noise[perm])
    assert torch.equal(x0p, x0[perm]) and torch.equal(imgp, img[perm])


def test_plan_time_autotune_freezes_a_valid_tile(ops):
    """recording a plan times every large implicit-GEMM launch per candidate tile (plan.autotune_igemm): the frozen
    choice must be one of the candidates, be cached per problem, and leave the result unchanged to rounding"""
    from mv_ldm_amd import plan as P
    dtype = torch.bfloat16
    x = _randn((36, 32, 32, 320), 11, dtype)
    w = _randn((320, 320, 3, 3), 12, torch.float32, 1.0 / math.sqrt(2880))
    pw = ops.pack_weight(w, dtype)
    ref = ops.conv2d(x, pw)
    P._TUNE_CACHE.clear()
    outs = []
    for tune in (True, False):
        bld = P.Builder(x.device, dtype, record=True)
        y = bld.conv(x, pw, name="probe")
        y2 = bld.conv(x, pw, name="probe_again")            # same problem: served from the cache
        plan = bld.finalize(autotune=tune)
        if tune:
            # (one entry for the problem itself; the parts a batch split tried -- plan.autosplit_igemm, round 6 -- have their own image counts)
            whole = [k for k in P._TUNE_CACHE if k[0] == 36]
            assert len(whole) == 1 and P._unpack_choice(P._TUNE_CACHE[whole[0]])[0] in P._TUNE_TILES      # ([tile, split-K] since round 4)
        plan.run()
        torch.cuda.synchronize()
        outs.append((y.clone(), y2.clone()))
    for y, y2 in outs:
        assert torch.equal(y, y2)
        assert (y.float() - ref.float()).norm() / ref.float().norm() < 4e-3


def test_every_candidate_the_tuner_may_freeze_reproduces_the_rules_result(ops):
    """the plan-time tuner freezes whatever (tile, split-K) ran fastest: every candidate it may try must either REFUSE the problem or
    compute it -- same result as the library's rule to rounding, and the same bits on a second run.  (Round 4: a new tile let the GEGLU
    epilogue through that its wave tile cannot pair; it read a neighbouring wave's parked block, right in most runs, and won a timing.)
    Problems: the epilogues and geometries the UNet carries, at the row counts the small / mid tuners see."""
    from mv_ldm_amd import plan as P
    dtype = torch.bfloat16
    tiles = sorted(set(P._TUNE_TILES) | set(P._SMALL_TILES) | {18})
    splits = sorted(set(s_ for s_ in P._SPLITS if s_ <= 8) | set(P._MID_SPLITS))
    cases = [  # name, n_img, h, c_in, n_out, ksize, epilogue, residual, row_bias
        ("geglu 320->2560", 9, 8, 320, 2560, 1, 2, False, False),
        ("geglu 1280->10240 few rows", 9, 4, 1280, 10240, 1, 2, False, False),
        ("linear 1280->1280 + residual", 9, 8, 1280, 1280, 1, 0, True, False),
        ("conv3x3 1280->320 + temb row", 9, 8, 1280, 320, 3, 0, False, True),
        ("conv3x3 320->320 + residual, 16x16", 10, 16, 320, 320, 3, 0, True, False),
    ]
    tried = refused = 0
    for name, n, h, ci, co, k, epi, res, rb in cases:
        x = _randn((n, h, h, ci), 500, dtype)
        w = _randn((co, ci, k, k), 501, torch.float32, 1.0 / math.sqrt(ci * k * k))
        pw = ops.pack_weight(w if k == 3 else w[:, :, 0, 0], dtype, geglu=epi == 2)
        nd = co // 2 if epi == 2 else co
        b = _randn((co,), 502, torch.float32, 0.1)
        r = _randn((n, h, h, nd), 503, dtype) if res else None
        rowb = _randn((n, nd), 504, torch.float32) if rb else None
        kw = dict(residual=r, row_bias=rowb, epilogue=epi)
        ref = ops.conv2d(x, pw, b, **kw).float()
        for tile in tiles:
            for sk in splits:
                try:
                    y = ops.conv2d(x, pw, b, tile=tile, splitk=sk, **kw)
                except RuntimeError:
                    refused += 1
                    continue
                tried += 1
                e = float((y.float() - ref).norm() / ref.norm())
                assert e < 6e-3, (name, tile, sk, e)
                assert torch.equal(y, ops.conv2d(x, pw, b, tile=tile, splitk=sk, **kw)), (name, tile, sk, "second run differs")
    assert tried > 200 and refused > 0, (tried, refused)


def test_batch_split_cuts_a_launch_at_an_image_boundary(ops, monkeypatch):
    """plan.autosplit_igemm (round 6): a big igemm launch becomes [images 0 .. n1) + [n1 .. n) when two launches are faster.  Forced here (the timing
    helper is patched to prefer every split): the two launches together must write what the whole launch writes -- every operand that is indexed
    by the image (both sources, the residual, the per-image row bias, the output) gets its pointer offset -- and the decision is cached per problem."""
    from mv_ldm_amd import _lib as L
    from mv_ldm_amd import plan as P
    dtype = torch.bfloat16
    n, hw, c0, c1, co = 80, 16, 320, 192, 640
    x = _randn((n, hw, hw, c0), 21, dtype)
    x2 = _randn((n, hw, hw, c1), 22, dtype)
    w = _randn((co, c0 + c1, 3, 3), 23, torch.float32, 1.0 / math.sqrt(9 * (c0 + c1)))
    pw = ops.pack_weight(w, dtype, c_split=c0)
    bias = _randn((co,), 24, torch.float32)
    row_bias = _randn((n, co), 25, torch.float32)
    res = _randn((n, hw, hw, co), 26, dtype)
    ref = ops.conv2d(x, pw, bias, x2=x2, row_bias=row_bias, residual=res)
    wl = _randn((co, c0), 27, torch.float32, 1.0 / math.sqrt(c0))
    pwl = ops.pack_weight(wl, dtype)
    xl = x.view(-1, c0)
    ref_l = ops.linear(xl, pwl, bias, residual=res.view(-1, co))

    calls = []
    real = P._time_ops
    monkeypatch.setattr(P, "_time_ops", lambda ops_, iters: (calls.append(len(ops_)), real(ops_, 1) * (0.5 if len(ops_) == 2 else 1.0))[1])
    monkeypatch.setattr(P, "_split_candidates", lambda d: [d.n_img * 3 // 4 // (256 // math.gcd(256, d.h_out * d.w_out)) * (256 // math.gcd(256, d.h_out * d.w_out))])
    monkeypatch.setattr(P, "_SPLIT_MIN_ROWS", 4096)
    P._SPLIT_CACHE.clear()
    bld = P.Builder(x.device, dtype, record=True)
    y = bld.conv(x, pw, bias, x2=x2, row_bias=row_bias, residual=res, name="conv")
    yl = bld.linear(xl, pwl, bias, residual=res.view(-1, co), name="linear")
    y_again = bld.conv(x, pw, bias, x2=x2, row_bias=row_bias, residual=res, name="conv_again")
    plan = bld.finalize(autotune=True)
    names = [m.name for m in plan.meta]
    assert names == ["conv", "conv[rest]", "linear", "linear[rest]", "conv_again", "conv_again[rest]"], names
    assert len(P._SPLIT_CACHE) == 2 and all(v > 0 for v in P._SPLIT_CACHE.values())         # (conv_again: from the cache)
    d0, d1 = plan.ops[0].u.igemm, plan.ops[1].u.igemm
    assert d0.n_img + d1.n_img == n and d1.src0 - d0.src0 == d0.n_img * hw * hw * c0 * 2 and d1.src1 - d0.src1 == d0.n_img * hw * hw * c1 * 2
    assert d1.row_bias - d0.row_bias == d0.n_img * co * 4 and d1.dst - d0.dst == d0.n_img * hw * hw * co * 2 == d1.residual - d0.residual
    assert abs(sum(m.flops for m in plan.meta[:2]) - 2.0 * n * hw * hw * co * 9 * (c0 + c1)) < 1.0
    for t in (y, yl, y_again):
        t.zero_()
    plan.run()
    torch.cuda.synchronize()
    for got, want in ((y, ref), (y_again, ref), (yl, ref_l)):
        assert torch.isfinite(got.float()).all()
        assert (got.float() - want.float()).norm() / want.float().norm() < 4e-3      # (tile / K-split of the parts: rounding only)
    assert torch.equal(y, y_again)
    # the kill switch keeps every launch whole
    monkeypatch.setattr(P, "_BATCH_SPLIT", False)
    bld = P.Builder(x.device, dtype, record=True)
    bld.conv(x, pw, bias, x2=x2, row_bias=row_bias, residual=res, name="conv")
    assert [m.name for m in bld.finalize(autotune=True).meta] == ["conv"]
