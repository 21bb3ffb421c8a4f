"""GPU parity tests, kernel by kernel, through the C ABI (mv_ldm_amd.ops -> libmvldm_hip.so).

Reference = the same op evaluated by plain torch on the CPU in fp64 from the SAME (already
dtype-rounded) inputs, so the only difference left is the kernel's own accumulation order and the
rounding of its output.  Tolerances (relative L2 / worst element relative to the output's RMS):
    f32 : 2e-5 / 2e-4     (fp32 MFMA accumulation over K up to ~3e3)
    f16 : 1e-3 / 1e-2     (output rounding 2^-11, P rounded to f16 in attention)
    bf16: 6e-3 / 5e-2     (output rounding 2^-8,  P rounded to bf16 in attention)
DDIM / scheduler arithmetic and all index work are bit-exact (asserted with torch.equal).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: (2e-5, 2e-4), torch.float16: (1e-3, 1e-2), torch.bfloat16: (6e-3, 5e-2)}
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


@pytest.fixture(scope="module")
def ops():
    from mv_ldm_amd import ops as O
    from mv_ldm_amd import _lib as L
    L.load()
    return O


def G(seed):
    return torch.Generator().manual_seed(seed)


def rnd(shape, seed, dtype, scale=1.0):
    """fp32 values already representable in `dtype`"""
    return (torch.randn(shape, generator=G(seed)) * scale).to(dtype).float()


def close(got, ref, dtype, what=""):
    got, ref = got.detach().double().cpu(), ref.double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    rl2, rmax = TOL[dtype]
    rms = ref.pow(2).mean().sqrt().clamp_min(1e-30)
    e2 = float((got - ref).norm() / ref.norm().clamp_min(1e-30))
    emax = float((got - ref).abs().max() / rms)
    assert e2 <= rl2 and emax <= rmax, f"{what}: rel-L2 {e2:.3e} (tol {rl2}), max/rms {emax:.3e} (tol {rmax})"


def nhwc(x, dtype):  # NCHW fp32 cpu -> NHWC dtype cuda
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def nchw(y):  # NHWC cuda -> NCHW fp64 cpu
    return y.float().cpu().permute(0, 3, 1, 2).double()


# ------------------------------------------------------------------------------------------------ igemm
CONV_CASES = [
    # name, n, cin, cin2, cout, h, w, k, stride, pad, upsample
    ("3x3", 2, 64, 0, 96, 8, 8, 3, 1, 1, False),
    ("3x3_wide_odd", 3, 40, 0, 72, 5, 7, 3, 1, 1, False),
    ("3x3_concat", 2, 64, 32, 64, 8, 8, 3, 1, 1, False),
    ("3x3_concat_aligned", 2, 128, 64, 96, 8, 8, 3, 1, 1, False),
    ("3x3_upsample_wide", 1, 128, 0, 128, 8, 8, 3, 1, 1, True),
    ("3x3_stride2_wide", 2, 128, 0, 64, 16, 16, 3, 2, 1, False),
    ("3x3_stride2", 2, 64, 0, 64, 8, 8, 3, 2, 1, False),
    ("3x3_stride2_asym_vae", 1, 32, 0, 32, 8, 8, 3, 2, 0, False),
    ("3x3_upsample", 2, 64, 0, 64, 4, 4, 3, 1, 1, True),
    ("1x1_shortcut", 2, 96, 0, 64, 6, 6, 1, 1, 0, False),
    ("3x3_big", 1, 320, 0, 320, 16, 16, 3, 1, 1, False),
    ("3x3_lowres_wide", 3, 1280, 1280, 1280, 4, 4, 3, 1, 1, False),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_variants(ops, case, dtype):
    name, n, cin, cin2, cout, h, w, k, stride, pad, up = case
    if dtype != torch.bfloat16 and name in ("3x3_lowres_wide",):
        pytest.skip("large case on the throughput dtype only")
    ws = 1.0 / math.sqrt((cin + cin2) * k * k)
    x = rnd((n, cin, h, w), 1, dtype)
    x2 = rnd((n, cin2, h, w), 2, dtype) if cin2 else None
    wt = rnd((cout, cin + cin2, k, k), 3, dtype, ws)
    b = torch.randn(cout, generator=G(4)) * 0.1
    xin = x if x2 is None else torch.cat([x, x2], 1)
    xr = xin.double()
    if up:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    if k == 3 and stride == 2 and pad == 0:
        xr = F.pad(xr, (0, 1, 0, 1))
    ref = F.conv2d(xr, wt.double(), b.double(), stride=stride, padding=pad)
    pw = ops.pack_weight(wt.cuda(), dtype, c_split=cin if cin2 else None)
    y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), x2=None if x2 is None else nhwc(x2, dtype), stride=stride, pad=pad, upsample=up)
    close(nchw(y), ref, dtype, name)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_conv_tiles_and_splitk(ops, dtype):
    """every tile configuration and split-K depth gives the same result"""
    n, cin, cout, h = 2, 128, 192, 12
    x, wt = rnd((n, cin, h, h), 5, dtype), rnd((cout, cin, 3, 3), 6, dtype, 1 / math.sqrt(cin * 9))
    b = torch.randn(cout, generator=G(7)) * 0.1
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    pw = ops.pack_weight(wt.cuda(), dtype)
    xg = nhwc(x, dtype)
    tiles = (1, 2, 3, 4, 5) if dtype == torch.float32 else (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 18)
    for tile in tiles:
        for sk in (1, 2, 5):
            y = ops.conv2d(xg, pw, b.cuda(), tile=tile, splitk=sk)
            close(nchw(y), ref, dtype, f"tile{tile}/splitk{sk}")
    if dtype != torch.float32:   # the register-prefetch main loop on 16-bit data (bit 12), forced XCD grids (bits 8-11)
        for code in (2 | (1 << 12), 3 | (1 << 12), 2 | (2 << 8), 7 | (4 << 8)):
            close(nchw(ops.conv2d(xg, pw, b.cuda(), tile=code, splitk=2)), ref, dtype, f"tilecode {code:#x}")
    close(nchw(ops.conv2d(xg, pw, b.cuda())), ref, dtype, "auto")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_conv_epilogues(ops, dtype):
    """+bias +per-image bias (time embedding) +residual, SiLU, fp32 output, out_scale"""
    n, cin, cout, h = 3, 64, 96, 6
    x, wt = rnd((n, cin, h, h), 8, dtype), rnd((cout, cin, 3, 3), 9, dtype, 1 / math.sqrt(cin * 9))
    b = torch.randn(cout, generator=G(10)) * 0.1
    rb = torch.randn(n, cout, generator=G(11))
    res = rnd((n, cout, h, h), 12, dtype)
    pw = ops.pack_weight(wt.cuda(), dtype)
    base = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), row_bias=rb.cuda(), residual=nhwc(res, dtype))
    close(nchw(y), base + rb.double()[:, :, None, None] + res.double(), dtype, "temb+residual")
    y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), epilogue=1, out_dtype=torch.float32, out_scale=0.5)
    assert y.dtype == torch.float32
    close(nchw(y), F.silu(base) * 0.5, dtype, "silu/f32/scale")
    for sk in (2, 3):  # same epilogues through the split-K reduce kernel
        y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), row_bias=rb.cuda(), residual=nhwc(res, dtype), splitk=sk)
        close(nchw(y), base + rb.double()[:, :, None, None] + res.double(), dtype, f"temb+residual splitk{sk}")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_conv_in_out_channel_padding(ops, dtype):
    """conv_in: 11 real channels in a 16-channel NHWC buffer; conv_out: 4 output channels"""
    n, h = 2, 8
    x = rnd((n, 11, h, h), 13, dtype)
    w_in = rnd((64, 11, 3, 3), 14, dtype, 0.1)
    pw = ops.pack_weight(w_in.cuda(), dtype, c_pad=16)
    xp = torch.zeros(n, h, h, 16, dtype=dtype, device="cuda")
    xp[..., :11] = nhwc(x, dtype)
    close(nchw(ops.conv2d(xp, pw)), F.conv2d(x.double(), w_in.double(), padding=1), dtype, "conv_in")
    x2 = rnd((n, 64, h, h), 15, dtype)
    w_out = rnd((4, 64, 3, 3), 16, dtype, 0.05)
    pw2 = ops.pack_weight(w_out.cuda(), dtype)
    y = ops.conv2d(nhwc(x2, dtype), pw2, out_dtype=torch.float32)
    assert y.shape == (n, h, h, 4)
    close(nchw(y), F.conv2d(x2.double(), w_out.double(), padding=1), dtype, "conv_out")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("rows,cin,cout", [(9, 1280, 320), (80, 320, 960), (300, 64, 64), (1000, 640, 640)])
def test_linear(ops, dtype, rows, cin, cout):
    x, wt = rnd((rows, cin), 17, dtype), rnd((cout, cin), 18, dtype, 1 / math.sqrt(cin))
    b = torch.randn(cout, generator=G(19)) * 0.1
    res = rnd((rows, cout), 20, dtype)
    pw = ops.pack_weight(wt.cuda(), dtype)
    y = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), residual=res.to(dtype).cuda())
    close(y.float().cpu().double(), F.linear(x.double(), wt.double(), b.double()) + res.double(), dtype, "linear")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("rows,c", [(50, 64), (300, 320)])
def test_geglu(ops, dtype, rows, c):
    x, wt = rnd((rows, c), 21, dtype), rnd((8 * c, c), 22, dtype, 1 / math.sqrt(c))
    b = torch.randn(8 * c, generator=G(23)) * 0.1
    pw = ops.pack_weight(wt.cuda(), dtype, geglu=True)
    hgate = F.linear(x.double(), wt.double(), b.double())
    a, g = hgate.chunk(2, -1)
    ref = a * F.gelu(g)
    for sk in (1, 2):
        y = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, splitk=sk)
        assert y.shape == (rows, 4 * c)
        close(y.float().cpu().double(), ref, dtype, f"geglu splitk{sk}")
    if dtype != torch.float32:   # the 8-wave tiles (staged epilogue only) with the GEGLU column pairing
        for tile in (1, 7, 8, 9):
            y = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=tile)
            close(y.float().cpu().double(), ref, dtype, f"geglu tile{tile}")


# ---- tile 15: the skinny-M weight-streaming GEMM (csrc/skinny.hip).  Configurations ride in bits 8-13 of `tile`:
#   3x3: 1 = 48 rows, 2 = 64 rows, 18 = 192 rows; stride 2: 5 = 48 output rows; 2x2 phase: 6 = 48 rows, 7 = 64 rows, 19 = 192 rows;
#   1x1 / Linear: 8 = 48 x 64, 9 = 48 x 32, 10 = 16 x 64, 20 = 16 x 32, 12 = 48 x 16, 13 = 96 x 32;
#   38 ... 46: the independent-wave-streams form (no barrier in the loop) of the 1x1 / Linear: 38 / 39 / 40 = 48 rows x 16 / 32 / 64 columns,
#   41 / 42 = 16 rows x 32 / 64, 43 / 46 = 48 x 16 / 32 with 8 waves, 44 = 96 x 32
def sk(cfg=0):
    return 15 | (cfg << 8)


SKINNY_CONV = [
    # name, n, cin, cin2, cout, h, w, k, stride, pad, configurations
    ("3x3_4x4_nine", 9, 128, 0, 192, 4, 4, 3, 1, 1, (0, 1, 18)),
    ("3x3_4x4_ragged_images", 7, 192, 0, 320, 4, 4, 3, 1, 1, (1, 18)),        # 7 images: groups of 3 / 9 are ragged
    ("3x3_8x8", 5, 128, 0, 128, 8, 8, 3, 1, 1, (0, 2, 18)),
    ("3x3_2x2", 9, 64, 0, 64, 2, 2, 3, 1, 1, (0, 1)),                         # configs[0]'s deepest levels (8x8 latents)
    ("3x3_1x1", 9, 64, 0, 64, 1, 1, 3, 1, 1, (0, 1)),
    ("3x3_stride2", 9, 128, 0, 128, 8, 8, 3, 2, 1, (0, 5)),
    ("3x3_stride2_asym_vae", 3, 64, 0, 64, 8, 8, 3, 2, 0, (5,)),
    ("1x1_shortcut_concat", 9, 128, 192, 320, 4, 4, 1, 1, 0, (0, 8, 9, 10, 12, 13, 20, 38, 39, 40, 44)),
    ("1x1_8x8", 9, 320, 0, 100, 8, 8, 1, 1, 0, (0, 8, 10, 13, 38, 42, 43)),                  # n_out 100 -> n_pad 128: columns past n_out are masked
    ("3x3_wide_K", 9, 2560, 0, 64, 4, 4, 3, 1, 1, (1,)),                         # 40 channel blocks: 4 rounds of the ring
    ("3x3_short_K", 9, 192, 0, 64, 4, 4, 3, 1, 1, (1,)),                      # 3 channel blocks: less than one round (stages past K read zeros)
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", SKINNY_CONV, ids=[c[0] for c in SKINNY_CONV])
def test_skinny_tile_convs(ops, case, dtype):
    """tile 15 against fp64 torch on every geometry it takes: 3x3 (stride 1 / 2, VAE's asymmetric pad), 1x1, one and two sources, images of
    1 ... 64 pixels, ragged image groups, K shorter and longer than the weight ring, every configuration that fits"""
    name, n, cin, cin2, cout, h, w, k, stride, pad, cfgs = case
    ws = 1.0 / math.sqrt((cin + cin2) * k * k)
    x = rnd((n, cin, h, w), 101, dtype)
    x2 = rnd((n, cin2, h, w), 102, dtype) if cin2 else None
    wt = rnd((cout, cin + cin2, k, k), 103, dtype, ws)
    b = torch.randn(cout, generator=G(104)) * 0.1
    xr = (x if x2 is None else torch.cat([x, x2], 1)).double()
    if k == 3 and stride == 2 and pad == 0:
        xr = F.pad(xr, (0, 1, 0, 1))
    ref = F.conv2d(xr, wt.double(), b.double(), stride=stride, padding=pad)
    pw = ops.pack_weight(wt.cuda(), dtype, c_split=cin if cin2 else None)
    for cfg in cfgs:
        y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), x2=None if x2 is None else nhwc(x2, dtype), stride=stride, pad=pad, tile=sk(cfg))
        close(nchw(y), ref, dtype, f"{name} cfg{cfg}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_skinny_tile_epilogues(ops, dtype):
    """+bias +per-image time-embedding row +residual (also in place), SiLU / GELU, fp32 output, out_scale, a wider destination"""
    n, cin, cout, h = 9, 128, 192, 4
    x, wt = rnd((n, cin, h, h), 111, dtype), rnd((cout, cin, 3, 3), 112, dtype, 1 / math.sqrt(cin * 9))
    b = torch.randn(cout, generator=G(113)) * 0.1
    rb = torch.randn(n, cout, generator=G(114))
    res = rnd((n, cout, h, h), 115, dtype)
    pw = ops.pack_weight(wt.cuda(), dtype)
    base = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    for cfg in (1, 18):     # (18: more than 9 tiles per workgroup -- the epilogue's operands are fetched late)
        y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), row_bias=rb.cuda(), residual=nhwc(res, dtype), tile=sk(cfg))
        close(nchw(y), base + rb.double()[:, :, None, None] + res.double(), dtype, f"temb+residual cfg{cfg}")
    y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), epilogue=1, out_dtype=torch.float32, out_scale=0.5, tile=sk(1))
    assert y.dtype == torch.float32
    close(nchw(y), F.silu(base) * 0.5, dtype, "silu/f32/scale")
    y = ops.conv2d(nhwc(x, dtype), pw, None, epilogue=3, tile=sk(1))
    close(nchw(y), F.gelu(base - b.double()[None, :, None, None]), dtype, "gelu, no bias")
    r = nhwc(res, dtype)
    y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), residual=r, out=r, tile=sk(1))       # x += f(x)
    assert y.data_ptr() == r.data_ptr()
    close(nchw(y), base + res.double(), dtype, "in-place residual")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,cin,cout", [(144, 1280, 1280), (576, 640, 320), (9, 320, 1280), (100, 5120, 1280), (37, 64, 64)],
                         ids=["144x1280", "576x640", "temb_9", "ffout_5120", "ragged_37"])
def test_skinny_tile_linear(ops, dtype, rows, cin, cout):
    x, wt = rnd((rows, cin), 121, dtype), rnd((cout, cin), 122, dtype, 1 / math.sqrt(cin))
    b = torch.randn(cout, generator=G(123)) * 0.1
    res = rnd((rows, cout), 124, dtype)
    pw = ops.pack_weight(wt.cuda(), dtype)
    ref = F.linear(x.double(), wt.double(), b.double()) + res.double()
    for cfg in (0, 8, 9, 10, 12, 13, 20, 38, 39, 40, 41, 42, 43, 44, 46):
        y = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), residual=res.to(dtype).cuda(), tile=sk(cfg))
        close(y.float().cpu().double(), ref, dtype, f"linear cfg{cfg}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,c", [(144, 1280), (50, 64), (576, 320)])
def test_skinny_tile_geglu(ops, dtype, rows, c):
    """GEGLU: the fragment-order pack pairs value / gate tiles; odd column-tile counts per workgroup are refused"""
    import mv_ldm_amd._lib as L
    x, wt = rnd((rows, c), 131, dtype), rnd((8 * c, c), 132, dtype, 1 / math.sqrt(c))
    b = torch.randn(8 * c, generator=G(133)) * 0.1
    pw = ops.pack_weight(wt.cuda(), dtype, geglu=True)
    a, g = F.linear(x.double(), wt.double(), b.double()).chunk(2, -1)
    ref = a * F.gelu(g)
    for cfg in (0, 8, 9, 10, 13, 20, 39, 40, 41, 42, 44, 46):
        y = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=sk(cfg))
        assert y.shape == (rows, 4 * c)
        close(y.float().cpu().double(), ref, dtype, f"geglu cfg{cfg}")
    with pytest.raises(L.MvldmError):
        ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=sk(12))        # one column tile per workgroup cannot pair value and gate


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("n,c,cout,h,w", [(9, 128, 128, 4, 4), (9, 64, 320, 8, 8), (5, 128, 64, 2, 2)], ids=["4x4", "8x8", "2x2"])
def test_skinny_tile_upsample_phases(ops, dtype, n, c, cout, h, w):
    """the four 2x2 phase convs of nearest-2x + 3x3 through tile 15 (rows scattered to (2i+py, 2j+px))"""
    x = rnd((n, c, h, w), 141, dtype)
    wt = rnd((cout, c, 3, 3), 142, dtype, 1 / math.sqrt(c * 9))
    b = torch.randn(cout, generator=G(143)) * 0.1
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), wt.double(), b.double(), padding=1)
    pws = [ops.pack_weight(t.cuda(), dtype) for t in ops.upsample_phase_weights(wt)]
    for cfg in (0, 6, 19) if h * w <= 16 else (0, 7, 19):
        y = ops.conv2d_upsample_phases(nhwc(x, dtype), pws, b.cuda(), tile=sk(cfg))
        close(nchw(y), ref, dtype, f"phases cfg{cfg}")


def test_skinny_tile_is_deterministic_and_refuses_what_it_cannot_do(ops):
    """bit-identical launch after launch under memory-system noise (counted waits over LDS-DMA pieces and register loads in one in-order
    queue; the waves' partial tiles are folded in wave order); f32, channels that are not multiples of 64, a K-major pointer, a configuration
    that does not fit the problem are errors, never fallbacks"""
    import mv_ldm_amd._lib as L
    torch.manual_seed(0)
    side, noise = torch.cuda.Stream(), torch.randn(16 << 20, device="cuda")
    for n, hh, cin, cout, k in ((9, 4, 1280, 1280, 3), (9, 8, 1280, 640, 1), (36, 4, 640, 1280, 3)):
        x = torch.randn(n, hh, hh, cin, device="cuda").to(torch.bfloat16)
        pw = ops.pack_weight(torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5, torch.bfloat16)
        b = torch.randn(cout, device="cuda")
        ref = ops.conv2d(x, pw, b, tile=15).clone()
        ref2 = ops.conv2d(x, pw, b, tile=2)
        assert (ref.float() - ref2.float()).abs().max().item() <= 2e-2 * ref2.float().abs().max().item()
        for i in range(40):
            if i % 4 == 0:
                with torch.cuda.stream(side):
                    noise.mul_(1.0001)
            assert torch.equal(ops.conv2d(x, pw, b, tile=15), ref), (n, hh, i)
    torch.cuda.synchronize()
    x = torch.randn(9, 4, 4, 128, device="cuda")
    with pytest.raises((L.MvldmError, AssertionError)):
        ops.conv2d(x, ops.pack_weight(torch.randn(64, 128, 3, 3, device="cuda"), torch.float32), tile=15)           # f32
    with pytest.raises((L.MvldmError, AssertionError)):
        ops.conv2d(x[..., :96].contiguous().bfloat16(), ops.pack_weight(torch.randn(64, 96, 3, 3, device="cuda"), torch.bfloat16), tile=15)   # 96 channels
    pw = ops.pack_weight(torch.randn(64, 128, 3, 3, device="cuda"), torch.bfloat16)
    with pytest.raises(L.MvldmError):
        ops.conv2d(x.bfloat16(), ops.pack_weight(torch.randn(64, 192, 3, 3, device="cuda"), torch.bfloat16, c_split=128), x2=x[..., :64].contiguous().bfloat16(), tile=15)   # two sources: 1x1 only
    with pytest.raises(L.MvldmError):
        ops.conv2d(x.bfloat16(), pw, tile=sk(8))                 # a Linear configuration on a 3x3 conv
    with pytest.raises(L.MvldmError):
        ops.conv2d(torch.randn(2, 8, 8, 128, device="cuda").bfloat16(), pw, tile=sk(1))      # 64-pixel images do not fit 48 rows
    d = ops.igemm_desc(x.bfloat16(), None, pw, torch.empty(9, 4, 4, 64, device="cuda", dtype=torch.bfloat16), n_img=9, h_in=4, w_in=4, h_out=4, w_out=4)
    d.tile = 15                                                   # K-major pointer with k_order 1: refused
    import ctypes as C
    assert L.load().mvldm_igemm_fwd(C.byref(d), 0) != 0
    d.tile, d.k_order = 2, 2                                      # and the fragment-order pack is refused by every other tile
    assert L.load().mvldm_igemm_fwd(C.byref(d), 0) != 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_deep_ring_tile_for_small_launches(ops, dtype):
    """tile 18 (192 x 128, 8 waves of 96 x 32, 4-slot ring: the weight-bound launches of a few scenes -- up to 192 rows fetch every
    weight byte once): 3x3 conv at the 4x4 / 8x8 level shapes of one scene (ragged row tiles: 144 and 576 rows), K longer and shorter
    than the ring (1 .. 20 channel blocks x 9 taps), split-K, the skip-concat 1x1 shortcut (two sources), Linear with bias +
    residual; against fp64 and bit-identical to tile 7 (same MFMA K order and epilogue) where both apply"""
    for (n, h, cin, cout) in ((9, 4, 1280, 320), (9, 8, 64, 128), (9, 8, 192, 192)):
        x, wt = rnd((n, cin, h, h), 71, dtype), rnd((cout, cin, 3, 3), 72, dtype, 1 / math.sqrt(cin * 9))
        b = torch.randn(cout, generator=G(73)) * 0.1
        ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
        pw = ops.pack_weight(wt.cuda(), dtype)
        xg = nhwc(x, dtype)
        for sk in (1, 3, 0):
            y = ops.conv2d(xg, pw, b.cuda(), tile=18, splitk=sk)
            close(nchw(y), ref, dtype, f"conv3x3 {cin}->{cout} @ {h}x{h} tile18/splitk{sk}")
        assert torch.equal(ops.conv2d(xg, pw, b.cuda(), tile=18, splitk=1), ops.conv2d(xg, pw, b.cuda(), tile=7, splitk=1))
    # two sources (never-materialised channel concat), 1x1
    n, h, c0, c1, cout = 9, 8, 128, 64, 256
    x, x2 = rnd((n, c0, h, h), 74, dtype), rnd((n, c1, h, h), 75, dtype)
    wt = rnd((cout, c0 + c1, 1, 1), 76, dtype, 1 / math.sqrt(c0 + c1))
    b = torch.randn(cout, generator=G(77)) * 0.1
    ref = F.conv2d(torch.cat([x, x2], 1).double(), wt.double(), b.double())
    pw = ops.pack_weight(wt[:, :, 0, 0].cuda(), dtype, c_split=c0)
    close(nchw(ops.conv2d(nhwc(x, dtype), pw, b.cuda(), x2=nhwc(x2, dtype), tile=18)), ref, dtype, "dual-source 1x1 tile18")
    # Linear + bias + residual, rows not a multiple of 192, N not a multiple of 128
    rows, cin, cout = 577, 1280, 320
    x, wt, res = rnd((rows, cin), 78, dtype), rnd((cout, cin), 79, dtype, 1 / math.sqrt(cin)), rnd((rows, cout), 80, dtype)
    b = torch.randn(cout, generator=G(81)) * 0.1
    pw = ops.pack_weight(wt.cuda(), dtype)
    for sk in (1, 4):
        y = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), residual=res.to(dtype).cuda(), tile=18, splitk=sk)
        close(y.float().cpu().double(), F.linear(x.double(), wt.double(), b.double()) + res.double(), dtype, f"linear+res tile18/splitk{sk}")
    # GEGLU pairs value and gate blocks inside a wave's tile; tile 18's waves own 32 columns: refused, not remapped, not approximated
    wg = rnd((8 * 64, 64), 82, dtype, 1 / 8.0)
    with pytest.raises(RuntimeError, match="GEGLU"):
        ops.linear(rnd((300, 64), 83, dtype).to(dtype).cuda(), ops.pack_weight(wg.cuda(), dtype, geglu=True), None, epilogue=2, tile=18)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,cin,cout", [(300, 320, 320), (1000, 384, 960), (2500, 640, 200), (70000, 320, 64), (20000, 1280, 328)])
def test_linear_persistent_tile(ops, dtype, rows, cin, cout):
    """tile 12 (linear_pp.hip: persistent workgroups, the epilogue of a tile runs under the next tile's main loop): every
    epilogue it carries, ragged M / N, more tiles than workgroups (70000 rows -> 274 tiles on <= 256 workgroups)"""
    x, wt = rnd((rows, cin), 41, dtype), rnd((cout, cin), 42, dtype, 1 / math.sqrt(cin))
    b = torch.randn(cout, generator=G(43)) * 0.1
    res = rnd((rows, cout), 44, dtype)
    pw = ops.pack_weight(wt.cuda(), dtype)
    lin = F.linear(x.double(), wt.double(), b.double())
    xg = x.to(dtype).cuda()
    y = ops.linear(xg, pw, b.cuda(), residual=res.to(dtype).cuda(), tile=12)
    close(y.float().cpu().double(), lin + res.double(), dtype, "linear+res tile12")
    y = ops.linear(xg, pw, None, tile=12)
    close(y.float().cpu().double(), F.linear(x.double(), wt.double()), dtype, "linear nobias tile12")
    y = ops.linear(xg, pw, b.cuda(), epilogue=1, tile=12)
    close(y.float().cpu().double(), F.silu(lin), dtype, "linear silu tile12")
    y = ops.linear(xg, pw, b.cuda(), epilogue=3, tile=12)
    close(y.float().cpu().double(), F.gelu(lin), dtype, "linear gelu tile12")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,c", [(300, 320), (33000, 320), (5000, 640)])
def test_geglu_persistent_tile(ops, dtype, rows, c):
    x, wt = rnd((rows, c), 45, dtype), rnd((8 * c, c), 46, dtype, 1 / math.sqrt(c))
    b = torch.randn(8 * c, generator=G(47)) * 0.1
    pw = ops.pack_weight(wt.cuda(), dtype, geglu=True)
    a, g = F.linear(x.double(), wt.double(), b.double()).chunk(2, -1)
    y = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=12)
    assert y.shape == (rows, 4 * c)
    close(y.float().cpu().double(), a * F.gelu(g), dtype, "geglu tile12")
    y7 = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=7)      # same MFMA K order; the bias enters first here, last there
    assert (y.float() - y7.float()).abs().max().item() <= 2e-2 * y7.float().abs().max().item()
    assert (y != y7).float().mean().item() < 0.05


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("n,h,c0,c1,cout", [(3, 16, 320, 320, 320), (5, 8, 640, 320, 640), (2, 32, 64, 576, 200)])
def test_persistent_tile_two_sources(ops, dtype, n, h, c0, c1, cout):
    """tile 12 on the 1x1 shortcut conv of an up-block resnet: the input is the channel concat of two tensors that is never
    materialised -- K-tiles [0, c0/64) stream from the first, the rest from the second"""
    x, x2 = rnd((n, c0, h, h), 51, dtype), rnd((n, c1, h, h), 52, dtype)
    w = rnd((cout, c0 + c1), 53, dtype, 1 / math.sqrt(c0 + c1))
    b = torch.randn(cout, generator=G(54)) * 0.1
    pw = ops.pack_weight(w.cuda(), dtype, c_split=c0)
    ref = F.conv2d(torch.cat([x, x2], 1).double(), w.double()[:, :, None, None], b.double())
    y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), x2=nhwc(x2, dtype), tile=12)
    close(nchw(y), ref, dtype, "dual-source 1x1 tile12")
    y0 = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), x2=nhwc(x2, dtype))
    assert (y.float() - y0.float()).abs().max().item() <= 2e-2 * y0.float().abs().max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,cin,cout", [(300, 320, 320), (1000, 384, 960), (2500, 640, 200), (70000, 320, 64), (20000, 1280, 328),
                                           (66000, 320, 640), (40001, 640, 1280)])
def test_linear_wide_persistent_tile(ops, dtype, rows, cin, cout):
    """tile 13 (linear_pw.hip: persistent 256 x 320 / 256 x 256 tiles, 4-slot BK = 32 ring across tile boundaries, epilogue straight
    from the accumulators through the permuted-column transposed product): ragged M / N, N a multiple of 320 (5 column blocks per
    wave) and not (4), more tiles than workgroups, residual (rolling asm loads) and none, no bias"""
    x, wt = rnd((rows, cin), 41, dtype), rnd((cout, cin), 42, dtype, 1 / math.sqrt(cin))
    b = torch.randn(cout, generator=G(43)) * 0.1
    res = rnd((rows, cout), 44, dtype)
    pw = ops.pack_weight(wt.cuda(), dtype)
    lin = F.linear(x.double(), wt.double(), b.double())
    xg = x.to(dtype).cuda()
    y = ops.linear(xg, pw, b.cuda(), residual=res.to(dtype).cuda(), tile=13)
    close(y.float().cpu().double(), lin + res.double(), dtype, "linear+res tile13")
    y = ops.linear(xg, pw, None, tile=13)
    close(y.float().cpu().double(), F.linear(x.double(), wt.double()), dtype, "linear nobias tile13")
    y = ops.linear(xg, pw, b.cuda(), tile=13)
    close(y.float().cpu().double(), lin, dtype, "linear tile13")
    y0 = ops.linear(xg, pw, b.cuda(), tile=7)       # same MFMA K order; the bias enters first here, last there
    assert (y.float() - y0.float()).abs().max().item() <= 2e-2 * y0.float().abs().max().item()
    assert (y != y0).float().mean().item() < 0.05


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,c", [(300, 320), (33000, 320), (5000, 640)])
def test_geglu_wide_persistent_tile(ops, dtype, rows, c):
    """tile 13 with the GEGLU pairing: value and gate of a column sit in the same lane (same row permutation for both blocks)"""
    x, wt = rnd((rows, c), 45, dtype), rnd((8 * c, c), 46, dtype, 1 / math.sqrt(c))
    b = torch.randn(8 * c, generator=G(47)) * 0.1
    pw = ops.pack_weight(wt.cuda(), dtype, geglu=True)
    a, g = F.linear(x.double(), wt.double(), b.double()).chunk(2, -1)
    y = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=13)
    assert y.shape == (rows, 4 * c)
    close(y.float().cpu().double(), a * F.gelu(g), dtype, "geglu tile13")
    y7 = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=7)
    assert (y.float() - y7.float()).abs().max().item() <= 2e-2 * y7.float().abs().max().item()
    assert (y != y7).float().mean().item() < 0.05


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("n,h,c0,c1,cout", [(3, 16, 320, 320, 320), (5, 8, 640, 320, 640), (2, 32, 64, 576, 200)])
def test_wide_persistent_tile_two_sources(ops, dtype, n, h, c0, c1, cout):
    """tile 13 on the 1x1 shortcut conv of an up-block resnet (never-materialised channel concat): K-steps [0, c0/32) stream from the
    first tensor, the rest from the second"""
    x, x2 = rnd((n, c0, h, h), 51, dtype), rnd((n, c1, h, h), 52, dtype)
    w = rnd((cout, c0 + c1), 53, dtype, 1 / math.sqrt(c0 + c1))
    b = torch.randn(cout, generator=G(54)) * 0.1
    pw = ops.pack_weight(w.cuda(), dtype, c_split=c0)
    ref = F.conv2d(torch.cat([x, x2], 1).double(), w.double()[:, :, None, None], b.double())
    y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), x2=nhwc(x2, dtype), tile=13)
    close(nchw(y), ref, dtype, "dual-source 1x1 tile13")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,cout", [(300, 320), (1000, 960), (70001, 320), (40000, 640), (33, 320)])
def test_linear_weight_stationary_tile(ops, dtype, rows, cout):
    """tile 14 (linear_ws.hip: K = 320, a wave keeps its 32 columns' weights in registers, 64-row slots stream through a 3-slot ring,
    the epilogue of a 32-row block runs under the next block's MFMAs): ragged M (partial slots, fewer slots than workgroups), one to
    three column slices, residual / none / no bias"""
    cin = 320
    x, wt = rnd((rows, cin), 71, dtype), rnd((cout, cin), 72, dtype, 1 / math.sqrt(cin))
    b = torch.randn(cout, generator=G(73)) * 0.1
    res = rnd((rows, cout), 74, dtype)
    pw = ops.pack_weight(wt.cuda(), dtype)
    lin = F.linear(x.double(), wt.double(), b.double())
    xg = x.to(dtype).cuda()
    y = ops.linear(xg, pw, b.cuda(), residual=res.to(dtype).cuda(), tile=14)
    close(y.float().cpu().double(), lin + res.double(), dtype, "linear+res tile14")
    y = ops.linear(xg, pw, None, tile=14)
    close(y.float().cpu().double(), F.linear(x.double(), wt.double()), dtype, "linear nobias tile14")
    y = ops.linear(xg, pw, b.cuda(), tile=14)
    close(y.float().cpu().double(), lin, dtype, "linear tile14")
    y0 = ops.linear(xg, pw, b.cuda(), tile=7)
    assert (y.float() - y0.float()).abs().max().item() <= 2e-2 * y0.float().abs().max().item()
    assert (y != y0).float().mean().item() < 0.05


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("rows", [300, 33000])
def test_geglu_weight_stationary_tile(ops, dtype, rows):
    """tile 14 with GEGLU: a wave's 32 weight rows are 16 value + 16 gate columns, the exact-erf GELU runs under the next block's MFMAs"""
    c = 320
    x, wt = rnd((rows, c), 75, dtype), rnd((8 * c, c), 76, dtype, 1 / math.sqrt(c))
    b = torch.randn(8 * c, generator=G(77)) * 0.1
    pw = ops.pack_weight(wt.cuda(), dtype, geglu=True)
    a, g = F.linear(x.double(), wt.double(), b.double()).chunk(2, -1)
    y = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=14)
    assert y.shape == (rows, 4 * c)
    close(y.float().cpu().double(), a * F.gelu(g), dtype, "geglu tile14")
    y7 = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=7)
    assert (y.float() - y7.float()).abs().max().item() <= 2e-2 * y7.float().abs().max().item()


def test_weight_stationary_tile_race_screen_and_refusals(ops):
    """tile 14: bit-identical results launch after launch under memory-system noise (counted vmcnt waits over ring refills, asm residual
    loads and asm stores in one in-order queue); K != 320 / a second source / an activation epilogue are errors, not fallbacks"""
    import mv_ldm_amd._lib as L
    torch.manual_seed(0)
    side, noise = torch.cuda.Stream(), torch.randn(16 << 20, device="cuda")
    for rows, n, epi, res in ((150000, 2560, 2, False), (200001, 320, 0, True), (90000, 960, 0, False)):
        x = torch.randn(rows, 320, device="cuda").to(torch.bfloat16)
        pw = ops.pack_weight(torch.randn(n, 320, device="cuda") / 320 ** 0.5, torch.bfloat16, geglu=epi == 2)
        b = torch.randn(n, device="cuda")
        r = torch.randn(rows, n, device="cuda").to(torch.bfloat16) if res else None
        ref = ops.linear(x, pw, b, residual=r, epilogue=epi, tile=14).clone()
        ref7 = ops.linear(x, pw, b, residual=r, epilogue=epi, tile=7)
        assert (ref.float() - ref7.float()).abs().max().item() <= 2e-2 * ref7.float().abs().max().item()
        for i in range(40):
            if i % 4 == 0:
                with torch.cuda.stream(side):
                    noise.mul_(1.0001)
            assert torch.equal(ops.linear(x, pw, b, residual=r, epilogue=epi, tile=14), ref), (rows, i)
    # in place: x += f(x)
    h = torch.randn(30000, 320, device="cuda").to(torch.bfloat16)
    x = torch.randn(30000, 320, device="cuda").to(torch.bfloat16)
    pw = ops.pack_weight(torch.randn(320, 320, device="cuda") / 320 ** 0.5, torch.bfloat16)
    want = ops.linear(x, pw, None, residual=h, tile=7)
    got = ops.linear(x, pw, None, residual=h, out=h, tile=14)
    assert (got.float() - want.float()).abs().max().item() <= 2e-2 * want.float().abs().max().item()
    torch.cuda.synchronize()
    pw640 = ops.pack_weight(torch.randn(320, 640, device="cuda"), torch.bfloat16)
    with pytest.raises(L.MvldmError):
        ops.linear(torch.randn(128, 640, device="cuda").to(torch.bfloat16), pw640, tile=14)
    with pytest.raises(L.MvldmError):
        ops.linear(x, pw, None, epilogue=1, tile=14)


def test_wide_persistent_tile_race_screen(ops):
    """like the tile-12 screen: the counted-vmcnt ring across tile boundaries, the rolling asm residual loads and the asm stores of
    tile 13 must give bit-identical results launch after launch while another stream keeps the memory system busy"""
    torch.manual_seed(0)
    side, noise = torch.cuda.Stream(), torch.randn(16 << 20, device="cuda")
    for rows, k, n, epi, res in ((150000, 320, 2560, 2, False), (70001, 384, 200, 0, True), (200000, 320, 320, 0, True), (90000, 640, 1920, 0, False)):
        x = torch.randn(rows, k, device="cuda").to(torch.bfloat16)
        pw = ops.pack_weight(torch.randn(n, k, device="cuda") / k ** 0.5, torch.bfloat16, geglu=epi == 2)
        b = torch.randn(n, device="cuda")
        r = torch.randn(rows, n, device="cuda").to(torch.bfloat16) if res else None
        ref = ops.linear(x, pw, b, residual=r, epilogue=epi, tile=13).clone()
        ref7 = ops.linear(x, pw, b, residual=r, epilogue=epi, tile=7)
        assert (ref.float() - ref7.float()).abs().max().item() <= 2e-2 * ref7.float().abs().max().item()
        for i in range(40):
            if i % 4 == 0:
                with torch.cuda.stream(side):
                    noise.mul_(1.0001)
            assert torch.equal(ops.linear(x, pw, b, residual=r, epilogue=epi, tile=13), ref), (rows, i)
    torch.cuda.synchronize()


def test_wide_persistent_tile_in_place_residual(ops):
    """x += f(x): the residual aliases the destination (every transformer sub-block of the UNet runs this way)"""
    dtype = torch.bfloat16
    x, wt = rnd((30000, 320), 61, dtype), rnd((320, 320), 62, dtype, 1 / math.sqrt(320))
    h = rnd((30000, 320), 63, dtype)
    pw = ops.pack_weight(wt.cuda(), dtype)
    hg = h.to(dtype).cuda()
    y = ops.linear(x.to(dtype).cuda(), pw, None, residual=hg, out=hg, tile=13)
    close(y.float().cpu().double(), F.linear(x.double(), wt.double()) + h.double(), dtype, "in-place residual tile13")


# ---- tile 19 (round 6): register-staged persistent Linear, 4 waves of 128 x 128 (csrc/linear_rs.hip) ----------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,cin,cout", [(300, 256, 320), (1000, 384, 960), (2500, 640, 200), (70000, 256, 64), (20000, 1280, 328),
                                           (66000, 512, 640), (40001, 640, 1280), (9216, 5120, 1280)])
def test_linear_register_staged_tile(ops, dtype, rows, cin, cout):
    """tile 19 (linear_rs.hip: persistent 256 x 256 tiles, two K-steps of operands in flight in registers, 2-slot LDS ring written by
    ds_write_b128, epilogue straight from the accumulators): ragged M / N, 4 .. 80 K-steps, more tiles than workgroups and fewer,
    residual and none, no bias"""
    x, wt = rnd((rows, cin), 41, dtype), rnd((cout, cin), 42, dtype, 1 / math.sqrt(cin))
    b = torch.randn(cout, generator=G(43)) * 0.1
    res = rnd((rows, cout), 44, dtype)
    pw = ops.pack_weight(wt.cuda(), dtype)
    lin = F.linear(x.double(), wt.double(), b.double())
    xg = x.to(dtype).cuda()
    y = ops.linear(xg, pw, b.cuda(), residual=res.to(dtype).cuda(), tile=19)
    close(y.float().cpu().double(), lin + res.double(), dtype, "linear+res tile19")
    y = ops.linear(xg, pw, None, tile=19)
    close(y.float().cpu().double(), F.linear(x.double(), wt.double()), dtype, "linear nobias tile19")
    y = ops.linear(xg, pw, b.cuda(), tile=19)
    close(y.float().cpu().double(), lin, dtype, "linear tile19")
    y0 = ops.linear(xg, pw, b.cuda(), tile=7)       # same MFMA K order; the bias enters last here
    assert (y.float() - y0.float()).abs().max().item() <= 2e-2 * y0.float().abs().max().item()
    assert (y != y0).float().mean().item() < 0.05


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,c", [(300, 256), (33000, 384), (5000, 640), (9216, 1280)])
def test_geglu_register_staged_tile(ops, dtype, rows, c):
    """tile 19 with the GEGLU pairing: value and gate of a column sit in the same lane (same row permutation for both blocks)"""
    x, wt = rnd((rows, c), 45, dtype), rnd((8 * c, c), 46, dtype, 1 / math.sqrt(c))
    b = torch.randn(8 * c, generator=G(47)) * 0.1
    pw = ops.pack_weight(wt.cuda(), dtype, geglu=True)
    a, g = F.linear(x.double(), wt.double(), b.double()).chunk(2, -1)
    y = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=19)
    assert y.shape == (rows, 4 * c)
    close(y.float().cpu().double(), a * F.gelu(g), dtype, "geglu tile19")
    y7 = ops.linear(x.to(dtype).cuda(), pw, b.cuda(), epilogue=2, tile=7)
    assert (y.float() - y7.float()).abs().max().item() <= 2e-2 * y7.float().abs().max().item()
    assert (y != y7).float().mean().item() < 0.05


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("n,h,c0,c1,cout", [(3, 16, 320, 320, 320), (5, 8, 640, 384, 640), (2, 32, 64, 576, 200), (9, 8, 1280, 1280, 1280)])
def test_register_staged_tile_two_sources(ops, dtype, n, h, c0, c1, cout):
    """tile 19 on the 1x1 shortcut conv of an up-block resnet (never-materialised channel concat): K-steps [0, c0/64) stream from the
    first tensor, the rest from the second (the switch may fall on an odd step)"""
    x, x2 = rnd((n, c0, h, h), 51, dtype), rnd((n, c1, h, h), 52, dtype)
    w = rnd((cout, c0 + c1), 53, dtype, 1 / math.sqrt(c0 + c1))
    b = torch.randn(cout, generator=G(54)) * 0.1
    pw = ops.pack_weight(w.cuda(), dtype, c_split=c0)
    ref = F.conv2d(torch.cat([x, x2], 1).double(), w.double()[:, :, None, None], b.double())
    y = ops.conv2d(nhwc(x, dtype), pw, b.cuda(), x2=nhwc(x2, dtype), tile=19)
    close(nchw(y), ref, dtype, "dual-source 1x1 tile19")


def test_register_staged_tile_race_screen_and_refusals(ops):
    """tile 19 keeps two K-steps in flight in registers across tile boundaries and through the epilogue's loads and stores, all in one
    in-order queue counted by the compiler: the same launch, repeated while another stream keeps the memory system busy, must be
    bit-identical every time.  In place (x += f(x)).  An odd number of K-steps / K < 256 / an activation epilogue are errors."""
    import mv_ldm_amd._lib as L
    torch.manual_seed(0)
    side, noise = torch.cuda.Stream(), torch.randn(16 << 20, device="cuda")
    for rows, k, n, epi, res in ((36864, 1280, 10240, 2, False), (70001, 384, 200, 0, True), (36864, 5120, 1280, 0, True),
                                 (147456, 640, 1920, 0, False)):
        x = torch.randn(rows, k, device="cuda").to(torch.bfloat16)
        pw = ops.pack_weight(torch.randn(n, k, device="cuda") / k ** 0.5, torch.bfloat16, geglu=epi == 2)
        b = torch.randn(n, device="cuda")
        r = torch.randn(rows, n, device="cuda").to(torch.bfloat16) if res else None
        ref = ops.linear(x, pw, b, residual=r, epilogue=epi, tile=19).clone()
        ref7 = ops.linear(x, pw, b, residual=r, epilogue=epi, tile=7)
        assert (ref.float() - ref7.float()).abs().max().item() <= 2e-2 * ref7.float().abs().max().item()
        for i in range(30):
            if i % 4 == 0:
                with torch.cuda.stream(side):
                    noise.mul_(1.0001)
            assert torch.equal(ops.linear(x, pw, b, residual=r, epilogue=epi, tile=19), ref), (rows, i)
    torch.cuda.synchronize()
    dtype = torch.bfloat16
    x, wt = rnd((30000, 640), 61, dtype), rnd((640, 640), 62, dtype, 1 / math.sqrt(640))
    h = rnd((30000, 640), 63, dtype)
    pw = ops.pack_weight(wt.cuda(), dtype)
    hg = h.to(dtype).cuda()
    y = ops.linear(x.to(dtype).cuda(), pw, None, residual=hg, out=hg, tile=19)
    close(y.float().cpu().double(), F.linear(x.double(), wt.double()) + h.double(), dtype, "in-place residual tile19")
    xs = torch.randn(512, 320, device="cuda").to(dtype)
    with pytest.raises(L.MvldmError):     # 5 K-steps
        ops.linear(xs, ops.pack_weight(torch.randn(320, 320, device="cuda"), dtype), tile=19)
    with pytest.raises(L.MvldmError):     # K = 128
        ops.linear(xs[:, :128].contiguous(), ops.pack_weight(torch.randn(320, 128, device="cuda"), dtype), tile=19)
    with pytest.raises(L.MvldmError):     # SiLU epilogue
        ops.linear(x.to(dtype).cuda(), pw, None, epilogue=1, tile=19)


def test_persistent_tile_race_screen(ops):
    """the counted-vmcnt ring, the prefetch across tile boundaries and the asm stores of tile 12 are the kind of code whose
    hazards show up as RARE wrong tiles: the same launch, repeated while another stream keeps the memory system busy, must be
    bit-identical every time (tools/race_screen.py is the long form: 1500 launches, 0 mismatches)"""
    torch.manual_seed(0)
    side, noise = torch.cuda.Stream(), torch.randn(16 << 20, device="cuda")
    for rows, k, n, epi, res in ((150000, 320, 2560, 2, False), (70001, 384, 200, 0, True)):
        x = torch.randn(rows, k, device="cuda").to(torch.bfloat16)
        pw = ops.pack_weight(torch.randn(n, k, device="cuda") / k ** 0.5, torch.bfloat16, geglu=epi == 2)
        b = torch.randn(n, device="cuda")
        r = torch.randn(rows, n, device="cuda").to(torch.bfloat16) if res else None
        ref = ops.linear(x, pw, b, residual=r, epilogue=epi, tile=12).clone()
        for i in range(40):
            if i % 4 == 0:
                with torch.cuda.stream(side):
                    noise.mul_(1.0001)
            assert torch.equal(ops.linear(x, pw, b, residual=r, epilogue=epi, tile=12), ref), (rows, i)
    torch.cuda.synchronize()


def test_two_source_3x3_conv_tiles(ops):
    """a 3x3 conv over a never-materialised channel concat: every tile that takes it agrees with the reference; the 256x256 /
    256x320 tiles (no room for the second source's offsets without spilling) refuse instead of running slowly"""
    dtype = torch.bfloat16
    x, x2 = rnd((2, 128, 16, 16), 1, dtype), rnd((2, 64, 16, 16), 2, dtype)
    wt = rnd((320, 192, 3, 3), 3, dtype, 0.05)
    b = torch.randn(320, generator=G(4)) * 0.1
    ref = F.conv2d(torch.cat([x, x2], 1).double(), wt.double(), b.double(), padding=1)
    pw = ops.pack_weight(wt.cuda(), dtype, c_split=128)
    xg, x2g = nhwc(x, dtype), nhwc(x2, dtype)
    close(nchw(ops.conv2d(xg, pw, b.cuda(), x2=x2g)), ref, dtype, "two-source 3x3, rules")
    for tile in (1, 2, 3, 4, 5, 6, 7, 8, 11):
        close(nchw(ops.conv2d(xg, pw, b.cuda(), x2=x2g, tile=tile, splitk=1)), ref, dtype, f"two-source 3x3 tile{tile}")
    for tile in (9, 10):
        with pytest.raises(RuntimeError, match="two-source 3x3"):
            ops.conv2d(xg, pw, b.cuda(), x2=x2g, tile=tile, splitk=1)


def test_persistent_tile_refuses_what_it_cannot_do(ops):
    """tile 12 is Linear-only (1x1, one source, K a multiple of 64 and >= 320): anything else is an error, not a silent fallback"""
    import mv_ldm_amd._lib as L
    x = torch.randn(2, 8, 8, 64, device="cuda").to(torch.bfloat16)
    pw3 = ops.pack_weight(torch.randn(64, 64, 3, 3, device="cuda"), torch.bfloat16)
    with pytest.raises(L.MvldmError):
        ops.conv2d(x, pw3, tile=12)
    pw1 = ops.pack_weight(torch.randn(64, 256, device="cuda"), torch.bfloat16)
    with pytest.raises(L.MvldmError):
        ops.linear(torch.randn(128, 256, device="cuda").to(torch.bfloat16), pw1, tile=12)     # K = 256: 4 K-tiles, the epilogue needs 5 steps


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_block_major_packers_bit_exact(ops, dtype):
    """the LDS-tiled packers of the 16-bit block-major layout (forward 1x1 incl. GEGLU row interleave, forward 3x3, transposed
    1x1 / 3x3 with a channel window) against the layout written out in torch: k = (64-channel block, tap, channel)"""
    g = G(77)
    # forward 3x3: n_out 70 (padded to 128), c_in 128
    w = torch.randn(70, 128, 3, 3, generator=g)
    pw = ops.pack_weight(w.cuda(), dtype)
    assert pw.k_order == 1
    ref = torch.zeros(pw.n_pad, 2, 9, 64)
    ref[:70] = w.reshape(70, 2, 64, 9).permute(0, 1, 3, 2)
    assert torch.equal(pw.data.float().cpu(), ref.reshape(pw.n_pad, -1).to(dtype).float())
    # forward 1x1 with the GEGLU interleave: rows alternate [value 32 | gate 32]
    wl = torch.randn(256, 192, generator=g)
    pl = ops.pack_weight(wl.cuda(), dtype, geglu=True)
    rows = torch.arange(256)
    blk, wi = rows // 32, rows % 32
    orig = torch.where(blk % 2 == 1, 128 + (blk // 2) * 32 + wi, (blk // 2) * 32 + wi)
    assert torch.equal(pl.data.float().cpu(), wl[orig].to(dtype).float())
    # transposed packs: rows = input channels [c_off, c_off + n_rows), K = (output block, flipped tap, output channel)
    for wt, c_off, n_rows in ((torch.randn(128, 200, 3, 3, generator=g), 8, 150), (torch.randn(192, 96, generator=g), 0, 96)):
        pt = ops.pack_weight_t(wt.cuda(), dtype, c_off, n_rows)
        assert pt.k_order == 1
        n_out = wt.shape[0]
        taps = 9 if wt.ndim == 4 else 1
        w4 = wt.reshape(n_out, wt.shape[1], taps)
        ref = torch.zeros(pt.n_pad, n_out // 64, taps, 64)
        src = w4[:, c_off:c_off + n_rows].flip(-1)                       # [n, r, tap']
        ref[:n_rows] = src.permute(1, 0, 2).reshape(n_rows, n_out // 64, 64, taps).permute(0, 1, 3, 2)
        assert torch.equal(pt.data.float().cpu(), ref.reshape(pt.n_pad, -1).to(dtype).float()), (wt.shape, c_off, n_rows)


def test_pack_weight_layout(ops):
    w = torch.randn(70, 24, 3, 3, generator=G(24))
    pw = ops.pack_weight(w.cuda(), torch.float32)
    assert pw.k_order == 0                      # 24 channels: tap-major K order
    ref = torch.zeros(pw.n_pad, pw.k_pad)
    ref[:70, :9 * 24] = w.permute(0, 2, 3, 1).reshape(70, -1)
    assert torch.equal(pw.data.cpu(), ref)
    w2 = torch.randn(8, 128, 3, 3, generator=G(26))
    p2 = ops.pack_weight(w2.cuda(), torch.bfloat16)
    assert p2.k_order == 1                      # 128 channels: (64-channel block, tap, channel) K order
    ref2 = torch.zeros(p2.n_pad, 2, 9, 64)
    ref2[:8] = w2.reshape(8, 2, 64, 9).permute(0, 1, 3, 2)
    assert torch.equal(p2.data.float().cpu(), ref2.reshape(p2.n_pad, -1).to(torch.bfloat16).float())
    assert ops.pack_weight(w2.cuda(), torch.bfloat16, c_split=40).k_order == 0   # unaligned concat split
    wl = torch.randn(128, 16, generator=G(25))
    pg = ops.pack_weight(wl.cuda(), torch.float32, geglu=True).data.cpu()
    assert torch.equal(pg[0:32, :16], wl[0:32]) and torch.equal(pg[32:64, :16], wl[64:96])
    assert torch.equal(pg[64:96, :16], wl[32:64]) and torch.equal(pg[96:128, :16], wl[96:128])


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("n,c,h,w,silu", [(2, 32, 4, 4, True), (3, 64, 8, 8, False), (5, 320, 32, 32, True),
                                          (2, 2560, 4, 4, True), (1, 128, 64, 48, True), (2, 960, 16, 16, False)])
def test_groupnorm(ops, dtype, n, c, h, w, silu):
    x = rnd((n, c, h, w), 26, dtype) * 2 + 0.7
    x = x.to(dtype).float()
    gm, bt = 1 + 0.2 * torch.randn(c, generator=G(27)), 0.2 * torch.randn(c, generator=G(28))
    ref = F.group_norm(x.double(), 32, gm.double(), bt.double(), 1e-5)
    if silu:
        ref = F.silu(ref)
    y = ops.groupnorm(nhwc(x, dtype), gm.cuda(), bt.cuda(), 32, 1e-5, silu)
    close(nchw(y), ref, dtype, "groupnorm")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("rows,c", [(7, 64), (1024, 320), (80, 1280), (33, 2560), (4099, 320), (2050, 640), (1027, 1280), (1500, 1024),
                                    (1300, 384), (1111, 192), (70000, 320)])
def test_layernorm(ops, dtype, rows, c):
    # (>= 1024 rows of 8/16/32 x {3,4,5} chunks: the multi-row kernel, ragged last row group; else one wave per row)
    x = (rnd((rows, c), 29, dtype) * 1.5 + 0.3).to(dtype).float()
    gm, bt = 1 + 0.2 * torch.randn(c, generator=G(30)), 0.2 * torch.randn(c, generator=G(31))
    y = ops.layernorm(x.to(dtype).cuda(), gm.cuda(), bt.cuda(), 1e-5)
    close(y.float().cpu().double(), F.layer_norm(x.double(), (c,), gm.double(), bt.double(), 1e-5), dtype, "layernorm")


# ------------------------------------------------------------------------------------------------ attention
def ref_attention(q, k, v, heads, d, q_lens, kv_lens):
    outs, q0, k0 = [], 0, 0
    for ql, kl in zip(q_lens, kv_lens):
        qq = q[q0:q0 + ql].double().view(ql, heads, d).transpose(0, 1)
        kk = k[k0:k0 + kl].double().view(kl, heads, d).transpose(0, 1)
        vv = v[k0:k0 + kl].double().view(kl, heads, d).transpose(0, 1)
        sim = (qq @ kk.transpose(1, 2)) * d ** -0.5
        outs.append((sim.softmax(-1) @ vv).transpose(0, 1).reshape(ql, heads * d))
        q0, k0 = q0 + ql, k0 + kl
    return torch.cat(outs)


ATTN_CASES = [  # heads, d, q_lens (kv = q: self-attention)
    (2, 8, [5]), (2, 16, [64, 64]), (3, 32, [100]), (8, 40, [320, 256]), (5, 64, [17, 256, 1]),
    (8, 80, [1280]), (8, 160, [320, 80]), (8, 40, [1029]), (1, 64, [129, 127, 128]),
    # long query segments (>= 256 rows) at every padded width with and without the ones column, ragged query and key tails,
    # one / two / many key tiles, short segments next to long ones
    (2, 16, [700]), (4, 8, [513]), (3, 32, [256, 300]), (2, 24, [300]), (2, 48, [384]), (2, 56, [260, 64]), (5, 64, [1024, 1]),
    (8, 40, [1280, 1024]), (1, 64, [256]), (2, 40, [257, 255, 63, 64, 65]),
    # the VAE mid-block head: one head of 512 columns (16-bit: head dimension split over the four waves on the matrix cores;
    # f32: the VALU kernel), whole and ragged query / key tiles, two heads
    (1, 512, [1024]), (1, 512, [100, 33, 1]), (2, 512, [65]),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("heads,d,lens", ATTN_CASES, ids=[f"h{c[0]}d{c[1]}L{'_'.join(map(str, c[2]))}" for c in ATTN_CASES])
def test_attention_self(ops, dtype, heads, d, lens):
    n, C = sum(lens), heads * d
    qkv = rnd((n, 3 * C), 32, dtype)           # one fused projection buffer: q|k|v column slices
    ref = ref_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], heads, d, lens, lens)
    g = qkv.to(dtype).cuda()
    seg = ops.make_segments(lens)
    y = ops.attention(g[:, :C], g[:, C:2 * C], g[:, 2 * C:], heads, d, seg, max(lens))
    close(y.float().cpu().double(), ref, dtype, "attention")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_attention_cross_and_spike(ops, dtype):
    """kv length != q length, and a key that dominates one query row late in the sequence (forces
    the online-softmax rescale branch with a large max jump)."""
    heads, d, ql, kl = 4, 40, [70, 33], [200, 1]
    q, k, v = rnd((sum(ql), heads * d), 33, dtype), rnd((sum(kl), heads * d), 34, dtype), rnd((sum(kl), heads * d), 35, dtype)
    k[150] = q[5] * 6.0   # large logit at key 150 (third tile) for query 5
    k = k.to(dtype).float()
    ref = ref_attention(q, k, v, heads, d, ql, kl)
    seg = ops.make_segments(ql, kl)
    y = ops.attention(q.to(dtype).cuda(), k.to(dtype).cuda(), v.to(dtype).cuda(), heads, d, seg, max(ql))
    close(y.float().cpu().double(), ref, dtype, "attention cross/spike")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_attention_long_cross_spike_and_lse(ops, dtype):
    """long segments with kv length != q length (both ways), a late dominating key (online-softmax rescale with a large max jump in
    the middle of the key stream), and the log-sum-exp saved for the backward pass"""
    heads, d = 4, 40
    ql, kl = [300, 290, 33], [200, 1000, 1]
    q, k, v = rnd((sum(ql), heads * d), 43, dtype), rnd((sum(kl), heads * d), 44, dtype), rnd((sum(kl), heads * d), 45, dtype)
    k[150] = q[5] * 6.0          # third key tile of segment 0, query 5
    k[200 + 777] = q[300 + 280] * 5.0   # late in segment 1, a query of the last wave
    k = k.to(dtype).float()
    ref = ref_attention(q, k, v, heads, d, ql, kl)
    seg = ops.make_segments(ql, kl)
    lse = torch.zeros(heads, sum(ql), device="cuda")
    y = ops.attention(q.to(dtype).cuda(), k.to(dtype).cuda(), v.to(dtype).cuda(), heads, d, seg, max(ql), lse=lse)
    close(y.float().cpu().double(), ref, dtype, "attention long cross/spike")
    q0, k0 = 0, 0
    for a, b in zip(ql, kl):
        sc = (q[q0:q0 + a].view(a, heads, d).transpose(0, 1).double() @ k[k0:k0 + b].view(b, heads, d).transpose(0, 1).double().transpose(1, 2)) * d ** -0.5
        assert (lse[:, q0:q0 + a].double().cpu() - torch.logsumexp(sc, -1) / math.log(2)).abs().max() < 3e-2
        q0, k0 = q0 + a, k0 + b


# ------------------------------------------------------------------------------------------------ small ops
def test_timestep_embedding_matches_oracle(ops):
    from oracle.blocks import Timesteps
    t = torch.tensor([0, 1, 20, 333, 980, 999], dtype=torch.int64)
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    y = ops.timestep_embed(t.cuda(), freqs.cuda(), 320, True)
    ref = Timesteps(320)(t)
    assert (y.cpu() - ref).abs().max() < 2e-6      # sinf/cosf of an identical fp32 argument
    y16 = ops.timestep_embed(t.cuda(), freqs.cuda(), 320, True, dtype=torch.bfloat16)
    assert (y16.float().cpu() - ref).abs().max() < 4e-3


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_eltwise_and_layout(ops, dtype):
    x = rnd((3, 1280), 36, dtype)
    close(ops.silu(x.to(dtype).cuda()).float().cpu().double(), F.silu(x.double()), dtype, "silu")
    assert torch.equal(ops.convert(x.cuda(), dtype).cpu(), x.to(dtype))
    img = torch.randn(2, 5, 4, 6, generator=G(37))
    buf = torch.zeros(2, 4, 6, 16, dtype=dtype, device="cuda")
    ops.nchw_to_nhwc(img.cuda(), dtype, dst=buf, c_off=3)
    assert torch.equal(buf[..., 3:8].cpu(), img.permute(0, 2, 3, 1).to(dtype))
    assert float(buf[..., :3].abs().max()) == 0 and float(buf[..., 8:].abs().max()) == 0
    back = ops.nhwc_to_nchw(buf, c=5, c_off=3)
    assert torch.equal(back.cpu(), img.to(dtype).float())
    clamped = ops.nhwc_to_nchw(buf, c=5, c_off=3, scale=0.5, shift=0.5, clamp01=True)
    assert torch.equal(clamped.cpu(), (img.to(dtype).float() * 0.5 + 0.5).clamp(0, 1))


def test_ddim_cfg_step_bit_exact(ops):
    """CFG compose + DDIM update: bit-identical to the torch CPU fp32 expression of the oracle
    scheduler (diffusion_wrapper.py:444,451 / DDIMScheduler.step), every step of a 50-step run."""
    from oracle.scheduler import DDIMScheduler
    s = DDIMScheduler(clip_sample=False)
    s.set_timesteps(50)
    ac = s.alphas_cumprod
    coef = []
    for t in s.timesteps.tolist():
        a_t = ac[t]
        a_p = ac[t - 20] if t - 20 >= 0 else s.final_alpha_cumprod
        coef.append(torch.stack([(1 - a_t) ** 0.5, a_t ** 0.5, a_p ** 0.5, (1 - a_p) ** 0.5]))
    coef = torch.stack(coef).float().contiguous()
    n_tgt, n_img, h, w, c = 3, 7, 4, 4, 4
    g = G(38)
    eps = torch.randn(n_img, h, w, c, generator=g)
    x = torch.randn(n_tgt, h, w, c, generator=g)
    cond = torch.tensor([1, 2, 3], dtype=torch.int32)
    unc = torch.tensor([4, 5, 6], dtype=torch.int32)
    unet_in = torch.zeros(n_img, h, w, 16, dtype=torch.bfloat16, device="cuda")
    for step in (0, 1, 25, 49):
        sp = torch.tensor([step], dtype=torch.int32).cuda()
        t = s.timesteps[step]
        pred = eps[unc.long()] + 3.0 * (eps[cond.long()] - eps[unc.long()])
        ref = s.step(pred, t, x).prev_sample
        got = ops.ddim_cfg_step(eps.cuda(), x.cuda(), cond.cuda(), unc.cuda(), 3.0, coef.cuda(), sp, unet_in)
        assert torch.equal(got.cpu(), ref), f"step {step}"
        assert torch.equal(unet_in[1:4, ..., :4].cpu(), ref.to(torch.bfloat16))
        assert torch.equal(unet_in[4:7, ..., :4].cpu(), ref.to(torch.bfloat16))
        ref_nocfg = s.step(eps[cond.long()], t, x).prev_sample
        got = ops.ddim_cfg_step(eps.cuda(), x.cuda(), cond.cuda(), None, 3.0, coef.cuda(), sp, None)
        assert torch.equal(got.cpu(), ref_nocfg)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("n,c0,c1,cout,h,w", [(3, 64, 0, 96, 5, 7), (2, 128, 64, 128, 16, 16), (5, 320, 0, 320, 32, 32), (9, 640, 640, 640, 8, 8),
                                              (40, 128, 0, 64, 4, 4), (1, 64, 0, 64, 40, 44)],
                         ids=["odd_5x7", "dual_16x16", "L0_32x32", "dual_8x8", "many_4x4", "wide_44"])
def test_conv3x3_halo_tile(ops, dtype, n, c0, c1, cout, h, w):
    """tile 11: the LDS-resident pixel-halo kernel (all 9 taps served from one halo tile per channel block): image
    borders, tiles spanning several images, ragged last tile, two sources, epilogues"""
    x = rnd((n, c0, h, w), 31, dtype)
    x2 = rnd((n, c1, h, w), 32, dtype) if c1 else None
    wt = rnd((cout, c0 + c1, 3, 3), 33, dtype, 1 / math.sqrt((c0 + c1) * 9))
    b = torch.randn(cout, generator=G(34)) * 0.1
    rb = torch.randn(n, cout, generator=G(35))
    res = rnd((n, cout, h, w), 36, dtype)
    xin = x if x2 is None else torch.cat([x, x2], 1)
    ref = F.conv2d(xin.double(), wt.double(), b.double(), padding=1)
    pw = ops.pack_weight(wt.cuda(), dtype, c_split=c0 if c1 else None)
    xg, x2g = nhwc(x, dtype), (None if x2 is None else nhwc(x2, dtype))
    y = ops.conv2d(xg, pw, b.cuda(), x2=x2g, tile=11, splitk=1)
    close(nchw(y), ref, dtype, "halo")
    y7 = ops.conv2d(xg, pw, b.cuda(), x2=x2g, tile=7, splitk=1)
    assert torch.equal(y, y7)                     # same K order as the streaming kernel: bit-identical
    y = ops.conv2d(xg, pw, b.cuda(), x2=x2g, row_bias=rb.cuda(), residual=nhwc(res, dtype), tile=11, splitk=1)
    close(nchw(y), ref + rb.double()[:, :, None, None] + res.double(), dtype, "halo temb+residual")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("n,c0,cout,h,w", [(3, 64, 96, 5, 7), (9, 640, 640, 16, 16), (20, 128, 320, 8, 8), (40, 128, 64, 4, 4), (2, 64, 320, 20, 24),
                                           (1, 64, 640, 3, 3)],
                         ids=["odd_5x7", "L1_16x16", "8x8_spans_images", "many_4x4", "widest_24", "one_ragged_tile"])
def test_conv3x3_wide_halo_tile(ops, dtype, n, c0, cout, h, w):
    """tile 17: the pixel halo under a 256 x 320 tile (maps up to 24 pixels wide): image borders, tiles spanning several images, ragged
    last row tile, column overhang (96 / 64 / 640 columns under 320-wide tiles), epilogues; bit-identical to the streaming kernels (same
    K order); what it cannot take (two sources, a 32-wide map) silently runs as tile 7 -- same values"""
    x = rnd((n, c0, h, w), 131, dtype)
    wt = rnd((cout, c0, 3, 3), 133, dtype, 1 / math.sqrt(c0 * 9))
    b = torch.randn(cout, generator=G(134)) * 0.1
    rb = torch.randn(n, cout, generator=G(135))
    res = rnd((n, cout, h, w), 136, dtype)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    pw = ops.pack_weight(wt.cuda(), dtype)
    xg = nhwc(x, dtype)
    y = ops.conv2d(xg, pw, b.cuda(), tile=17, splitk=1)
    close(nchw(y), ref, dtype, "wide halo")
    assert torch.equal(y, ops.conv2d(xg, pw, b.cuda(), tile=7, splitk=1))
    assert torch.equal(y, ops.conv2d(xg, pw, b.cuda(), tile=17, splitk=1))          # deterministic
    y = ops.conv2d(xg, pw, b.cuda(), row_bias=rb.cuda(), residual=nhwc(res, dtype), epilogue=1 if cout == 96 else 0, tile=17, splitk=1)
    want = ref + rb.double()[:, :, None, None]
    want = (F.silu(want) if cout == 96 else want) + res.double()
    close(nchw(y), want, dtype, "wide halo temb+residual")


def test_wide_halo_tile_falls_back_where_it_does_not_apply(ops):
    dtype = torch.bfloat16
    x, x2 = rnd((2, 64, 32, 32), 141, dtype), rnd((2, 64, 8, 8), 142, dtype)
    wt = rnd((320, 64, 3, 3), 143, dtype, 1 / 24)
    pw = ops.pack_weight(wt.cuda(), dtype)
    y = ops.conv2d(nhwc(x, dtype), pw, None, tile=17, splitk=1)                      # 32-wide map: 164 KB of LDS -> tile 7
    assert torch.equal(y, ops.conv2d(nhwc(x, dtype), pw, None, tile=7, splitk=1))
    wt2 = rnd((320, 128, 3, 3), 144, dtype, 1 / 34)
    pw2 = ops.pack_weight(wt2.cuda(), dtype, c_split=64)
    y = ops.conv2d(nhwc(x2, dtype), pw2, None, x2=nhwc(x2, dtype), tile=17, splitk=1)  # two sources -> tile 7
    close(nchw(y), F.conv2d(torch.cat([x2, x2], 1).double(), wt2.double(), padding=1), dtype, "wide halo fallback")


def test_empty_and_degenerate_inputs(ops):
    """zero images / rows / segments are no-ops (no launch, no error); one-pixel images and one-token sequences work;
    bad arguments are refused by the C side (negative return code -> exception), never silently computed elsewhere"""
    from mv_ldm_amd._lib import MvldmError
    dt_ = torch.bfloat16
    w = torch.randn(64, 64, 3, 3, generator=G(41))
    pw = ops.pack_weight(w.cuda(), dt_)
    y = ops.conv2d(torch.empty(0, 8, 8, 64, dtype=dt_, device="cuda"), pw)
    assert y.shape == (0, 8, 8, 64)
    x1 = rnd((2, 64, 1, 1), 42, dt_)                       # 1x1 images: only the centre tap is inside
    y = ops.conv2d(nhwc(x1, dt_), pw)
    close(nchw(y), F.conv2d(x1.double(), w.to(dt_).double(), padding=1), dt_, "1x1 image")
    g, b = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    assert ops.groupnorm(torch.empty(0, 4, 4, 64, dtype=dt_, device="cuda"), g, b, 32, 1e-5, True).shape == (0, 4, 4, 64)
    assert ops.layernorm(torch.empty(0, 64, dtype=dt_, device="cuda"), g, b).shape == (0, 64)
    q = rnd((5, 64), 43, dt_).to(dt_).cuda()
    o = ops.attention(q, q, q, 1, 64, ops.make_segments([1, 1, 3]), 3)      # one-token sequences: softmax over one key = V
    assert torch.isfinite(o.float()).all() and float((o[:2].float() - q[:2].float()).abs().max()) == 0.0
    o = ops.attention(q[:0], q[:0], q[:0], 1, 64, ops.make_segments([]), 0)
    assert o.shape == (0, 64)
    with pytest.raises(MvldmError):
        ops.attention(q, q, q, 1, 63, ops.make_segments([5]), 5)             # head_dim not a multiple of 8
    with pytest.raises(MvldmError):
        ops.groupnorm(torch.empty(1, 2, 2, 64, dtype=dt_, device="cuda"), g, b, 48, 1e-5, False)   # 64 % 48 != 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("n,c,cout,h,w,tile", [(2, 64, 64, 4, 4, 0), (3, 128, 96, 5, 7, 0), (5, 320, 320, 16, 16, 0), (9, 64, 320, 8, 8, 10),
                                               (2, 128, 128, 33, 20, 7)],
                         ids=["4x4", "odd_5x7", "320_16x16", "tile10", "tile7_ragged"])
def test_upsample_conv_as_four_phase_convs(ops, dtype, n, c, cout, h, w, tile):
    """nearest-2x + 3x3/pad-1 conv == four 2x2 convs on the low-resolution image with pre-summed taps
    (ops.upsample_phase_weights): against F.interpolate + F.conv2d, and against the gather form of the same kernel"""
    x = rnd((n, c, h, w), 51, dtype)
    wt = rnd((cout, c, 3, 3), 52, dtype, 1 / math.sqrt(c * 9))
    b = torch.randn(cout, generator=G(53)) * 0.1
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), wt.double(), b.double(), padding=1)
    # the identity itself, in fp64 on the CPU (each phase is a 2x2 conv with padding chosen per parity)
    wp = [t.double() for t in ops.upsample_phase_weights(wt)]          # (taps summed in fp32)
    chk = torch.zeros_like(ref)
    for ph, wph in enumerate(wp):
        py, px = ph >> 1, ph & 1
        xp = F.pad(x.double(), (1 - px, px, 1 - py, py))
        chk[:, :, py::2, px::2] = F.conv2d(xp, wph, b.double())
    assert (chk - ref).abs().max() < 1e-5
    pws = [ops.pack_weight(t.cuda(), dtype) for t in ops.upsample_phase_weights(wt)]
    y = ops.conv2d_upsample_phases(nhwc(x, dtype), pws, b.cuda(), tile=tile)
    assert y.shape == (n, 2 * h, 2 * w, cout)
    close(nchw(y), ref, dtype, "phases")
    y_gather = ops.conv2d(nhwc(x, dtype), ops.pack_weight(wt.cuda(), dtype), b.cuda(), upsample=True)
    close(nchw(y), nchw(y_gather), dtype, "phases vs gather")
    # small batches split K (K = 4C over few output tiles): partial slabs by low-resolution row, scattered by the reduce kernel
    for sk in (1, 3):
        ys = ops.conv2d_upsample_phases(nhwc(x, dtype), pws, b.cuda(), tile=tile, splitk=sk)
        close(nchw(ys), ref, dtype, f"phases, split-K {sk}")


def test_ddpm_step_bit_exact(ops):
    """`SCHEDULER["ddpm"]`: the ancestral update of `mv_ldm_amd.scheduler.DDPMScheduler.step` (HIP kernel, host-side scalars) is
    bit-identical to the restated diffusers `DDPMScheduler.step` on the CPU, with and without `clip_sample`, with the CFG compose
    fused in front (against compose-then-step on the CPU), at t = 0 (no noise) and for a device-side draw (same generator seed)"""
    from mv_ldm_amd.scheduler import DDPMScheduler
    from oracle.scheduler import DDPMScheduler as OracleDDPM
    g = G(11)
    x, ec, eu, z = (torch.randn(2, 4, 4, 16, 16, generator=g) for _ in range(4))
    x = x * 1.7
    for clip in (False, True):
        h, o = DDPMScheduler(clip_sample=clip), OracleDDPM(clip_sample=clip)
        h.set_timesteps(50)
        o.set_timesteps(50)
        for t in (980, 500, 20, 0):
            want = o.step(ec, t, x, variance_noise=z).prev_sample
            got = h.step(ec.cuda(), torch.tensor(t), x.cuda(), variance_noise=z.cuda()).prev_sample
            assert torch.equal(got.cpu(), want), (clip, t, float((got.cpu() - want).abs().max()))
            e = eu + 3.0 * (ec - eu)
            want = o.step(e, t, x, variance_noise=z).prev_sample
            got = h.step(ec.cuda(), t, x.cuda(), variance_noise=z.cuda(), model_output_uncond=eu.cuda(), cfg_scale=3.0).prev_sample
            assert torch.equal(got.cpu(), want), ("cfg", clip, t)
    h = DDPMScheduler(clip_sample=False)
    h.set_timesteps(50)
    a = h.step(ec.cuda(), 500, x.cuda(), generator=torch.Generator(device="cuda").manual_seed(5)).prev_sample
    b = h.step(ec.cuda(), 500, x.cuda(), generator=torch.Generator(device="cuda").manual_seed(5)).prev_sample
    c = h.step(ec.cuda(), 500, x.cuda(), generator=torch.Generator(device="cuda").manual_seed(6)).prev_sample
    assert torch.equal(a, b) and not torch.equal(a, c)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        h.step(ec, 500, x)


def test_ddim_clip_sample_and_step_clamp_bit_exact(ops):
    """diffusers' default `clip_sample=True` (x0 clamped to +-1 inside the fused kernel) against the oracle scheduler,
    bit for bit; and a step counter beyond the table re-applies the LAST row instead of reading out of bounds"""
    from oracle.scheduler import DDIMScheduler
    s = DDIMScheduler(clip_sample=True)
    s.set_timesteps(5)
    from mv_ldm_amd.scheduler import DDIMScheduler as HS
    h = HS()                                  # the mirrored defaults: clip_sample=True
    h.set_timesteps(5)
    assert h.clip_range == 1.0
    g = G(61)
    x, e = torch.randn(2, 3, 4, 8, 8, generator=g) * 2, torch.randn(2, 3, 4, 8, 8, generator=g)
    for t in s.timesteps.tolist():
        ref = s.step(e, torch.tensor(t), x).prev_sample
        got = h.step(e.cuda(), t, x.cuda()).prev_sample
        assert torch.equal(got.cpu(), ref), t
    coef = h.coefficient_table().cuda()
    xs, es = x.reshape(1, 1, -1, 1).cuda().contiguous(), e.reshape(1, 1, -1, 1).cuda().contiguous()
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    last = ops.ddim_cfg_step(es, xs, zero, None, 0.0, coef, torch.tensor([4], dtype=torch.int32, device="cuda"), None)
    past = ops.ddim_cfg_step(es, xs, zero, None, 0.0, coef, torch.tensor([11], dtype=torch.int32, device="cuda"), None)
    assert torch.equal(last, past) and torch.isfinite(past).all()


def test_ray_grid_kernel_vs_reference_golden(ops, golden):
    """mvldm_ray_encode against G3 (the reference's own get_world_rays / sample_image_grid outputs) and the NHWC slice form"""
    g = golden("g3_rays")
    extr, intr = torch.from_numpy(g["extrinsics"]).cuda(), torch.from_numpy(g["intrinsics"]).cuda()
    b, v = extr.shape[:2]
    for (h, w) in [(8, 8), (4, 6)]:
        r = ops.ray_encode(extr.view(b * v, 4, 4), intr.view(b * v, 3, 3), h, w).view(b, v, 6, h * w).permute(0, 1, 3, 2).cpu()
        o, d = torch.from_numpy(g[f"origins_{h}x{w}"]), torch.from_numpy(g[f"directions_{h}x{w}"])
        assert (r[..., :3] - o).abs().max() == 0 and (r[..., 3:] - d).abs().max() < 5e-7, (h, w)
    buf = torch.zeros(b * v + 2, 8, 8, 16, dtype=torch.bfloat16, device="cuda")
    rows = torch.tensor([b * v + 1 - i for i in range(b * v)], dtype=torch.int32, device="cuda")      # reversed, offset by 2
    ops.ray_encode(extr.view(b * v, 4, 4), intr.view(b * v, 3, 3), 8, 8, out_nhwc=buf, c_off=5, img_map=rows)
    ref = ops.ray_encode(extr.view(b * v, 4, 4), intr.view(b * v, 3, 3), 8, 8)                          # [n, 6, 8, 8] fp32
    assert torch.equal(buf[rows.long()][..., 5:11].cpu(), ref.permute(0, 2, 3, 1).to(torch.bfloat16).cpu())
    assert float(buf[..., :5].abs().max()) == 0 and float(buf[..., 11:].abs().max()) == 0 and float(buf[:2].abs().max()) == 0


def test_posterior_sample_and_mapped_layout(ops):
    g = G(62)
    mom = torch.randn(3, 8, 4, 4, generator=g) * 3
    mom[0, 4:] = 40.0
    mom[1, 4:] = -50.0                                   # exercise both clamps of logvar
    noise = torch.randn(3, 4, 4, 4, generator=g)
    ref = (mom[:, :4] + torch.exp(0.5 * mom[:, 4:].clamp(-30.0, 20.0)) * noise) * 0.18215
    got = ops.posterior_sample(mom.cuda(), noise.cuda(), 0.18215).cpu()
    assert ((got - ref).abs() <= 2e-6 * ref.abs() + 1e-30).all()
    # nchw_to_nhwc with an affine map and a destination image map (the UNet input assembly)
    img = torch.randn(3, 4, 4, 6, generator=g)
    buf = torch.zeros(5, 4, 6, 8, dtype=torch.float16, device="cuda")
    rows = torch.tensor([4, 0, 2], dtype=torch.int32, device="cuda")
    ops.nchw_to_nhwc(img.cuda(), torch.float16, dst=buf, c_off=1, scale=2.0, shift=-1.0, img_map=rows)
    assert torch.equal(buf[rows.long()][..., 1:5].cpu(), (img * 2.0 - 1.0).permute(0, 2, 3, 1).to(torch.float16))
    assert float(buf[[1, 3]].abs().max()) == 0 and float(buf[..., 0].abs().max()) == 0 and float(buf[..., 5:].abs().max()) == 0
