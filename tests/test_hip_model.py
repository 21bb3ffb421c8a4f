"""GPU parity tests at block / model / pipeline level (through the drop-in module surface and the
C-side plans) against (a) the committed golden vectors produced by the REFERENCE's own code and
(b) the CPU oracle on seeded inputs.

Tolerances (relative L2 of the output tensor vs the fp32 reference / oracle):
    f32 path  : 2e-4 per block, 1e-3 through a whole UNet forward / DDIM loop  (north star: latents
                within 1e-3 rel-err of the reference path)
    bf16 path : 3e-2 per block / forward (8-bit mantissa storage between ~300 chained ops)
    f16 path  : 4e-3 per block / forward
"""
import numpy as np
import pytest
import torch

from conftest import record_err, rel_err
from seeded import load_seeded, random_cameras

pytestmark = pytest.mark.gpu

# 16-bit tolerances = 2 x the maximum measured on MI355X (tests/golden/measured_errors_r04.json / _r05.json, written by MVLDM_TEST_REPORT: g1_block,
# g4_model, g11_*), f32 = the north-star 1e-3 or tighter: a 1 % regression of a 16-bit kernel fails these, the round numbers of rounds 1-3 did not
TOL_BLOCK = {torch.float32: 2e-4, torch.float16: 1.5e-3, torch.bfloat16: 1.2e-2}      # G1: measured 5.8e-7 / 7.2e-4 / 5.8e-3
TOL_G4 = {torch.float32: 1e-3, torch.float16: 3.7e-3, torch.bfloat16: 2.9e-2}         # G4: measured 1.6e-6 / 1.8e-3 / 1.4e-2
# round 5 (tests/golden/measured_errors_r05.json): the last round-number tolerances are gone too
TOL_MODEL = {torch.float32: 1e-3, torch.float16: 3.2e-3, torch.bfloat16: 2.5e-2}      # G5 step / sample (x 2), full topology: measured 1.5e-6 / 1.55e-3 / 1.2e-2
TOL_VAE = {torch.float32: 1e-3, torch.float16: 3.6e-3, torch.bfloat16: 2.8e-2}        # reduced-width VAE: measured 1.6e-6 / 1.76e-3 / 1.4e-2
TOL_VAE_FULL = {torch.float32: 1e-4, torch.float16: 3.2e-3, torch.bfloat16: 2.6e-2}   # SD-2.1-width decoder, one 256 x 256 view: measured 2.8e-6 / 1.59e-3 / 1.28e-2
TOL_G11 = {torch.float32: 5e-4, torch.float16: 2.3e-3, torch.bfloat16: 2.2e-2}        # whole samples; measured 1.8e-4 (the fixture's f16 storage) / 1.1e-3 / 1.06e-2
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
IDS = ["f32", "bf16", "f16"]
GRAD_ENABLED = False      # tests/conftest.py::_grad_mode: no autograd graphs in this module


@pytest.fixture(scope="module")
def M():
    import mv_ldm_amd
    from mv_ldm_amd import _lib
    _lib.load()
    return mv_ldm_amd


def sd_cfg(M, widths):
    from mv_ldm_amd.mvunet import MultiViewUNetCfg, UNet2DModelCfg
    over = dict(block_out_channels=tuple(widths), attention_head_dim=tuple(max(1, c // 64) for c in widths))
    return MultiViewUNetCfg(autoencoder=UNet2DModelCfg(block_out_channels=tuple(widths)),
                            pretrained_from="stabilityai/stable-diffusion-2-1", pretrained_overrides=over)


def scratch_cfg(M, widths):
    from mv_ldm_amd.mvunet import MultiViewUNetCfg, UNet2DModelCfg
    return MultiViewUNetCfg(autoencoder=UNet2DModelCfg(block_out_channels=tuple(widths)), pretrained_from=None)


# ------------------------------------------------------------------------------------------------ G1
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_spatial_transformer_3d_vs_reference_golden(M, golden, dtype):
    """G1: outputs of the reference's own SpatialTransformer3D (mvdream/attention.py:371-439)"""
    from mv_ldm_amd.mvunet import SpatialTransformer3D, SpatialTransformer3DCfg
    g = golden("g1_spatial_transformer_3d")
    for i in range(int(g["n"])):
        C, V, h, w, b, seed = (int(v) for v in g[f"c{i}_meta"])
        m = SpatialTransformer3D(SpatialTransformer3DCfg(num_heads=8), C)
        cs = load_seeded(m, seed)
        assert abs(cs - float(g[f"c{i}_checksum"])) <= 1e-9 * cs
        m = m.cuda()
        with M.compute_dtype(dtype):
            y = m(torch.from_numpy(g[f"c{i}_x"]).cuda())
        assert y.shape == g[f"c{i}_y"].shape
        e = record_err(f"g1_block/{str(dtype)[6:]}", rel_err(y.float().cpu(), g[f"c{i}_y"]))
        assert e < TOL_BLOCK[dtype], (i, e)


# ------------------------------------------------------------------------------------------------ G4
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_mvunet_forward_vs_reference_golden(M, golden, dtype):
    """G4: the reference's MultiViewUNet.forward walk (mvunet.py:90-208); fused plan AND the literal
    module-by-module walk."""
    from mv_ldm_amd.mvunet import MultiViewUNet
    g = golden("g4_mvunet_forward")
    ran = 0
    for i in range(int(g["n"])):
        topo, widths = str(g[f"c{i}_topology"]), [int(v) for v in g[f"c{i}_widths"]]
        if widths[0] // 8 % (4 if dtype == torch.float32 else 8):
            continue  # multi-view head dim 4: below the 16-byte chunk of the 16-bit paths
        m = MultiViewUNet(sd_cfg(M, widths) if topo == "sd" else scratch_cfg(M, widths), 11, 4)
        assert len(m.state_dict()) == int(g[f"c{i}_nkeys"])
        cs = load_seeded(m, int(g[f"c{i}_seed"]))
        assert abs(cs - float(g[f"c{i}_checksum"])) <= 1e-9 * cs
        m = m.cuda()
        x, t = torch.from_numpy(g[f"c{i}_x"]).cuda(), torch.from_numpy(g[f"c{i}_t"]).cuda()
        with M.compute_dtype(dtype):
            y = m(x, t)
            yw = m.forward_walk(x, t)
        e, ew = rel_err(y.cpu(), g[f"c{i}_y"]), rel_err(yw.cpu(), g[f"c{i}_y"])
        record_err(f"g4_model/{str(dtype)[6:]}", max(e, ew))
        assert e < TOL_G4[dtype] and ew < TOL_G4[dtype], (i, topo, e, ew)
        ran += 1
    assert ran >= 3


def test_mvunet_scratch_topology_f32(M, golden):
    """scratch UNet cases of G4 (multi-view head dim 4/8/16): fp32 path"""
    from mv_ldm_amd.mvunet import MultiViewUNet
    g = golden("g4_mvunet_forward")
    for i in range(int(g["n"])):
        if str(g[f"c{i}_topology"]) != "scratch":
            continue
        m = MultiViewUNet(scratch_cfg(M, [int(v) for v in g[f"c{i}_widths"]]), 11, 4)
        load_seeded(m, int(g[f"c{i}_seed"]))
        m = m.cuda()
        with M.compute_dtype(torch.float32):
            y = m(torch.from_numpy(g[f"c{i}_x"]).cuda(), torch.from_numpy(g[f"c{i}_t"]).cuda())
        assert record_err("forward_walk_f32", rel_err(y.cpu(), g[f"c{i}_y"])) < TOL_MODEL[torch.float32]


def test_parallel_lanes_give_the_serial_result(M, monkeypatch):
    """plans with MVLDM_OP_PAR_* lanes (opt-in: the upsamplers' phase convs and the resnet shortcuts on side streams / parallel
    hipGraph branches, each lane with its own split-K workspace) compute exactly what the serial plan computes -- eagerly and
    through graph replay"""
    from mv_ldm_amd import _lib as L
    from mv_ldm_amd.mvunet import MultiViewUNet
    m = MultiViewUNet(sd_cfg(M, (64, 128, 256, 256)), 11, 4)
    load_seeded(m, 77)
    m = m.cuda()
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 11, 16, 16, generator=gen).cuda()
    t = torch.tensor([[0, 500, 500], [0, 20, 20]]).cuda()
    outs = {}
    monkeypatch.setenv("MVLDM_AUTOTUNE", "0")       # (lanes own smaller split-K workspaces: another tuning signature, possibly another tile -- the rules do not care)
    for rows in ("0", "16384"):
        monkeypatch.setenv("MVLDM_PAR_ROWS", rows)
        m._plans.clear()
        with M.compute_dtype(torch.bfloat16):
            y = m(x, t).float().clone()
            assert torch.equal(y, m(x, t).float())          # replay of the captured graph
            eager = m.compile(2, 3, 16, 16)["plan"]
            eager.run()                                     # the same plan launched op by op (real side streams)
            torch.cuda.synchronize()
            assert torch.equal(m.compile(2, 3, 16, 16)["out"].view_as(y), y)
        outs[rows] = y
        kinds = [mm.kind for mm in eager.meta]
        assert (L.OP_PAR_BEGIN in kinds) == (rows != "0")
        if rows != "0":
            # a range cut INSIDE a group is refused before anything is launched (it used to be noticed after the lanes
            # had been enqueued on side streams, with no join recorded)
            i0 = kinds.index(L.OP_PAR_BEGIN)
            i1 = kinds.index(L.OP_PAR_END, i0)
            before = m.compile(2, 3, 16, 16)["out"].clone()
            for first, last in ((0, i0 + 1), (i0 + 1, len(kinds)), (i0 + 1, i1)):
                with pytest.raises(L.MvldmError, match="parallel group"):
                    eager.run(first, last)
            torch.cuda.synchronize()
            assert torch.equal(m.compile(2, 3, 16, 16)["out"], before)      # nothing ran
    assert torch.equal(outs["0"], outs["16384"])


# ------------------------------------------------------------------------------------------------ G5
def _pipeline(M, g, p, dtype):
    from mv_ldm_amd.mvunet import MultiViewUNet
    from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
    from mv_ldm_amd.scheduler import DDIMScheduler
    from mv_ldm_amd.vae import AutoencoderKL
    widths = [int(v) for v in g["widths"]]
    den = MultiViewUNet(sd_cfg(M, widths), 11, 4)
    vae = AutoencoderKL.from_pretrained("x", config_overrides=dict(block_out_channels=tuple(int(v) for v in g["vae_widths"]),
                                                                   layers_per_block=1))
    assert abs(load_seeded(den, 400) - float(g[p + "checksum_denoiser"])) < 1e-6
    assert abs(load_seeded(vae, 401) - float(g[p + "checksum_vae"])) < 1e-6
    sch = DDIMScheduler(clip_sample=False)
    pipe = MVLDMPipeline(den.cuda(), vae.cuda(), sch, SamplerCfg(use_cfg=bool(g[p + "use_cfg"]), cfg_scale=3.0,
                                                                 num_inference_steps=5))
    pipe.set_timesteps(5)
    return pipe


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_step_and_sample_vs_reference_golden(M, golden, dtype):
    """G5: DiffusionWrapper.step / .sample of the reference (diffusion_wrapper.py:413-490), 5 DDIM
    steps, with and without classifier-free guidance, explicit noise."""
    from mv_ldm_amd.pipeline import ray_encode
    g = golden("g5_step_sample")
    for ci in range(int(g["n"])):
        p = f"c{ci}_"
        with M.compute_dtype(dtype):
            pipe = _pipeline(M, g, p, dtype)
            extr, intr = torch.from_numpy(g[p + "extr"]), torch.from_numpy(g[p + "intr"])
            v_c = g[p + "ctx_img"].shape[1]
            batch = {"context": {"image": torch.from_numpy(g[p + "ctx_img"]), "extrinsics": extr[:, :v_c], "intrinsics": intr[:, :v_c]},
                     "target": {"extrinsics": extr[:, v_c:], "intrinsics": intr[:, v_c:]}}
            img, x0 = pipe.sample(batch, x_T=torch.from_numpy(g[p + "x_T"]), encode_noise=torch.from_numpy(g[p + "enc_noise"]))
            e_img = rel_err(img.cpu(), g[p + "img"])
            # the literal two-forward step of the reference on the drop-in modules
            ctx_lat, x_t = torch.from_numpy(g[p + "step_ctx_lat"]).cuda(), torch.from_numpy(g[p + "step_x_t"]).cuda()
            hl = x_t.shape[-1]
            rays = ray_encode(extr[:, :v_c], intr[:, :v_c], extr[:, v_c:], intr[:, v_c:], hl, hl)
            assert rel_err(rays.cpu(), g[p + "rays"]) < 2e-6      # the HIP ray kernel (adjugate K^-1 in fp32) vs the reference
            ctx_in = torch.cat([ctx_lat, torch.zeros_like(ctx_lat[:, :, :1])], dim=2)
            x_prev = pipe.step(pipe.denoiser, x_t, torch.tensor(int(g[p + "step_ts"])), ctx_in, rays,
                               torch.ones_like(x_t[:, :, :1]))
            e_step = rel_err(x_prev.cpu(), g[p + "step_x_prev"])
        record_err(f"g5_step/{str(dtype)[6:]}", e_step)
        record_err(f"g5_sample_img/{str(dtype)[6:]}", e_img)
        assert e_step < TOL_MODEL[dtype], (ci, "step", e_step)
        assert e_img < 2 * TOL_MODEL[dtype], (ci, "sample", e_img)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_configs0_and_configs4_shapes_vs_reference_golden(M, golden, dtype):
    """G11 = the reference's own DiffusionWrapper.sample at the geometry of BASELINE.json configs[0] (1 ctx + 1 tgt view, 64x64 images,
    8x8 latents, 5 DDIM steps: the literal CPU-reference case) and of configs[4] (1 + 8 = 9 views, 64x64 LATENTS, above the
    `h <= 32` gate of mvunet.py:137,190, so the level-0 multi-view blocks are skipped and the deeper 3-D attentions see 9 views),
    reduced width, through the fused sampler AND the literal loop (`sample_literal`)"""
    from mv_ldm_amd.mvunet import MultiViewUNet
    from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
    from mv_ldm_amd.scheduler import DDIMScheduler
    from mv_ldm_amd.vae import AutoencoderKL
    g = golden("g11_configs")
    widths = [int(v) for v in g["widths"]]
    for ci in range(int(g["n"])):
        p = f"c{ci}_"
        n_steps = int(g[p + "n_steps"])
        with M.compute_dtype(dtype):
            den = MultiViewUNet(sd_cfg(M, widths), 11, 4)
            vae = AutoencoderKL.from_pretrained("x", config_overrides=dict(block_out_channels=tuple(int(v) for v in g[p + "vae_widths"]), layers_per_block=1))
            assert abs(load_seeded(den, 400) - float(g[p + "checksum_denoiser"])) < 1e-6
            assert abs(load_seeded(vae, 401) - float(g[p + "checksum_vae"])) < 1e-6
            pipe = MVLDMPipeline(den.cuda(), vae.cuda(), DDIMScheduler(clip_sample=False), SamplerCfg(True, 3.0, n_steps))
            pipe.set_timesteps(n_steps)
            extr, intr = torch.from_numpy(g[p + "extr"]), torch.from_numpy(g[p + "intr"])
            v_c = g[p + "ctx_img"].shape[1]
            batch = {"context": {"image": torch.from_numpy(g[p + "ctx_img"]), "extrinsics": extr[:, :v_c], "intrinsics": intr[:, :v_c]},
                     "target": {"extrinsics": extr[:, v_c:], "intrinsics": intr[:, v_c:]}}
            kw = dict(x_T=torch.from_numpy(g[p + "x_T"]), encode_noise=torch.from_numpy(g[p + "enc_noise"]))
            img, _ = pipe.sample(batch, **kw)
            img_l, _ = pipe.sample_literal(batch, **kw)
        want = torch.from_numpy(g[p + "img"].astype("float32"))
        e, el = rel_err(img.cpu(), want), rel_err(img_l.cpu(), want)
        record_err(f"g11_c{ci}/{str(dtype)[6:]}", max(e, el))
        # (f32: the fixture's own f16 storage is 2.4e-4 absolute on [0, 1]; 16-bit: 2 x the measured maximum, tests/golden/measured_errors_r04.json)
        assert max(e, el) < TOL_G11[dtype], (ci, e, el)


def test_ddpm_ancestral_sampling_vs_oracle(M):
    """`SCHEDULER["ddpm"]` through `sample()`: the reference's loop (diffusion_wrapper.py:455-490) with diffusers' DDPMScheduler --
    5 ancestral steps with CFG, explicit x_T and per-step variance noise -- on the HIP path (f32) against `oracle.pipeline.sample`
    with the restated DDPMScheduler consuming the same noise; latents within the north-star 1e-3"""
    from mv_ldm_amd.mvunet import MultiViewUNet
    from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
    from mv_ldm_amd.scheduler import DDPMScheduler
    from mv_ldm_amd.vae import AutoencoderKL
    from oracle import multiview as MV
    from oracle import pipeline as OPL
    from oracle.scheduler import DDPMScheduler as OracleDDPM
    from oracle.vae import AutoencoderKL as OracleVAE
    widths = (64, 64, 128, 128)
    over = dict(block_out_channels=widths, attention_head_dim=tuple(max(1, c // 64) for c in widths))
    o = MV.MultiViewUNet(MV.MVUNetCfg(autoencoder=MV.UNetCfg(block_out_channels=widths), pretrained_from="stabilityai/stable-diffusion-2-1",
                                      pretrained_overrides=over), 11, 4).eval()
    ovae = OracleVAE(block_out_channels=(32, 32, 64, 64), layers_per_block=1).eval()
    den = MultiViewUNet(sd_cfg(M, widths), 11, 4)
    vae = AutoencoderKL.from_pretrained("x", config_overrides=dict(block_out_channels=(32, 32, 64, 64), layers_per_block=1))
    load_seeded(o, 410), load_seeded(den, 410), load_seeded(ovae, 411), load_seeded(vae, 411)
    g = torch.Generator().manual_seed(9)
    b, v_c, v_t, res = 1, 1, 2, 64
    ctx_img = torch.rand(b, v_c, 3, res, res, generator=g)
    extr, intr = random_cameras(b, v_c + v_t, seed=17)
    x_T = torch.randn(b, v_t, 4, res // 8, res // 8, generator=g)
    enc_noise = torch.randn(b * v_c, 4, res // 8, res // 8, generator=g)
    z = torch.randn(5, b, v_t, 4, res // 8, res // 8, generator=g)

    class Queued(OracleDDPM):           # the oracle scheduler fed with the same per-step draws
        def step(self, model_output, timestep, sample, **kw):
            i = self.timesteps.tolist().index(int(timestep))
            return super().step(model_output, timestep, sample, variance_noise=z[i])

    osch = Queued(clip_sample=False)
    osch.set_timesteps(5)
    with torch.no_grad():
        _, want = OPL.sample(o, ovae, osch, ctx_img, extr[:, :v_c], intr[:, :v_c], extr[:, v_c:], intr[:, v_c:], x_T=x_T,
                             encode_noise=enc_noise, decode=False)
    sch = DDPMScheduler(clip_sample=False)
    pipe = MVLDMPipeline(den.cuda(), vae.cuda(), sch, SamplerCfg(True, 3.0, 5))
    pipe.set_timesteps(5)
    batch = {"context": {"image": ctx_img, "extrinsics": extr[:, :v_c], "intrinsics": intr[:, :v_c]},
             "target": {"extrinsics": extr[:, v_c:], "intrinsics": intr[:, v_c:]}}
    with M.compute_dtype(torch.float32):
        img, x0 = pipe.sample(batch, x_T=x_T, encode_noise=enc_noise, step_noise=z.cuda())
    e = rel_err(x0.cpu(), want)
    print(f"DDPM 5-step ancestral sample, f32: latent rel-err {e:.3e}")
    assert e < 1e-3 and torch.isfinite(img).all() and img.shape == (b, v_t, 3, res, res)
    # a different draw gives a different sample (the noise really enters)
    with M.compute_dtype(torch.float32):
        _, x1 = pipe.sample(batch, x_T=x_T, encode_noise=enc_noise, step_generator=torch.Generator(device="cuda").manual_seed(1), decode=False)
    assert rel_err(x1.cpu(), want) > 1e-2


# ------------------------------------------------------------------------------------------------ VAE vs oracle
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_vae_decode_encode_vs_oracle(M, dtype):
    from mv_ldm_amd.vae import AutoencoderKL
    from oracle.vae import AutoencoderKL as OracleVAE
    over = dict(block_out_channels=(32, 64, 64), layers_per_block=2)
    v, o = AutoencoderKL.from_pretrained("x", config_overrides=over), OracleVAE.from_pretrained("x", config_overrides=over).eval()
    load_seeded(v, 7)
    load_seeded(o, 7)
    v = v.cuda()
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(1))
    img = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(2)) * 2 - 1
    with M.compute_dtype(dtype):
        dec = v.decode(z.cuda()).sample
        enc = v.encode(img.cuda()).latent_dist
    assert dec.shape == (2, 3, 32, 32)
    n = str(dtype)[6:]
    assert record_err(f"vae_small_decode/{n}", rel_err(dec.cpu(), o.decode(z).sample)) < TOL_VAE[dtype]
    ref = o.encode(img).latent_dist
    assert record_err(f"vae_small_encode_mean/{n}", rel_err(enc.mean.cpu(), ref.mean)) < TOL_VAE[dtype]
    assert record_err(f"vae_small_encode_std/{n}", rel_err(enc.std.cpu(), ref.std)) < TOL_VAE[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_full_width_vae_decode_of_one_256x256_view_vs_oracle(M, dtype):
    """`last_stage_decode` (diffusion_wrapper.py:289-298) at the released SD-2.1 VAE widths (128 / 256 / 512 / 512, mid-block attention with
    one 512-wide head): ONE 32 x 32 latent -> 256 x 256 image against the fp32 CPU oracle (0.7 s there).  f32 within the north-star 1e-3,
    16-bit at 2 x the error measured on MI355X (the reduced-width case above cannot see a full-width tile choice or the 512-wide head)."""
    from mv_ldm_amd.vae import AutoencoderKL
    from oracle.vae import AutoencoderKL as OracleVAE
    v, o = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1", allow_random_init=True), OracleVAE.from_pretrained("x").eval()
    load_seeded(v, 9)
    load_seeded(o, 9)
    assert sum(p.numel() for p in v.parameters()) == sum(p.numel() for p in o.parameters()) == 83_653_863
    v = v.cuda()
    z = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(4)) / 0.18215 * 0.2
    ref = o.decode(z).sample
    with M.compute_dtype(dtype):
        dec = v.decode(z.cuda()).sample
    assert dec.shape == (1, 3, 256, 256) and torch.isfinite(dec).all()
    e = record_err(f"vae_full_decode_256/{str(dtype)[6:]}", rel_err(dec.cpu(), ref))
    print(f"full-width VAE decode of one 256x256 view [{dtype}]: rel-err {e:.3e}")
    assert e < TOL_VAE_FULL[dtype], e


# ------------------------------------------------------------------------------------------------ real topology
def test_full_sd21_topology_bf16_vs_oracle(M):
    """The real config-2 network (SD-2.1 widths 320/640/1280/1280, heads 5/10/20/20 x 64, multi-view heads
    8 x 40/80/160; 1.07 B parameters) at a small latent (16x16, 3 views: 1 context + 2 targets):
    fused bf16 plan vs the fp32 CPU oracle."""
    from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg, UNet2DModelCfg
    from oracle import multiview as MV
    cfg = MultiViewUNetCfg(autoencoder=UNet2DModelCfg(), pretrained_from="stabilityai/stable-diffusion-2-1")
    m = MultiViewUNet(cfg, 11, 4)
    o = MV.MultiViewUNet(MV.MVUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1"), 11, 4).eval()
    assert sorted(m.state_dict()) == sorted(o.state_dict())
    sd = __import__("seeded").seeded_state(o, 11)
    o.load_state_dict(sd)
    m.load_state_dict(sd)
    del sd
    m = m.cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, 11, 16, 16, generator=g)
    t = torch.tensor([[0, 500, 500]])
    ref = o(x, t)
    with M.compute_dtype(torch.bfloat16):
        y = m(x.cuda(), t.cuda())
    e = record_err("full_sd21_topology_16x16/bfloat16", rel_err(y.cpu(), ref))
    assert e < TOL_MODEL[torch.bfloat16], e


def test_install_fused_sampler_hook_on_a_reference_style_wrapper(M):
    """`pipeline.install_fused_sampler(wrapper)`: ONE call re-points `wrapper.sample` (diffusion_wrapper.py:455-490) at the fused sampler
    without touching the wrapper's class: same images as `MVLDMPipeline.sample` for the same CPU-generator noise, the reference's
    `(images, batch)` return contract, live `model_cfg` (CFG scale / use_cfg changes are picked up), a foreign denoiser is refused."""
    from types import SimpleNamespace
    from mv_ldm_amd.mvunet import MultiViewUNet
    from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg, install_fused_sampler
    from mv_ldm_amd.scheduler import DDIMScheduler
    from mv_ldm_amd.vae import AutoencoderKL
    den = MultiViewUNet(scratch_cfg(M, (32, 64, 64, 64)), 11, 4)
    vae = AutoencoderKL.from_pretrained("x", config_overrides=dict(block_out_channels=(32, 64, 64), layers_per_block=1))
    load_seeded(den, 31)
    load_seeded(vae, 32)
    den, vae = den.cuda(), vae.cuda()
    sch = DDIMScheduler()
    sch.set_timesteps(4)
    g = torch.Generator().manual_seed(8)
    ext, intr = random_cameras(1, 3, 5)
    batch = {"context": {"image": torch.rand(1, 1, 3, 32, 32, generator=g), "extrinsics": ext[:, :1], "intrinsics": intr[:, :1]},
             "target": {"image": torch.rand(1, 2, 3, 32, 32, generator=g), "extrinsics": ext[:, 1:], "intrinsics": intr[:, 1:]}}
    wrapper = SimpleNamespace(model_cfg=SimpleNamespace(use_cfg=True, cfg_scale=3.0, use_ema_sampling=False), denoiser=den, autoencoder=vae,
                              scheduler=sch, ema=None)
    with M.compute_dtype(torch.float32):
        pipe = install_fused_sampler(wrapper)
        torch.manual_seed(5)
        img, same = wrapper.sample(batch)
        assert same is batch and img.shape == (1, 2, 3, 32, 32) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0
        ref = MVLDMPipeline(den, vae, sch, SamplerCfg(True, 3.0, 4))
        torch.manual_seed(5)
        want, _ = ref.sample({"context": batch["context"], "target": batch["target"]})
        assert torch.equal(img, want)
        wrapper.model_cfg.cfg_scale = 1.0                      # read at every call, like the reference's step()
        torch.manual_seed(5)
        img1, _ = wrapper.sample(batch)
        assert pipe.cfg.cfg_scale == 1.0 and not torch.equal(img1, img)
        # the wrapper's ray-encoding switches are read too (diffusion_wrapper.py:301-320): Pluecker origins keep the channel count, so a
        # hook that ignored `use_plucker` would sample silently different images than the reference wrapper
        from mv_ldm_amd.pipeline import RayEncodingCfg
        wrapper.model_cfg.use_plucker = True
        torch.manual_seed(5)
        img_p, _ = wrapper.sample(batch)
        ref_p = MVLDMPipeline(den, vae, sch, SamplerCfg(True, 1.0, 4), rays=RayEncodingCfg(use_plucker=True))
        torch.manual_seed(5)
        want_p, _ = ref_p.sample({"context": batch["context"], "target": batch["target"]})
        assert pipe.rays.use_plucker and torch.equal(img_p, want_p) and not torch.equal(img_p, img1)
        wrapper.model_cfg.use_plucker = False
        # config/main.yaml's default `use_ray_encoding: true` needs a denoiser with 4 + 1 + 6 * 15 + 6 * 15 input channels: refused loudly
        wrapper.model_cfg.use_ray_encoding = True
        wrapper.model_cfg.ray_encodings = SimpleNamespace(num_origin_octaves=15, num_direction_octaves=15)
        with pytest.raises(ValueError):
            wrapper.sample(batch)
        with pytest.raises(ValueError):
            install_fused_sampler(SimpleNamespace(model_cfg=wrapper.model_cfg, denoiser=den, autoencoder=vae, scheduler=sch, ema=None))
        wrapper.model_cfg.use_ray_encoding = False
    with pytest.raises(TypeError):
        install_fused_sampler(SimpleNamespace(model_cfg=wrapper.model_cfg, denoiser=torch.nn.Linear(2, 2), autoencoder=vae, scheduler=sch, ema=None))
