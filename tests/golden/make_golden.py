"""Generate the committed golden vectors by running the REFERENCE's own code (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

What is executed from /root/reference (imported in place, never copied; see ref_import.py):
  G1  src.model.denoiser.mvdream.attention.SpatialTransformer3D.forward
  G2  src.model.denoiser.mvdream.attention.CrossAttention.forward
  G3  src.geometry.projection.{sample_image_grid,get_world_rays}, src.misc.camera_utils.absolute_to_relative_camera
  G4  src.model.denoiser.mvunet.MultiViewUNet.{__init__,forward}  (walking the oracle's diffusers blocks)
  G5  src.model.diffusion_wrapper.DiffusionWrapper.{step,sample,ray_encode,first_stage_encode,last_stage_decode}
  G7  src.model.diffusion_wrapper.DiffusionWrapper.{test_video_anchored,test_video_autoregressive} bookkeeping
      with `sample()` stubbed out (index schedules only)
G6 (DDIM tables / one step KAT) comes from the oracle's own scheduler: diffusers is absent, so that
arithmetic is "parity unpinned" by the reference; the fixture pins it against drift.

Weights are regenerated from seeds (seeded.py); inputs and outputs are stored explicitly (float32).
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))
sys.dont_write_bytecode = True

import ref_import as R  # noqa: E402
from seeded import checksum, load_seeded, random_cameras, seeded_state  # noqa: E402

torch.set_num_threads(8)
torch.set_grad_enabled(False)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(f"wrote {name}.npz  ({sum(a.nbytes for a in out.values()) / 1e3:.0f} kB raw)")


# ----------------------------------------------------------------------------------------------- G1 / G2
def g1_g2():
    A = R.ref("src.model.denoiser.mvdream.attention")
    cases = {}
    for i, (C, V, h, w, b, seed) in enumerate([(64, 1, 4, 4, 1, 0), (64, 2, 8, 8, 2, 1), (64, 5, 4, 8, 1, 2),
                                                (320, 2, 4, 4, 1, 3), (320, 5, 8, 8, 1, 4)]):
        cfg = A.SpatialTransformer3DCfg(name="spatial_transformer_3d", num_heads=8)
        m = A.SpatialTransformer3D(cfg, C).eval()
        cs = load_seeded(m, 100 + seed)
        x = torch.randn(b, V, C, h, w, generator=torch.Generator().manual_seed(seed))
        y = m(x)
        cases[f"c{i}_meta"] = np.array([C, V, h, w, b, 100 + seed])
        cases[f"c{i}_checksum"] = cs
        cases[f"c{i}_x"], cases[f"c{i}_y"] = x, y
    cases["n"] = 5
    save("g1_spatial_transformer_3d", **cases)

    cases = {}
    for i, (heads, d, L, bsz, seed) in enumerate([(2, 40, 24, 2, 0), (2, 80, 17, 1, 1), (2, 160, 33, 1, 2),
                                                   (5, 64, 16, 3, 3)]):
        m = A.CrossAttention(query_dim=heads * d, heads=heads, dim_head=d).eval()
        cs = load_seeded(m, 200 + seed)
        x = torch.randn(bsz, L, heads * d, generator=torch.Generator().manual_seed(seed))
        cases[f"c{i}_meta"] = np.array([heads, d, L, bsz, 200 + seed])
        cases[f"c{i}_checksum"] = cs
        cases[f"c{i}_x"], cases[f"c{i}_y"] = x, m(x)
    cases["n"] = 4
    save("g2_cross_attention", **cases)


# ----------------------------------------------------------------------------------------------- G3
def g3():
    P = R.ref("src.geometry.projection")
    Cm = R.ref("src.misc.camera_utils")
    from einops import rearrange
    extr, intr = random_cameras(2, 4, seed=7)
    # make the first camera non-identity too so that inv() is exercised
    extr = extr[:, [1, 0, 2, 3]]
    out = {}
    for (h, w) in [(8, 8), (4, 6)]:
        xy, _ = P.sample_image_grid((h, w))
        o, d = P.get_world_rays(rearrange(xy, "h w xy -> (h w) xy"),
                                rearrange(extr, "b v i j -> b v () i j"),
                                rearrange(intr, "b v i j -> b v () i j"))
        out[f"xy_{h}x{w}"], out[f"origins_{h}x{w}"], out[f"directions_{h}x{w}"] = xy, o, d
    out["extrinsics"], out["intrinsics"] = extr, intr
    for idx in (0, 1, 3):
        out[f"relative_{idx}"] = Cm.absolute_to_relative_camera(extr, idx)
    save("g3_rays", **out)


# ----------------------------------------------------------------------------------------------- G4
SD_TINY = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4))


def _ref_mvunet(topology: str, widths):
    """Build the REFERENCE MultiViewUNet (its __init__ runs, mvunet.py:43-88)."""
    R.install()
    import diffusers
    M = R.ref("src.model.denoiser.mvunet")
    A = R.ref("src.model.denoiser.mvdream.attention")
    mv = A.SpatialTransformer3DCfg(name="spatial_transformer_3d", num_heads=8)
    if topology == "scratch":
        ae = M.UNet2DModelCfg(name="unet", down_block_types=["DownBlock2D"] * 4, mid_block_type="UNetMidBlock2D",
                              up_block_types=["UpBlock2D"] * 4, only_cross_attention=False,
                              block_out_channels=list(widths))
        cfg = M.MultiViewUNetCfg(name="mv_unet", autoencoder=ae, multi_view_attention=mv, pretrained_from=None)
        return M.MultiViewUNet(cfg, 11, 4).eval()
    ae = M.UNet2DModelCfg(name="unet", down_block_types=[], mid_block_type="", up_block_types=[],
                          only_cross_attention=False, block_out_channels=list(widths))
    cfg = M.MultiViewUNetCfg(name="mv_unet", autoencoder=ae, multi_view_attention=mv,
                             pretrained_from="stabilityai/stable-diffusion-2-1")
    orig = diffusers.UNet2DConditionModel.from_pretrained
    over = dict(block_out_channels=tuple(widths), attention_head_dim=tuple(max(1, c // 64) for c in widths))
    diffusers.UNet2DConditionModel.from_pretrained = classmethod(
        lambda cls, path, subfolder="unet": orig.__func__(cls, path, subfolder, config_overrides=over))
    try:
        return M.MultiViewUNet(cfg, 11, 4).eval()
    finally:
        diffusers.UNet2DConditionModel.from_pretrained = orig


def g4():
    out = {}
    cases = [("scratch", (32, 64, 128, 128), 1, 2, 8, "2d", 0),
             ("scratch", (32, 64, 128, 128), 1, 3, 16, "1d", 1),
             ("sd", (64, 128, 256, 256), 1, 2, 8, "2d", 2),
             ("sd", (64, 128, 256, 256), 2, 3, 8, "2d", 3),
             ("sd", (64, 128, 256, 256), 1, 1, 64, "2d", 4),   # 64x64: exercises the h<=32 gate
             ]
    for i, (topo, widths, b, V, h, tform, seed) in enumerate(cases):
        m = _ref_mvunet(topo, widths)
        cs = load_seeded(m, 300 + seed)
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(b, V, 11, h, h, generator=g)
        t = torch.randint(0, 1000, (b, V) if tform == "2d" else (b,), generator=g)
        if tform == "2d":
            t[:, 0] = 0
        with R.cpu_cuda():
            y = m.forward(x, t)
        out[f"c{i}_topology"] = topo
        out[f"c{i}_widths"] = np.array(widths)
        out[f"c{i}_seed"] = 300 + seed
        out[f"c{i}_checksum"] = cs
        out[f"c{i}_nkeys"] = len(m.state_dict())
        out[f"c{i}_x"], out[f"c{i}_t"], out[f"c{i}_y"] = x, t, y
        print(f"  g4 case {i}: {topo} {widths} b{b} V{V} h{h} -> |y| {y.abs().mean():.4f}")
    out["n"] = len(cases)
    save("g4_mvunet_forward", **out)


# ----------------------------------------------------------------------------------------------- G5
def _ref_wrapper(use_cfg: bool, widths=(64, 128, 256, 256), vae_widths=(32, 64), n_steps=5):
    """Build the REFERENCE DiffusionWrapper (its __init__ runs, diffusion_wrapper.py:71-150) on the
    SD-like topology at reduced widths."""
    R.install()
    import diffusers
    W = R.ref("src.model.diffusion_wrapper")
    M = R.ref("src.model.denoiser.mvunet")
    A = R.ref("src.model.denoiser.mvdream.attention")
    S = R.ref("src.model.scheduler")
    Sd = R.ref("src.model.scheduler.ddim")
    mv = A.SpatialTransformer3DCfg(name="spatial_transformer_3d", num_heads=8)
    ae = M.UNet2DModelCfg(name="unet", down_block_types=[], mid_block_type="", up_block_types=[],
                          only_cross_attention=False, block_out_channels=list(widths))
    den = M.MultiViewUNetCfg(name="mv_unet", autoencoder=ae, multi_view_attention=mv,
                             pretrained_from="stabilityai/stable-diffusion-2-1")
    sched = S.SchedulerCfg(name="ddim", num_train_timesteps=1000, num_inference_steps=n_steps, pretrained_from=None,
                           kwargs=Sd.DDIMSchedulerCfg(clip_sample=False))
    vae_cfg = SimpleNamespace(name="kl", pretrained_from="stabilityai/stable-diffusion-2-1",
                              kwargs=SimpleNamespace(latent_channels=4))
    model_cfg = SimpleNamespace(denoiser=den, scheduler=sched, autoencoder=vae_cfg,
                                ray_encodings=SimpleNamespace(num_origin_octaves=15, num_direction_octaves=15),
                                use_cfg=use_cfg, cfg_scale=3.0, cfg_train=True, use_ray_encoding=False,
                                srt_ray_encoding=False, use_plucker=False, ema=False, use_ema_sampling=False,
                                enable_xformers_memory_efficient_attention=False)
    o_unet = diffusers.UNet2DConditionModel.from_pretrained
    o_vae = diffusers.AutoencoderKL.from_pretrained
    over = dict(block_out_channels=tuple(widths), attention_head_dim=tuple(max(1, c // 64) for c in widths))
    vover = dict(block_out_channels=tuple(vae_widths), layers_per_block=1)
    diffusers.UNet2DConditionModel.from_pretrained = classmethod(
        lambda cls, path, subfolder="unet": o_unet.__func__(cls, path, subfolder, config_overrides=over))
    diffusers.AutoencoderKL.from_pretrained = classmethod(
        lambda cls, path, subfolder="vae": o_vae.__func__(cls, path, subfolder, config_overrides=vover))
    try:
        w = W.DiffusionWrapper(model_cfg, SimpleNamespace(), SimpleNamespace(), SimpleNamespace(),
                               SimpleNamespace(denoiser=False, autoencoder=True), None, None)
    finally:
        diffusers.UNet2DConditionModel.from_pretrained = o_unet
        diffusers.AutoencoderKL.from_pretrained = o_vae
    return w.eval()


def g5():
    out = {}
    # (use_cfg, b, v_c, v_t): cases 2, 3 are the (2 context + 3 target) calls that make up 25 of the 26 sample() calls of an
    # anchored / autoregressive sequence (diffusion_wrapper.py:841-902, 961-1040), the second with two scenes per call
    cases = [(False, 1, 1, 2), (True, 1, 1, 2), (True, 1, 2, 3), (True, 2, 2, 3)]
    for ci, (use_cfg, b, v_c, v_t) in enumerate(cases):
        w = _ref_wrapper(use_cfg)
        cs_d = load_seeded(w.denoiser, 400)
        cs_v = load_seeded(w.autoencoder, 401)
        H = 32                                # VAE widths (32,64): one downsample -> latents 16x16
        g = torch.Generator().manual_seed(5 + ci)
        ctx_img = torch.rand(b, v_c, 3, H, H, generator=g)
        extr, intr = random_cameras(b, v_c + v_t, seed=11 + ci)
        hl = H // 2
        enc_noise = torch.randn(b * v_c, 4, hl, hl, generator=g)
        x_T = torch.randn(b, v_t, 4, hl, hl, generator=g)
        batch = {"context": {"image": ctx_img, "extrinsics": extr[:, :v_c], "intrinsics": intr[:, :v_c]},
                 "target": {"image": torch.zeros(b, v_t, 3, H, H), "extrinsics": extr[:, v_c:],
                            "intrinsics": intr[:, v_c:]},
                 "scene": ["synthetic"] * b}
        w.set_timesteps(5)
        # the reference draws its noise from the global CPU generator (diffusion_wrapper.py:283,473):
        # feed it the recorded draws by patching randn for the duration of the call
        draws = [enc_noise, x_T]
        o_randn = torch.randn
        import oracle.vae as ovae

        def fake_randn(*shape, **kw):
            t = draws.pop(0)
            shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
            assert tuple(t.shape) == shp, (t.shape, shp)
            return t.clone()
        torch.randn = fake_randn
        try:
            with R.cpu_cuda():
                img, _ = w.sample(batch)
        finally:
            torch.randn = o_randn
        assert not draws
        # one isolated step as well (diffusion_wrapper.py:413-453)
        ctx_lat = torch.randn(b, v_c, 4, hl, hl, generator=g)
        x_t = torch.randn(b, v_t, 4, hl, hl, generator=g)
        with R.cpu_cuda():
            rays = w.ray_encode(batch, ctx_lat, x_t)
            ctx_in = torch.cat([ctx_lat, torch.zeros(b, v_c, 1, hl, hl)], dim=2)
            x_prev = w.step(w.denoiser, x_t, w.scheduler.timesteps[1], ctx_in, rays, torch.ones(b, v_t, 1, hl, hl))
        p = f"c{ci}_"
        out.update({p + "use_cfg": int(use_cfg), p + "checksum_denoiser": cs_d, p + "checksum_vae": cs_v,
                    p + "ctx_img": ctx_img, p + "extr": extr, p + "intr": intr, p + "enc_noise": enc_noise,
                    p + "x_T": x_T, p + "img": img, p + "rays": rays, p + "step_ctx_lat": ctx_lat,
                    p + "step_x_t": x_t, p + "step_ts": int(w.scheduler.timesteps[1]), p + "step_x_prev": x_prev})
        print(f"  g5 cfg={use_cfg} b={b} v_c={v_c} v_t={v_t}: img mean {img.mean():.4f}")
    out["n"] = len(cases)
    out["widths"] = np.array([64, 128, 256, 256])
    out["vae_widths"] = np.array([32, 64])
    save("g5_step_sample", **out)


# ----------------------------------------------------------------------------------------------- G6
def g6():
    from oracle.scheduler import DDIMScheduler
    s = DDIMScheduler(clip_sample=False)
    out = {"alphas_cumprod": s.alphas_cumprod, "betas": s.betas}
    for n in (5, 25, 50, 70):
        s.set_timesteps(n)
        out[f"timesteps_{n}"] = s.timesteps
    s.set_timesteps(50)
    g = torch.Generator().manual_seed(6)
    x, e = torch.randn(2, 3, 4, 8, 8, generator=g), torch.randn(2, 3, 4, 8, 8, generator=g)
    out["kat_x"], out["kat_eps"] = x, e
    for t in (980, 500, 0):
        out[f"kat_prev_{t}"] = s.step(e, torch.tensor(t), x).prev_sample
    out["kat_add_noise"] = s.add_noise(x, e, torch.tensor([10, 900]))
    save("g6_ddim", **out)


# ----------------------------------------------------------------------------------------------- G9
G9_FULL = ("unet.conv_in.weight", "unet.conv_out.weight", "unet.conv_out.bias", "unet.time_embedding.linear_1.weight",
           "unet.down_blocks.0.resnets.0.time_emb_proj.weight", "unet.down_blocks.0.resnets.0.norm1.weight",
           "unet.down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight",
           "unet.down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.bias",
           "unet.down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_out.0.bias",
           "unet.down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.bias",
           "unet.down_blocks.1.downsamplers.0.conv.weight", "unet.mid_block.resnets.1.conv2.bias",
           "unet.up_blocks.2.resnets.0.conv_shortcut.weight", "unet.up_blocks.1.upsamplers.0.conv.weight",
           "unet.up_blocks.3.resnets.2.norm1.bias", "cross_attn_blocks_encoder.0.transformer_blocks.0.attn1.to_k.weight",
           "cross_attn_blocks_mid.0.transformer_blocks.0.norm2.weight", "cross_attn_blocks_decoder.3.proj_out.weight",
           "cross_attn_blocks_decoder.2.transformer_blocks.0.ff.net.2.weight")


def g9():
    """`DiffusionWrapper.training_step` of the REFERENCE (diffusion_wrapper.py:324-411) + autograd, with its random draws
    (context count :336, view choice :225/:235, relative-pose coin :346, noise :362, timesteps :363, CFG coin :381, the VAE
    posterior noise) replaced by recorded values.  Stored: the inputs, the draws, the loss, the L2 norm of EVERY parameter
    gradient and a handful of gradients in full."""
    import numpy as onp
    out = {}
    # (b, index draw, second randint draw, relative_pose coin, unconditional coin)
    cases = [(2, 2, 1, False, False), (1, 1, 1, True, False), (2, 2, 0, False, True)]
    for ci, (b, index, second, rel_coin, unc_coin) in enumerate(cases):
        w = _ref_wrapper(True)
        w.train()
        w.train_cfg = SimpleNamespace(cfg_train=True, step_offset=0)
        cs_d = load_seeded(w.denoiser, 500)
        cs_v = load_seeded(w.autoencoder, 501)
        H = 32
        g = torch.Generator().manual_seed(90 + ci)
        img = torch.rand(b, 5, 3, H, H, generator=g)
        extr, intr = random_cameras(b, 5, seed=95 + ci)
        extr = extr.roll(1, dims=1)                     # no identity pose at view 0
        view = lambda sl: {"image": img[:, sl].clone(), "extrinsics": extr[:, sl].clone(), "intrinsics": intr[:, sl].clone(),
                           "near": torch.ones(b, img[:, sl].shape[1]), "far": torch.full((b, img[:, sl].shape[1]), 100.0),
                           "index": torch.arange(5)[sl][None].expand(b, -1).clone()}
        batch = {"context": view(slice(0, 2)), "target": view(slice(2, 5)), "scene": ["synthetic"] * b}
        hl = H // 2
        v_c = index if index > 1 else 1
        v_t = 5 - v_c
        enc_noise = torch.randn(b * 5, 4, hl, hl, generator=g)
        noise = torch.randn(b, v_t, 4, hl, hl, generator=g)
        tsteps = torch.randint(0, 1000, (b,), generator=g)
        ints = [torch.tensor([index]), torch.tensor([second]), tsteps]
        coins = [onp.array([rel_coin]), onp.array([unc_coin])]
        o_randint, o_randn, o_randn_like, o_choice = torch.randint, torch.randn, torch.randn_like, onp.random.choice

        def fake_randint(*a, **k):
            return ints.pop(0).clone()

        def fake_randn(*shape, **k):
            shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
            assert shp == tuple(enc_noise.shape), shp
            return enc_noise.clone()

        def fake_randn_like(t, **k):
            assert tuple(t.shape) == tuple(noise.shape)
            return noise.clone()
        torch.randint, torch.randn, torch.randn_like, onp.random.choice = fake_randint, fake_randn, fake_randn_like, lambda *a, **k: coins.pop(0)
        try:
            with R.cpu_cuda(), torch.enable_grad():
                for prm in w.denoiser.parameters():
                    prm.requires_grad_(True)
                loss = w.training_step(batch, 0)
        finally:
            torch.randint, torch.randn, torch.randn_like, onp.random.choice = o_randint, o_randn, o_randn_like, o_choice
        assert not ints and not coins
        w.denoiser.zero_grad()
        with torch.enable_grad():
            loss.backward()
        p = f"c{ci}_"
        names, norms = [], []
        for n_, prm in w.denoiser.named_parameters():
            names.append(n_)
            norms.append(-1.0 if prm.grad is None else float(prm.grad.double().norm()))
        out.update({p + "b": b, p + "index": index, p + "second": second, p + "relative_coin": int(rel_coin), p + "unconditional": int(unc_coin),
                    p + "image": img, p + "extr": extr, p + "intr": intr, p + "enc_noise": enc_noise, p + "noise": noise, p + "timesteps": tsteps,
                    p + "loss": float(loss), p + "grad_norms": onp.array(norms), p + "checksum_denoiser": cs_d, p + "checksum_vae": cs_v})
        sd = dict(w.denoiser.named_parameters())
        for k_ in G9_FULL:       # up to 2048 evenly strided entries of the flattened gradient (the norms above cover the rest)
            gflat = sd[k_].grad.detach().reshape(-1)
            out[p + "grad/" + k_] = gflat[::max(1, gflat.numel() // 2048)][:2048].clone()
        unused = [n_ for n_, v_ in zip(names, norms) if v_ < 0]
        zero = [n_ for n_, v_ in zip(names, norms) if v_ == 0]
        print(f"  g9 case {ci}: b={b} index={index} uncond={unc_coin}: loss {float(loss):.6f}; {len(unused)} params without grad, {len(zero)} with zero grad")
        if ci == 0:
            out["names"] = onp.array(names)
    out["n"] = len(cases)
    out["widths"] = np.array([64, 128, 256, 256])
    out["vae_widths"] = np.array([32, 64])
    save("g9_training_step", **out)


# ----------------------------------------------------------------------------------------------- G10
def g10():
    """N4: the reference's `StandardTransformer` (src/model/denoiser/standard/transformer.py:45-136 over
    src/model/transformer/{transformer,attention,feed_forward,pre_norm}.py), its MultiViewUNet walk with those blocks, and the
    ray encodings of `DiffusionWrapper.ray_encode` (diffusion_wrapper.py:98-127,301-322: positional_encoding.py:8-36,
    srt/layers.py:11-58, Pluecker).  transformer/attention.py:94 pins the CUDA-only EFFICIENT_ATTENTION SDPA backend:
    replaced by a null context here (same arithmetic on the CPU math path)."""
    import contextlib
    R.install()
    ST = R.ref("src.model.denoiser.standard.transformer")
    TA = R.ref("src.model.transformer.attention")
    TA.sdpa_kernel = lambda *a, **k: contextlib.nullcontext()
    out = {}
    cases = [(64, 8, 1, None, 1, 2, 3, 4), (128, 4, 2, 16, 2, 1, 5, 4), (320, 8, 1, None, 1, 1, 2, 8)]      # C, heads, layers, d_dot, mlp mult, b, V, h
    for i, (C, heads, layers, d_dot, mult, b, V, h) in enumerate(cases):
        cfg = ST.CrossAttentionCfg(name="standard", num_heads=heads, num_layers=layers, d_dot=d_dot, d_mlp=None, d_mlp_multiplier=mult)
        m = ST.StandardTransformer(cfg, C).eval()
        cs = load_seeded(m, 600 + i)
        x = torch.randn(b, V, C, h, h, generator=torch.Generator().manual_seed(610 + i))
        y = m(x)
        out.update({f"st{i}_meta": np.array([C, heads, layers, -1 if d_dot is None else d_dot, mult, b, V, h, 600 + i]),
                    f"st{i}_checksum": cs, f"st{i}_x": x, f"st{i}_y": y, f"st{i}_nkeys": len(m.state_dict())})
        print(f"  g10 standard transformer {i}: C={C} heads={heads} layers={layers} -> |y| {y.abs().mean():.4f}")
    out["n_st"] = len(cases)
    # the reference's UNet walk with "standard" multi-view blocks (scratch topology, reduced width)
    M = R.ref("src.model.denoiser.mvunet")
    widths = [64, 64, 128, 128]
    ae = M.UNet2DModelCfg(name="unet", down_block_types=["DownBlock2D"] * 4, mid_block_type="UNetMidBlock2D", up_block_types=["UpBlock2D"] * 4,
                          only_cross_attention=False, block_out_channels=widths)
    cfg = M.MultiViewUNetCfg(name="mv_unet", autoencoder=ae, multi_view_attention=ST.CrossAttentionCfg(
        name="standard", num_heads=8, num_layers=1, d_dot=None, d_mlp=None, d_mlp_multiplier=1), pretrained_from=None)
    m = M.MultiViewUNet(cfg, 11, 4).eval()
    cs = load_seeded(m, 620)
    g = torch.Generator().manual_seed(621)
    x, t = torch.randn(1, 3, 11, 16, 16, generator=g), torch.tensor([[0, 400, 400]])
    with R.cpu_cuda():
        y = m.forward(x, t)
    out.update({"unet_widths": np.array(widths), "unet_checksum": cs, "unet_nkeys": len(m.state_dict()), "unet_x": x, "unet_t": t, "unet_y": y})
    print(f"  g10 mv-unet with standard blocks: |y| {y.abs().mean():.4f}")
    # ray encodings through the reference's DiffusionWrapper.ray_encode
    W = R.ref("src.model.diffusion_wrapper")
    PE = R.ref("src.model.encodings.positional_encoding")
    SRT = R.ref("src.model.srt.layers")
    extr, intr = random_cameras(2, 3, seed=630)
    extr = extr.roll(1, dims=1)
    hl = 6
    batch = {"context": {"extrinsics": extr[:, :1], "intrinsics": intr[:, :1]}, "target": {"extrinsics": extr[:, 1:], "intrinsics": intr[:, 1:]}}
    modes = [("raw", False, False, False, 0, 0), ("plucker", False, False, True, 0, 0), ("positional", True, False, False, 4, 3),
             ("positional_plucker_15", True, False, True, 15, 15), ("positional_dir_only", True, False, False, 0, 5), ("srt", False, True, False, 5, 2)]
    for name, use_pe, srt, plucker, no, nd in modes:
        w = W.DiffusionWrapper.__new__(W.DiffusionWrapper)
        torch.nn.Module.__init__(w)
        w.model_cfg = SimpleNamespace(use_plucker=plucker, srt_ray_encoding=srt, use_ray_encoding=use_pe)
        if srt:
            w.ray_encoder = SRT.RayEncoder(pos_octaves=no, ray_octaves=nd)
        else:
            w.ori_encoder = PE.PositionalEncoding(no) if (use_pe and no > 0) else torch.nn.Identity()
            w.dir_encoder = PE.PositionalEncoding(nd) if (use_pe and nd > 0) else torch.nn.Identity()
        with R.cpu_cuda():
            enc = w.ray_encode(batch, torch.zeros(2, 1, 4, hl, hl), torch.zeros(2, 2, 4, hl, hl))
        out[f"rays_{name}"] = enc
        out[f"rays_{name}_cfg"] = np.array([int(use_pe), int(srt), int(plucker), no, nd])
        print(f"  g10 rays {name}: {tuple(enc.shape)}")
    out["rays_extr"], out["rays_intr"], out["rays_modes"] = extr, intr, np.array([m_[0] for m_ in modes])
    save("g10_standard_and_encodings", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g3", "g4", "g5", "g6", "g7", "g9", "g10"]
    if "g1" in which:
        g1_g2()
    if "g3" in which:
        g3()
    if "g4" in which:
        g4()
    if "g5" in which:
        g5()
    if "g6" in which:
        g6()
    if "g9" in which:
        g9()
    if "g10" in which:
        g10()
    if "g7" in which:
        from make_golden_schedules import g7
        g7()
    assert not list(Path(R.REF_ROOT).rglob("__pycache__")), "reference tree was written to"
