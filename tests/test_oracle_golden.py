"""CPU: the oracle restatement vs the vectors produced by the reference's own code (G1-G5).

Tolerances: both sides are fp32 CPU; the restatement uses differently-ordered but equivalent torch
ops, so agreement is to fp32 round-off (rel-L2 <= 2e-6 per block, <= 2e-5 through a whole UNet /
5-step sampling loop)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from seeded import load_seeded

from oracle import multiview as MV
from oracle import pipeline as PL
from oracle.scheduler import DDIMScheduler
from oracle.vae import AutoencoderKL

torch.set_grad_enabled(False)


def test_g1_spatial_transformer_3d(golden):
    g = golden("g1_spatial_transformer_3d")
    for i in range(int(g["n"])):
        C, V, h, w, b, seed = (int(v) for v in g[f"c{i}_meta"])
        m = MV.SpatialTransformer3D(MV.MVAttnCfg(num_heads=8), C).eval()
        cs = load_seeded(m, seed)
        assert abs(cs - float(g[f"c{i}_checksum"])) <= 1e-9 * cs, "seeded weights drifted / key layout differs"
        y = m(torch.from_numpy(g[f"c{i}_x"]))
        assert rel_err(y, g[f"c{i}_y"]) < 2e-6


def test_g2_cross_attention(golden):
    g = golden("g2_cross_attention")
    for i in range(int(g["n"])):
        heads, d, L, bsz, seed = (int(v) for v in g[f"c{i}_meta"])
        m = MV.MVCrossAttention(heads * d, heads, d).eval()
        cs = load_seeded(m, seed)
        assert abs(cs - float(g[f"c{i}_checksum"])) <= 1e-9 * cs
        assert rel_err(m(torch.from_numpy(g[f"c{i}_x"])), g[f"c{i}_y"]) < 2e-6


def test_g3_rays(golden):
    g = golden("g3_rays")
    extr, intr = torch.from_numpy(g["extrinsics"]), torch.from_numpy(g["intrinsics"])
    for (h, w) in [(8, 8), (4, 6)]:
        xy = PL.sample_image_grid(h, w)
        assert torch.equal(xy, torch.from_numpy(g[f"xy_{h}x{w}"]))
        o, d = PL.image_rays(h, w, extr, intr)
        assert rel_err(o, g[f"origins_{h}x{w}"]) < 1e-7
        assert rel_err(d, g[f"directions_{h}x{w}"]) < 1e-6
    for idx in (0, 1, 3):
        assert rel_err(PL.absolute_to_relative_camera(extr, idx), g[f"relative_{idx}"]) < 1e-6


def _oracle_mvunet(topology, widths):
    if topology == "scratch":
        cfg = MV.MVUNetCfg(autoencoder=MV.UNetCfg(block_out_channels=tuple(widths)), pretrained_from=None)
    else:
        over = dict(block_out_channels=tuple(widths), attention_head_dim=tuple(max(1, c // 64) for c in widths))
        cfg = MV.MVUNetCfg(autoencoder=MV.UNetCfg(block_out_channels=tuple(widths)),
                           pretrained_from="stabilityai/stable-diffusion-2-1", pretrained_overrides=over)
    return MV.MultiViewUNet(cfg, 11, 4).eval()


def test_g4_mvunet_forward(golden):
    g = golden("g4_mvunet_forward")
    for i in range(int(g["n"])):
        m = _oracle_mvunet(str(g[f"c{i}_topology"]), [int(v) for v in g[f"c{i}_widths"]])
        assert len(m.state_dict()) == int(g[f"c{i}_nkeys"]), "state-dict key layout differs from the reference"
        cs = load_seeded(m, int(g[f"c{i}_seed"]))
        assert abs(cs - float(g[f"c{i}_checksum"])) <= 1e-9 * cs
        y = m(torch.from_numpy(g[f"c{i}_x"]), torch.from_numpy(g[f"c{i}_t"]))
        assert rel_err(y, g[f"c{i}_y"]) < 2e-5, i


def test_g5_step_and_sample(golden):
    g = golden("g5_step_sample")
    widths = [int(v) for v in g["widths"]]
    vae_widths = tuple(int(v) for v in g["vae_widths"])
    for ci in range(int(g["n"])):
        p = f"c{ci}_"
        use_cfg = bool(g[p + "use_cfg"])
        den = _oracle_mvunet("sd", widths)
        vae = AutoencoderKL.from_pretrained("x", config_overrides=dict(block_out_channels=vae_widths,
                                                                       layers_per_block=1)).eval()
        assert abs(load_seeded(den, 400) - float(g[p + "checksum_denoiser"])) < 1e-6
        assert abs(load_seeded(vae, 401) - float(g[p + "checksum_vae"])) < 1e-6
        sch = DDIMScheduler(clip_sample=False)
        sch.set_timesteps(5)
        extr, intr = torch.from_numpy(g[p + "extr"]), torch.from_numpy(g[p + "intr"])
        v_c = g[p + "ctx_img"].shape[1]
        img, _ = PL.sample(den, vae, sch, torch.from_numpy(g[p + "ctx_img"]), extr[:, :v_c], intr[:, :v_c],
                           extr[:, v_c:], intr[:, v_c:], x_T=torch.from_numpy(g[p + "x_T"]),
                           encode_noise=torch.from_numpy(g[p + "enc_noise"]), use_cfg=use_cfg)
        assert rel_err(img, g[p + "img"]) < 2e-5
        # isolated step + ray encoding
        ctx_lat, x_t = torch.from_numpy(g[p + "step_ctx_lat"]), torch.from_numpy(g[p + "step_x_t"])
        hl = x_t.shape[-1]
        rays = PL.ray_encode(extr[:, :v_c], intr[:, :v_c], extr[:, v_c:], intr[:, v_c:], hl, hl)
        assert rel_err(rays, g[p + "rays"]) < 1e-6
        ctx_in = torch.cat([ctx_lat, torch.zeros_like(ctx_lat[:, :, :1])], dim=2)
        x_prev = PL.step(den, sch, x_t, torch.tensor(int(g[p + "step_ts"])), ctx_in, rays,
                         torch.ones_like(x_t[:, :, :1]), use_cfg=use_cfg)
        assert rel_err(x_prev, g[p + "step_x_prev"]) < 2e-5


def test_g11_configs0_and_configs4_shapes(golden):
    """G11 (tests/golden/make_golden_configs.py, the reference's own DiffusionWrapper.sample): BASELINE.json configs[0] literally
    (1 ctx + 1 tgt view, 64x64 -> 8x8 latents, 5 DDIM steps) and configs[4]'s geometry (9 views, 64x64 latents: above the `h <= 32`
    gate of mvunet.py:137,190) -- pins the oracle's walk at both ends of the resolution gate"""
    g = golden("g11_configs")
    widths = [int(v) for v in g["widths"]]
    for ci in range(int(g["n"])):
        p = f"c{ci}_"
        den = _oracle_mvunet("sd", widths)
        vae = AutoencoderKL.from_pretrained("x", config_overrides=dict(block_out_channels=tuple(int(v) for v in g[p + "vae_widths"]),
                                                                       layers_per_block=1)).eval()
        assert abs(load_seeded(den, 400) - float(g[p + "checksum_denoiser"])) < 1e-6
        assert abs(load_seeded(vae, 401) - float(g[p + "checksum_vae"])) < 1e-6
        sch = DDIMScheduler(clip_sample=False)
        sch.set_timesteps(int(g[p + "n_steps"]))
        extr, intr = torch.from_numpy(g[p + "extr"]), torch.from_numpy(g[p + "intr"])
        v_c = g[p + "ctx_img"].shape[1]
        img, _ = PL.sample(den, vae, sch, torch.from_numpy(g[p + "ctx_img"]), extr[:, :v_c], intr[:, :v_c], extr[:, v_c:], intr[:, v_c:],
                           x_T=torch.from_numpy(g[p + "x_T"]), encode_noise=torch.from_numpy(g[p + "enc_noise"]), use_cfg=True)
        want = torch.from_numpy(g[p + "img"].astype("float32"))
        assert img.shape == want.shape
        assert float((img - want).abs().max()) < 1e-3, ci          # (the fixture stores the image in f16: 2^-11 absolute on [0, 1])
        assert rel_err(img, want) < 6e-4, ci
