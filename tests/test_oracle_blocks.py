"""CPU: G8 -- every restated diffusers block vs an INDEPENDENT composition of torch primitives in
fp64 (`F.group_norm`, `F.conv2d`, `F.scaled_dot_product_attention`, `F.layer_norm`, ...).  The
diffusers package itself is absent, so this is the second CPU witness for that
arithmetic, not a pin against the reference."""
import math

import torch
import torch.nn.functional as F

from conftest import rel_err
from seeded import load_seeded
from oracle import blocks as B
from oracle.vae import AutoencoderKL

GRAD_ENABLED = False      # tests/conftest.py::_grad_mode: no autograd graphs in this module
G = lambda s: torch.Generator().manual_seed(s)


def test_timesteps_and_embedding():
    t = torch.tensor([0, 1, 20, 500, 980, 999])
    e = B.Timesteps(320)(t)
    assert e.dtype == torch.float32 and e.shape == (6, 320)
    i = torch.arange(160, dtype=torch.float64)
    arg = t.double()[:, None] * torch.exp(-math.log(10000.0) * i / 160)[None]
    ref = torch.cat([arg.cos(), arg.sin()], dim=-1)     # flip_sin_to_cos
    assert (e.double() - ref).abs().max() < 2e-4        # fp32 argument round-off at t~1000
    m = B.TimestepEmbedding(320, 1280).double(); load_seeded(m, 1)
    sd = m.state_dict()
    x = e.double()
    ref = F.linear(F.silu(F.linear(x, sd["linear_1.weight"], sd["linear_1.bias"])), sd["linear_2.weight"], sd["linear_2.bias"])
    assert rel_err(m(e), ref) < 1e-12


def test_resnet_block():
    for cin, cout, temb in [(64, 64, 128), (96, 64, 128), (64, 128, None)]:
        m = B.ResnetBlock2D(cin, cout, temb, 32, 1e-5).double(); load_seeded(m, 2)
        sd = m.state_dict()
        x = torch.randn(3, cin, 6, 5, generator=G(3), dtype=torch.float64)
        e = torch.randn(3, temb, generator=G(4), dtype=torch.float64) if temb else None
        h = F.conv2d(F.silu(F.group_norm(x, 32, sd["norm1.weight"], sd["norm1.bias"], 1e-5)), sd["conv1.weight"], sd["conv1.bias"], padding=1)
        if temb:
            h = h + F.linear(F.silu(e), sd["time_emb_proj.weight"], sd["time_emb_proj.bias"])[:, :, None, None]
        h = F.conv2d(F.silu(F.group_norm(h, 32, sd["norm2.weight"], sd["norm2.bias"], 1e-5)), sd["conv2.weight"], sd["conv2.bias"], padding=1)
        sc = F.conv2d(x, sd["conv_shortcut.weight"], sd["conv_shortcut.bias"]) if cin != cout else x
        assert rel_err(m(x, e), sc + h) < 1e-12


def _tblock_ref(sd, p, x, ctx, heads):
    def attn(pref, q_in, kv_in):
        q = F.linear(q_in, sd[pref + "to_q.weight"]); k = F.linear(kv_in, sd[pref + "to_k.weight"]); v = F.linear(kv_in, sd[pref + "to_v.weight"])
        sp = lambda t: t.unflatten(-1, (heads, -1)).transpose(1, 2)
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).flatten(2)
        return F.linear(o, sd[pref + "to_out.0.weight"], sd[pref + "to_out.0.bias"])
    C = x.shape[-1]
    ln = lambda t, n: F.layer_norm(t, (C,), sd[p + n + ".weight"], sd[p + n + ".bias"], 1e-5)
    x = x + attn(p + "attn1.", ln(x, "norm1"), ln(x, "norm1"))
    x = x + attn(p + "attn2.", ln(x, "norm2"), ctx)
    hgate = F.linear(ln(x, "norm3"), sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])
    a, gate = hgate.chunk(2, -1)
    return x + F.linear(a * F.gelu(gate), sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])


def test_transformer2d_linear_and_conv_projection():
    for use_linear in (True, False):
        m = B.Transformer2DModel(2, 32, 64, 48, use_linear_projection=use_linear).double(); load_seeded(m, 5)
        sd = m.state_dict()
        x = torch.randn(2, 64, 4, 3, generator=G(6), dtype=torch.float64)
        ctx = torch.randn(2, 5, 48, generator=G(7), dtype=torch.float64)
        h = F.group_norm(x, 32, sd["norm.weight"], sd["norm.bias"], 1e-6)
        if use_linear:
            t = F.linear(h.flatten(2).transpose(1, 2), sd["proj_in.weight"], sd["proj_in.bias"])
        else:
            t = F.conv2d(h, sd["proj_in.weight"], sd["proj_in.bias"]).flatten(2).transpose(1, 2)
        t = _tblock_ref(sd, "transformer_blocks.0.", t, ctx, 2)
        if use_linear:
            o = F.linear(t, sd["proj_out.weight"], sd["proj_out.bias"]).transpose(1, 2).unflatten(2, (4, 3))
        else:
            o = F.conv2d(t.transpose(1, 2).unflatten(2, (4, 3)), sd["proj_out.weight"], sd["proj_out.bias"])
        assert rel_err(m(x, encoder_hidden_states=ctx).sample, o + x) < 1e-12
        # zero single-token context (mvunet.py:128): attn2 == to_out bias exactly
        z = torch.zeros(2, 1, 48, dtype=torch.float64)
        blk = m.transformer_blocks[0]
        out = blk.attn2(torch.randn(2, 12, 64, generator=G(8), dtype=torch.float64), z)
        assert torch.equal(out, blk.attn2.to_out[0].bias.expand_as(out))


def test_resampling():
    d = B.Downsample2D(32, 32, padding=1).double(); load_seeded(d, 9)
    x = torch.randn(2, 32, 8, 8, generator=G(10), dtype=torch.float64)
    assert rel_err(d(x), F.conv2d(x, d.conv.weight, d.conv.bias, stride=2, padding=1)) < 1e-13
    d0 = B.Downsample2D(32, 32, padding=0).double(); load_seeded(d0, 11)
    assert rel_err(d0(x), F.conv2d(F.pad(x, (0, 1, 0, 1)), d0.conv.weight, d0.conv.bias, stride=2)) < 1e-13
    assert d0(x).shape[-1] == 4
    u = B.Upsample2D(32, 32).double(); load_seeded(u, 12)
    up = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    assert rel_err(u(x), F.conv2d(up, u.conv.weight, u.conv.bias, padding=1)) < 1e-13


def test_unet_topology_sd21_channel_plan():
    """skip-concat input widths of the up path and head counts."""
    with torch.device("meta"):
        u = B.UNet2DConditionModel.from_pretrained("x")
    got = [[r.in_channels for r in b.resnets] for b in u.up_blocks]
    assert got == [[2560, 2560, 2560], [2560, 2560, 1920], [1920, 1280, 960], [960, 640, 640]]
    assert [b.attentions[0].transformer_blocks[0].attn1.heads for b in u.down_blocks[:3]] == [5, 10, 20]
    assert u.mid_block.attentions[0].transformer_blocks[0].attn1.heads == 20
    assert u.down_blocks[0].attentions[0].transformer_blocks[0].attn2.to_k.in_features == 1024
    assert not hasattr(u.down_blocks[3], "has_cross_attention") and u.down_blocks[3].downsamplers is None
    n = sum(p.numel() for p in u.parameters())
    assert 860e6 < n < 870e6, n     # SD-2.1 UNet ~= 865.9 M parameters


def test_vae_mid_attention_and_decoder_shapes():
    vae = AutoencoderKL.from_pretrained("x", config_overrides=dict(block_out_channels=(32, 64, 64), layers_per_block=1)).double()
    load_seeded(vae, 13)
    a = vae.decoder.mid_block.attentions[0]
    sd = a.state_dict()
    x = torch.randn(2, 64, 4, 4, generator=G(14), dtype=torch.float64)
    h = F.group_norm(x, 32, sd["group_norm.weight"], sd["group_norm.bias"], 1e-6).flatten(2).transpose(1, 2)
    q, k, v = (F.linear(h, sd[f"to_{n}.weight"], sd[f"to_{n}.bias"]) for n in "qkv")
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = F.linear(o, sd["to_out.0.weight"], sd["to_out.0.bias"]).transpose(1, 2).unflatten(2, (4, 4)) + x
    assert rel_err(a(x), o) < 1e-12
    z = torch.randn(1, 4, 4, 4, generator=G(15), dtype=torch.float64)
    img = vae.decode(z).sample
    assert img.shape == (1, 3, 16, 16)
    lat = vae.encode(img).latent_dist
    assert lat.mean.shape == (1, 4, 4, 4)
    full = AutoencoderKL.from_pretrained("x")
    n = sum(p.numel() for p in full.parameters())
    assert 83e6 < n < 84.5e6, n     # SD VAE ~= 83.65 M parameters
