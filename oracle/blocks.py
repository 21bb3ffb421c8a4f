"""Oracle (test infrastructure): CPU restatement of the diffusers==0.27.2 building blocks the
reference walks by hand in `src/model/denoiser/mvunet.py:90-208`.

The package itself is absent from /root/reference (requirements.txt:8 pins it) and cannot be
installed here, so every class below restates the published algorithm
and exposes exactly the attribute names the reference touches (`resnets`, `attentions`,
`downsamplers`, `upsamplers`, `has_cross_attention`, `.sample`, `time_proj`, ...), with the
state-dict key layout of SURVEY.md App. A.9.  Parity for these blocks is unpinned by the reference
(it has no tests); they are cross-checked against independent torch-primitive compositions in
tests/test_oracle_blocks.py.

Everything runs in the dtype of the module parameters (call `.double()` for an fp64 witness).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn


# ----------------------------------------------------------------------------- A.1 time embedding
class Timesteps(nn.Module):
    """diffusers `Timesteps` / `get_timestep_embedding` (called at mvunet.py:107)."""

    def __init__(self, num_channels: int, flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0.0):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps: torch.Tensor) -> torch.Tensor:
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.downscale_freq_shift)
        emb = torch.exp(exponent)
        emb = timesteps[:, None].float() * emb[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb  # always fp32, like the reference's use at mvunet.py:107 (no cast)


class TimestepEmbedding(nn.Module):
    """diffusers `TimestepEmbedding` (called at mvunet.py:108)."""

    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        sample = sample.to(self.linear_1.weight.dtype)
        return self.linear_2(self.act(self.linear_1(sample)))


# ----------------------------------------------------------------------------- A.2 resnet
class ResnetBlock2D(nn.Module):
    """diffusers `ResnetBlock2D` (called at mvunet.py:121,150,159,177)."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, temb_channels: Optional[int] = 512,
                 groups: int = 32, eps: float = 1e-6, output_scale_factor: float = 1.0):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1, stride=1, padding=0, bias=True)

    def forward(self, input_tensor, temb=None):
        h = self.conv1(self.nonlinearity(self.norm1(input_tensor)))
        if self.time_emb_proj is not None:
            t = self.time_emb_proj(self.nonlinearity(temb.to(h.dtype)))[:, :, None, None]
            h = h + t
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


# ----------------------------------------------------------------------------- A.3 transformer
def sdpa(q, k, v, scale: float):
    """softmax(q k^T * scale) v on [B, H, L, d] tensors, computed in the tensors' dtype."""
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    return torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)


class Attention(nn.Module):
    """diffusers `Attention` with the PyTorch-2 processor (`F.scaled_dot_product_attention`,
    scale d^-1/2, no mask).  Also covers the VAE mid-block form (`group_norm`, biases on q/k/v,
    residual connection) used by `AutoencoderKL`."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8,
                 dim_head: int = 64, bias: bool = False, out_bias: bool = True,
                 norm_num_groups: Optional[int] = None, eps: float = 1e-5,
                 residual_connection: bool = False, rescale_output_factor: float = 1.0):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.residual_connection = residual_connection
        self.rescale_output_factor = rescale_output_factor
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.group_norm = (nn.GroupNorm(norm_num_groups, query_dim, eps=eps, affine=True)
                           if norm_num_groups is not None else None)
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])

    def forward(self, hidden_states, encoder_hidden_states=None, temb=None):
        residual = hidden_states
        ndim = hidden_states.ndim
        if ndim == 4:
            b, c, hh, ww = hidden_states.shape
            hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
        if self.group_norm is not None:
            hidden_states = self.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states.to(hidden_states.dtype)
        B, L, _ = hidden_states.shape
        q = self.to_q(hidden_states).view(B, L, self.heads, self.dim_head).transpose(1, 2)
        k = self.to_k(ctx).view(B, -1, self.heads, self.dim_head).transpose(1, 2)
        v = self.to_v(ctx).view(B, -1, self.heads, self.dim_head).transpose(1, 2)
        o = sdpa(q, k, v, self.scale).transpose(1, 2).reshape(B, L, self.heads * self.dim_head)
        o = self.to_out[1](self.to_out[0](o))
        if ndim == 4:
            o = o.transpose(-1, -2).reshape(b, c, hh, ww)
        if self.residual_connection:
            o = o + residual
        return o / self.rescale_output_factor


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)  # exact (erf) GELU


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """diffusers `BasicTransformerBlock` (layer_norm flavour): self-attn, cross-attn, GEGLU FF."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, cross_attention_dim: Optional[int]):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, num_attention_heads, attention_head_dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim, num_attention_heads, attention_head_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states=None):
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states


class Transformer2DModel(nn.Module):
    """diffusers `Transformer2DModel`, continuous input (called at mvunet.py:131-134,158,185-188).
    Returns an object with `.sample`."""

    def __init__(self, num_attention_heads: int, attention_head_dim: int, in_channels: int,
                 cross_attention_dim: Optional[int], use_linear_projection: bool = False,
                 norm_num_groups: int = 32, num_layers: int = 1):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        if use_linear_projection:
            self.proj_in = nn.Linear(in_channels, inner)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim)
            for _ in range(num_layers)])
        if use_linear_projection:
            self.proj_out = nn.Linear(inner, in_channels)
        else:
            self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, hidden_states, encoder_hidden_states=None, **_unused):
        b, _, h, w = hidden_states.shape
        residual = hidden_states
        x = self.norm(hidden_states)
        if not self.use_linear_projection:
            x = self.proj_in(x)
            inner = x.shape[1]
            x = x.permute(0, 2, 3, 1).reshape(b, h * w, inner)
        else:
            inner = x.shape[1]
            x = x.permute(0, 2, 3, 1).reshape(b, h * w, inner)
            x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states)
        if not self.use_linear_projection:
            x = x.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
            x = self.proj_out(x)
        else:
            x = self.proj_out(x)
            x = x.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
        return SimpleNamespace(sample=x + residual)


# ----------------------------------------------------------------------------- A.4 resampling
class Downsample2D(nn.Module):
    """diffusers `Downsample2D(use_conv=True)`: 3x3 stride-2 conv; padding=0 => F.pad(0,1,0,1) first
    (the VAE encoder variant)."""

    def __init__(self, channels: int, out_channels: Optional[int] = None, padding: int = 1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(x)


class Upsample2D(nn.Module):
    """diffusers `Upsample2D(use_conv=True)`: nearest x2 then 3x3 conv."""

    def __init__(self, channels: int, out_channels: Optional[int] = None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x)


# ----------------------------------------------------------------------------- A.5 blocks
class DownBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers=2, resnet_eps=1e-5,
                 resnet_groups=32, add_downsample=True, downsample_padding=1):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_groups, resnet_eps)
            for i in range(num_layers)])
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, out_channels, downsample_padding)])
                             if add_downsample else None)

    def forward(self, hidden_states, temb=None):
        outs = ()
        for r in self.resnets:
            hidden_states = r(hidden_states, temb)
            outs += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class CrossAttnDownBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers=2, resnet_eps=1e-5,
                 resnet_groups=32, num_attention_heads=1, cross_attention_dim=1280,
                 use_linear_projection=False, add_downsample=True, downsample_padding=1):
        super().__init__()
        self.has_cross_attention = True
        self.num_attention_heads = num_attention_heads
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_groups, resnet_eps)
            for i in range(num_layers)])
        self.attentions = nn.ModuleList([
            Transformer2DModel(num_attention_heads, out_channels // num_attention_heads, out_channels,
                               cross_attention_dim, use_linear_projection, resnet_groups)
            for _ in range(num_layers)])
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, out_channels, downsample_padding)])
                             if add_downsample else None)

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None):
        outs = ()
        for r, a in zip(self.resnets, self.attentions):
            hidden_states = r(hidden_states, temb)
            hidden_states = a(hidden_states, encoder_hidden_states=encoder_hidden_states).sample
            outs += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, in_channels, temb_channels, num_layers=1, resnet_eps=1e-5, resnet_groups=32,
                 num_attention_heads=1, cross_attention_dim=1280, use_linear_projection=False):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels, in_channels, temb_channels, resnet_groups, resnet_eps)
                                      for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([
            Transformer2DModel(num_attention_heads, in_channels // num_attention_heads, in_channels,
                               cross_attention_dim, use_linear_projection, resnet_groups)
            for _ in range(num_layers)])

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        for a, r in zip(self.attentions, self.resnets[1:]):
            hidden_states = a(hidden_states, encoder_hidden_states=encoder_hidden_states).sample
            hidden_states = r(hidden_states, temb)
        return hidden_states


class UNetMidBlock2D(nn.Module):
    """diffusers `UNetMidBlock2D`.  The UNet builds it with `num_layers=0, add_attention=False`
    (one resnet, no attention: SURVEY.md App. A.0 "scratch UNet"); the VAE builds it with
    `num_layers=1`, one single-head attention of width `in_channels` (App. A.8)."""

    def __init__(self, in_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 add_attention=True, attention_head_dim=None):
        super().__init__()
        attention_head_dim = attention_head_dim or in_channels
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels, in_channels, temb_channels, resnet_groups, resnet_eps)
                                      for _ in range(num_layers + 1)])
        attns = []
        for _ in range(num_layers):
            if add_attention:
                attns.append(Attention(in_channels, None, in_channels // attention_head_dim, attention_head_dim,
                                       bias=True, out_bias=True, norm_num_groups=resnet_groups, eps=resnet_eps,
                                       residual_connection=True))
            else:
                attns.append(None)
        self.attentions = nn.ModuleList(attns)

    def forward(self, hidden_states, temb=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        for a, r in zip(self.attentions, self.resnets[1:]):
            if a is not None:
                hidden_states = a(hidden_states)
            hidden_states = r(hidden_states, temb)
        return hidden_states


class UpBlock2D(nn.Module):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=3,
                 resnet_eps=1e-5, resnet_groups=32, add_upsample=True):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            res.append(ResnetBlock2D(rin + skip, out_channels, temb_channels, resnet_groups, resnet_eps))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None):
        for r in self.resnets:
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = r(torch.cat([hidden_states, skip], dim=1), temb)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class CrossAttnUpBlock2D(nn.Module):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=3,
                 resnet_eps=1e-5, resnet_groups=32, num_attention_heads=1, cross_attention_dim=1280,
                 use_linear_projection=False, add_upsample=True):
        super().__init__()
        self.has_cross_attention = True
        res, att = [], []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            res.append(ResnetBlock2D(rin + skip, out_channels, temb_channels, resnet_groups, resnet_eps))
            att.append(Transformer2DModel(num_attention_heads, out_channels // num_attention_heads, out_channels,
                                          cross_attention_dim, use_linear_projection, resnet_groups))
        self.resnets = nn.ModuleList(res)
        self.attentions = nn.ModuleList(att)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None):
        for r, a in zip(self.resnets, self.attentions):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = r(torch.cat([hidden_states, skip], dim=1), temb)
            hidden_states = a(hidden_states, encoder_hidden_states=encoder_hidden_states).sample
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


# ----------------------------------------------------------------------------- A.0 / A.6 the UNet
SD21_UNET_CONFIG = dict(
    in_channels=4, out_channels=4,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, attention_head_dim=(5, 10, 20, 20),
    cross_attention_dim=1024, use_linear_projection=True, norm_num_groups=32, norm_eps=1e-5,
)


def _per_block(v, n):
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * n


class UNet2DConditionModel(nn.Module):
    """diffusers `UNet2DConditionModel` restricted to what the reference instantiates
    (mvunet.py:54-72): the SD-2.1 topology (`from_pretrained`, App. A.0) and the scratch topology
    built from `config/model/denoiser/autoencoder/unet.yaml`."""

    def __init__(self, in_channels=4, out_channels=4,
                 down_block_types: Sequence[str] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                 mid_block_type: Optional[str] = "UNetMidBlock2DCrossAttn",
                 up_block_types: Sequence[str] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
                 only_cross_attention=False, block_out_channels: Sequence[int] = (320, 640, 1280, 1280),
                 layers_per_block=2, downsample_padding=1, norm_num_groups=32, norm_eps=1e-5,
                 cross_attention_dim=1280, attention_head_dim=8, use_linear_projection=False,
                 flip_sin_to_cos=True, freq_shift=0):
        super().__init__()
        assert not only_cross_attention, "only_cross_attention=True is not on the reference's path"
        n = len(down_block_types)
        boc = tuple(block_out_channels)
        heads = _per_block(attention_head_dim, n)  # diffusers quirk: "attention_head_dim" = number of heads
        xdim = _per_block(cross_attention_dim, n)
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=boc, cross_attention_dim=cross_attention_dim)
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(boc[0], temb)

        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, t in enumerate(down_block_types):
            in_c, out_c, final = out_c, boc[i], i == n - 1
            if t == "DownBlock2D":
                self.down_blocks.append(DownBlock2D(in_c, out_c, temb, layers_per_block, norm_eps, norm_num_groups,
                                                    not final, downsample_padding))
            elif t == "CrossAttnDownBlock2D":
                self.down_blocks.append(CrossAttnDownBlock2D(in_c, out_c, temb, layers_per_block, norm_eps,
                                                             norm_num_groups, heads[i], xdim[i],
                                                             use_linear_projection, not final, downsample_padding))
            else:
                raise ValueError(t)

        if mid_block_type == "UNetMidBlock2DCrossAttn":
            self.mid_block = UNetMidBlock2DCrossAttn(boc[-1], temb, 1, norm_eps, norm_num_groups, heads[-1], xdim[-1],
                                                     use_linear_projection)
        elif mid_block_type == "UNetMidBlock2D":
            self.mid_block = UNetMidBlock2D(boc[-1], temb, num_layers=0, resnet_eps=norm_eps,
                                            resnet_groups=norm_num_groups, add_attention=False)
        else:
            raise ValueError(mid_block_type)

        self.up_blocks = nn.ModuleList()
        rboc, rheads, rxdim = boc[::-1], heads[::-1], xdim[::-1]
        out_c = rboc[0]
        for i, t in enumerate(up_block_types):
            prev, out_c, in_c, final = out_c, rboc[i], rboc[min(i + 1, n - 1)], i == n - 1
            if t == "UpBlock2D":
                self.up_blocks.append(UpBlock2D(in_c, prev, out_c, temb, layers_per_block + 1, norm_eps,
                                                norm_num_groups, not final))
            elif t == "CrossAttnUpBlock2D":
                self.up_blocks.append(CrossAttnUpBlock2D(in_c, prev, out_c, temb, layers_per_block + 1, norm_eps,
                                                         norm_num_groups, rheads[i], rxdim[i],
                                                         use_linear_projection, not final))
            else:
                raise ValueError(t)

        self.conv_norm_out = nn.GroupNorm(norm_num_groups, boc[0], eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @classmethod
    def from_pretrained(cls, path, subfolder="unet", config_overrides=None):
        """No checkpoints are reachable offline: build the SD-2.1 topology (App. A.0) with the
        module's default random init.  `config_overrides` lets tests shrink the widths."""
        cfg = dict(SD21_UNET_CONFIG)
        cfg.update(config_overrides or {})
        return cls(**cfg)

    def enable_xformers_memory_efficient_attention(self):  # diffusion_wrapper.py:147 (no-op here)
        return None

    def forward(self, sample, timestep, encoder_hidden_states=None):
        """diffusers' own forward (not used by the reference, which walks the sub-modules by hand)."""
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.long, device=sample.device)
        timestep = timestep.reshape(-1).expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(timestep).to(sample.dtype))
        h = self.conv_in(sample)
        skips = (h,)
        for blk in self.down_blocks:
            if getattr(blk, "has_cross_attention", False):
                h, outs = blk(h, emb, encoder_hidden_states)
            else:
                h, outs = blk(h, emb)
            skips += outs
        if getattr(self.mid_block, "has_cross_attention", False):
            h = self.mid_block(h, emb, encoder_hidden_states)
        else:
            h = self.mid_block(h, emb)
        for blk in self.up_blocks:
            k = len(blk.resnets)
            res, skips = skips[-k:], skips[:-k]
            if getattr(blk, "has_cross_attention", False):
                h = blk(h, res, emb, encoder_hidden_states)
            else:
                h = blk(h, res, emb)
        h = self.conv_out(self.conv_act(self.conv_norm_out(h)))
        return SimpleNamespace(sample=h)
