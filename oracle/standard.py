"""Oracle (test infrastructure): the "standard" multi-view attention block and the ray encodings.

Follows
* `src/model/denoiser/standard/transformer.py:13-22,45-136` (`CrossAttentionCfg`, `StandardTransformer`, downscale = 1,
  pos_enc = False: the only configuration the reference can run -- `pos_enc=True` hits an undefined name at :102),
* `src/model/transformer/transformer.py:33-69`, `attention.py:36-101` (self-attention form), `feed_forward.py:29-40`,
  `pre_norm.py:30-37`,
* `src/model/encodings/positional_encoding.py:8-36`, `src/model/srt/layers.py:11-58`, and their use in
  `src/model/diffusion_wrapper.py:98-127,301-322` (incl. the Pluecker switch).
PINNED: tests/golden/g10_standard_and_encodings.npz holds outputs of the reference's own modules (imported in the build
container); tests/test_oracle_standard.py compares.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn


@dataclass
class StdAttnCfg:
    name: str = "standard"
    num_heads: int = 8
    num_layers: int = 1
    d_dot: Optional[int] = None
    d_mlp: Optional[int] = None
    d_mlp_multiplier: Optional[int] = 1
    downscale: int = 1
    pos_enc: bool = False


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x):
        return self.fn(self.norm(x))


class _Attention(nn.Module):
    def __init__(self, dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(0.0)) if not (heads == 1 and dim_head == dim) else nn.Identity()

    def forward(self, x):
        B, L_, _ = x.shape
        q, k, v = (t.reshape(B, L_, self.heads, -1).transpose(1, 2) for t in self.to_qkv(x).chunk(3, dim=-1))
        out = F.scaled_dot_product_attention(q, k, v)
        return self.to_out(out.transpose(1, 2).reshape(B, L_, -1).float())


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0), nn.Linear(hidden, dim), nn.Dropout(0.0))

    def forward(self, x):
        return self.net(x)


class _Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([_PreNorm(dim, _Attention(dim, heads, dim_head)), _PreNorm(dim, _FeedForward(dim, mlp_dim))])
                                     for _ in range(depth)])

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn(x) + x
            x = ff(x) + x
        return x


class StandardTransformer(nn.Module):
    def __init__(self, cfg: StdAttnCfg, d_in: int):
        super().__init__()
        assert (cfg.d_mlp is None) != (cfg.d_mlp_multiplier is None)
        assert cfg.downscale == 1 and not cfg.pos_enc
        self.transformer = _Transformer(d_in, cfg.num_layers, cfg.num_heads, cfg.d_dot or d_in // cfg.num_heads, cfg.d_mlp or d_in * cfg.d_mlp_multiplier)

    def forward(self, features):
        b, v, c, h, w = features.shape
        x = features.permute(0, 1, 3, 4, 2).reshape(b, v * h * w, c)
        x = self.transformer(x)
        return x.reshape(b, v, h, w, c).permute(0, 1, 4, 2, 3)


# ---------------------------------------------------------------------------------------------------- ray encodings
def positional_encoding(x: torch.Tensor, num_octaves: int) -> torch.Tensor:
    """positional_encoding.py:8-36: [..., d] -> [..., d * octaves * 2], order (d, frequency, phase)"""
    freq = (2 * torch.pi * 2 ** torch.arange(num_octaves).float())[:, None].expand(-1, 2)
    phase = torch.tensor([0, 0.5 * torch.pi], dtype=torch.float32)[None].expand(num_octaves, -1)
    s = x[..., None, None] * freq
    return torch.sin(s + phase).flatten(-3)


def srt_encoding(x: torch.Tensor, num_octaves: int) -> torch.Tensor:
    """srt/layers.py:11-33: [b, n, d] -> [b, n, 2 * d * octaves] = [sines | cosines], each ordered (d, octave)"""
    mult = 2 ** torch.arange(num_octaves).float() * math.pi
    s = x[..., None] * mult
    return torch.cat([torch.sin(s).flatten(-2), torch.cos(s).flatten(-2)], dim=-1)


def encode_rays(origins, directions, use_ray_encoding=False, srt_ray_encoding=False, use_plucker=False, num_origin_octaves=15,
                num_direction_octaves=15) -> torch.Tensor:
    """diffusion_wrapper.py:306-320 on [b, v, hw, 3] origins / directions -> [b, v, hw, C]"""
    if use_plucker:
        origins = torch.cross(origins, directions, dim=-1)
    if srt_ray_encoding:
        b, v = origins.shape[:2]
        enc = torch.cat([srt_encoding(origins.flatten(0, 1), num_origin_octaves), srt_encoding(directions.flatten(0, 1), num_direction_octaves)], dim=-1)
        return enc.reshape(b, v, *enc.shape[1:])
    o = positional_encoding(origins, num_origin_octaves) if (use_ray_encoding and num_origin_octaves > 0) else origins
    d = positional_encoding(directions, num_direction_octaves) if (use_ray_encoding and num_direction_octaves > 0) else directions
    return torch.cat([o, d], dim=-1)
