"""Oracle (test infrastructure): restatement of the reference-owned multi-view denoiser.

Follows
* `src/model/denoiser/mvdream/attention.py:60-87`  (GEGLU feed-forward, mult 4),
* `.../attention.py:156-205` (CrossAttention: bias-free q/k/v, QK^T in fp32, softmax, out Linear),
* `.../attention.py:357-368` (BasicTransformerBlock3D: attn1 over all views' tokens, attn2 per view),
* `.../attention.py:371-439` (SpatialTransformer3D: GN(32, eps 1e-6), 1x1 proj_in, block,
  zero-initialised 1x1 proj_out, residual),
* `src/model/denoiser/attention.py:8-27` (`get_attn_blocks`: one block per UNet block, width
  `block.resnets[-1].out_channels`),
* `src/model/denoiser/mvunet.py:43-88` (construction) and `:90-208` (the hand-rolled UNet walk).

PINNED: tests/golden/make_golden.py runs the reference's own classes (imported from
/root/reference in the build container) on seeded inputs; tests/test_oracle_golden.py checks this
restatement against those committed vectors (G1, G2, G4).
State-dict keys match the reference's so weights move between the two freely.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import torch
from torch import nn

from . import blocks as B


@dataclass
class MVAttnCfg:
    """Mirror of `SpatialTransformer3DCfg` (mvdream/attention.py:24-32); only the consumed fields."""
    name: str = "spatial_transformer_3d"
    num_heads: int = 8
    num_layers: int = 1
    d_dot: Optional[int] = None


class MVCrossAttention(nn.Module):
    """mvdream/attention.py:156-205 with `context=None` (self-attention) and no mask."""

    def __init__(self, dim: int, heads: int, dim_head: int):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(dim, inner, bias=False)
        self.to_v = nn.Linear(dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(0.0))

    def forward(self, x, context=None):
        ctx = x if context is None else context
        Bn, L, _ = x.shape
        split = lambda t: t.view(Bn, -1, self.heads, self.dim_head).transpose(1, 2)
        q, k, v = split(self.to_q(x)), split(self.to_k(ctx)), split(self.to_v(ctx))
        # the reference forces q,k to fp32 for QK^T (:185-188); in an fp32/fp64 oracle that is the
        # working dtype already
        o = B.sdpa(q, k, v, self.scale).transpose(1, 2).reshape(Bn, L, self.heads * self.dim_head)
        return self.to_out(o)


class MVFeedForward(nn.Module):
    """mvdream/attention.py:60-87: GEGLU(dim -> 4 dim) then Linear(4 dim -> dim); keys `net.0.proj`, `net.2`."""

    def __init__(self, dim: int):
        super().__init__()
        self.net = nn.Sequential(B.GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim))

    def forward(self, x):
        return self.net(x)


class MVBlock3D(nn.Module):
    """mvdream/attention.py:257-296 + 357-368."""

    def __init__(self, dim: int, heads: int, dim_head: int):
        super().__init__()
        self.attn1 = MVCrossAttention(dim, heads, dim_head)
        self.ff = MVFeedForward(dim)
        self.attn2 = MVCrossAttention(dim, heads, dim_head)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, x, num_frames: int):
        bf, l, c = x.shape
        x = x.reshape(bf // num_frames, num_frames * l, c)         # all views of a scene = one sequence
        x = self.attn1(self.norm1(x)) + x
        x = x.reshape(bf, l, c)                                    # back to one sequence per view
        x = self.attn2(self.norm2(x)) + x
        x = self.ff(self.norm3(x)) + x
        return x


class SpatialTransformer3D(nn.Module):
    """mvdream/attention.py:371-439 (`use_linear=False` branch, the only one the reference builds)."""

    def __init__(self, cfg: MVAttnCfg, d_in: int):
        super().__init__()
        d_head = cfg.d_dot or d_in // cfg.num_heads
        self.in_channels = d_in
        self.norm = nn.GroupNorm(32, d_in, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(d_in, d_in, 1)
        self.transformer_blocks = nn.ModuleList([MVBlock3D(d_in, cfg.num_heads, d_head) for _ in range(cfg.num_layers)])
        self.proj_out = nn.Conv2d(d_in, d_in, 1)
        nn.init.zeros_(self.proj_out.weight)   # zero_module (:407-411)
        nn.init.zeros_(self.proj_out.bias)

    def forward(self, x):
        b, v, c, h, w = x.shape
        x = x.reshape(b * v, c, h, w)
        x_in = x
        x = self.proj_in(self.norm(x))
        x = x.permute(0, 2, 3, 1).reshape(b * v, h * w, c)
        for blk in self.transformer_blocks:
            x = blk(x, num_frames=v)
        x = x.reshape(b * v, h, w, c).permute(0, 3, 1, 2)
        x = self.proj_out(x) + x_in
        return x.reshape(b, v, c, h, w)


def get_attn_blocks(cfg, unet_blocks) -> nn.ModuleList:
    """src/model/denoiser/attention.py:8-27."""
    if cfg.name == "standard":
        from .standard import StandardTransformer
        return nn.ModuleList([StandardTransformer(cfg, blk.resnets[-1].out_channels) for blk in unet_blocks])
    if cfg.name != "spatial_transformer_3d":
        raise NotImplementedError(cfg.name)
    return nn.ModuleList([SpatialTransformer3D(cfg, blk.resnets[-1].out_channels) for blk in unet_blocks])


@dataclass
class UNetCfg:
    """Mirror of `UNet2DModelCfg` (mvunet.py:22-29)."""
    down_block_types: Sequence[str] = ("DownBlock2D",) * 4
    mid_block_type: str = "UNetMidBlock2D"
    up_block_types: Sequence[str] = ("UpBlock2D",) * 4
    only_cross_attention: bool = False
    block_out_channels: Sequence[int] = (320, 640, 1280, 1280)
    name: str = "unet"


@dataclass
class MVUNetCfg:
    """Mirror of `MultiViewUNetCfg` (mvunet.py:31-40).  `pretrained_overrides` is oracle-only: it
    lets tests build the SD-2.1 *topology* at reduced widths (no checkpoint is reachable)."""
    autoencoder: UNetCfg = field(default_factory=UNetCfg)
    multi_view_attention: object = field(default_factory=MVAttnCfg)
    use_ray_encoding: bool = True
    encoder_conditioning: bool = True
    mid_conditioning: bool = True
    decoder_conditioning: bool = True
    pretrained_from: Optional[str] = None
    name: str = "mv_unet"
    pretrained_overrides: Optional[dict] = None


class MultiViewUNet(nn.Module):
    """mvunet.py:42-208."""

    def __init__(self, cfg: MVUNetCfg, in_channels: int, out_channels: int):
        super().__init__()
        self.cfg = cfg
        self.pretrained_from = cfg.pretrained_from
        a = cfg.autoencoder
        if cfg.pretrained_from is None:
            self.unet = B.UNet2DConditionModel(
                in_channels=in_channels, out_channels=out_channels, down_block_types=a.down_block_types,
                mid_block_type=a.mid_block_type, up_block_types=a.up_block_types,
                only_cross_attention=a.only_cross_attention, block_out_channels=a.block_out_channels,
                cross_attention_dim=list(a.block_out_channels))
        else:
            self.unet = B.UNet2DConditionModel.from_pretrained(cfg.pretrained_from, subfolder="unet",
                                                                config_overrides=cfg.pretrained_overrides)
            c0 = self.unet.config.block_out_channels[0]
            self.unet.conv_in = nn.Conv2d(in_channels, c0, 3, padding=1)
            self.unet.conv_out = nn.Conv2d(c0, out_channels, 3, padding=1)
        if cfg.encoder_conditioning:
            self.cross_attn_blocks_encoder = get_attn_blocks(cfg.multi_view_attention, self.unet.down_blocks)
        if cfg.mid_conditioning:
            self.cross_attn_blocks_mid = get_attn_blocks(cfg.multi_view_attention, [self.unet.mid_block])
        if cfg.decoder_conditioning:
            self.cross_attn_blocks_decoder = get_attn_blocks(cfg.multi_view_attention, self.unet.up_blocks)

    # -- helpers -----------------------------------------------------------------------------------
    def _zero_context(self, h, n_img):
        """mvunet.py:124-128: pretrained => one all-zero 1024-d token; scratch => zeros shaped like
        the feature map's token matrix."""
        if self.pretrained_from is not None:
            return torch.zeros(n_img, 1, self.unet.config.cross_attention_dim, dtype=h.dtype, device=h.device)
        return torch.zeros_like(h).flatten(2).transpose(1, 2)

    def _mv(self, block, h, views):
        n, c, hh, ww = h.shape
        return block(h.reshape(n // views, views, c, hh, ww)).reshape(n, c, hh, ww)

    def forward(self, latents, timestep, cond_state=None):
        b, views = latents.shape[:2]
        t = timestep.reshape(b, -1)
        t = (t.expand(b, views) if t.shape[1] == 1 else t).reshape(b * views)
        emb = self.unet.time_embedding(self.unet.time_proj(t))

        h = self.unet.conv_in(latents.reshape(b * views, *latents.shape[2:]))
        skips = [h]
        for lvl, blk in enumerate(self.unet.down_blocks):
            for i, resnet in enumerate(blk.resnets):
                h = resnet(h, emb)
                if getattr(blk, "has_cross_attention", False):
                    ctx = cond_state if cond_state is not None else self._zero_context(h, b * views)
                    if self.pretrained_from is not None:
                        ctx = self._zero_context(h, b * views)
                    h = blk.attentions[i](h, encoder_hidden_states=ctx).sample
                skips.append(h)
            if h.shape[-2] <= 32 and h.shape[-1] <= 32 and self.cfg.encoder_conditioning:
                h = self._mv(self.cross_attn_blocks_encoder[lvl], h, views)
            if blk.downsamplers is not None:
                for d in blk.downsamplers:
                    h = d(h)
                skips.append(h)

        mid = self.unet.mid_block
        h = mid.resnets[0](h, emb)
        for attn, resnet in zip(mid.attentions, mid.resnets[1:]):
            ctx = cond_state if cond_state is not None else self._zero_context(h, b * views)
            if self.pretrained_from is not None:
                ctx = self._zero_context(h, b * views)
            h = attn(h, encoder_hidden_states=ctx).sample
            h = resnet(h, emb)
        if self.cfg.mid_conditioning:
            h = self._mv(self.cross_attn_blocks_mid[0], h, views)

        for lvl, blk in enumerate(self.unet.up_blocks):
            for i, resnet in enumerate(blk.resnets):
                h = resnet(torch.cat([h, skips.pop()], dim=1), emb)
                if getattr(blk, "has_cross_attention", False) and self.pretrained_from is None:
                    ctx = cond_state if cond_state is not None else self._zero_context(h, b * views)
                    h = blk.attentions[i](h, encoder_hidden_states=ctx).sample
            if h.shape[-2] <= 32 and h.shape[-1] <= 32 and self.cfg.decoder_conditioning:
                h = self._mv(self.cross_attn_blocks_decoder[lvl], h, views)
            if blk.upsamplers is not None:
                for u in blk.upsamplers:
                    h = u(h)

        h = self.unet.conv_out(self.unet.conv_act(self.unet.conv_norm_out(h)))
        return h.reshape(b, views, *h.shape[1:])
