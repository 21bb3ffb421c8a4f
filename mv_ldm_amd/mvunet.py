"""The multi-view denoiser behind the reference's plug-in registries.

Mirrors (same names, argument meaning and error behaviour):
  * `DENOISER` / `get_denoiser`          src/model/denoiser/__init__.py:7-18
  * `Denoiser.forward(latents[b,v,c,h,w], timestep[b | b,v], cond_state=None)`   denoiser.py:12-29
  * `MultiViewUNetCfg`, `UNet2DModelCfg`, `MultiViewUNet`                         mvunet.py:22-208
  * `get_attn_blocks(cfg, unet_blocks)`                                            denoiser/attention.py:8-27
  * `SpatialTransformer3DCfg`, `SpatialTransformer3D`                              mvdream/attention.py:24-32,371-439
State-dict keys are the reference's (`unet.*`, `cross_attn_blocks_{encoder,mid,decoder}.*`).

`MultiViewUNet.forward` runs the WHOLE walk of mvunet.py:90-208 as one C-side plan (see `emit`):
  - all 22 `time_emb_proj(silu(emb))` of the pass are one GEMM issued up front;
  - the skip concat (mvunet.py:176) is never materialised: GroupNorm / the 1x1 shortcut read two sources;
  - SD cross-attention to the all-zero context (mvunet.py:124-128) is exactly `to_out.bias`
    (k = v = 0): folded into the preceding projection's bias, no kernel at all;
  - views that attend to each other are described by `groups` (views per scene), so the conditional
    and unconditional CFG passes of DiffusionWrapper.step can share ONE forward ([v_c+v_t, v_t]).
`forward_walk` is the literal module-by-module walk (each kernel launched eagerly through the
diffusers-style surface), kept as the readable specification and for `cond_state` != None.
"""
from __future__ import annotations

import os

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import _lib as L
from . import ops
from .modules import (Attention, Builder, Conv2d, FeedForward, GroupNorm, LayerNorm, Linear, UNet2DConditionModel, _PackMixin,
                      _segments, eager_builder, weights_version)
from .runtime import from_nhwc, get_compute_dtype, require_gpu, to_nhwc


# ------------------------------------------------------------------------------------------ configs
@dataclass
class SpatialTransformer3DCfg:
    name: str = "spatial_transformer_3d"
    num_heads: int = 8
    num_layers: int = 1
    d_dot: Optional[int] = None
    d_mlp: Optional[int] = None
    d_mlp_multiplier: Optional[int] = None   # ignored by the reference as well
    downscale: int = 1
    pos_enc: bool = False


@dataclass
class CrossAttentionCfg:
    """src/model/denoiser/standard/transformer.py:13-22 (`name: standard`)"""
    name: str = "standard"
    num_heads: int = 8
    num_layers: int = 1
    d_dot: Optional[int] = None
    d_mlp: Optional[int] = None
    d_mlp_multiplier: Optional[int] = 1
    downscale: int = 1
    pos_enc: bool = False


MultiViewAttentionCfg = SpatialTransformer3DCfg      # | CrossAttentionCfg (src/model/denoiser/attention.py:6)


@dataclass
class UNet2DModelCfg:
    name: str = "unet"
    down_block_types: Sequence[str] = ("DownBlock2D",) * 4
    mid_block_type: str = "UNetMidBlock2D"
    up_block_types: Sequence[str] = ("UpBlock2D",) * 4
    only_cross_attention: bool = False
    block_out_channels: Sequence[int] = (320, 640, 1280, 1280)


@dataclass
class MultiViewUNetCfg:
    name: str = "mv_unet"
    autoencoder: UNet2DModelCfg = field(default_factory=UNet2DModelCfg)
    multi_view_attention: object = field(default_factory=SpatialTransformer3DCfg)     # SpatialTransformer3DCfg | CrossAttentionCfg
    use_ray_encoding: bool = True
    encoder_conditioning: bool = True
    mid_conditioning: bool = True
    decoder_conditioning: bool = True
    pretrained_from: Optional[str] = None
    # not in the reference: lets tests / benches build the SD-2.1 *topology* at other widths, and
    # hands a diffusers-layout state dict to `from_pretrained` (no hub access offline)
    pretrained_overrides: Optional[dict] = None
    pretrained_state_dict: Optional[dict] = None
    allow_random_init: bool = False          # silence the "no pretrained weights found" warning (benches / tests)


# ------------------------------------------------------------------------------------------ MV attention
class CrossAttention(Attention):
    """mvdream/attention.py:156-205 (`to_out` is Sequential(Linear, Dropout): same keys as ModuleList)"""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__(query_dim, context_dim, heads, dim_head)


class BasicTransformerBlock3D(nn.Module):
    """mvdream/attention.py:257-296,357-368: attn1 over ALL views' tokens of a scene, attn2 per view."""

    def __init__(self, dim, n_heads, d_head):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, n_heads, d_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, None, n_heads, d_head)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)
        self.norm3 = LayerNorm(dim)

    def emit(self, b: Builder, hs, groups: Sequence[int], tokens: int, keep=None, pair=None):
        """`keep = (keep_rows, drops)`: only the views `keep_rows` (device int32 image rows; per group the first `drops[g]` views are
        dropped) are needed downstream -- the last multi-view block of the sampler, whose context-view outputs nothing reads: the
        3-D attention skips the dropped views' queries (their keys / values still serve the others) and everything behind it runs
        on the kept views, compacted.  Returns the compact token matrix [len(keep_rows) * tokens, C] then."""
        scene_lens = [g * tokens for g in groups]
        n1 = self.norm1.emit(b, hs, name="norm1")
        if pair is not None:
            n_views = sum(groups)
            h1 = self._emit_attn3d_pair(b, n1, hs, groups, tokens, pair)
        elif keep is None:
            n_views = sum(groups)
            h1 = self.attn1.emit_self(b, n1, hs, _segments(b, scene_lens), scene_lens, name="attn1_3d")
        else:
            keep_rows, drops = keep
            n_views = keep_rows.numel()
            seg, row0 = [], 0
            for g, d in zip(groups, drops):
                seg.append([row0 + d * tokens, (g - d) * tokens, row0, g * tokens])
                row0 += g * tokens
            seg_t = torch.tensor(seg, dtype=torch.int32, device=b.device)
            b.keep.append(seg_t)
            full = self.attn1.emit_self(b, n1, hs, seg_t, [(g - d) * tokens for g, d in zip(groups, drops)], name="attn1_3d", kv_lens=scene_lens)
            c = full.shape[-1]
            h1 = b.empty(n_views * tokens, c, dtype=full.dtype)
            b.gather_rows(full.view(sum(groups), tokens * c), h1.view(n_views, tokens * c), src_index=keep_rows, name="keep_views")
            b.free(full)
        view_lens = [tokens] * n_views
        b.free(n1)
        n2 = self.norm2.emit(b, h1, name="norm2")
        h2 = self.attn2.emit_self(b, n2, h1, _segments(b, view_lens), view_lens, name="attn2_view")
        b.free(n2)
        b.free(h1)
        n3 = self.norm3.emit(b, h2, name="norm3")
        out = self.ff.emit(b, n3, h2, name="ff")
        b.free(n3)
        b.free(h2)
        return out


def _emit_attn3d_pair(self, b: Builder, n1, hs, groups, tokens, pair):
    """3-D attention of the FIRST multi-view block of the fused CFG forward.  `pair = (n_cond, cond_img, unc_img)`: the groups are
    [conditional scenes (context + target views)] + [unconditional scenes (the same target views)] and -- the layers in front
    being per-image -- the target views' rows hold identical features in both (MultiViewUNet.emit, `dup`).  Their queries and
    keys / values are therefore identical too, and the conditional softmax over {context, target} keys is the unconditional
    one over {target} keys extended by the context keys:
        launch 1:  unconditional scenes as usual (-> final rows + lse)
        launch 2:  context queries x all keys of their scene (final rows)
        launch 3:  conditional target queries x the context views' keys only (-> partial rows + lse)
        merge:     conditional target rows = combine(unconditional rows, partial rows)       (mvldm_attention_merge)
    25 instead of 41 view x view blocks of scores at 1 context + 4 target views."""
    a1 = self.attn1
    n_cond, cond_img, unc_img = pair
    C = a1.inner
    half = len(groups) // 2
    cond_g, unc_g = list(groups[:half]), list(groups[half:])
    assert sum(cond_g) == n_cond and len(cond_g) == len(unc_g) and all(c > u for c, u in zip(cond_g, unc_g))
    qkv = b.linear(n1, a1._packed_cat(b.dtype, [a1.to_q, a1.to_k, a1.to_v]), a1._bias_cat([a1.to_q, a1.to_k, a1.to_v]), name="attn1_3d.to_qkv")
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    m_full = qkv.shape[0]
    a = b.empty(m_full, C, dtype=qkv.dtype)
    lse1 = b.empty(a1.heads, m_full, dtype=torch.float32)
    seg0, q0, kv0, seg1, q1, kv1, seg2, q2, kv2 = [], [], [], [], [], [], [], [], []
    row0 = 0
    for c, u in zip(cond_g, unc_g):                  # conditional scenes: context views first, then the targets
        vc = c - u
        seg0.append([row0, vc * tokens, row0, c * tokens])                      # context queries x all keys
        q0.append(vc * tokens); kv0.append(c * tokens)
        seg2.append([row0 + vc * tokens, u * tokens, row0, vc * tokens])        # target queries x context keys
        q2.append(u * tokens); kv2.append(vc * tokens)
        row0 += c * tokens
    for u in unc_g:                                  # unconditional scenes
        seg1.append([row0, u * tokens, row0, u * tokens])
        q1.append(u * tokens); kv1.append(u * tokens)
        row0 += u * tokens
    seg0_t, seg1_t, seg2_t = (torch.tensor(sg, dtype=torch.int32, device=b.device) for sg in (seg0, seg1, seg2))
    b.keep.extend([seg0_t, seg1_t, seg2_t])
    b.attention(q, k, v, a1.heads, a1.dim_head, seg1_t, q1, kv1, name="attn1_3d.sdpa", lse=lse1, out=a)
    # (its own launch: a few query tiles per (head, scene) -- the short-sequence launch geometry keeps them on the XCD whose L2 holds
    #  that scene's K / V; inside the long launch they ran at 0.35 ms per view x view block instead of 0.15)
    b.attention(q, k, v, a1.heads, a1.dim_head, seg0_t, q0, kv0, name="attn1_3d.sdpa.ctx_queries", out=a)
    n_cond_rows = n_cond * tokens
    part = b.empty(n_cond_rows, C, dtype=qkv.dtype)
    lse2 = b.empty(a1.heads, n_cond_rows, dtype=torch.float32)
    b.attention(q, k, v, a1.heads, a1.dim_head, seg2_t, q2, kv2, name="attn1_3d.sdpa.ctx_keys", lse=lse2, out=part)
    b.free(qkv)
    b.attention_merge(a, lse1, part, lse2, a, unc_img, cond_img, cond_img, tokens, a1.heads, a1.dim_head, name="attn1_3d.merge")
    b.free(part)
    b.free(lse1)
    b.free(lse2)
    out = b.linear(a, a1.to_out[0].packed(b.dtype), a1.out_bias(None), residual=hs, name="attn1_3d.to_out")
    b.free(a)
    return out


BasicTransformerBlock3D._emit_attn3d_pair = _emit_attn3d_pair


class SpatialTransformer3D(nn.Module):
    """mvdream/attention.py:371-439 (`use_linear=False`).  `proj_out` is zero-initialised like the
    reference's `zero_module` (:407-411)."""

    def __init__(self, cfg: SpatialTransformer3DCfg, d_in: int):
        super().__init__()
        n_heads = cfg.num_heads
        d_head = cfg.d_dot or d_in // n_heads
        self.in_channels = d_in
        self.norm = GroupNorm(32, d_in, eps=1e-6)
        self.proj_in = Conv2d(d_in, d_in, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock3D(d_in, n_heads, d_head) for _ in range(cfg.num_layers)])
        self.proj_out = Conv2d(d_in, d_in, 1)
        nn.init.zeros_(self.proj_out.weight)
        nn.init.zeros_(self.proj_out.bias)

    def emit(self, b: Builder, x, groups: Sequence[int], keep=None, pair=None):
        """x: NHWC [sum(groups), h, w, C]; `groups` = number of views of each scene.  `keep`: see BasicTransformerBlock3D.emit --
        the result then holds the kept views only, [len(keep_rows), h, w, C]."""
        n, h, w, c = x.shape
        assert n == sum(groups)
        g = self.norm.emit(b, x, name="norm")
        hs = b.linear(g.view(n * h * w, c), self.proj_in.packed(b.dtype, c), self.proj_in._f32("bias"), name="proj_in")
        b.free(g)
        last = len(self.transformer_blocks) - 1
        for i, blk in enumerate(self.transformer_blocks):
            with b.scope(f"transformer_blocks.{i}"):
                nxt = blk.emit(b, hs, groups, h * w, keep=keep if i == last else None, pair=pair if i == 0 else None)
            b.free(hs)
            hs = nxt
        res, n_out = x, n
        if keep is not None:
            n_out = keep[0].numel()
            res = b.empty(n_out, h, w, c, dtype=x.dtype)
            b.gather_rows(x, res, src_index=keep[0], name="keep_views.residual")
        out = b.linear(hs, self.proj_out.packed(b.dtype, c), self.proj_out._f32("bias"), residual=res.view(n_out * h * w, c), name="proj_out")
        b.free(hs)
        if keep is not None:
            b.free(res)
        return out.view(n_out, h, w, c)

    def forward(self, x, context=None):
        bsz, v, c, h, w = x.shape
        b = eager_builder(x)
        y = self.emit(b, to_nhwc(x.reshape(bsz * v, c, h, w), b.dtype), [v] * bsz)
        return from_nhwc(y).reshape(bsz, v, c, h, w)


# ------------------------------------------------------------------------------------------ "standard" MV attention
class _PreNorm(nn.Module):
    """src/model/transformer/pre_norm.py:30-37 (keys `norm.*`, `fn.*`)"""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = LayerNorm(dim)
        self.fn = fn


class _ViTAttention(nn.Module, _PackMixin):
    """src/model/transformer/attention.py:36-101, self-attention form: fused bias-free `to_qkv`, `to_out` = Sequential(Linear, Dropout)
    unless heads == 1 and dim_head == dim (then identity)."""

    def __init__(self, dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.inner = heads, dim_head, inner
        self.to_qkv = Linear(dim, 3 * inner, bias=False)
        self.to_out = nn.Sequential(Linear(inner, dim), nn.Dropout(0.0)) if not (heads == 1 and dim_head == dim) else nn.Identity()


class _ViTFeedForward(nn.Module):
    """src/model/transformer/feed_forward.py:29-40: Linear -> GELU -> Dropout -> Linear -> Dropout (keys net.0, net.3)"""

    def __init__(self, dim, hidden):
        super().__init__()
        self.net = nn.Sequential(Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0), Linear(hidden, dim), nn.Dropout(0.0))


class _ViTTransformer(nn.Module):
    """src/model/transformer/transformer.py:33-69: x = attn(norm(x)) + x; x = ff(norm(x)) + x per layer"""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([_PreNorm(dim, _ViTAttention(dim, heads, dim_head)),
                                                    _PreNorm(dim, _ViTFeedForward(dim, mlp_dim))]) for _ in range(depth)])


class StandardTransformer(nn.Module):
    """`StandardTransformer` (src/model/denoiser/standard/transformer.py:45-136): a pre-norm ViT stack over ALL views' tokens of
    a scene -- no GroupNorm / projections / outer residual around it (the block's output replaces the feature map).
    `downscale > 1` (strided conv, transposed conv, 7x7 refinement convs) is not in the kernel family and refused;
    `pos_enc=True` raises NameError exactly like the reference (`cond_features` is undefined there, :102)."""

    def __init__(self, cfg: CrossAttentionCfg, d_in: int, d_kv: Optional[int] = None):
        super().__init__()
        assert (cfg.d_mlp is None) != (cfg.d_mlp_multiplier is None), "Expected exactly one of d_mlp and d_mlp_multiplier"
        if d_kv is not None:
            raise NotImplementedError("StandardTransformer(d_kv=...): the reference never passes it (denoiser/attention.py:13-19)")
        if cfg.downscale > 1:
            raise NotImplementedError("StandardTransformer downscale > 1 (strided / transposed / 7x7 convs) is not on the released path")
        self.cfg, self.pos_enc, self.num_heads, self.d_in = cfg, cfg.pos_enc, cfg.num_heads, d_in
        self.transformer = _ViTTransformer(d_in, cfg.num_layers, cfg.num_heads, cfg.d_dot or d_in // cfg.num_heads,
                                           cfg.d_mlp or d_in * cfg.d_mlp_multiplier)

    def emit(self, b: Builder, x, groups: Sequence[int]):
        """x: NHWC [sum(groups), h, w, C]; one attention segment per scene (all its views' tokens)"""
        if self.pos_enc:
            raise NameError("name 'cond_features' is not defined")     # standard/transformer.py:102
        n, h, w, c = x.shape
        assert n == sum(groups)
        lens = [g * h * w for g in groups]
        hs = x.view(n * h * w, c)
        first = True
        for i, (attn, ff) in enumerate(self.transformer.layers):
            with b.scope(f"transformer.layers.{i}"):
                a = attn.fn
                n1 = attn.norm.emit(b, hs, name="attn.norm")
                qkv = a.to_qkv.emit(b, n1, name="attn.to_qkv")
                b.free(n1)
                o = b.attention(qkv[:, :a.inner], qkv[:, a.inner:2 * a.inner], qkv[:, 2 * a.inner:], a.heads, a.dim_head,
                                _segments(b, lens), lens, lens, name="attn.sdpa")
                b.free(qkv)
                if isinstance(a.to_out, nn.Identity):
                    raise NotImplementedError("heads == 1 with dim_head == dim (no output projection)")
                h1 = a.to_out[0].emit(b, o, residual=hs, name="attn.to_out")
                b.free(o)
                if not first:
                    b.free(hs)
                n2 = ff.norm.emit(b, h1, name="ff.norm")
                g = ff.fn.net[0].emit(b, n2, epilogue=L.EPI_GELU, name="ff.net.0+gelu")
                b.free(n2)
                hs = ff.fn.net[3].emit(b, g, residual=h1, name="ff.net.3")
                b.free(g)
                b.free(h1)
                first = False
        return hs.view(n, h, w, c)

    def forward(self, features):
        bsz, v, c, h, w = features.shape
        b = eager_builder(features)
        y = self.emit(b, to_nhwc(features.reshape(bsz * v, c, h, w), b.dtype), [v] * bsz)
        return from_nhwc(y).reshape(bsz, v, c, h, w)


def get_attn_blocks(cfg, unet_blocks) -> nn.ModuleList:
    """src/model/denoiser/attention.py:8-27"""
    if cfg.name == "standard":
        return nn.ModuleList([StandardTransformer(cfg=cfg, d_in=block.resnets[-1].out_channels) for block in unet_blocks])
    if cfg.name == "spatial_transformer_3d":
        return nn.ModuleList([SpatialTransformer3D(cfg=cfg, d_in=block.resnets[-1].out_channels) for block in unet_blocks])
    raise NotImplementedError(f"multi_view_attention '{cfg.name}': the reference's get_attn_blocks knows 'standard' and "
                              "'spatial_transformer_3d' (src/model/denoiser/attention.py:12,21)")


# ------------------------------------------------------------------------------------------ denoiser
class Denoiser(nn.Module):
    """src/model/denoiser/denoiser.py:12-29"""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg


class _PrefixDone(Exception):
    """`MultiViewUNet.emit(prefix_only=True)` reached the first multi-view block"""

    def __init__(self, skips):
        super().__init__()
        self.skips = skips


class MultiViewUNet(Denoiser, _PackMixin):
    def __init__(self, cfg: MultiViewUNetCfg, in_channels: int, out_channels: int):
        super().__init__(cfg)
        self.use_ray_encoding = cfg.use_ray_encoding
        self.pretrained_from = cfg.pretrained_from
        a = cfg.autoencoder
        if self.pretrained_from is None:
            self.unet = UNet2DConditionModel(
                in_channels=in_channels, out_channels=out_channels, down_block_types=a.down_block_types,
                mid_block_type=a.mid_block_type, up_block_types=a.up_block_types,
                only_cross_attention=a.only_cross_attention, block_out_channels=a.block_out_channels,
                cross_attention_dim=list(a.block_out_channels))
        else:
            self.unet = UNet2DConditionModel.from_pretrained(self.pretrained_from, subfolder="unet",
                                                             config_overrides=getattr(cfg, "pretrained_overrides", None),
                                                             state_dict=getattr(cfg, "pretrained_state_dict", None),
                                                             allow_random_init=getattr(cfg, "allow_random_init", False))
            c0 = self.unet.config.block_out_channels[0]
            self.unet.conv_in = Conv2d(in_channels, c0, kernel_size=3, padding=1)
            self.unet.conv_out = Conv2d(c0, out_channels, kernel_size=3, padding=1)
        if cfg.encoder_conditioning:
            self.cross_attn_blocks_encoder = get_attn_blocks(cfg.multi_view_attention, self.unet.down_blocks)
        if cfg.mid_conditioning:
            self.cross_attn_blocks_mid = get_attn_blocks(cfg.multi_view_attention, [self.unet.mid_block])
        if cfg.decoder_conditioning:
            self.cross_attn_blocks_decoder = get_attn_blocks(cfg.multi_view_attention, self.unet.up_blocks)
        self.in_channels, self.out_channels = in_channels, out_channels
        self._plans = {}

    # ---- fused whole-forward emission -------------------------------------------------------------
    def _resnets_in_order(self) -> List:
        u = self.unet
        rs = [r for blk in u.down_blocks for r in blk.resnets] + list(u.mid_block.resnets)
        return rs + [r for blk in u.up_blocks for r in blk.resnets]

    def _temb_proj_all(self, b: Builder, emb_act):
        """all `time_emb_proj` of the pass in ONE GEMM -> {resnet: fp32 [n_img, C_out] column slice}"""
        rs = self._resnets_in_order()
        ws = [r.time_emb_proj.weight for r in rs]
        bs = [r.time_emb_proj.bias for r in rs]
        pw = self._cache(("tproj", b.dtype), ws, lambda: ops.pack_weight(torch.cat([w.detach() for w in ws], 0), b.dtype))
        bias = self._cache(("tprojb",), bs, lambda: torch.cat([x.detach().float() for x in bs], 0).contiguous())
        allp = b.linear(emb_act, pw, bias, out_dtype=torch.float32, name="time_emb_proj[all]")
        out, off = {}, 0
        for r in rs:
            out[id(r)] = allp[:, off:off + r.out_channels]
            off += r.out_channels
        return out

    def _zero_ctx_ok(self, cond_state) -> bool:
        # cond_state is None on every call the reference makes (diffusion_wrapper.py:401,435,441)
        return cond_state is None

    def emit(self, b: Builder, x_in, timesteps, groups: Sequence[int], out: Optional[torch.Tensor] = None,
             dup: Optional[tuple] = None, prefix_only: bool = False, tail: Optional[tuple] = None):
        """x_in: NHWC [n_img, h, w, c_pad] (11 real channels, zero padded); timesteps: int64 [n_img];
        groups: views per scene for the 3-D attention.  Returns eps, fp32 NHWC [n_img, h, w, out_channels].

        `dup = (n_src, src_rows)`: images [n_src, n_img) carry exactly the inputs (latents, mask, rays, timestep) of images
        `src_rows[k]` (device int32 [n_img - n_src]) -- the unconditional pass of classifier-free guidance re-submits the target
        views of the conditional pass (diffusion_wrapper.py:437-441).  Every layer in front of the first multi-view attention block
        works per image (convs, GroupNorm per image, per-view self-attention, time-embedding rows), so it is evaluated once per
        distinct input and the feature maps -- the running activation and the skip connections collected so far -- are copied
        to the rows that share it by row gathers right before that block.  Same arithmetic on the same values: the result is what
        the full-batch walk gives (to the rounding of a different tile choice).  Two forms:
          * `(n_src, src_rows)`: the shared layers run on images [0, n_src) (conditional pass: context + target views), the
            duplicates [n_src, n_img) are filled in: 320 instead of 576 images at 1 context + 4 target views;
          * `(n_src, src_rows, const_rows, const_skips)`: additionally the images `const_rows` (the context views: their latents,
            mask, rays and timestep 0 do not change during sampling) take their feature maps from `const_skips`, the tensors an
            `emit(..., prefix_only=True)` over those images returned (the sampler's loader plan runs it once per sample); the
            shared layers then run on the contiguous duplicate block [n_src, n_img) = one copy of every target view (256 images)
            and both the rows `src_rows` and `const_rows` are filled in.
        `prefix_only=True`: walk the shared layers only and return the list of feature maps collected up to the first multi-view
        block (conv_in output, the down blocks' skip tensors) instead of eps.

        `tail = (keep_rows, drops)`: only the eps of the images `keep_rows` is read (the sampler: the target views; the context
        views' prediction is never used, diffusion_wrapper.py:444-451).  When the last multi-view block sits in the last up block,
        nothing behind its 3-D attention mixes images any more: that attention skips the other views' queries, the rest of the
        block and the output stage run on the kept views only, and their eps is scattered back to the full layout (the other
        rows of eps are left undefined).  `drops[g]` = number of leading views of group g that are not kept."""
        u = self.unet
        n_img = x_in.shape[0]
        assert n_img == sum(groups)
        if dup is not None and (os.environ.get("MVLDM_CFG_SHARE", "1") == "0" or dup[0] >= n_img):
            dup = None
        const = None
        if dup is not None and len(dup) == 4:
            const = (dup[2], dup[3]) if os.environ.get("MVLDM_CFG_SHARE", "1") != "1a" else None
        n_src = n_img if dup is None else int(dup[0])
        shared = dup is not None          # True while only the distinct images are being computed
        lo, hi = (n_src, n_img) if const is not None else (0, n_src)      # rows of the batch the shared layers run on
        with b.scope("time"):
            t_emb = u.time_proj.emit(b, timesteps, dtype=b.dtype)
            emb_act = u.time_embedding.emit(b, t_emb, silu_out=True)   # every consumer applies SiLU first
            b.free(t_emb)
            tproj = self._temb_proj_all(b, emb_act)
            b.free(emb_act)

        def dest(shape_of):        # `out=` for the op that produces a skip / the running activation while the prefix is shared
            if not shared:
                return {}, None
            full = b.empty(n_img, *shape_of, dtype=b.dtype)
            return {"out": full[lo:hi]}, full

        def resnet(r, name, h, skip=None, **kw):
            with b.scope(name):
                tp_ = tproj[id(r)]
                return r.emit(b, h, None, x2=skip, temb_proj=tp_[lo:hi] if shared else tp_, **kw)

        def sd_attn(attn, name, h, **kw):
            with b.scope(name):
                return attn.emit(b, h, None, zero_ctx=True, **kw)

        fulls = {}                 # id(prefix view) -> its full-batch buffer

        def fill(full, i, name):
            """rows of a full-batch feature map that did not compute it themselves"""
            if const is None:
                b.gather_rows(full, full[n_src:], src_index=dup[1], name=name)                     # duplicates <- their sources
            else:
                b.gather_rows(full[n_src:], full, dst_index=dup[1], name=name)                     # sources <- the duplicate block
                b.gather_rows(const[1][i], full, dst_index=const[0], name=name + ".const")         # constant rows <- the per-sample store

        def expand(h):
            """the prefix ends here: fill the other images of the running activation and of every skip collected so far"""
            nonlocal shared
            if prefix_only:
                raise _PrefixDone(list(skips))
            if not shared:
                return h
            shared = False
            with b.scope("cfg_share"):
                for i, t in enumerate(skips):
                    full = fulls[id(t)]
                    fill(full, i, f"skip{i}")
                    skips[i] = full
                assert id(h) in fulls and any(fulls[id(h)] is t for t in skips), "the running activation at a multi-view block is a skip"
                return fulls[id(h)]

        def mv(blocks, idx, name, h, **kw):
            was_shared = shared
            h = expand(h)
            if (was_shared and isinstance(blocks[idx], SpatialTransformer3D) and len(groups) % 2 == 0
                    and os.environ.get("MVLDM_CFG_SHARE_ATTN", "1") != "0"):
                # the first multi-view block: the target views' rows are still identical in both passes -- share their scores
                unc_img = torch.arange(n_src, n_img, dtype=torch.int32, device=b.device)
                b.keep.append(unc_img)
                kw = dict(kw, pair=(n_src, dup[1], unc_img))
            with b.scope(name):
                return blocks[idx].emit(b, h, groups, **kw)

        with b.scope("conv_in"):
            kw, full = dest((x_in.shape[1], x_in.shape[2], u.conv_in.out_channels))
            h = u.conv_in.emit(b, x_in[lo:hi] if shared else x_in, name="conv", **kw)
            if full is not None:
                fulls[id(h)] = full
        skips = [h]
        try:
            if tail is not None and os.environ.get("MVLDM_TAIL_DROP", "1") == "0":
                tail = None
            return self._emit_body(b, u, h, skips, fulls, dest, resnet, sd_attn, mv, expand, out, groups, tail)
        except _PrefixDone as done:
            return done.skips

    def _emit_body(self, b, u, h, skips, fulls, dest, resnet, sd_attn, mv, expand, out, groups, tail=None):
        live_mv = None  # an MV-block output that is not a skip (freed once consumed)
        compact = False    # True once the last multi-view block has dropped the views whose eps nobody reads (`tail`)
        n_img_full = sum(groups)
        for lvl, blk in enumerate(u.down_blocks):
            has_attn = getattr(blk, "has_cross_attention", False)
            for i, r in enumerate(blk.resnets):
                # (while the prefix is shared, the op that produces a skip tensor writes into the head of a full-batch buffer)
                kw, full = ({}, None) if has_attn else dest((h.shape[1], h.shape[2], r.conv2.out_channels))
                nh = resnet(r, f"down{lvl}.resnets.{i}", h, **kw)
                if live_mv is not None:
                    b.free(live_mv)
                    live_mv = None
                h = nh
                if has_attn:
                    kw, full = dest(tuple(h.shape[1:]))
                    h2 = sd_attn(blk.attentions[i], f"down{lvl}.attentions.{i}", h, **kw)
                    b.free(h)
                    h = h2
                if full is not None:
                    fulls[id(h)] = full
                skips.append(h)
            if h.shape[1] <= 32 and h.shape[2] <= 32 and self.cfg.encoder_conditioning:
                h = mv(self.cross_attn_blocks_encoder, lvl, f"mv_encoder.{lvl}", h)
                live_mv = h
            if blk.downsamplers is not None:
                for d in blk.downsamplers:
                    with b.scope(f"down{lvl}.downsample"):
                        kw, full = dest(((h.shape[1] + 1) // 2, (h.shape[2] + 1) // 2, d.conv.out_channels))
                        nh = d.emit(b, h, **kw)
                        if full is not None:
                            fulls[id(nh)] = full
                    if live_mv is not None:
                        b.free(live_mv)
                        live_mv = None
                    h = nh
                skips.append(h)

        mid = u.mid_block
        h = expand(h)              # (a model without encoder-side multi-view blocks: the shared prefix ends at the mid block)
        nh = resnet(mid.resnets[0], "mid.resnets.0", h)
        if live_mv is not None:
            b.free(live_mv)
            live_mv = None
        h = nh
        for i, (attn, r) in enumerate(zip(mid.attentions, mid.resnets[1:])):
            h2 = sd_attn(attn, f"mid.attentions.{i}", h)
            b.free(h)
            h3 = resnet(r, f"mid.resnets.{i + 1}", h2)
            b.free(h2)
            h = h3
        if self.cfg.mid_conditioning:
            h2 = mv(self.cross_attn_blocks_mid, 0, "mv_mid", h)
            b.free(h)
            h = h2

        for lvl, blk in enumerate(u.up_blocks):
            has_attn = getattr(blk, "has_cross_attention", False) and self.pretrained_from is None
            for i, r in enumerate(blk.resnets):
                skip = skips.pop()
                nh = resnet(r, f"up{lvl}.resnets.{i}", h, skip)
                b.free(h)
                b.free(skip)
                h = nh
                if has_attn:
                    h2 = sd_attn(blk.attentions[i], f"up{lvl}.attentions.{i}", h)
                    b.free(h)
                    h = h2
            if h.shape[1] <= 32 and h.shape[2] <= 32 and self.cfg.decoder_conditioning:
                # the last multi-view block in the last up block: only the kept views go on (see `tail`)
                drop_here = (tail is not None and lvl == len(u.up_blocks) - 1 and blk.upsamplers is None
                             and isinstance(self.cross_attn_blocks_decoder[lvl], SpatialTransformer3D))
                h2 = mv(self.cross_attn_blocks_decoder, lvl, f"mv_decoder.{lvl}", h, **({"keep": tail} if drop_here else {}))
                b.free(h)
                h = h2
                compact = drop_here
            if blk.upsamplers is not None:
                for up in blk.upsamplers:
                    with b.scope(f"up{lvl}.upsample"):
                        h2 = up.emit(b, h)
                    b.free(h)
                    h = h2
        with b.scope("out"):
            g = u.conv_norm_out.emit(b, h, silu=True, name="conv_norm_out+silu")
            b.free(h)
            if compact:
                eps_c = u.conv_out.emit(b, g, out_dtype=torch.float32, name="conv_out")
                eps = out if out is not None else b.empty(n_img_full, *eps_c.shape[1:], dtype=torch.float32)
                b.gather_rows(eps_c, eps, dst_index=tail[0], name="eps.scatter")
                b.free(eps_c)
            else:
                eps = u.conv_out.emit(b, g, out_dtype=torch.float32, out=out, name="conv_out")
            b.free(g)
        return eps

    # ---- drop-in forward ---------------------------------------------------------------------------
    def compile(self, b_scenes: int, views: int, h: int, w: int, dtype=None, graph: bool = True):
        """build (and cache) the plan for latents [b, v, in_channels, h, w]"""
        dtype = dtype or get_compute_dtype()
        dev = next(self.parameters()).device
        key = (b_scenes, views, h, w, dtype, str(dev), graph)
        st = self._plans.get(key)
        if st is not None and st["weights_version"] == self.weights_version():
            return st
        n = b_scenes * views
        bld = Builder(dev, dtype, record=True)
        c_pad = (self.in_channels + ops.epc(dtype) - 1) // ops.epc(dtype) * ops.epc(dtype)
        lat = torch.zeros(n, self.in_channels, h, w, dtype=torch.float32, device=dev)
        ts = torch.zeros(n, dtype=torch.int64, device=dev)
        x_in = torch.zeros(n, h, w, c_pad, dtype=dtype, device=dev)
        out = torch.zeros(n, self.out_channels, h, w, dtype=torch.float32, device=dev)
        bld.nchw_to_nhwc(lat, x_in)
        eps = self.emit(bld, x_in, ts, [views] * b_scenes)
        bld.nhwc_to_nchw(eps, out)
        plan = bld.finalize()
        if graph:
            plan.capture()
        st = dict(plan=plan, lat=lat, ts=ts, out=out, graph=graph, weights_version=self.weights_version())
        self._plans[key] = st
        return st

    def weights_version(self) -> int:
        return weights_version(self)

    def forward(self, latents, timestep, cond_state=None):
        """latents [b, v, c, h, w]; timestep int64 [b] or [b, v] -> [b, v, out_channels, h, w] (fp32)."""
        require_gpu(latents)
        if not self._zero_ctx_ok(cond_state):
            return self.forward_walk(latents, timestep, cond_state)
        bsz, v, c, h, w = latents.shape
        st = self.compile(bsz, v, h, w)
        t = timestep.reshape(bsz, -1)
        t = (t.expand(bsz, v) if t.shape[1] == 1 else t).reshape(bsz * v)
        st["lat"].copy_(latents.reshape(bsz * v, c, h, w))
        st["ts"].copy_(t)
        st["plan"].replay() if st["graph"] else st["plan"].run()
        return st["out"].view(bsz, v, self.out_channels, h, w).clone()

    def forward_walk(self, latents, timestep, cond_state=None):
        """The walk of mvunet.py:90-208, one diffusers-style module call at a time (eager kernels)."""
        bsz, v = latents.shape[:2]
        u = self.unet
        t = timestep.reshape(bsz, -1)
        t = (t.expand(bsz, v) if t.shape[1] == 1 else t).reshape(bsz * v).to(torch.int64)
        emb = u.time_embedding(u.time_proj(t))
        hs = u.conv_in(latents.reshape(bsz * v, *latents.shape[2:]))
        skips = [hs]

        def ctx_for(x):
            if self.pretrained_from is not None:            # mvunet.py:127-128
                return torch.zeros(bsz * v, 1, 1024, device=x.device)
            if cond_state is None:                           # mvunet.py:124-125
                return torch.zeros(x.shape[0], x.shape[2] * x.shape[3], x.shape[1], device=x.device)
            return cond_state

        def mv(block, x):
            n, c, hh, ww = x.shape
            return block(x.reshape(bsz, v, c, hh, ww)).reshape(n, c, hh, ww)

        for lvl, blk in enumerate(u.down_blocks):
            for i, r in enumerate(blk.resnets):
                hs = r(hs, emb)
                if getattr(blk, "has_cross_attention", False):
                    hs = blk.attentions[i](hs, encoder_hidden_states=ctx_for(hs)).sample
                skips.append(hs)
            if hs.shape[-2] <= 32 and hs.shape[-1] <= 32 and self.cfg.encoder_conditioning:
                hs = mv(self.cross_attn_blocks_encoder[lvl], hs)
            if blk.downsamplers is not None:
                for d in blk.downsamplers:
                    hs = d(hs)
                skips.append(hs)
        hs = u.mid_block.resnets[0](hs, emb)
        for attn, r in zip(u.mid_block.attentions, u.mid_block.resnets[1:]):
            hs = attn(hs, encoder_hidden_states=ctx_for(hs)).sample
            hs = r(hs, emb)
        if self.cfg.mid_conditioning:
            hs = mv(self.cross_attn_blocks_mid[0], hs)
        for lvl, blk in enumerate(u.up_blocks):
            for i, r in enumerate(blk.resnets):
                hs = r(torch.cat((hs, skips.pop()), dim=1), emb)
                if getattr(blk, "has_cross_attention", False) and self.pretrained_from is None:
                    hs = blk.attentions[i](hs, encoder_hidden_states=ctx_for(hs)).sample
            if hs.shape[-2] <= 32 and hs.shape[-1] <= 32 and self.cfg.decoder_conditioning:
                hs = mv(self.cross_attn_blocks_decoder[lvl], hs)
            if blk.upsamplers is not None:
                for up in blk.upsamplers:
                    hs = up(hs)
        hs = u.conv_out(u.conv_act(u.conv_norm_out(hs)))
        return hs.float().reshape(bsz, v, *hs.shape[1:])


DENOISER = {"mv_unet": MultiViewUNet}
DenoiserCfg = MultiViewUNetCfg


def get_denoiser(denoiser_cfg: DenoiserCfg, in_channels: int, out_channels: int) -> Denoiser:
    """src/model/denoiser/__init__.py:13-18"""
    return DENOISER[denoiser_cfg.name](denoiser_cfg, in_channels, out_channels)
