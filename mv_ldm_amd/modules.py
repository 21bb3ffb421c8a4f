"""diffusers-style module surface over the HIP kernels.

These classes expose exactly the attributes the reference touches when it walks a
`diffusers.UNet2DConditionModel` by hand (src/model/denoiser/mvunet.py:90-208):
`time_proj`, `time_embedding`, `conv_in`, `down_blocks[i].resnets[j](x, emb)`, `.resnets[-1].out_channels`,
`.has_cross_attention`, `.attentions[j](x, encoder_hidden_states=ctx).sample`, `.downsamplers`,
`mid_block.resnets/.attentions`, `up_blocks[i].resnets/.attentions/.upsamplers`, `conv_norm_out`,
`conv_act`, `conv_out` -- with the state-dict key layout of diffusers==0.27.2,
so a Lightning `.ckpt` / diffusers state dict loads by key name.

Every module has two entry points that share one code path:
  * `forward(...)`  -- NCHW-shaped tensors in/out (channels_last storage, activation dtype); each
                       kernel is launched eagerly.  This is the drop-in for the reference's own walk.
  * `emit(b, ...)`  -- NHWC tensors through a `plan.Builder`; with a recording builder the whole
                       network becomes one C-side plan / hipGraph (mvunet.MultiViewUNet.compile).
All arithmetic happens in libmvldm_hip.so.  Parameters stay fp32 in torch layout (the checkpoint
format); K-major packed copies in the activation dtype are built lazily per dtype and re-built when
a parameter changes.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import _lib as L
from . import ops
from .ops import PackedWeight
from .plan import Builder
from .runtime import from_nhwc, get_compute_dtype, require_gpu, to_nhwc


# Kernels that update parameters in place (train.py: fused AdamW on the flat buffers) do not move torch's version counters;
# they call `bump_weights_epoch(module)` instead: every parameter of THAT module gets the new epoch (`p._mvldm_epoch`), and every
# pack / plan fingerprint below carries the epochs of the parameters it was built from.  Other modules (the frozen VAE next to a
# trained denoiser) keep theirs, so their packs and recorded plans survive the optimizer step.
_EPOCH_COUNTER = [0]


def bump_weights_epoch(module=None) -> int:
    """mark the parameters of `module` (None: every later fingerprint, via the global floor) as changed in place"""
    _EPOCH_COUNTER[0] += 1
    e = _EPOCH_COUNTER[0]
    if module is None:
        _GLOBAL_FLOOR[0] = e
    else:
        for p in module.parameters():
            p._mvldm_epoch = e
    return e


_GLOBAL_FLOOR = [0]


def _epoch(p) -> int:
    return max(getattr(p, "_mvldm_epoch", 0), _GLOBAL_FLOOR[0])


def _ver(*params):
    return tuple((p.data_ptr(), p._version, _epoch(p), str(p.device)) for p in params if p is not None)


def weights_version(module) -> int:
    """fingerprint of every parameter's (storage, in-place version, in-place-kernel epoch): recorded plans hold raw pointers to
    PACKED copies of the weights, so they are keyed on this and re-recorded after `load_state_dict`, an optimizer step, an EMA copy ..."""
    return hash(tuple((p.data_ptr(), p._version, _epoch(p)) for p in module.parameters()))


class _PackMixin:
    """lazy, version-checked cache of kernel-layout copies of the module's parameters"""

    def _cache(self, key, params, fn):
        store = self.__dict__.setdefault("_packs", {})
        ver = _ver(*params)
        ent = store.get(key)
        if ent is None or ent[0] != ver:
            with torch.no_grad():
                ent = (ver, fn())
            store[key] = ent
        return ent[1]

    def _f32(self, name):
        p = getattr(self, name)
        if p is None:
            return None
        require_gpu(p)
        if p.dtype == torch.float32:
            return p.detach()
        return self._cache(("f32", name), [p], lambda: p.detach().float().contiguous())


def eager_builder(t: torch.Tensor) -> Builder:
    require_gpu(t)
    return Builder(t.device, get_compute_dtype(), record=False)


# ------------------------------------------------------------------------------------------ leaves
class Conv2d(nn.Conv2d, _PackMixin):
    def packed(self, dtype, c_pad=None, c_split=None) -> PackedWeight:
        require_gpu(self.weight)
        aligned = c_split is None or c_split % ops.block_k(dtype) == 0
        return self._cache(("w", dtype, c_pad, aligned), [self.weight],
                           lambda: ops.pack_weight(self.weight, dtype, c_pad=c_pad, c_split=c_split))

    def emit(self, b: Builder, x, x2=None, **kw):
        c_tot = x.shape[-1] + (0 if x2 is None else x2.shape[-1])
        return b.conv(x, self.packed(b.dtype, c_tot, None if x2 is None else x.shape[-1]), self._f32("bias"), x2=x2,
                      stride=self.stride[0], pad=self.padding[0], **kw)

    def forward(self, x):
        b = eager_builder(x)
        return from_nhwc(self.emit(b, to_nhwc(x, b.dtype, ops.epc(b.dtype))))


class Linear(nn.Linear, _PackMixin):
    def packed(self, dtype, geglu=False) -> PackedWeight:
        require_gpu(self.weight)
        return self._cache(("w", dtype, geglu), [self.weight], lambda: ops.pack_weight(self.weight, dtype, geglu=geglu))

    def emit(self, b: Builder, x, **kw):
        return b.linear(x, self.packed(b.dtype), self._f32("bias"), **kw)

    def forward(self, x):
        b = eager_builder(x)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.dtype == b.dtype and x2.is_contiguous() else ops.convert(x2.contiguous(), b.dtype)
        return self.emit(b, x2).reshape(*lead, -1)


class GroupNorm(nn.GroupNorm, _PackMixin):
    def emit(self, b: Builder, x, silu=False, x2=None, name="groupnorm"):
        return b.groupnorm(x, self._f32("weight"), self._f32("bias"), self.num_groups, self.eps, silu, x2=x2, name=name)

    def forward(self, x):
        b = eager_builder(x)
        return from_nhwc(self.emit(b, to_nhwc(x, b.dtype)))


class LayerNorm(nn.LayerNorm, _PackMixin):
    def emit(self, b: Builder, x, name="layernorm"):
        return b.layernorm(x, self._f32("weight"), self._f32("bias"), self.eps, name=name)


class SiLU(nn.Module):
    def forward(self, x):
        require_gpu(x)
        if x.ndim == 4:  # NCHW-shaped, any storage order: run on the NHWC storage and hand back the same view
            return from_nhwc(ops.silu(to_nhwc(x, x.dtype if x.dtype != torch.float64 else torch.float32)))
        return ops.silu(x.contiguous())


# ------------------------------------------------------------------------------------------ time embedding
class Timesteps(nn.Module):
    """diffusers `Timesteps` (mvunet.py:107): fp32 sinusoid, `flip_sin_to_cos`, `freq_shift`."""

    def __init__(self, num_channels: int, flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0.0):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift
        half = num_channels // 2
        # the table is built with the reference's own fp32 torch ops on the host
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32) / (half - downscale_freq_shift)
        self.register_buffer("freqs", torch.exp(exponent), persistent=False)

    def emit(self, b: Builder, timesteps, dtype=torch.float32):
        return b.timestep_embed(timesteps, self.freqs, self.num_channels, self.flip_sin_to_cos, dtype)

    def forward(self, timesteps):
        require_gpu(timesteps)
        return ops.timestep_embed(timesteps.to(torch.int64).contiguous(), self.freqs, self.num_channels,
                                  self.flip_sin_to_cos, torch.float32)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = Linear(time_embed_dim, time_embed_dim)

    def emit(self, b: Builder, t_emb, silu_out=False):
        h = self.linear_1.emit(b, t_emb, epilogue=L.EPI_SILU, name="time_embedding.linear_1")
        out = self.linear_2.emit(b, h, epilogue=L.EPI_SILU if silu_out else L.EPI_NONE, name="time_embedding.linear_2")
        b.free(h)
        return out

    def forward(self, sample):
        b = eager_builder(sample)
        x = sample if sample.dtype == b.dtype else ops.convert(sample.contiguous(), b.dtype)
        return self.emit(b, x)


# ------------------------------------------------------------------------------------------ resnet
class ResnetBlock2D(nn.Module):
    """diffusers `ResnetBlock2D`."""

    def __init__(self, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = GroupNorm(groups, out_channels, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = Conv2d(out_channels, out_channels, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def emit(self, b: Builder, x, temb=None, x2=None, temb_proj=None, out=None):
        """x (+ x2 = skip tensor, concatenated along C on the fly).  `temb_proj`: precomputed
        time_emb_proj(silu(emb)) rows [n_img, out_channels] fp32 (plan mode computes all resnets' at once)."""
        if self.time_emb_proj is not None and temb_proj is None:
            ta = b.eltwise(temb, L.ELT_SILU, name="temb_silu")
            temb_proj = self.time_emb_proj.emit(b, ta, out_dtype=torch.float32, name="time_emb_proj")
            b.free(ta)
        def main_chain():
            g1 = self.norm1.emit(b, x, silu=True, x2=x2, name="norm1+silu")
            h = self.conv1.emit(b, g1, row_bias=temb_proj, name="conv1")
            b.free(g1)
            g2 = self.norm2.emit(b, h, silu=True, name="norm2+silu")
            b.free(h)
            return g2

        if self.conv_shortcut is None:
            assert x2 is None
            g2, sc = main_chain(), x
        elif b.small_launch(x.shape[0] * x.shape[1] * x.shape[2], self.conv_shortcut.out_channels):
            # a few scenes: the 1x1 shortcut (12 - 27 us at one scene, 14 of them) is independent of norm1 -> conv1 -> norm2 and
            # runs beside it as a second lane; both are needed only by conv2
            with b.parallel() as par:
                par.lane()
                g2 = main_chain()
                par.lane()
                sc = self.conv_shortcut.emit(b, x, x2=x2, name="conv_shortcut")
        else:
            g2 = main_chain()
            sc = self.conv_shortcut.emit(b, x, x2=x2, name="conv_shortcut")
        out = self.conv2.emit(b, g2, residual=sc, name="conv2", **({} if out is None else {"out": out}))
        b.free(g2)
        if sc is not x:
            b.free(sc)
        return out

    def forward(self, input_tensor, temb=None):
        b = eager_builder(input_tensor)
        x = to_nhwc(input_tensor, b.dtype)
        if temb is not None and (temb.dtype != b.dtype or not temb.is_contiguous()):
            temb = ops.convert(temb.contiguous(), b.dtype)
        return from_nhwc(self.emit(b, x, temb))


# ------------------------------------------------------------------------------------------ attention
class Attention(nn.Module, _PackMixin):
    """diffusers `Attention` (q/k/v bias-free by default, `to_out.0` with bias).  `emit_*` return
    residual + to_out(attention(...)) fused into the output projection's epilogue."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, out_bias=True):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.inner = heads, dim_head, inner
        kv = query_dim if cross_attention_dim is None else cross_attention_dim
        self.to_q = Linear(query_dim, inner, bias=bias)
        self.to_k = Linear(kv, inner, bias=bias)
        self.to_v = Linear(kv, inner, bias=bias)
        self.to_out = nn.ModuleList([Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])

    def _packed_cat(self, dtype, mods: Sequence[Linear]):
        ws = [m.weight for m in mods]
        return self._cache(("cat", dtype, len(mods)), ws, lambda: ops.pack_weight(torch.cat([w.detach() for w in ws], 0), dtype))

    def _bias_cat(self, mods: Sequence[Linear]):
        if mods[0].bias is None:
            return None
        bs = [m.bias for m in mods]
        return self._cache(("bcat", len(mods)), bs, lambda: torch.cat([x.detach().float() for x in bs], 0).contiguous())

    def out_bias(self, extra: Optional[torch.Tensor] = None):
        bo = self.to_out[0].bias
        if extra is None:
            return self.to_out[0]._f32("bias")
        return self._cache(("bsum",), [bo, extra], lambda: (bo.detach().float() + extra.detach().float()).contiguous())

    def emit_self(self, b: Builder, xn, residual, seg, lens, extra_bias=None, name="attn", kv_lens=None):
        """xn: normalised tokens [M, C]; one fused QKV projection, flash attention, out-proj + residual.  `kv_lens` (with a `seg`
        whose query ranges are sub-ranges of the key ranges): only those query rows are attended and written."""
        C = self.inner
        qkv = b.linear(xn, self._packed_cat(b.dtype, [self.to_q, self.to_k, self.to_v]),
                       self._bias_cat([self.to_q, self.to_k, self.to_v]), name=name + ".to_qkv")
        a = b.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], self.heads, self.dim_head, seg, lens, lens if kv_lens is None else kv_lens,
                        name=name + ".sdpa")
        b.free(qkv)
        out = b.linear(a, self.to_out[0].packed(b.dtype), self.out_bias(extra_bias), residual=residual, name=name + ".to_out")
        b.free(a)
        return out

    def emit_cross(self, b: Builder, xn, ctx, residual, seg, q_lens, kv_lens, name="attn"):
        """ctx: context tokens [Mc, ctx_dim]"""
        C = self.inner
        q = self.to_q.emit(b, xn, name=name + ".to_q")
        kv = b.linear(ctx, self._packed_cat(b.dtype, [self.to_k, self.to_v]), self._bias_cat([self.to_k, self.to_v]), name=name + ".to_kv")
        a = b.attention(q, kv[:, :C], kv[:, C:], self.heads, self.dim_head, seg, q_lens, kv_lens, name=name + ".sdpa")
        b.free(q)
        b.free(kv)
        out = b.linear(a, self.to_out[0].packed(b.dtype), self.out_bias(), residual=residual, name=name + ".to_out")
        b.free(a)
        return out


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    """GEGLU(dim -> 4 dim) -> Linear(4 dim -> dim); keys `net.0.proj`, `net.2` (diffusers and
    mvdream/attention.py:60-87 are the same arithmetic).  The GELU-gate product is the epilogue of the
    first GEMM, the residual add the epilogue of the second."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), Linear(dim * mult, dim)])

    def emit(self, b: Builder, xn, residual, name="ff"):
        p = self.net[0].proj
        g = b.linear(xn, p.packed(b.dtype, geglu=True), p._f32("bias"), epilogue=L.EPI_GEGLU, name=name + ".geglu")
        out = self.net[2].emit(b, g, residual=residual, name=name + ".out")
        b.free(g)
        return out


def _segments(b: Builder, q_lens, kv_lens=None):
    key = (tuple(q_lens), None if kv_lens is None else tuple(kv_lens))
    cache = b.__dict__.setdefault("_segcache", {})
    if key not in cache:
        cache[key] = ops.make_segments(list(q_lens), None if kv_lens is None else list(kv_lens), device=b.device)
        b.keep.append(cache[key])
    return cache[key]


class BasicTransformerBlock(nn.Module):
    """diffusers `BasicTransformerBlock`: LN -> self-attn -> LN -> cross-attn -> LN -> GEGLU FF."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, cross_attention_dim):
        super().__init__()
        self.norm1 = LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, num_attention_heads, attention_head_dim)
        self.norm2 = LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim, num_attention_heads, attention_head_dim)
        self.norm3 = LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def emit(self, b: Builder, hs, n_img, tokens, ctx=None, ctx_tokens=0, zero_ctx=False):
        """hs: [n_img*tokens, C].  zero_ctx: the context is known to be all-zero (mvunet.py:124-128) so
        attn2 == to_out.bias exactly; the bias is folded into attn1's output projection."""
        lens = [tokens] * n_img
        n1 = self.norm1.emit(b, hs, name="norm1")
        extra = self.attn2.to_out[0].bias if zero_ctx else None
        h1 = self.attn1.emit_self(b, n1, hs, _segments(b, lens), lens, extra_bias=extra, name="attn1")
        b.free(n1)
        if not zero_ctx:
            n2 = self.norm2.emit(b, h1, name="norm2")
            kv_lens = [ctx_tokens] * n_img
            h2 = self.attn2.emit_cross(b, n2, ctx, h1, _segments(b, lens, kv_lens), lens, kv_lens, name="attn2")
            b.free(n2)
            b.free(h1)
            h1 = h2
        n3 = self.norm3.emit(b, h1, name="norm3")
        out = self.ff.emit(b, n3, h1, name="ff")
        b.free(n3)
        b.free(h1)
        return out


class Transformer2DModel(nn.Module):
    """diffusers `Transformer2DModel` (continuous input).  `.forward` returns an object with `.sample`."""

    def __init__(self, num_attention_heads, attention_head_dim, in_channels, cross_attention_dim,
                 use_linear_projection=False, norm_num_groups=32, num_layers=1):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.use_linear_projection, self.cross_attention_dim = use_linear_projection, cross_attention_dim
        self.norm = GroupNorm(norm_num_groups, in_channels, eps=1e-6)
        self.proj_in = Linear(in_channels, inner) if use_linear_projection else Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim)
            for _ in range(num_layers)])
        self.proj_out = Linear(inner, in_channels) if use_linear_projection else Conv2d(inner, in_channels, 1)

    def _proj(self, b, mod, x2d, **kw):
        if isinstance(mod, Linear):
            return mod.emit(b, x2d, **kw)
        return b.linear(x2d, mod.packed(b.dtype, x2d.shape[-1]), mod._f32("bias"), **kw)  # 1x1 conv == linear on tokens

    def emit(self, b: Builder, x, ctx=None, zero_ctx=False, out=None):
        """`out`: optional NHWC destination of the block's result (a view of a larger batch buffer)"""
        n, h, w, c = x.shape
        g = self.norm.emit(b, x, name="norm")
        hs = self._proj(b, self.proj_in, g.view(n * h * w, c), name="proj_in")
        b.free(g)
        ctx2d, ctx_tokens = None, 0
        if not zero_ctx:
            assert ctx is not None and ctx.shape[0] == n, "encoder_hidden_states must be [n_img, tokens, dim]"
            ctx_tokens = ctx.shape[1]
            ctx2d = ctx.reshape(n * ctx_tokens, ctx.shape[2])
        for i, blk in enumerate(self.transformer_blocks):
            with b.scope(f"transformer_blocks.{i}"):
                nxt = blk.emit(b, hs, n, h * w, ctx2d, ctx_tokens, zero_ctx)
            b.free(hs)
            hs = nxt
        out = self._proj(b, self.proj_out, hs, residual=x.view(n * h * w, c), name="proj_out", **({} if out is None else {"out": out.view(n * h * w, c)}))
        b.free(hs)
        return out.view(n, h, w, c)

    def forward(self, hidden_states, encoder_hidden_states=None, **_unused):
        b = eager_builder(hidden_states)
        x = to_nhwc(hidden_states, b.dtype)
        ctx = encoder_hidden_states
        if ctx is None:
            raise ValueError("Transformer2DModel needs encoder_hidden_states (the reference always passes one)")
        require_gpu(ctx)
        if ctx.dtype != b.dtype or not ctx.is_contiguous():
            ctx = ops.convert(ctx.contiguous(), b.dtype)
        return SimpleNamespace(sample=from_nhwc(self.emit(b, x, ctx)))


# ------------------------------------------------------------------------------------------ resampling
class Downsample2D(nn.Module):
    def __init__(self, channels, out_channels=None, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def emit(self, b: Builder, x, out=None):
        return self.conv.emit(b, x, name="downsample", **({} if out is None else {"out": out}))   # padding 0 => asymmetric (0,1,0,1) zero pad in-kernel

    def forward(self, x):
        b = eager_builder(x)
        return from_nhwc(self.emit(b, to_nhwc(x, b.dtype)))


class Upsample2D(nn.Module):
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = Conv2d(channels, out_channels or channels, 3, padding=1)

    def emit(self, b: Builder, x):
        c = x.shape[-1]
        if b.dtype != torch.float32 and c % ops.block_k(b.dtype) == 0 and self.conv.out_channels % 8 == 0:
            # nearest x2 + 3x3 = four 2x2 convs on the low-resolution image (4/9 of the multiply-adds)
            pws = self.conv._cache(("wphase", b.dtype), [self.conv.weight],
                                   lambda: [ops.pack_weight(w, b.dtype) for w in ops.upsample_phase_weights(self.conv.weight)])
            return b.conv_upsample_phases(x, pws, self.conv._f32("bias"), name="upsample")
        return self.conv.emit(b, x, upsample=True, name="upsample")  # f32 / odd channel counts: nearest x2 folded into the gather

    def forward(self, x):
        b = eager_builder(x)
        return from_nhwc(self.emit(b, to_nhwc(x, b.dtype)))


# ------------------------------------------------------------------------------------------ UNet blocks
class DownBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers=2, resnet_eps=1e-5, resnet_groups=32,
                 add_downsample=True, downsample_padding=1):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels,
                                                    resnet_groups, resnet_eps) for i in range(num_layers)])
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, out_channels, downsample_padding)])
                             if add_downsample else None)


class CrossAttnDownBlock2D(DownBlock2D):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers=2, resnet_eps=1e-5, resnet_groups=32,
                 num_attention_heads=1, cross_attention_dim=1280, use_linear_projection=False, add_downsample=True,
                 downsample_padding=1):
        super().__init__(in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_groups, add_downsample,
                         downsample_padding)
        self.has_cross_attention = True
        self.attentions = nn.ModuleList([
            Transformer2DModel(num_attention_heads, out_channels // num_attention_heads, out_channels, cross_attention_dim,
                               use_linear_projection, resnet_groups) for _ in range(num_layers)])


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, in_channels, temb_channels, num_layers=1, resnet_eps=1e-5, resnet_groups=32,
                 num_attention_heads=1, cross_attention_dim=1280, use_linear_projection=False):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels, in_channels, temb_channels, resnet_groups, resnet_eps)
                                      for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([
            Transformer2DModel(num_attention_heads, in_channels // num_attention_heads, in_channels, cross_attention_dim,
                               use_linear_projection, resnet_groups) for _ in range(num_layers)])


class VaeAttention(nn.Module, _PackMixin):
    """The VAE mid-block attention (diffusers `Attention` built from the deprecated AttnBlock:
    `group_norm`, biased q/k/v, one head of width C, residual)."""

    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.channels = channels
        self.group_norm = GroupNorm(groups, channels, eps=eps)
        self.to_q = Linear(channels, channels)
        self.to_k = Linear(channels, channels)
        self.to_v = Linear(channels, channels)
        self.to_out = nn.ModuleList([Linear(channels, channels), nn.Dropout(0.0)])

    def emit(self, b: Builder, x):
        n, h, w, c = x.shape
        g = self.group_norm.emit(b, x, name="group_norm")
        ws = [self.to_q.weight, self.to_k.weight, self.to_v.weight]
        bs = [self.to_q.bias, self.to_k.bias, self.to_v.bias]
        pw = self._cache(("qkv", b.dtype), ws, lambda: ops.pack_weight(torch.cat([t.detach() for t in ws], 0), b.dtype))
        bias = self._cache(("qkvb",), bs, lambda: torch.cat([t.detach().float() for t in bs], 0).contiguous())
        qkv = b.linear(g.view(n * h * w, c), pw, bias, name="to_qkv")
        b.free(g)
        lens = [h * w] * n
        a = b.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], 1, c, _segments(b, lens), lens, lens, name="sdpa")
        b.free(qkv)
        out = self.to_out[0].emit(b, a, residual=x.view(n * h * w, c), name="to_out")
        b.free(a)
        return out.view(n, h, w, c)


class UNetMidBlock2D(nn.Module):
    """UNet flavour: `num_layers=0, add_attention=False` -> one resnet (scratch topology, App. A.0);
    VAE flavour: `num_layers=1` with the single-head attention."""

    def __init__(self, in_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, add_attention=True):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels, in_channels, temb_channels, resnet_groups, resnet_eps)
                                      for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([VaeAttention(in_channels, resnet_groups, resnet_eps) for _ in range(num_layers)]
                                        if add_attention else [])

    def emit(self, b: Builder, x, temb=None):
        h = self.resnets[0].emit(b, x, temb)
        for i, (a, r) in enumerate(zip(self.attentions, self.resnets[1:])):
            if a is not None:
                with b.scope(f"attentions.{i}"):
                    h2 = a.emit(b, h)
                b.free(h)
                h = h2
            with b.scope(f"resnets.{i + 1}"):
                h2 = r.emit(b, h, temb)
            b.free(h)
            h = h2
        return h


class UpBlock2D(nn.Module):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=3, resnet_eps=1e-5,
                 resnet_groups=32, add_upsample=True):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            res.append(ResnetBlock2D(rin + skip, out_channels, temb_channels, resnet_groups, resnet_eps))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, out_channels)]) if add_upsample else None


class CrossAttnUpBlock2D(UpBlock2D):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=3, resnet_eps=1e-5,
                 resnet_groups=32, num_attention_heads=1, cross_attention_dim=1280, use_linear_projection=False,
                 add_upsample=True):
        super().__init__(in_channels, prev_output_channel, out_channels, temb_channels, num_layers, resnet_eps,
                         resnet_groups, add_upsample)
        self.has_cross_attention = True
        self.attentions = nn.ModuleList([
            Transformer2DModel(num_attention_heads, out_channels // num_attention_heads, out_channels, cross_attention_dim,
                               use_linear_projection, resnet_groups) for _ in range(num_layers)])


# ------------------------------------------------------------------------------------------ the UNet
SD21_UNET_CONFIG = dict(
    in_channels=4, out_channels=4,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, attention_head_dim=(5, 10, 20, 20),
    cross_attention_dim=1024, use_linear_projection=True, norm_num_groups=32, norm_eps=1e-5)


def _per_block(v, n):
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * n


class UNet2DConditionModel(nn.Module):
    """Container with the diffusers `UNet2DConditionModel` constructor surface the reference uses
    (mvunet.py:54-72) and the sub-module attribute names it walks.  Topologies: SD-2.1
    (`from_pretrained`) and the scratch one (`mid_block_type="UNetMidBlock2D"`)."""

    def __init__(self, in_channels=4, out_channels=4,
                 down_block_types: Sequence[str] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                 mid_block_type: Optional[str] = "UNetMidBlock2DCrossAttn",
                 up_block_types: Sequence[str] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
                 only_cross_attention=False, block_out_channels: Sequence[int] = (320, 640, 1280, 1280),
                 layers_per_block=2, downsample_padding=1, norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280,
                 attention_head_dim=8, use_linear_projection=False, flip_sin_to_cos=True, freq_shift=0):
        super().__init__()
        if only_cross_attention:
            raise NotImplementedError("only_cross_attention=True is not on the reference's path")
        n, boc = len(down_block_types), tuple(block_out_channels)
        heads, xdim = _per_block(attention_head_dim, n), _per_block(cross_attention_dim, n)
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, block_out_channels=boc,
                                      cross_attention_dim=cross_attention_dim, norm_num_groups=norm_num_groups,
                                      norm_eps=norm_eps)
        temb = boc[0] * 4
        self.conv_in = Conv2d(in_channels, boc[0], 3, padding=1)
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
                self.down_blocks.append(CrossAttnDownBlock2D(in_c, out_c, temb, layers_per_block, norm_eps, norm_num_groups,
                                                             heads[i], xdim[i], use_linear_projection, not final,
                                                             downsample_padding))
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
                self.up_blocks.append(UpBlock2D(in_c, prev, out_c, temb, layers_per_block + 1, norm_eps, norm_num_groups,
                                                not final))
            elif t == "CrossAttnUpBlock2D":
                self.up_blocks.append(CrossAttnUpBlock2D(in_c, prev, out_c, temb, layers_per_block + 1, norm_eps,
                                                         norm_num_groups, rheads[i], rxdim[i], use_linear_projection,
                                                         not final))
            else:
                raise ValueError(t)
        self.conv_norm_out = GroupNorm(norm_num_groups, boc[0], eps=norm_eps)
        self.conv_act = SiLU()
        self.conv_out = Conv2d(boc[0], out_channels, 3, padding=1)

    @classmethod
    def from_pretrained(cls, path, subfolder="unet", config_overrides=None, state_dict=None, allow_random_init=False):
        """Builds the SD-2.1 topology.  Weights: `state_dict` (diffusers key layout), or a LOCAL snapshot at `path`
        (`<path>/<subfolder>/diffusion_pytorch_model.safetensors|.bin`); no hub access exists offline, so anything
        else keeps torch's random init -- with a warning unless `allow_random_init` (the reference loads SD-2.1 here,
        mvunet.py:66)."""
        cfg = dict(SD21_UNET_CONFIG)
        cfg.update(config_overrides or {})
        m = cls(**cfg)
        if state_dict is not None:
            m.load_state_dict(state_dict)
            return m
        from .checkpoint import find_local_weights, load_unet_checkpoint
        f = find_local_weights(path, subfolder)
        if f is not None:
            load_unet_checkpoint(m, f)
        elif not allow_random_init:
            import warnings
            warnings.warn(f"UNet2DConditionModel.from_pretrained({path!r}): no local weights found and no state_dict given "
                          "-- the module keeps RANDOM initial weights (pass allow_random_init=True to silence)", stacklevel=2)
        return m

    def enable_xformers_memory_efficient_attention(self):  # diffusion_wrapper.py:147: already flash-style
        return None
