"""Tensor-level wrappers over the C ABI (include/mvldm.h).  torch is plumbing here: device memory,
the current HIP stream and dtype tags.  Every function enqueues hand-written HIP kernels from
libmvldm_hip.so on `torch.cuda.current_stream()`; nothing falls back to torch math.

Conventions: activations are NHWC tensors `[n_img, h, w, c]` (or token matrices `[rows, c]`) in the
activation dtype; biases / norm parameters / statistics are fp32.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import math

import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}


def dt(t) -> int:
    return _DT[t if isinstance(t, torch.dtype) else t.dtype]


def epc(dtype: torch.dtype) -> int:
    """elements per 16-byte chunk: channel counts must be multiples of this"""
    return 4 if dtype == torch.float32 else 8


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _roundup(a: int, b: int) -> int:
    return (a + b - 1) // b * b


# ------------------------------------------------------------------------------------------ workspaces
_WS: dict = {}


def workspace(nbytes: int, device, key: str = "splitk") -> torch.Tensor:
    """grow-only scratch buffer per (device, key); eager ops share it (stream-ordered reuse)."""
    k = (str(device), key)
    cur = _WS.get(k)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[k] = cur
    return cur


# ------------------------------------------------------------------------------------------ weights
@dataclass
class PackedWeight:
    """K-major [n_pad][k_pad] weight in the activation dtype + what the kernel needs to know."""
    data: torch.Tensor
    n_out: int
    n_pad: int
    k_pad: int
    c_pad: int      # channels per tap as seen by the kernel (sum of source channels)
    ksize: int
    geglu: bool = False
    k_order: int = 0    # 1: K stored as (channel block, tap, channel) -- needs every source's channels % BK == 0
    _skinny: Optional[torch.Tensor] = None
    _skinny_pinned: bool = False     # an op of some plan points at the fragment-order copy: it stays with the pack (drop_skinny is a no-op)

    def can_skinny(self) -> bool:
        """does igemm tile 15 (csrc/skinny.hip) read this weight?  16-bit block-major packs whose channel count is a multiple of 64"""
        return self.k_order == 1 and self.data.dtype != torch.float32 and self.c_pad % 64 == 0 and self.k_pad == self.ksize * self.ksize * self.c_pad \
            and self.ksize in (1, 2, 3) and self.n_out % 4 == 0

    def skinny(self) -> torch.Tensor:
        """the same weight in MFMA-fragment order (`mvldm_pack_skinny`), made on first use from the K-major pack and kept with it: the
        operand of igemm tile 15.  A re-pack produces a new PackedWeight (modules._PackMixin), so the copy cannot go stale."""
        if self._skinny is None:
            assert self.can_skinny(), "tile 15 needs a 16-bit block-major pack with channels in multiples of 64"
            out = torch.empty_like(self.data)
            L.check(L.load().mvldm_pack_skinny(self.data.data_ptr(), out.data_ptr(), self.n_pad, self.k_pad, int(self.geglu), dt(self.data), stream()))
            self._skinny = out
        return self._skinny

    def drop_skinny(self):
        if not self._skinny_pinned:
            self._skinny = None


def block_k(dtype: torch.dtype) -> int:
    return 32 if dtype == torch.float32 else 64


def pack_weight_t(w: torch.Tensor, dtype: torch.dtype, c_off: int = 0, n_rows: Optional[int] = None,
                  out: Optional[torch.Tensor] = None) -> PackedWeight:
    """the weight of the DATA-GRADIENT convolution of `w` (fp32 `[n_out, c_in, k, k]` / `[n_out, c_in]`): rows = input
    channels [c_off, c_off + n_rows), K = (flipped tap, output channel).  Fed to the forward implicit GEMM with the
    upstream gradient as its source, it yields dL/dx (stride-1 convs and Linears).  `out`: repack in place."""
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
    n_out, c_in = w.shape[0], w.shape[1]
    ksize = w.shape[2] if w.ndim == 4 else 1
    n_rows = c_in - c_off if n_rows is None else n_rows
    c_pad = _roundup(n_out, epc(dtype))
    bk = block_k(dtype)
    k_pad = _roundup(ksize * ksize * c_pad, bk)
    n_pad = _roundup(n_rows, 64)
    k_order = int(c_pad % bk == 0)
    data = torch.zeros(n_pad, k_pad, dtype=dtype, device=w.device) if out is None else out
    L.check(L.load().mvldm_pack_weight(w.data_ptr(), data.data_ptr(), n_out, c_in, ksize, c_pad, n_pad, k_pad, 0, k_order, dt(dtype),
                                       1, c_off, n_rows, stream()))
    return PackedWeight(data, n_rows, n_pad, k_pad, c_pad, ksize, False, k_order)


def pack_weight(w: torch.Tensor, dtype: torch.dtype, c_pad: Optional[int] = None, geglu: bool = False,
                c_split: Optional[int] = None, out: Optional[torch.Tensor] = None) -> PackedWeight:
    """w: fp32 `[n_out, c_in, k, k]` (conv) or `[n_out, c_in]` (linear), on the GPU.  `c_split`: channel
    count of the first of two concatenated sources (decides whether the block-major K order applies)."""
    assert w.is_cuda, "pack_weight needs a device tensor (no CPU path)"
    w = w.detach().to(torch.float32).contiguous()
    n_out, c_in = w.shape[0], w.shape[1]
    ksize = w.shape[2] if w.ndim == 4 else 1
    e = epc(dtype)
    c_pad = _roundup(c_in, e) if c_pad is None else c_pad
    bk = block_k(dtype)
    k_pad = _roundup(ksize * ksize * c_pad, bk)
    n_pad = _roundup(n_out, 64)
    k_order = int(c_pad % bk == 0 and (c_split is None or c_split % bk == 0))
    out = torch.empty(n_pad, k_pad, dtype=dtype, device=w.device) if out is None else out
    L.check(L.load().mvldm_pack_weight(w.data_ptr(), out.data_ptr(), n_out, c_in, ksize, c_pad, n_pad, k_pad,
                                       int(geglu), k_order, dt(dtype), 0, 0, n_out, stream()))
    return PackedWeight(out, n_out, n_pad, k_pad, c_pad, ksize, geglu, k_order)


def pack_job(w: torch.Tensor, pw: PackedWeight, *, transpose: bool = False, c_off: int = 0) -> "L.PackJob":
    """the re-pack of `pw` (made by `pack_weight` / `pack_weight_t` from the fp32 weight `w`) as one job of `PackBatch`"""
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
    j = L.PackJob()
    j.src, j.dst = w.data_ptr(), pw.data.data_ptr()
    j.n_out, j.c_in, j.ksize = w.shape[0], w.shape[1], pw.ksize
    j.c_pad, j.n_pad, j.k_pad, j.geglu, j.k_order = pw.c_pad, pw.n_pad, pw.k_pad, int(pw.geglu), pw.k_order
    j.transpose, j.c_off, j.n_rows = int(transpose), c_off, (pw.n_out if transpose else w.shape[0])
    L.check(L.load().mvldm_pack_job_prepare(C.byref(j), dt(pw.data.dtype)))
    return j


class PackBatch:
    """many weight re-packs as ONE launch (`mvldm_pack_weight_batch`): the job list lives on the device, a workgroup finds its
    job by its first-workgroup index.  Same bytes out as one `pack_weight(..., out=)` call per job."""

    def __init__(self, jobs: Sequence["L.PackJob"], dtype: torch.dtype, device):
        self.n, self.dtype = len(jobs), dtype
        arr = (L.PackJob * max(len(jobs), 1))()
        b0 = 0
        for i, j in enumerate(jobs):
            C.memmove(C.byref(arr[i]), C.byref(j), C.sizeof(L.PackJob))
            arr[i].block0 = b0
            b0 += j.blocks
        self.total_blocks = b0
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        self.table = raw.to(device)
        # workgroup -> job table (without it the kernel searches the job list: measured slower, HISTORY 3.2)
        self.block_job = None
        if jobs:
            self.block_job = torch.repeat_interleave(torch.arange(len(jobs), dtype=torch.int32), torch.tensor([j.blocks for j in jobs])).to(device)
            assert self.block_job.numel() == b0

    def run(self):
        L.check(L.load().mvldm_pack_weight_batch(self.table.data_ptr(), self.n, None if self.block_job is None else self.block_job.data_ptr(),
                                                 self.total_blocks, dt(self.dtype), stream()))


# ------------------------------------------------------------------------------------------ igemm
def igemm_desc(src0, src1, pw: PackedWeight, dst, *, n_img, h_in, w_in, h_out, w_out, stride=1, pad=None, upsample=False,
               bias=None, row_bias=None, residual=None, epilogue=L.EPI_NONE, out_scale=1.0, ws=None, splitk=0,
               tile=0) -> L.IgemmDesc:
    d = L.IgemmDesc()
    c0 = src0.shape[-1]
    c1 = 0 if src1 is None else src1.shape[-1]
    assert c0 + c1 == pw.c_pad, f"weight packed for {pw.c_pad} channels, sources give {c0}+{c1}"
    d.src0, d.src1, d.weight = ptr(src0), ptr(src1), ptr(pw.data)
    d.bias, d.row_bias, d.residual, d.dst = ptr(bias), ptr(row_bias), ptr(residual), ptr(dst)
    d.c0, d.c1 = c0, c1
    d.n_img, d.h_in, d.w_in, d.h_out, d.w_out = n_img, h_in, w_in, h_out, w_out
    d.ksize, d.stride, d.upsample = pw.ksize, stride, int(upsample)      # 0/1, or 2 + phase (see upsample_phase_weights)
    d.pad = (pw.ksize // 2) if pad is None else pad
    d.n_out, d.n_pad, d.k_pad = pw.n_out, pw.n_pad, pw.k_pad
    d.row_bias_ld = 0 if row_bias is None else row_bias.stride(0)
    d.epilogue, d.act_dtype, d.dst_dtype = epilogue, dt(src0), dt(dst)
    d.splitk, d.tile, d.out_scale = splitk, tile, out_scale
    d.dst_ld = 0
    d.k_order = pw.k_order
    assert not pw.k_order or c0 % block_k(src0.dtype) == 0, "weight packed block-major but the source split is unaligned"
    if (tile & 63) == 15:       # the skinny weight-streaming kernel reads the fragment-order copy of the pack
        d.weight, d.k_order = ptr(pw.skinny()), 2
    if ws is not None:
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    else:
        d.workspace, d.workspace_bytes = None, 0
        if splitk == 0:
            d.splitk = 1
    return d


def conv2d(x: torch.Tensor, pw: PackedWeight, bias=None, *, x2=None, stride=1, pad=None, upsample=False, row_bias=None,
           residual=None, epilogue=L.EPI_NONE, out_dtype=None, out_scale=1.0, splitk=0, tile=0, out=None) -> torch.Tensor:
    """x: NHWC `[n, h, w, c]`; x2: optional second source concatenated along c.  Returns NHWC.  `out`: write into this tensor
    (it may be the residual: x += f(x))."""
    assert x.is_cuda and x.is_contiguous() and (x2 is None or x2.is_contiguous())
    n, h, w, _ = x.shape
    pad = (pw.ksize // 2) if pad is None else pad
    hs, ws_ = (2 * h, 2 * w) if upsample else (h, w)
    if pw.ksize == 3 and stride == 2 and pad == 0:       # VAE encoder: F.pad(0,1,0,1) then stride-2 conv
        ho, wo = (hs + 1 - 3) // 2 + 1, (ws_ + 1 - 3) // 2 + 1
    else:
        ho, wo = (hs + 2 * pad - pw.ksize) // stride + 1, (ws_ + 2 * pad - pw.ksize) // stride + 1
    n_dst = pw.n_out // 2 if epilogue == L.EPI_GEGLU else pw.n_out
    if out is None:
        out = torch.empty(n, ho, wo, n_dst, dtype=out_dtype or x.dtype, device=x.device)
    else:
        assert out.is_contiguous() and out.numel() == n * ho * wo * n_dst and out.dtype == (out_dtype or x.dtype)
        out = out.view(n, ho, wo, n_dst)
    scratch = None
    if splitk != 1:
        # split-K only pays when M*N is small: the C side lowers the split count to what fits
        scratch = workspace(min(16 * n * ho * wo * pw.n_pad * 4, 256 << 20), x.device)
    d = igemm_desc(x, x2, pw, out, n_img=n, h_in=h, w_in=w, h_out=ho, w_out=wo, stride=stride, pad=pad,
                   upsample=upsample, bias=bias, row_bias=row_bias, residual=residual, epilogue=epilogue,
                   out_scale=out_scale, ws=scratch, splitk=splitk, tile=tile)
    L.check(L.load().mvldm_igemm_fwd(C.byref(d), stream()))
    return out


def upsample_phase_weights(w: torch.Tensor) -> list:
    """nearest-2x upsampling followed by a 3x3 / pad-1 conv equals four 2x2 convs on the LOW-resolution image, one per
    output parity (py, px): output (2i+py, 2j+px) reads input rows {i-1+py, i+py} and columns {j-1+px, j+px} with the 3x3
    taps that land on the same source pixel pre-summed (rows: py=0 -> [w0, w1+w2], py=1 -> [w0+w1, w2]; columns alike).
    4/9 of the multiply-adds, exact up to the rounding of the summed weights.  Returns the fp32 `[n, c, 2, 2]` weights of
    phases 0..3 (phase = 2*py + px)."""
    assert w.ndim == 4 and w.shape[2:] == (3, 3)
    w = w.detach().to(torch.float32)
    rows = {0: (w[:, :, 0], w[:, :, 1] + w[:, :, 2]), 1: (w[:, :, 0] + w[:, :, 1], w[:, :, 2])}     # [n, c, 3(kx)] each
    out = []
    for py in (0, 1):
        for px in (0, 1):
            taps = []
            for r in rows[py]:
                cols = (r[:, :, 0], r[:, :, 1] + r[:, :, 2]) if px == 0 else (r[:, :, 0] + r[:, :, 1], r[:, :, 2])
                taps.append(torch.stack(cols, dim=-1))
            out.append(torch.stack(taps, dim=-2).contiguous())       # [n, c, 2(ky), 2(kx)]
    return out


def conv2d_upsample_phases(x: torch.Tensor, pws, bias=None, tile=0, splitk=0) -> torch.Tensor:
    """x NHWC `[n, h, w, c]` (16-bit); `pws`: the 4 packed phase weights -> NHWC `[n, 2h, 2w, n_out]`"""
    n, h, w, _ = x.shape
    out = torch.empty(n, 2 * h, 2 * w, pws[0].n_out, dtype=x.dtype, device=x.device)
    scratch = workspace(min(16 * n * h * w * pws[0].n_pad * 4, 256 << 20), x.device)
    for phase, pw in enumerate(pws):
        d = igemm_desc(x, None, pw, out, n_img=n, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, pad=0, upsample=2 + phase,
                       bias=bias, splitk=splitk, tile=tile, ws=scratch)
        L.check(L.load().mvldm_igemm_fwd(C.byref(d), stream()))
    return out


def linear(x: torch.Tensor, pw: PackedWeight, bias=None, *, residual=None, epilogue=L.EPI_NONE, out_dtype=None,
           splitk=0, tile=0, out=None) -> torch.Tensor:
    """x: `[rows, c]` token matrix."""
    rows, c = x.shape
    y = conv2d(x.view(rows, 1, 1, c), pw, bias, residual=None if residual is None else residual.view(rows, 1, 1, -1),
               epilogue=epilogue, out_dtype=out_dtype, splitk=splitk, tile=tile, out=out)
    return y.view(rows, -1)


# ------------------------------------------------------------------------------------------ norms
def groupnorm(x: torch.Tensor, gamma, beta, groups: int, eps: float, silu: bool, x2=None, stats_out=None) -> torch.Tensor:
    """x NHWC `[n, h, w, c]` (or `[n, hw, c]`); x2: optional second source concatenated along c.
    stats_out: optional fp32 `[n, groups, 2]` receiving (mean, rstd) for the backward pass."""
    assert x.is_cuda and x.is_contiguous() and (x2 is None or x2.is_contiguous())
    n, c0 = x.shape[0], x.shape[-1]
    c1 = 0 if x2 is None else x2.shape[-1]
    hw = math.prod(x.shape[1:-1])
    y = torch.empty(*x.shape[:-1], c0 + c1, dtype=x.dtype, device=x.device)
    ws = workspace(max(n, 1) * L.GN_MAX_CHUNKS * groups * 2 * 8, x.device, "gn")
    L.check(L.load().mvldm_groupnorm_fwd(x.data_ptr(), ptr(x2), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), n, hw, c0, c1,
                                         groups, eps, int(silu), dt(x), ws.data_ptr(), ptr(stats_out), stream()))
    return y


def layernorm(x: torch.Tensor, gamma, beta, eps: float = 1e-5) -> torch.Tensor:
    assert x.is_cuda and x.is_contiguous()
    c = x.shape[-1]
    rows = x.numel() // c
    y = torch.empty_like(x)
    L.check(L.load().mvldm_layernorm_fwd(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, c, eps,
                                         dt(x), stream()))
    return y


# ------------------------------------------------------------------------------------------ attention
def make_segments(q_lens, kv_lens=None, device="cuda") -> torch.Tensor:
    """contiguous segments -> int32 [n_seg, 4] = {q_row0, q_len, kv_row0, kv_len}"""
    kv_lens = q_lens if kv_lens is None else kv_lens
    rows, q0, k0 = [], 0, 0
    for ql, kl in zip(q_lens, kv_lens):
        rows.append([q0, ql, k0, kl])
        q0, k0 = q0 + ql, k0 + kl
    return torch.tensor(rows, dtype=torch.int32, device=device)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, head_dim: int, seg: torch.Tensor,
              max_q_len: int, scale: Optional[float] = None, lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q/k/v: 2-D row-major views `[tokens, >= heads*head_dim]` (may be column slices of one fused
    projection: only the row stride is used).  Returns `[q_tokens, heads*head_dim]`."""
    assert q.is_cuda and q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
    out = torch.empty(q.shape[0], heads * head_dim, dtype=q.dtype, device=q.device)
    scale = head_dim ** -0.5 if scale is None else scale
    L.check(L.load().mvldm_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q.stride(0), k.stride(0),
                                         v.stride(0), out.stride(0), heads, head_dim, seg.data_ptr(), seg.shape[0],
                                         max_q_len, scale, dt(q), ptr(lse), 0 if lse is None else lse.stride(0), stream()))
    return out


# ------------------------------------------------------------------------------------------ small ops
def timestep_embed(timesteps: torch.Tensor, freqs: torch.Tensor, dim: int, flip_sin_to_cos: bool,
                   dtype=torch.float32) -> torch.Tensor:
    assert timesteps.dtype == torch.int64 and timesteps.is_cuda
    out = torch.empty(timesteps.numel(), dim, dtype=dtype, device=timesteps.device)
    L.check(L.load().mvldm_timestep_embed_fwd(timesteps.data_ptr(), freqs.data_ptr(), out.data_ptr(), timesteps.numel(),
                                              dim, int(flip_sin_to_cos), dt(dtype), stream()))
    return out


def eltwise(x: torch.Tensor, op: int, out_dtype=None) -> torch.Tensor:
    assert x.is_cuda and x.is_contiguous()
    y = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    L.check(L.load().mvldm_eltwise_fwd(x.data_ptr(), y.data_ptr(), x.numel(), op, dt(x), dt(y), stream()))
    return y


def silu(x, out_dtype=None):
    return eltwise(x, L.ELT_SILU, out_dtype)


def convert(x, out_dtype):
    return eltwise(x, L.ELT_COPY, out_dtype)


def nchw_to_nhwc(src: torch.Tensor, dtype, dst: Optional[torch.Tensor] = None, c_off: int = 0,
                 dst_c: Optional[int] = None, scale: float = 1.0, shift: float = 0.0,
                 img_map: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 NCHW `[n, c, h, w]` -> NHWC `[n, h, w, dst_c]` (channels [c_off, c_off+c) written), `* scale + shift`;
    `img_map` (int32 `[n]`): source image i lands in destination image img_map[i]."""
    assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous()
    n, c, h, w = src.shape
    if dst is None:
        dst_c = c if dst_c is None else dst_c
        dst = torch.zeros(n, h, w, dst_c, dtype=dtype, device=src.device)
    L.check(L.load().mvldm_nchw_to_nhwc(src.data_ptr(), dst.data_ptr(), n, c, h * w, dst.shape[-1], c_off, dt(dst), scale, shift,
                                        ptr(img_map), stream()))
    return dst


def ray_channels(mode: int = 0, n_origin_octaves: int = 0, n_dir_octaves: int = 0) -> int:
    return int(L.load(required=False).mvldm_ray_channels(mode, n_origin_octaves, n_dir_octaves)) if L.load(required=False) is not None else \
        {0: 6, 1: (6 * n_origin_octaves or 3) + (6 * n_dir_octaves or 3), 2: 6 * (n_origin_octaves + n_dir_octaves)}[mode]


def ray_encode(extrinsics: torch.Tensor, intrinsics: torch.Tensor, h: int, w: int, out_nchw: Optional[torch.Tensor] = None,
               out_nhwc: Optional[torch.Tensor] = None, c_off: int = 0, img_map: Optional[torch.Tensor] = None,
               mode: int = 0, n_origin_octaves: int = 0, n_dir_octaves: int = 0, plucker: bool = False):
    """extrinsics fp32 `[n, 4, 4]` (camera-to-world), intrinsics fp32 `[n, 3, 3]` on the device -> per-pixel
    [origin | direction] of the `h x w` latent grid: fp32 `[n, 6, h, w]` and/or channels [c_off, c_off+6) of an NHWC
    buffer `[.., h, w, C]` (camera i -> image img_map[i])."""
    n = extrinsics.shape[0]
    assert extrinsics.is_cuda and extrinsics.dtype == torch.float32 and extrinsics.is_contiguous() and extrinsics.shape[1:] == (4, 4)
    assert intrinsics.dtype == torch.float32 and intrinsics.is_contiguous() and intrinsics.shape == (n, 3, 3)
    if out_nchw is None and out_nhwc is None:
        out_nchw = torch.empty(n, ray_channels(mode, n_origin_octaves, n_dir_octaves), h, w, dtype=torch.float32, device=extrinsics.device)
    L.check(L.load().mvldm_ray_encode(extrinsics.data_ptr(), intrinsics.data_ptr(), n, h, w, ptr(out_nchw), ptr(out_nhwc),
                                      0 if out_nhwc is None else out_nhwc.shape[-1], c_off,
                                      L.F32 if out_nhwc is None else dt(out_nhwc), ptr(img_map), mode, n_origin_octaves, n_dir_octaves,
                                      int(plucker), stream()))
    return out_nchw if out_nchw is not None else out_nhwc


def posterior_sample(moments: torch.Tensor, noise: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """moments fp32 NCHW `[n, 2c, h, w]` = [mean | logvar]; `(mean + exp(0.5 clamp(logvar, -30, 20)) * noise) * scale`"""
    n, c2, h, w = moments.shape
    assert moments.is_cuda and moments.dtype == torch.float32 and moments.is_contiguous()
    assert noise.dtype == torch.float32 and noise.is_contiguous() and noise.numel() == n * (c2 // 2) * h * w
    out = torch.empty(n, c2 // 2, h, w, dtype=torch.float32, device=moments.device)
    L.check(L.load().mvldm_posterior_sample(moments.data_ptr(), noise.data_ptr(), out.data_ptr(), n, c2 // 2, h * w, scale, stream()))
    return out


def nhwc_to_nchw(src: torch.Tensor, c: Optional[int] = None, c_off: int = 0, scale=1.0, shift=0.0,
                 clamp01=False) -> torch.Tensor:
    assert src.is_cuda and src.is_contiguous()
    n, h, w, sc = src.shape
    c = sc if c is None else c
    out = torch.empty(n, c, h, w, dtype=torch.float32, device=src.device)
    L.check(L.load().mvldm_nhwc_to_nchw(src.data_ptr(), out.data_ptr(), n, c, h * w, sc, c_off, dt(src), scale, shift,
                                        int(clamp01), stream()))
    return out


def ddim_cfg_step(eps, x_t, cond_img, uncond_img, cfg_scale, coef, step_ptr, unet_in=None, clip_range: float = 0.0) -> torch.Tensor:
    """eps: fp32 NHWC `[n_img, h, w, c]`; x_t: fp32 NHWC `[n_tgt, h, w, c]` -> x_{t-1} (same shape).
    coef: fp32 `[n_steps, 4]`; `clip_range` > 0 clamps the predicted x0 (diffusers `clip_sample`)."""
    n_tgt, h, w, c = x_t.shape
    out = torch.empty_like(x_t)
    L.check(L.load().mvldm_ddim_cfg_step(eps.data_ptr(), x_t.data_ptr(), out.data_ptr(), cond_img.data_ptr(), ptr(uncond_img),
                                         n_tgt, h * w, c, cfg_scale, coef.data_ptr(), step_ptr.data_ptr(), ptr(unet_in),
                                         0 if unet_in is None else unet_in.shape[-1],
                                         dt(unet_in) if unet_in is not None else L.F32, coef.shape[0], clip_range, stream()))
    return out


def ddpm_cfg_step(eps_c, eps_u, x_t, noise, cfg_scale, coef, clip_range: float = 0.0) -> torch.Tensor:
    """diffusers `DDPMScheduler.step` (epsilon, fixed_small) behind the optional CFG compose, on flat fp32 tensors of equal size.
    coef: device fp32 [5] = {sqrt(1-a_t), sqrt(a_t), c_x0, c_xt, sigma}; `noise` None when sigma = 0 (t = 0)."""
    for t in (eps_c, eps_u, x_t, noise):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == x_t.numel())
    out = torch.empty_like(x_t)
    L.check(L.load().mvldm_ddpm_cfg_step(eps_c.data_ptr(), ptr(eps_u), x_t.data_ptr(), ptr(noise), out.data_ptr(), x_t.numel(),
                                         cfg_scale, coef.data_ptr(), clip_range, stream()))
    return out


# ------------------------------------------------------------------------------------------ training kernels
def conv_wgrad(x, dy, grad, *, ksize, stride=1, pad=None, upsample=False, x2=None, c_in=None, accumulate=False, n_out=None, form: int = 0) -> torch.Tensor:
    """x (x2): NHWC forward input(s); dy: NHWC / `[m, ld]` upstream gradient (columns [0, n_out)); grad: fp32 PyTorch-layout
    weight gradient `[n_out, c_in, k, k]` / `[n_out, c_in]`, written or accumulated in place."""
    n, h, w, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[-1]
    pad = ksize // 2 if pad is None else pad
    hs, ws_ = (2 * h, 2 * w) if upsample else (h, w)
    ho, wo = (hs + 2 * pad - ksize) // stride + 1, (ws_ + 2 * pad - ksize) // stride + 1
    n_out = grad.shape[0] if n_out is None else n_out
    d = L.WgradDesc()
    d.src0, d.src1, d.dy, d.grad = ptr(x), ptr(x2), ptr(dy), ptr(grad)
    need = n_out * ksize * ksize * (c0 + c1) * 4
    scratch = workspace(min(max(need * 8, 1 << 20), max(need, 512 << 20)), x.device, "wgrad")
    d.workspace, d.workspace_bytes = scratch.data_ptr(), scratch.numel()
    d.c0, d.c1, d.c_in = c0, c1, (c0 + c1) if c_in is None else c_in
    d.n_img, d.h_in, d.w_in, d.h_out, d.w_out = n, h, w, ho, wo
    d.ksize, d.stride, d.pad, d.upsample = ksize, stride, pad, int(upsample)
    # form: 0 = the library's rule, 1 = register-staged small tile, 2 = wide LDS-DMA tile (bits 8-9 of `accumulate`)
    d.n_out, d.dy_ld, d.act_dtype, d.accumulate = n_out, dy.stride(-2), dt(x), int(accumulate) | (int(form) << 8)
    L.check(L.load().mvldm_igemm_wgrad(C.byref(d), stream()))
    return grad


def colsum(x2d, dst, rows_per_seg=None, per_seg=False, accumulate=False, n=None) -> torch.Tensor:
    rows = x2d.shape[0]
    n = x2d.shape[1] if n is None else n
    rows_per_seg = rows if rows_per_seg is None else rows_per_seg
    n_seg = rows // rows_per_seg
    ws = workspace(max((1024 + n_seg) * ((n + 7) // 8 * 8) * 4, 1 << 16), x2d.device, "colsum")
    L.check(L.load().mvldm_colsum(x2d.data_ptr(), dst.data_ptr(), ws.data_ptr(), ws.numel(), n_seg, rows_per_seg, n, x2d.stride(0),
                                  dst.stride(0) if dst.ndim == 2 else n, int(per_seg), int(accumulate), dt(x2d), stream()))
    return dst


def groupnorm_bwd(x, dy, gamma, beta, stats, dgamma, dbeta, groups, silu, x2=None):
    n, c0 = x.shape[0], x.shape[-1]
    c1 = 0 if x2 is None else x2.shape[-1]
    hw = math.prod(x.shape[1:-1])
    dx, dx2 = torch.empty_like(x), (None if x2 is None else torch.empty_like(x2))
    ws = workspace(n * (L.GN_MAX_CHUNKS * (c0 + c1) + groups) * 2 * 4, x.device, "gnb")
    L.check(L.load().mvldm_groupnorm_bwd(x.data_ptr(), ptr(x2), dy.data_ptr(), dx.data_ptr(), ptr(dx2), gamma.data_ptr(), beta.data_ptr(),
                                         stats.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), n, hw, c0, c1, groups, int(silu), dt(x),
                                         ws.data_ptr(), ws.numel(), stream()))
    return dx, dx2


def layernorm_bwd(x, dy, gamma, dgamma, dbeta, eps=1e-5):
    c = x.shape[-1]
    rows = x.numel() // c
    dx = torch.empty_like(x)
    ws = workspace(512 * c * 2 * 4, x.device, "lnb")
    L.check(L.load().mvldm_layernorm_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                         rows, c, eps, dt(x), ws.data_ptr(), ws.numel(), stream()))
    return dx


def attention_bwd(q, k, v, out, dout, lse, heads, head_dim, seg, max_q_len, max_kv_len, scale=None, dqkv=None):
    """returns (dq, dk, dv) `[tokens, heads*head_dim]` (views of `dqkv` `[tokens, 3C]` when given)"""
    C_ = heads * head_dim
    if dqkv is None:
        dq, dk, dv = (torch.empty(t.shape[0], C_, dtype=q.dtype, device=q.device) for t in (q, k, v))
    else:
        dq, dk, dv = dqkv[:, :C_], dqkv[:, C_:2 * C_], dqkv[:, 2 * C_:]
    delta = torch.empty_like(lse)
    d = L.AttnBwdDesc()
    d.q, d.k, d.v, d.out, d.dout, d.dq, d.dk, d.dv = (t.data_ptr() for t in (q, k, v, out, dout, dq, dk, dv))
    d.lse, d.delta, d.seg = lse.data_ptr(), delta.data_ptr(), seg.data_ptr()
    d.ld_q, d.ld_k, d.ld_v, d.ld_o, d.ld_do, d.ld_dq, d.ld_dk, d.ld_dv = (t.stride(0) for t in (q, k, v, out, dout, dq, dk, dv))
    d.heads, d.head_dim, d.n_seg, d.max_q_len, d.max_kv_len = heads, head_dim, seg.shape[0], max_q_len, max_kv_len
    d.total_q_rows, d.stat_ld, d.dtype = q.shape[0], lse.stride(0), dt(q)
    d.scale = head_dim ** -0.5 if scale is None else scale
    L.check(L.load().mvldm_attention_bwd(C.byref(d), stream()))
    return dq, dk, dv


def train_eltwise(op: int, a, b, out, rows: int, d: int):
    L.check(L.load().mvldm_train_eltwise(op, a.data_ptr(), ptr(b), out.data_ptr(), rows, d, dt(a), dt(out), stream()))
    return out


def silu_bwd(x, dy):
    return train_eltwise(L.TE_SILU_BWD, x, dy, torch.empty_like(dy), 1, x.numel())


def geglu_fwd(ag):
    rows, d2 = ag.shape
    return train_eltwise(L.TE_GEGLU_FWD, ag, None, torch.empty(rows, d2 // 2, dtype=ag.dtype, device=ag.device), rows, d2 // 2)


def geglu_bwd(ag, dh):
    rows, d2 = ag.shape
    return train_eltwise(L.TE_GEGLU_BWD, ag, dh, torch.empty_like(ag), rows, d2 // 2)


def pool2x2_sum(du):
    n, h2, w2, c = du.shape
    dx = torch.empty(n, h2 // 2, w2 // 2, c, dtype=du.dtype, device=du.device)
    L.check(L.load().mvldm_pool2x2_sum(du.data_ptr(), dx.data_ptr(), n, h2 // 2, w2 // 2, c, dt(du), stream()))
    return dx


def zero_insert2x(x):
    n, h, w, c = x.shape
    out = torch.empty(n, 2 * h, 2 * w, c, dtype=x.dtype, device=x.device)
    L.check(L.load().mvldm_zero_insert2x(x.data_ptr(), out.data_ptr(), n, h, w, c, dt(x), stream()))
    return out


def grad_norm(flat_grad, max_norm: float, norm_out=None, sumsq_in=None):
    """norm_out fp32 [4]: [total norm, clip coefficient, this buffer's sum of squares, -]"""
    norm_out = torch.zeros(4, dtype=torch.float32, device=flat_grad.device) if norm_out is None else norm_out
    ws = workspace(1024 * 8, flat_grad.device, "norm")
    L.check(L.load().mvldm_grad_norm(flat_grad.data_ptr(), flat_grad.numel(), ptr(sumsq_in), max_norm, norm_out.data_ptr(), ws.data_ptr(), stream()))
    return norm_out


def adamw_step(p, g, m, v, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, step=1, grad_scale=1.0, clip=None):
    assert all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel() for t in (p, g, m, v))
    L.check(L.load().mvldm_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, betas[0], betas[1], eps,
                                      weight_decay, step, grad_scale, ptr(clip), stream()))


def ema_update(avg, p, weight: float):
    """avg.lerp_(p, weight) on flat fp32 buffers (torch.optim.swa_utils EMA: weight = 1 - decay)"""
    assert avg.dtype == p.dtype == torch.float32 and avg.is_contiguous() and p.is_contiguous() and avg.numel() == p.numel()
    L.check(L.load().mvldm_ema_update(avg.data_ptr(), p.data_ptr(), avg.numel(), float(weight), stream()))
