"""Host-side graph builder for the C plan executor (include/mvldm.h "Plans").

A `Builder` turns module-level calls (conv, groupnorm, attention, ...) into `mvldm_op` records with
all device pointers resolved.  Two modes share one code path:

  * eager  -- every op is launched immediately through `mvldm_op_run` on the current stream (this is
              what the diffusers-style module objects use when the reference walks them one by one);
  * record -- ops are appended to a list; `finalize()` hands the list to `mvldm_plan_create` and
              returns a `Plan` that runs the whole forward in C++ (optionally as one hipGraph) with
              no Python in the loop.

torch is only the allocator here (`torch.empty`) and the owner of the HIP stream.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _lib as L
from .ops import PackedWeight, dt, ptr


def _PIN_TILE() -> int:
    """MVLDM_IGEMM_TILE=<n> (read per call: tests set it with monkeypatch): every 16-bit block-major conv / Linear a Builder emits
    without an explicit tile takes tile n and a single K pass instead of the size rules / the plan-time tuning (the split-K count follows the
    workgroup count, i.e. the row count, and changes the summation order like a tile does).  With both pinned a row's dot product
    is the same instruction sequence whatever the row count of the launch, so two plans that push the same rows through differently
    sized launches (the shared CFG prefix against the full walk) can be compared far below the 16-bit rounding of a re-tiled sum."""
    v = os.environ.get("MVLDM_IGEMM_TILE")
    return int(v) if v else 0


@dataclass
class OpMeta:
    name: str
    kind: int
    flops: float = 0.0        # algorithmic 2*MAC
    bytes: float = 0.0        # algorithmic HBM bytes (compulsory reads + writes)


class Builder:
    def __init__(self, device, dtype: torch.dtype, record: bool = False, splitk_ws_bytes: int = 256 << 20):
        self.device, self.dtype, self.record = torch.device(device), dtype, record
        self.ops: List[L.Op] = []
        self.meta: List[OpMeta] = []
        self.keep: list = []            # tensors referenced by recorded ops
        self._pool: dict = {}           # (numel_bytes) -> [tensors] free list for temporaries
        self._ws = None
        self._ws_bytes = splitk_ws_bytes
        self._gn_ws = None
        self._scope: List[str] = []
        self._lane = 0                  # > 0 inside a parallel group (`parallel()`): which lane the next ops belong to
        self._lane_ws: dict = {}        # lane -> its own split-K / GroupNorm workspaces (lanes run concurrently)
        self._deferred: Optional[list] = None
        self.skinny = True              # may launches of a few hundred rows take igemm tile 15 (the fragment-order pack of the weight)?
        self._skinny_pw: dict = {}      # op index -> PackedWeight of the convs / Linears tile 15 could compute

    # ---- memory ---------------------------------------------------------------------------------
    def empty(self, *shape, dtype=None) -> torch.Tensor:
        dtype = dtype or self.dtype
        if self.record:
            nbytes = int(torch.Size(shape).numel()) * torch.empty(0, dtype=dtype).element_size()
            free = self._pool.get(nbytes)
            if free:
                base = free.pop()
                return base.view(dtype).view(*shape)
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        if self.record:
            self.keep.append(t)
            self.__dict__.setdefault("_own_ptrs", set()).add(t.data_ptr())
        return t

    def free(self, t: Optional[torch.Tensor]):
        """return a temporary to the pool (record mode): later ops may overwrite it.  Execution is
        stream-ordered in op order, so reuse after the last consumer was emitted is safe."""
        if t is None or not self.record:
            return
        base = t.reshape(-1).view(torch.uint8)
        if self._deferred is not None:      # inside a parallel group another lane could pick the buffer up while this lane's
            self._deferred.append(base)     # op is still running: hand it back when the lanes have joined
            return
        self._pool.setdefault(base.numel(), []).append(base)

    # ---- parallel lanes -------------------------------------------------------------------------
    class _Parallel:
        """`with b.parallel() as par:` -- the ops emitted after each `par.lane()` form one lane; the lanes are declared mutually
        independent (MVLDM_OP_PAR_* markers: side streams / parallel hipGraph branches in the C executor).  Lanes get their own
        split-K / GroupNorm workspaces and temporaries freed inside the group return to the pool only after the join."""

        def __init__(self, b):
            self.b, self.n = b, 0

        def __enter__(self):
            assert self.b._deferred is None, "parallel groups do not nest"
            self.b._marker(L.OP_PAR_BEGIN, "par_begin")
            self.b._deferred = []
            return self

        def lane(self):
            if self.n:
                self.b._marker(L.OP_PAR_NEXT, "par_next")
            self.b._lane = self.n
            self.n += 1

        def __exit__(self, *a):
            b = self.b
            b._marker(L.OP_PAR_END, "par_end")
            b._lane = 0
            deferred, b._deferred = b._deferred, None
            for base in deferred:
                b._pool.setdefault(base.numel(), []).append(base)

    def parallel(self):
        return Builder._Parallel(self)

    def _marker(self, kind: int, name: str):
        op = L.Op()
        op.kind = kind
        self._emit(op, name)

    def splitk_ws(self) -> torch.Tensor:
        if self._lane:
            ws = self._lane_ws.get(("splitk", self._lane))
            if ws is None:
                ws = torch.empty(min(self._ws_bytes, 64 << 20), dtype=torch.uint8, device=self.device)
                self._lane_ws[("splitk", self._lane)] = ws
                self.keep.append(ws)
            return ws
        if self._ws is None:
            self._ws = torch.empty(self._ws_bytes, dtype=torch.uint8, device=self.device)
            self.keep.append(self._ws)
        return self._ws

    def gn_ws(self, n_img: int, groups: int) -> torch.Tensor:
        need = n_img * L.GN_MAX_CHUNKS * groups * 16
        if self._lane:
            ws = self._lane_ws.get(("gn", self._lane))
            if ws is None or ws.numel() < need:
                ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=self.device)
                self._lane_ws[("gn", self._lane)] = ws
                self.keep.append(ws)
            return ws
        if self._gn_ws is None or self._gn_ws.numel() < need:
            self._gn_ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=self.device)
            self.keep.append(self._gn_ws)
        return self._gn_ws

    # ---- bookkeeping ----------------------------------------------------------------------------
    class _Scope:
        def __init__(self, b, name):
            self.b, self.name = b, name

        def __enter__(self):
            self.b._scope.append(self.name)

        def __exit__(self, *a):
            self.b._scope.pop()

    def scope(self, name: str):
        return Builder._Scope(self, name)

    def _emit(self, op: L.Op, name: str, flops=0.0, nbytes=0.0, refs=()):
        if self.record:
            op.tag = len(self.ops)
            self.ops.append(op)
            self.meta.append(OpMeta("/".join(self._scope + [name]), op.kind, flops, nbytes))
            self.keep.extend(r for r in refs if r is not None)
        else:
            L.check(L.load().mvldm_op_run(C.byref(op), torch.cuda.current_stream().cuda_stream))

    # ---- ops ------------------------------------------------------------------------------------
    def conv(self, x, pw: PackedWeight, bias=None, *, x2=None, stride=1, pad=None, upsample=False, row_bias=None,
             residual=None, epilogue=L.EPI_NONE, out_dtype=None, out_scale=1.0, out=None, name="conv", splitk=0, tile=0):
        """x (and x2): NHWC `[n, h, w, c]` contiguous; returns NHWC."""
        n, h, w, c0 = x.shape
        c1 = 0 if x2 is None else x2.shape[-1]
        assert c0 + c1 == pw.c_pad, f"{name}: weight packed for {pw.c_pad} channels, got {c0}+{c1}"
        pad = (pw.ksize // 2) if pad is None else pad
        hs, ws_ = (2 * h, 2 * w) if upsample else (h, w)
        if pw.ksize == 3 and stride == 2 and pad == 0:
            ho, wo = (hs + 1 - 3) // 2 + 1, (ws_ + 1 - 3) // 2 + 1
        else:
            ho, wo = (hs + 2 * pad - pw.ksize) // stride + 1, (ws_ + 2 * pad - pw.ksize) // stride + 1
        n_dst = pw.n_out // 2 if epilogue == L.EPI_GEGLU else pw.n_out
        if out is None:
            out = self.empty(n, ho, wo, n_dst, dtype=out_dtype or x.dtype)
        ws = self.splitk_ws() if splitk != 1 else None
        op = L.Op()
        op.kind = L.OP_IGEMM
        d = op.u.igemm
        d.src0, d.src1, d.weight = ptr(x), ptr(x2), ptr(pw.data)
        d.bias, d.row_bias, d.residual, d.dst = ptr(bias), ptr(row_bias), ptr(residual), ptr(out)
        d.c0, d.c1 = c0, c1
        d.n_img, d.h_in, d.w_in, d.h_out, d.w_out = n, h, w, ho, wo
        d.ksize, d.stride, d.pad, d.upsample = pw.ksize, stride, pad, int(upsample)
        d.n_out, d.n_pad, d.k_pad = pw.n_out, pw.n_pad, pw.k_pad
        d.row_bias_ld = 0 if row_bias is None else row_bias.stride(0)
        d.epilogue, d.act_dtype, d.dst_dtype = epilogue, dt(x), dt(out)
        if tile == 0 and _PIN_TILE() and x.dtype != torch.float32 and pw.k_order == 1:
            tile, splitk = _PIN_TILE(), 1        # (test knob MVLDM_IGEMM_TILE: one tile, one K pass for every 16-bit block-major launch, see _PIN_TILE)
        m = n * ho * wo
        can_sk = self.skinny and skinny_candidate(pw, m, c0, c1, upsample, x.dtype)
        d.splitk, d.tile, d.out_scale = splitk, tile, out_scale
        d.dst_ld = out.shape[-1] if out.shape[-1] != n_dst else 0
        d.k_order = pw.k_order
        can_sk = can_sk and x.is_cuda and L.load().mvldm_igemm_skinny_config(C.byref(d)) > 0      # (an image may be larger than the kernel's row tiles)
        if tile == 0 and can_sk and skinny_rule(m, pw.n_pad):
            tile = d.tile = 15                   # the rule: launches of a few hundred rows stream their weights (csrc/skinny.hip)
        if (tile & 63) == 15:
            d.weight, d.k_order = ptr(pw.skinny()), 2
        elif can_sk and self.record and tile == 0:
            self._skinny_pw[len(self.ops)] = pw  # a plan-time candidate (autotune_igemm)
        assert not pw.k_order or c0 % (32 if x.dtype == torch.float32 else 64) == 0, f"{name}: block-major weight, unaligned source split"
        if ws is not None:
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        else:
            d.workspace, d.workspace_bytes, d.splitk = None, 0, 1
        k_real = pw.ksize * pw.ksize * (c0 + c1)
        es = x.element_size()
        nbytes = (pw.n_out * k_real + n * h * w * (c0 + c1)) * es + m * n_dst * out.element_size() \
            + (m * n_dst * es if residual is not None else 0)
        if self.record:
            # (only temporaries of this builder's own pool may be filled with timing data: a caller's tensor is left alone)
            own = self.__dict__.get("_own_ptrs", ())
            self.__dict__.setdefault("_tune_srcs", {})[len(self.ops)] = tuple(t if (t is not None and t.data_ptr() in own) else None
                                                                              for t in (x, x2))
        self._emit(op, name, 2.0 * m * pw.n_out * k_real, nbytes, (x, x2, pw.data, pw._skinny if (tile & 63) == 15 else None, bias, row_bias, residual, out))
        return out

    def small_launch(self, rows: int, n_out: int) -> bool:
        """should independent launches of this size run as parallel lanes?  OFF by default (MVLDM_PAR_ROWS=0): measured on MI355X /
        ROCm 7 the fork + join of a two-to-four-branch section of a hipGraph costs ~80 us, more than the lanes save -- with the
        3 upsamplers' phase convs and the 14 shortcut convs as lanes (MVLDM_PAR_ROWS=16384: launches of <= 16384 rows x 640 columns)
        one DDIM step at 1 scene went 6.10 -> 7.1 ms, at 4 scenes 12.2 -> 12.9 (profiles/r03_parallel_lanes.txt).  The executor
        support and the builder API stay (correct, tested); the threshold is what a runtime with cheaper graph edges would set."""
        limit = int(os.environ.get("MVLDM_PAR_ROWS", "0"))
        return self.record and rows * max(n_out, 1) <= limit * 640 and limit > 0

    def conv_upsample_phases(self, x, pws, bias=None, name="upsample"):
        """nearest-2x + 3x3 conv as four 2x2 phase convs on the low-resolution input (ops.upsample_phase_weights)"""
        n, h, w, c0 = x.shape
        out = self.empty(n, 2 * h, 2 * w, pws[0].n_out, dtype=x.dtype)
        # the four phases write disjoint pixels of `out`: at a few scenes one phase is far from filling the chip (b = 1: 36 us each
        # at 210 TFLOP/s) and they run as parallel lanes; big launches gain nothing from sharing the CUs and stay serial
        lanes = self.parallel() if self.small_launch(n * h * w, pws[0].n_out) else None
        if lanes is not None:
            lanes.__enter__()
        for phase, pw in enumerate(pws):
            if lanes is not None:
                lanes.lane()
            self._phase_conv(x, pw, bias, out, phase, name)
        if lanes is not None:
            lanes.__exit__(None, None, None)
        return out

    def _phase_conv(self, x, pw, bias, out, phase, name, tile=0, splitk=0):
        n, h, w, c0 = x.shape
        ws = self.splitk_ws()        # small batches: K = 4C is long and there are few output tiles -- let the library split K
        assert pw.ksize == 2 and pw.k_order == 1 and c0 == pw.c_pad
        op = L.Op()
        op.kind = L.OP_IGEMM
        d = op.u.igemm
        d.src0, d.src1, d.weight = ptr(x), None, ptr(pw.data)
        d.bias, d.row_bias, d.residual, d.dst = ptr(bias), None, None, ptr(out)
        d.c0, d.c1 = c0, 0
        d.n_img, d.h_in, d.w_in, d.h_out, d.w_out = n, h, w, h, w
        d.ksize, d.stride, d.pad, d.upsample = 2, 1, 0, 2 + phase
        d.n_out, d.n_pad, d.k_pad = pw.n_out, pw.n_pad, pw.k_pad
        d.row_bias_ld = 0
        d.epilogue, d.act_dtype, d.dst_dtype = L.EPI_NONE, dt(x), dt(out)
        m = n * h * w
        can_sk = self.skinny and skinny_candidate(pw, m, c0, 0, False, x.dtype)
        d.splitk, d.tile, d.out_scale = splitk, tile, 1.0
        d.dst_ld = 0
        d.k_order = 1
        can_sk = can_sk and x.is_cuda and L.load().mvldm_igemm_skinny_config(C.byref(d)) > 0
        if tile == 0 and can_sk and skinny_rule(m, pw.n_pad):
            tile = d.tile = 15
        if (tile & 63) == 15:
            d.weight, d.k_order = ptr(pw.skinny()), 2
        elif can_sk and self.record and tile == 0:
            self._skinny_pw[len(self.ops)] = pw
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        es = x.element_size()
        nbytes = (pw.n_out * 4 * c0 + (m * c0 if phase == 0 else 0)) * es + m * pw.n_out * es
        if self.record:
            own = self.__dict__.get("_own_ptrs", ())
            self.__dict__.setdefault("_tune_srcs", {})[len(self.ops)] = (x if x.data_ptr() in own else None, None)
        self._emit(op, f"{name}.p{phase}", 2.0 * m * pw.n_out * 4 * c0, nbytes, (x, pw.data, pw._skinny if (tile & 63) == 15 else None, bias, out))

    def linear(self, x, pw: PackedWeight, bias=None, *, residual=None, epilogue=L.EPI_NONE, out_dtype=None, out=None,
               name="linear", row_bias=None):
        """x: `[rows, c]`."""
        rows, c = x.shape
        y = self.conv(x.view(rows, 1, 1, c), pw, bias, residual=None if residual is None else residual.view(rows, 1, 1, -1),
                      epilogue=epilogue, out_dtype=out_dtype, out=None if out is None else out.view(rows, 1, 1, -1),
                      name=name, row_bias=row_bias)
        return y.view(rows, -1)

    def groupnorm(self, x, gamma, beta, groups, eps, silu, x2=None, name="groupnorm", stats_out=None):
        n, c0 = x.shape[0], x.shape[-1]
        c1 = 0 if x2 is None else x2.shape[-1]
        hw = x.numel() // (n * c0)
        y = self.empty(*x.shape[:-1], c0 + c1, dtype=x.dtype)
        op = L.Op()
        op.kind = L.OP_GROUPNORM
        g = op.u.groupnorm
        g.x, g.x1, g.y, g.gamma, g.beta = ptr(x), ptr(x2), ptr(y), ptr(gamma), ptr(beta)
        g.stats_ws = self.gn_ws(n, groups).data_ptr()
        g.n_img, g.hw, g.c0, g.c1, g.groups, g.silu, g.dtype, g.eps = n, hw, c0, c1, groups, int(silu), dt(x), eps
        g.stats_out = ptr(stats_out)
        # bytes: what the chosen kernel form moves (fused: 1 read + 1 write; statistics + apply launches: 2 reads + 1 write)
        passes = L.load().mvldm_groupnorm_passes(n, hw, c0 + c1, groups, dt(x))
        self._emit(op, name, 0.0, float(passes) * y.numel() * y.element_size(), (x, x2, y, gamma, beta, stats_out))
        return y

    def layernorm(self, x, gamma, beta, eps=1e-5, name="layernorm"):
        c = x.shape[-1]
        rows = x.numel() // c
        y = self.empty(*x.shape, dtype=x.dtype)
        op = L.Op()
        op.kind = L.OP_LAYERNORM
        l = op.u.layernorm
        l.x, l.y, l.gamma, l.beta, l.rows, l.c, l.dtype, l.eps = ptr(x), ptr(y), ptr(gamma), ptr(beta), rows, c, dt(x), eps
        self._emit(op, name, 0.0, 2.0 * y.numel() * y.element_size(), (x, y, gamma, beta))
        return y

    def attention(self, q, k, v, heads, head_dim, seg, q_lens, kv_lens, scale=None, name="attention", lse=None, out=None):
        """q/k/v: 2-D views with unit column stride (may be slices of one fused projection).  `out`: existing [rows, heads * head_dim]
        buffer (rows are addressed by the segments' query rows); `lse`: fp32 [heads, rows] log-sum-exp output (log2 domain)."""
        assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
        if out is None:
            out = self.empty(q.shape[0], heads * head_dim, dtype=q.dtype)
        op = L.Op()
        op.kind = L.OP_ATTENTION
        a = op.u.attention
        a.q, a.k, a.v, a.out, a.seg = ptr(q), ptr(k), ptr(v), ptr(out), ptr(seg)
        a.ld_q, a.ld_k, a.ld_v, a.ld_o = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
        a.heads, a.head_dim, a.n_seg, a.max_q_len, a.dtype = heads, head_dim, seg.shape[0], max(q_lens), dt(q)
        a.scale = head_dim ** -0.5 if scale is None else scale
        a.lse, a.lse_ld = ptr(lse), (0 if lse is None else lse.stride(0))
        pairs = sum(ql * kl for ql, kl in zip(q_lens, kv_lens))
        es = q.element_size()
        nbytes = (sum(q_lens) * 2 + sum(kv_lens) * 2) * heads * head_dim * es
        self._emit(op, name, 4.0 * pairs * heads * head_dim, nbytes, (q, k, v, out, seg, lse))
        return out

    def attention_merge(self, oa, lse_a, ob, lse_b, out, a_img, b_img, out_img, tokens, heads, head_dim, name="attention_merge"):
        """combine two attention results of the same queries over disjoint key sets (mvldm_attention_merge), per image of `tokens` rows"""
        n = a_img.numel()
        assert b_img.numel() == n and out_img.numel() == n and all(i.dtype == torch.int32 for i in (a_img, b_img, out_img))
        assert lse_a.dtype == torch.float32 and lse_b.dtype == torch.float32 and oa.dtype == ob.dtype == out.dtype
        op = L.Op()
        op.kind = L.OP_ATTN_MERGE
        m = op.u.attn_merge
        m.oa, m.ob, m.out, m.lse_a, m.lse_b, m.a_img, m.b_img, m.out_img = ptr(oa), ptr(ob), ptr(out), ptr(lse_a), ptr(lse_b), ptr(a_img), ptr(b_img), ptr(out_img)
        m.n_img, m.tokens, m.heads, m.head_dim = n, tokens, heads, head_dim
        m.ld_a, m.ld_b, m.ld_o, m.lse_ld_a, m.lse_ld_b, m.dtype = oa.stride(0), ob.stride(0), out.stride(0), lse_a.stride(0), lse_b.stride(0), dt(oa)
        self._emit(op, name, 0.0, 3.0 * n * tokens * heads * head_dim * oa.element_size(), (oa, lse_a, ob, lse_b, out, a_img, b_img, out_img))

    def timestep_embed(self, timesteps, freqs, dim, flip, dtype, name="time_proj"):
        out = self.empty(timesteps.numel(), dim, dtype=dtype)
        op = L.Op()
        op.kind = L.OP_TIMESTEP_EMBED
        t = op.u.temb
        t.timesteps, t.freqs, t.out, t.n, t.dim, t.flip, t.dst_dtype = ptr(timesteps), ptr(freqs), ptr(out), timesteps.numel(), dim, int(flip), dt(out)
        self._emit(op, name, 0.0, out.numel() * out.element_size(), (timesteps, freqs, out))
        return out

    def eltwise(self, x, op_code, out_dtype=None, name="eltwise"):
        y = self.empty(*x.shape, dtype=out_dtype or x.dtype)
        op = L.Op()
        op.kind = L.OP_ELTWISE
        e = op.u.eltwise
        e.x, e.y, e.n, e.op, e.src_dtype, e.dst_dtype = ptr(x), ptr(y), x.numel(), op_code, dt(x), dt(y)
        self._emit(op, name, 0.0, x.numel() * (x.element_size() + y.element_size()), (x, y))
        return y

    def ddim_step(self, eps, x_t, x_next, cond_img, uncond_img, cfg_scale, coef, step_ptr, unet_in, name="ddim_cfg_step",
                  clip_range: float = 0.0):
        n_tgt, h, w, c = x_t.shape
        op = L.Op()
        op.kind = L.OP_DDIM_STEP
        d = op.u.ddim
        d.eps, d.x_t, d.x_next, d.cond_img, d.uncond_img = ptr(eps), ptr(x_t), ptr(x_next), ptr(cond_img), ptr(uncond_img)
        d.coef, d.step_ptr, d.unet_in = ptr(coef), ptr(step_ptr), ptr(unet_in)
        d.n_tgt, d.hw, d.c, d.cfg_scale = n_tgt, h * w, c, cfg_scale
        d.unet_in_c = 0 if unet_in is None else unet_in.shape[-1]
        d.unet_in_dtype = L.F32 if unet_in is None else dt(unet_in)
        d.n_steps, d.clip_range = coef.shape[0], clip_range
        self._emit(op, name, 0.0, 5.0 * x_t.numel() * 4, (eps, x_t, x_next, cond_img, uncond_img, coef, step_ptr, unet_in))
        return x_next

    def ddim_advance(self, step_ptr, t_table, timesteps, tgt_rows, name="ddim_advance"):
        op = L.Op()
        op.kind = L.OP_DDIM_ADVANCE
        a = op.u.advance
        a.step_ptr, a.t_table, a.timesteps, a.tgt_rows = ptr(step_ptr), ptr(t_table), ptr(timesteps), ptr(tgt_rows)
        a.n_steps, a.n_rows = t_table.numel(), 0 if tgt_rows is None else tgt_rows.numel()
        self._emit(op, name, 0.0, 0.0, (step_ptr, t_table, timesteps, tgt_rows))

    def memcpy(self, dst, src, name="memcpy"):
        assert dst.is_contiguous() and src.is_contiguous() and dst.numel() * dst.element_size() == src.numel() * src.element_size()
        op = L.Op()
        op.kind = L.OP_MEMCPY
        op.u.memcpy_.src, op.u.memcpy_.dst, op.u.memcpy_.bytes = ptr(src), ptr(dst), src.numel() * src.element_size()
        self._emit(op, name, 0.0, 2.0 * src.numel() * src.element_size(), (dst, src))

    def gather_rows(self, src, dst, src_index=None, dst_index=None, n_rows=None, name="gather_rows"):
        """dst[dst_index[k] or k] = src[src_index[k] or k] over the leading dimension (rows = whole images), k < n_rows; src and dst
        may alias (disjoint rows)"""
        idx = src_index if src_index is not None else dst_index
        n = int(n_rows if n_rows is not None else (idx.numel() if idx is not None else dst.shape[0]))
        row_bytes = dst[0].numel() * dst.element_size()
        assert src[0].numel() * src.element_size() == row_bytes and row_bytes % 16 == 0 and dst.is_contiguous() and src.is_contiguous()
        assert all(i is None or (i.dtype == torch.int32 and i.numel() == n) for i in (src_index, dst_index))
        assert (src_index is not None or src.shape[0] >= n) and (dst_index is not None or dst.shape[0] >= n)
        op = L.Op()
        op.kind = L.OP_GATHER_ROWS
        g = op.u.gather
        g.src, g.dst, g.src_index, g.dst_index, g.row_bytes, g.n_rows = ptr(src), ptr(dst), ptr(src_index), ptr(dst_index), row_bytes, n
        self._emit(op, name, 0.0, 2.0 * n * row_bytes, (src, dst, src_index, dst_index))

    def nhwc_to_nchw(self, src, dst, c=None, c_off=0, scale=1.0, shift=0.0, clamp01=False, name="nhwc_to_nchw"):
        n, h, w, sc = src.shape
        op = L.Op()
        op.kind = L.OP_NHWC_TO_NCHW
        l = op.u.layout
        l.src, l.dst, l.n_img, l.c, l.hw, l.other_c, l.other_c_off, l.dtype = ptr(src), ptr(dst), n, c or sc, h * w, sc, c_off, dt(src)
        l.clamp01, l.scale, l.shift = int(clamp01), scale, shift
        self._emit(op, name, 0.0, src.numel() * src.element_size() + dst.numel() * 4, (src, dst))
        return dst

    def nchw_to_nhwc(self, src, dst, c_off=0, scale=1.0, shift=0.0, img_map=None, name="nchw_to_nhwc"):
        n, c, h, w = src.shape
        op = L.Op()
        op.kind = L.OP_NCHW_TO_NHWC
        l = op.u.layout
        l.src, l.dst, l.n_img, l.c, l.hw, l.other_c, l.other_c_off, l.dtype = ptr(src), ptr(dst), n, c, h * w, dst.shape[-1], c_off, dt(dst)
        l.scale, l.shift, l.img_map = scale, shift, ptr(img_map)
        self._emit(op, name, 0.0, src.numel() * 4 + src.numel() * dst.element_size(), (src, dst, img_map))
        return dst

    def ray_encode(self, extr, intr, h, w, out_nhwc, c_off, img_map=None, out_nchw=None, name="ray_encode", mode=0, n_origin_octaves=0,
                   n_dir_octaves=0, plucker=False):
        """extr fp32 [n,4,4], intr fp32 [n,3,3] device buffers (filled by the caller before each run)"""
        op = L.Op()
        op.kind = L.OP_RAY_ENCODE
        r = op.u.rays
        r.extrinsics, r.intrinsics, r.out_nchw, r.out_nhwc, r.img_map = ptr(extr), ptr(intr), ptr(out_nchw), ptr(out_nhwc), ptr(img_map)
        r.n_cam, r.h, r.w = extr.shape[0], h, w
        r.nhwc_c, r.nhwc_c_off, r.nhwc_dtype = (0, 0, L.F32) if out_nhwc is None else (out_nhwc.shape[-1], c_off, dt(out_nhwc))
        r.mode, r.n_origin_octaves, r.n_dir_octaves, r.plucker = mode, n_origin_octaves, n_dir_octaves, int(plucker)
        self._emit(op, name, 0.0, extr.shape[0] * h * w * 6 * 4.0, (extr, intr, out_nhwc, out_nchw, img_map))

    def posterior_sample(self, moments, noise, out, scale=1.0, name="posterior_sample"):
        n, c2, h, w = moments.shape
        op = L.Op()
        op.kind = L.OP_POSTERIOR_SAMPLE
        q = op.u.posterior
        q.moments, q.noise, q.out, q.n, q.c, q.hw, q.scale = ptr(moments), ptr(noise), ptr(out), n, c2 // 2, h * w, scale
        self._emit(op, name, 0.0, moments.numel() * 4 * 2.0, (moments, noise, out))
        return out

    # ---- finish ---------------------------------------------------------------------------------
    def finalize(self, autotune: Optional[bool] = None) -> "Plan":
        assert self.record
        if autotune is None:
            autotune = os.environ.get("MVLDM_AUTOTUNE", "1") != "0"
        if autotune and torch.device(self.device).type == "cuda":
            autotune_igemm(self.ops, srcs=self.__dict__.get("_tune_srcs"), skinny=self._skinny_pw, keep=self.keep)
            autotune_wgrad(self.ops)
            if type(self) is Builder and _BATCH_SPLIT:      # (training plans address their ops by index: grad_writes, collective segments)
                self.ops, self.meta = autosplit_igemm(self.ops, self.meta, skinny=self._skinny_pw, keep=self.keep)
        return Plan(self.ops, self.meta, self.keep, self.device)


# ---- plan-time tile selection ---------------------------------------------------------------------------
# `choose_config` in igemm.hip picks a tile from the problem size with rules distilled from sweeps; which of
# the 8-wave tiles wins near a boundary depends on whole rounds of 256 workgroups, N padding and the XCD
# partition.  When a plan is recorded every large implicit-GEMM launch is therefore timed once per candidate
# tile on its own buffers (they hold scratch at that point) and the fastest is frozen into the descriptor.
# Results are cached per problem signature for the life of the process.  MVLDM_AUTOTUNE=0 keeps the rules;
# MVLDM_TUNE_TILES=0,2,9,... restricts the candidates (the small 64x64 / 32x64 tiles only ever win below ~4 scenes: +2 % at b = 1).
_TUNE_CACHE = {}
_WGRAD_CACHE = {}
_SPLIT_CACHE = {}         # igemm problem signature -> images of part A of a batch split (0 = keep the launch whole): autosplit_igemm
_VALIDATED = set()         # igemm cache entries this process has launched once (entries loaded from a file are trial-launched before use)
_TUNE_TILES = tuple(int(t) for t in os.environ["MVLDM_TUNE_TILES"].split(",")) if os.environ.get("MVLDM_TUNE_TILES") else (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14)


# ---- the tune cache as data: a file (MVLDM_TUNE_CACHE=<path>.json: read at import, rewritten whenever a plan timed new problems) and a
# broadcast.  Every process times its candidates itself, so two ranks of a job -- or a profiler pass and the run it is meant to
# explain -- can freeze different tiles (different last-bit rounding, and rocprof tables full of trial launches).  With the file a
# second process records its plans WITHOUT a single trial launch; `broadcast_tune_cache` gives the ranks of a job rank 0's choices.
def tune_cache_state() -> dict:
    return {"version": 2, "igemm": {json.dumps(list(k)): (int(v) if isinstance(v, int) else list(v)) for k, v in _TUNE_CACHE.items()},
            "wgrad": {json.dumps(list(k)): int(v) for k, v in _WGRAD_CACHE.items()},
            "split": {json.dumps(list(k)): int(v) for k, v in _SPLIT_CACHE.items()}}


def set_tune_cache_state(state: dict, replace: bool = False) -> None:
    if replace:
        _TUNE_CACHE.clear()
        _WGRAD_CACHE.clear()
        _SPLIT_CACHE.clear()
    for name, cache in (("igemm", _TUNE_CACHE), ("wgrad", _WGRAD_CACHE), ("split", _SPLIT_CACHE)):
        for k, v in (state.get(name) or {}).items():
            cache[tuple(json.loads(k))] = int(v) if isinstance(v, int) else list(v)


def save_tune_cache(path: Optional[str] = None) -> Optional[str]:
    path = path or os.environ.get("MVLDM_TUNE_CACHE")
    if not path:
        return None
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        json.dump(tune_cache_state(), f, indent=0, sort_keys=True)
    os.replace(tmp, path)           # atomic: ranks sharing one file never see half of it
    return path


def load_tune_cache(path: Optional[str] = None) -> int:
    path = path or os.environ.get("MVLDM_TUNE_CACHE")
    if not path or not os.path.exists(path):
        return 0
    with open(path) as f:
        set_tune_cache_state(json.load(f))
    return len(_TUNE_CACHE) + len(_WGRAD_CACHE) + len(_SPLIT_CACHE)


def broadcast_tune_cache(dist, src: int = 0, device=None) -> None:
    """every rank takes rank `src`'s tile choices (call it after `src` has recorded -- and thereby tuned -- its plans and before
    the other ranks record theirs): the ranks of a job then run the same kernels on the same problems"""
    box = [tune_cache_state() if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src, device=device)
    set_tune_cache_state(box[0], replace=True)


load_tune_cache()


def _igemm_signature(d) -> tuple:
    return (d.n_img, d.h_in, d.w_in, d.h_out, d.w_out, d.c0, d.c1, d.ksize, d.stride, d.pad, d.upsample, d.n_out, d.n_pad,
            d.k_pad, d.epilogue, d.act_dtype, d.dst_dtype, d.k_order, d.dst_ld, bool(d.residual), bool(d.row_bias),
            bool(d.bias), d.splitk, d.workspace_bytes)


_SMALL_ROWS = int(os.environ.get("MVLDM_TUNE_SMALL_ROWS", "64"))       # launches below this many output rows keep the rules
_SPLITS = (0, 1, 2, 4, 8, 16, 32)
# launches of a few thousand to a few ten thousand rows (a training window's 16x16 / 8x8 levels, 4 - 16 scenes of sampling, the 8x8 / 4x4 levels
# at 64 scenes): every tile x a few split-K counts -- the rule's split (fill ~512 workgroups) is one guess there too: training 63.2 -> 61.8 ms per
# optimizer step, b = 4 11.24 -> 10.69 ms per DDIM step (same-box A/Bs); 0 = tile only, at the rule's split, as before
_MID_ROWS = int(os.environ.get("MVLDM_TUNE_MID_ROWS", "40000"))
_MID_SPLITS = (0, 1, 2, 3, 4, 6, 8)
# tiles tried on small launches: the 1 / 2 / 4-wave tiles.  The tall 2-slot tiles 6 / 7 (256x64, 256x128) and the deep-ring tile 18 (192x128, 4 slots:
# up to 192 rows read every weight byte once) were tried and never win a shape they compute correctly (tools/skinny_probe.py; tile 18's one
# apparent win, the 8x8 GEGLU projection, was an epilogue it cannot do -- refused now); MVLDM_TUNE_SMALL_TILES=0,1,2,3,4,5,18 puts it back.
_SMALL_TILES = tuple(int(t) for t in os.environ.get("MVLDM_TUNE_SMALL_TILES", "0,1,2,3,4,5").split(","))


# ---- igemm tile 15 (csrc/skinny.hip): the skinny-M weight-streaming kernel.  It reads the FRAGMENT-ORDER copy of the packed weight
# (`PackedWeight.skinny()`, made on first use), so a choice of tile 15 swaps the descriptor's weight pointer and sets k_order = 2.
# Candidates: 16-bit block-major convs / Linears of at most MVLDM_SKINNY_ROWS output rows whose channel counts are multiples of 64.
# With plan-time tuning on it is one more candidate, timed like the tiles in every configuration the library accepts; with tuning off
# (MVLDM_AUTOTUNE=0) it is the RULE for the shape classes it won in tools/skinny_bench.py (profiles/r05_skinny_bench.json): launches of at
# most 160 rows, and of at most 1152 rows when the packed width is at most 1280 (the wide QKV / GEGLU / FF projections of the 8x8 level
# re-read their activation block once per 16-column tile and stay with the tiled kernels).  MVLDM_SKINNY=0 keeps it out of the plans (A/B).
_SKINNY = os.environ.get("MVLDM_SKINNY", "1") != "0"
_SKINNY_ROWS = int(os.environ.get("MVLDM_SKINNY_ROWS", "2304"))
_SKINNY_CFGS = tuple(int(c) for c in os.environ.get("MVLDM_SKINNY_CFGS", "0,1,2,5,6,7,8,9,10,12,13,18,19,20,38,39,40,41,42,43,44,46").split(","))


def skinny_rule(rows: int, n_pad: int) -> bool:
    """rule-based choice of tile 15 (read per call: tests set the variables with monkeypatch)"""
    if os.environ.get("MVLDM_AUTOTUNE", "1") != "0" or os.environ.get("MVLDM_SKINNY_RULE", "1") == "0":
        return False
    return rows <= 160 or (rows <= 1152 and n_pad <= 1280)


def skinny_candidate(pw: PackedWeight, rows: int, c0: int, c1: int, upsample, dtype) -> bool:
    return (_SKINNY and dtype != torch.float32 and rows <= _SKINNY_ROWS and not (upsample is True or upsample == 1) and pw.can_skinny()
            and c0 % 64 == 0 and c1 % 64 == 0)


def _unpack_choice(v):
    """a cache entry is the tile (int: files of earlier rounds) or [tile, splitk]; splitk None = leave the descriptor's"""
    return (int(v), None) if isinstance(v, int) else (int(v[0]), None if v[1] is None else int(v[1]))


# Launches of at most MVLDM_TUNE_COLD output rows (default 9216; 0: never) are timed with COLD caches: a 640 MB fill runs in front of every
# trial launch.  Back-to-back trials of a small op find its weights and inputs in the L2 of the XCD that read them a moment ago; inside a DDIM
# step 1.85 GB of other weights pass between two uses of a layer and its input was written by the CUs of other XCDs (it comes from the
# Infinity Cache at best).  Hot trials rank the tiles of the one-scene step wrongly: same box, one scene 5.14 -> 4.93 ms per DDIM step,
# four scenes 9.84 -> 9.67, sixteen 27.85 -> 27.72; at 64 scenes cold trials for EVERY op measure 0.3 % slower than hot ones (large ops do
# find their operands where a back-to-back trial finds them), hence the row bound.  (Cold trials of the weight-gradient forms and of the
# training plans' igemm ops up to 40000 rows: 530 - 537 training views/s either way, not kept.)
_TUNE_COLD = int(os.environ.get("MVLDM_TUNE_COLD", "9216"))
_THRASH = []
_THRASH_FILLS = [0]        # cold-cache fills run so far (tests: the fill really ran; the buffer itself is freed after tuning)


def _thrash(device=None):
    if not _THRASH or (device is not None and _THRASH[0].device != torch.device(device)):
        _THRASH[:] = [torch.empty(640 << 20, dtype=torch.uint8, device=torch.cuda.current_device() if device is None else device)]   # > 2 x the 256 MB Infinity Cache
    return _THRASH[0]


def autotune_igemm(ops, min_rows: int = 2048, iters: int = 4, srcs=None, skinny=None, keep=None) -> int:
    """set `desc.tile` of every auto-tiled 16-bit block-major igemm op with >= `min_rows` output rows to the
    fastest candidate; returns the number of distinct problems timed.  `srcs`: {op index: (x, x2)} source tensors of
    the recorded convs -- they hold scratch at this point and are filled with N(0,1) first: on zeros / NaNs the
    chip draws less power and clocks higher, which ranks the candidates differently from real data.
    Round 4: launches with fewer rows (down to MVLDM_TUNE_SMALL_ROWS = 64: levels 2 - 4 at a few scenes, 45 % of the one-scene step,
    which the rules alone used to serve) are tuned too, over the small tiles AND the split-K count -- at a few hundred rows the
    number of K slices decides how much of the chip a launch fills, and the rule (fill ~512 workgroups) is one guess."""
    lib = L.load()
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    timed = 0
    tune_dev = next((t.device for pair in (srcs or {}).values() for t in pair if t is not None), None)      # the plan's device (cold-cache fill)
    if srcs:
        seen, gen = set(), None
        for pair in srcs.values():
            for t in pair:
                if t is not None and t.is_floating_point() and t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    if gen is None:     # a private generator: the caller's CUDA RNG stream must not move
                        gen = torch.Generator(device=t.device)
                        gen.manual_seed(0x5EED)
                    t.normal_(generator=gen)
    for idx, op in enumerate(ops):
        if op.kind != L.OP_IGEMM:
            continue
        d = op.u.igemm
        rows = d.n_img * d.h_out * d.w_out
        sk_pw = (skinny or {}).get(idx)      # tile 15 can compute this op: even launches below MVLDM_TUNE_SMALL_ROWS are worth timing then
        if d.tile != 0 or d.act_dtype == L.F32 or d.k_order != 1 or (rows < min(min_rows, _SMALL_ROWS) and sk_pw is None):
            continue
        small = rows < min_rows
        key = _igemm_signature(d) + (("sk",) if sk_pw is not None else ())
        best = _TUNE_CACHE.get(key)
        if best is not None and (_unpack_choice(best)[0] & 63) not in (set(_TUNE_TILES) | set(_SMALL_TILES) | {17, 19} | ({15} if sk_pw is not None else set())):
            best = None          # an entry of an older build / another candidate set (a tile this build no longer offers): time it again
        if best is None:
            trial = L.Op()
            C.memmove(C.byref(trial), C.byref(op), C.sizeof(L.Op))
            # small launches: the 4-wave / 2-wave / 1-wave tiles x split counts (only where the op carries a split-K workspace and lets
            # the library choose); large launches: every tile at the descriptor's own split
            can_split = small and bool(d.workspace) and d.splitk == 0
            mid = (not small) and rows < _MID_ROWS and bool(d.workspace) and d.splitk == 0
            cands = [(t, sk) for t in _TUNE_TILES for sk in _MID_SPLITS] if mid else [(t, None) for t in _TUNE_TILES] if not small else \
                    [(t, sk) for t in _SMALL_TILES if t in _TUNE_TILES or not os.environ.get("MVLDM_TUNE_TILES") for sk in (_SPLITS if can_split else (None,))]
            n_it = iters if not small else 3 * iters
            cold = _TUNE_COLD > 0 and rows <= _TUNE_COLD
            if not small and 17 not in _TUNE_TILES and not os.environ.get("MVLDM_TUNE_TILES") and d.ksize == 3 and d.stride == 1 \
                    and d.src1 is None and not d.upsample and d.w_in <= 24:
                cands = cands + [(17, None)]     # the wide pixel-halo tile: one-source 3x3 convs on maps up to 24 wide (elsewhere it is tile 7)
            if not small and d.ksize == 1 and not os.environ.get("MVLDM_TUNE_TILES"):
                # round 6: the register-staged Linear (tile 19) and tile 13 with the L2 prefetch of its activation rows (bit 13)
                cands = cands + [(19, None), (13 | (1 << 13), None)]
            if sk_pw is not None:            # tile 15 in every configuration the library accepts for this problem (others return an error)
                cands = cands + [(15 | (c << 8), 1) for c in _SKINNY_CFGS]
                sk_ptr = sk_pw.skinny().data_ptr()
            results = []
            for tile, sk in cands:
                trial.u.igemm.tile = tile
                trial.u.igemm.splitk = d.splitk if sk is None else sk
                trial.u.igemm.weight, trial.u.igemm.k_order = (sk_ptr, 2) if (tile & 63) == 15 else (d.weight, d.k_order)
                if lib.mvldm_op_run(C.byref(trial), stream) != 0:      # candidate not applicable to this problem
                    continue
                if cold:
                    # weight-bound launches (a few scenes): inside a DDIM step 1.85 GB of other weights pass between two uses of a layer, so
                    # its weights come from HBM; back-to-back trials would read them from the Infinity Cache and rank the tiles differently
                    t_sum = 0.0
                    for _ in range(n_it):
                        _thrash(tune_dev).zero_()
                        _THRASH_FILLS[0] += 1
                        e0.record()
                        lib.mvldm_op_run(C.byref(trial), stream)
                        e1.record()
                        e1.synchronize()
                        t_sum += e0.elapsed_time(e1)
                    results.append((t_sum, tile, sk))
                    continue
                e0.record()
                for _ in range(n_it):
                    lib.mvldm_op_run(C.byref(trial), stream)
                e1.record()
                e1.synchronize()
                results.append((e0.elapsed_time(e1), tile, sk))
            # keep the rules' choice unless a candidate is clearly (>3 %) faster: timing noise must not flip tiles
            t_rule = next(t for t, tile, sk in results if tile == 0 and sk in (None, 0))
            t_best, tile_best, sk_best = min(results, key=lambda r: r[0])
            best = [tile_best, sk_best] if t_best < 0.97 * t_rule else [0, None]
            _TUNE_CACHE[key] = best
            _VALIDATED.add(key)
            timed += 1
        tile, sk = _unpack_choice(best)
        if tile != 0 and key not in _VALIDATED:
            # an entry from a file / another rank: this build's tile must accept the problem (one trial launch), else the rule stands --
            # a plan that fails at run time is worse than an untuned op
            trial = L.Op()
            C.memmove(C.byref(trial), C.byref(op), C.sizeof(L.Op))
            trial.u.igemm.tile = tile
            if sk is not None:
                trial.u.igemm.splitk = sk
            if (tile & 63) == 15:
                if sk_pw is None:
                    tile, sk = 0, None
                else:
                    trial.u.igemm.weight, trial.u.igemm.k_order = sk_pw.skinny().data_ptr(), 2
            if tile != 0 and lib.mvldm_op_run(C.byref(trial), stream) != 0:
                tile, sk = 0, None
                _TUNE_CACHE[key] = [0, None]
            _VALIDATED.add(key)
        d.tile = tile
        if sk is not None:
            d.splitk = sk
        if (tile & 63) == 15:                # the fragment-order copy of the weight stays with the plan
            t = sk_pw.skinny()
            sk_pw._skinny_pinned = True      # the op holds a raw pointer: the copy stays with the pack whoever else shares it
            d.weight, d.k_order = t.data_ptr(), 2
            if keep is not None:
                keep.append(t)
        elif sk_pw is not None:
            sk_pw.drop_skinny()              # not chosen: the copy made for the trials is not kept (a no-op while another op points at it)
    if timed:
        save_tune_cache()
        _THRASH.clear()                      # the 640 MB cold-cache fill is a tuning-time buffer: not kept past the plan that needed it
    return timed


# ---- batch split of the big implicit-GEMM launches (round 6) ---------------------------------------------------------------
# A launch of T tiles on 256 CUs takes ceil(T / 256) rounds: the 16x16-level convs of the headline (640 tiles of 256 x 320) pay 3 rounds for 2.5,
# the 8x8-level ones 2 for 1.25 -- and a smaller tile for the WHOLE launch trades that for arithmetic intensity (256 x 128: 85 flop per L2->LDS
# byte instead of 142).  The images of a batch are independent rows of the GEMM: the launch is cut at an image boundary into a part whose big
# tiles fill whole rounds and a remainder that is tuned on its own (smaller tiles and / or a K split: `autotune_igemm` times every tile x split
# for launches below MVLDM_TUNE_MID_ROWS) -- the cheap half of what a stream-K GEMM does with its ragged last round.  Decided by timing, per problem
# signature, like the tiles: both parts are tuned, then [whole] is timed against [part A, part B]; the split is kept if it is > 3 % faster.
# Sampling plans only; MVLDM_BATCH_SPLIT=0 keeps every launch whole (A/B).  The values of a split launch differ from the whole launch's only
# through the remainder's tile / K-split (last-bit rounding, like any other tuner choice).
_BATCH_SPLIT = os.environ.get("MVLDM_BATCH_SPLIT", "1") != "0"
_SPLIT_MIN_ROWS = int(os.environ.get("MVLDM_BATCH_SPLIT_MIN_ROWS", "16384"))


def _split_candidates(d) -> list:
    """image counts n1 (0 < n1 < n_img) at which part A = whole rounds of 256 x BN tiles for a BN the big tiles offer"""
    rpi = d.h_out * d.w_out
    rows = d.n_img * rpi
    out = []
    for bn in (320, 256):
        n_tiles = -(-d.n_pad // bn)
        m_tiles = -(-rows // 256)
        if m_tiles * n_tiles <= 256 or (m_tiles * n_tiles) % 256 == 0 or (m_tiles * n_tiles) % 256 > 216:
            continue                                  # one round, or a last round that is (nearly) full already
        for m1 in range(m_tiles - 1, 0, -1):          # the largest whole-round prefix that ends on an image boundary
            if (m1 * n_tiles) % 256 == 0 and (m1 * 256) % rpi == 0:
                n1 = m1 * 256 // rpi
                frac = 1.0 - n1 / d.n_img
                if 0.03 <= frac <= 0.45 and n1 not in out:
                    out.append(n1)
                break
    return out


def _image_range(op, i0: int, n: int):
    """a copy of igemm op `op` restricted to images [i0, i0 + n) (NHWC tensors: a pointer offset per operand)"""
    q = L.Op()
    C.memmove(C.byref(q), C.byref(op), C.sizeof(L.Op))
    d = q.u.igemm
    es = 4 if d.act_dtype == L.F32 else 2
    ds = 4 if d.dst_dtype == L.F32 else 2
    n_dst = d.n_out // 2 if d.epilogue == L.EPI_GEGLU else d.n_out
    px_in, px_out = d.h_in * d.w_in, d.h_out * d.w_out

    def off(p, nbytes):
        return (p + nbytes) if p else p
    d.src0 = off(d.src0, i0 * px_in * d.c0 * es)
    d.src1 = off(d.src1, i0 * px_in * d.c1 * es)
    d.dst = off(d.dst, i0 * px_out * (d.dst_ld or n_dst) * ds)
    d.residual = off(d.residual, i0 * px_out * n_dst * es)
    d.row_bias = off(d.row_bias, i0 * d.row_bias_ld * 4)
    d.n_img = n
    d.tile, d.splitk = 0, 0
    return q


def _time_ops(ops, iters: int) -> float:
    lib = L.load()
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for o in ops:
        L.check(lib.mvldm_op_run(C.byref(o), stream))
    e0.record()
    for _ in range(iters):
        for o in ops:
            lib.mvldm_op_run(C.byref(o), stream)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def autosplit_igemm(ops, meta, skinny=None, keep=None, iters: int = 6):
    """cut big 16-bit igemm launches at an image boundary where two launches are faster than one (see above); returns the new (ops, meta)"""
    new_ops, new_meta, decided = [], [], 0
    for idx, (op, m) in enumerate(zip(ops, meta)):
        d = op.u.igemm if op.kind == L.OP_IGEMM else None
        rows = d.n_img * d.h_out * d.w_out if d is not None else 0
        ok = (d is not None and rows >= _SPLIT_MIN_ROWS and d.act_dtype != L.F32 and d.k_order == 1 and d.stride == 1 and not d.upsample
              and (d.tile & 63) != 15 and not _PIN_TILE() and bool(d.workspace) and idx not in (skinny or {}))
        n1 = 0
        if ok:
            key = _igemm_signature(d)[:-2] + (d.workspace_bytes,)          # (the tuned tile / split of the whole launch are not part of the problem)
            if key in _SPLIT_CACHE:
                n1 = _SPLIT_CACHE[key]
            else:
                best_t, t_whole = None, None
                for cand in _split_candidates(d):
                    parts = [_image_range(op, 0, cand), _image_range(op, cand, d.n_img - cand)]
                    try:
                        autotune_igemm(parts, iters=4)
                        if t_whole is None:
                            t_whole = _time_ops([op], iters)
                        t = _time_ops(parts, iters)
                    except L.MvldmError:          # a part the library refuses where it takes the whole launch: the launch stays whole
                        continue
                    if os.environ.get("MVLDM_BATCH_SPLIT_LOG"):
                        print(f"[batch split] {m.name}: {d.n_img} images x {d.h_out}x{d.w_out}, N {d.n_out}, K {d.k_pad}: whole {t_whole * 1e3:.1f} us (tile {d.tile & 63}, "
                              f"splitk {d.splitk}); {cand} + {d.n_img - cand}: {t * 1e3:.1f} us (tiles {parts[0].u.igemm.tile & 63} / {parts[1].u.igemm.tile & 63}, "
                              f"splitk {parts[1].u.igemm.splitk})", file=sys.stderr, flush=True)
                    if t < 0.97 * t_whole and (best_t is None or t < best_t):
                        best_t, n1 = t, cand
                _SPLIT_CACHE[key] = n1
                decided += 1
        if n1:
            parts = [_image_range(op, 0, n1), _image_range(op, n1, d.n_img - n1)]
            autotune_igemm(parts, iters=4)          # (from the cache: nothing is timed twice)
            for k, (q, i0, cnt) in enumerate(zip(parts, (0, n1), (n1, d.n_img - n1))):
                q.tag = len(new_ops)
                new_ops.append(q)
                share = cnt / d.n_img
                new_meta.append(OpMeta(m.name if k == 0 else m.name + "[rest]", m.kind, m.flops * share, m.bytes * share))
        else:
            op.tag = len(new_ops)
            new_ops.append(op)
            new_meta.append(m)
    if decided:
        save_tune_cache()
    return new_ops, new_meta


_WGRAD_TARGETS = tuple(int(t) for t in os.environ.get("MVLDM_TUNE_WGRAD_TARGETS", "0,1,2,3,4,5").split(","))      # 0 = library default, 64 << code


def autotune_wgrad(ops, iters: int = 3) -> int:
    """plan-time choice between the two weight-gradient kernels (csrc/wgrad.hip: register-staged small tile / wide LDS-DMA tile)
    per problem signature: both are timed on the op's own buffers (scratch at this point; the gradient it writes is zeroed before
    the first real run) and the winner goes into bits 8-9 of `desc.accumulate`.  Returns the number of problems timed.
    The two forms split the pixel range differently, i.e. sum in a different order: like the igemm tile choice this makes the last
    bits of a gradient depend on what a process measured -- MVLDM_TRAIN_AUTOTUNE=0 (rules only), a shared MVLDM_TUNE_CACHE file or
    `broadcast_tune_cache` give run-to-run / rank-to-rank identical gradients.  (A form the library refuses leaves its message in
    `mvldm_last_error()`; it is only ever read after a failing call, which overwrites it.)"""
    lib = L.load()
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    timed = 0
    scratch = None          # the trials write their gradient HERE: the real accumulator may already hold a micro-batch
    for op in ops:
        if op.kind != L.OP_WGRAD:
            continue
        d = op.u.wgrad
        if d.act_dtype == L.F32 or (d.accumulate >> 8) & 3:
            continue
        key = (d.c0, d.c1, d.n_img, d.h_in, d.w_in, d.h_out, d.w_out, d.ksize, d.stride, d.pad, d.upsample, d.n_out, d.dy_ld, d.act_dtype,
               int(d.workspace_bytes), d.accumulate & 1)      # (store-first one-split Linears take the direct path: another problem than the accumulating op)          # (the workspace bounds the split count, hence which form wins)
        best = _WGRAD_CACHE.get(key)
        if best is None:
            trial = L.Op()
            C.memmove(C.byref(trial), C.byref(op), C.sizeof(L.Op))
            need = d.n_out * d.c_in * d.ksize * d.ksize
            if scratch is None or scratch.numel() < need:
                scratch = torch.zeros(need, dtype=torch.float32, device=torch.cuda.current_device())
            else:
                scratch[:need].zero_()       # the trials accumulate: no Inf / NaN left over from an earlier, larger problem
            trial.u.wgrad.grad = scratch.data_ptr()
            results = []
            # form (bits 8-9) x workgroup target of the pixel split (bits 10-12: 0 = the library's 512 / 256, else 64 << code): fewer
            # splits halve a small layer's slab traffic, more fill the chip on a long pixel range -- one value per problem
            for form in (1, 2):
                for tcode in _WGRAD_TARGETS:
                    v = form + 4 * tcode
                    trial.u.wgrad.accumulate = (d.accumulate & 1) | (v << 8)
                    if lib.mvldm_op_run(C.byref(trial), stream) != 0:      # the wide form does not take this problem
                        break
                    e0.record()
                    for _ in range(iters):
                        lib.mvldm_op_run(C.byref(trial), stream)
                    e1.record()
                    e1.synchronize()
                    results.append((e0.elapsed_time(e1), v))
            best = min(results)[1] if results else 0
            _WGRAD_CACHE[key] = best
            timed += 1
        d.accumulate = (d.accumulate & 1) | (best << 8)
    if timed:
        save_tune_cache()
    return timed


class Plan:
    """A recorded op list living in the C executor.  Holds references to every tensor it touches."""

    def __init__(self, ops, meta, keep, device):
        self.meta, self._keep, self.device = list(meta), list(keep), device
        self.ops = list(ops)             # the recorded descriptors (host copies: the executor has its own), for inspection
        self.tiles = [int(op.u.igemm.tile) if op.kind == L.OP_IGEMM else None for op in ops]     # 0 = chosen by the library's rules
        arr = (L.Op * max(len(ops), 1))(*ops)
        h = C.c_void_p()
        L.check(L.load().mvldm_plan_create(arr, len(ops), C.byref(h)))
        self._h = h
        self._stream = None
        self.captured = False

    def __len__(self):
        return len(self.meta)

    @staticmethod
    def _cur():
        return torch.cuda.current_stream().cuda_stream

    def run(self, first: int = 0, last: Optional[int] = None):
        last = len(self.meta) if last is None else last
        L.check(L.load().mvldm_plan_run_range(self._h, first, last, self._cur()))

    def capture(self):
        """record the plan into a hipGraph on a private stream"""
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        self._stream.wait_stream(torch.cuda.current_stream())
        L.check(L.load().mvldm_plan_capture(self._h, self._stream.cuda_stream))
        torch.cuda.current_stream().wait_stream(self._stream)
        self.captured = True

    def replay(self):
        L.check(L.load().mvldm_plan_replay(self._h, self._cur()))

    def profile(self, iters: int = 3):
        """per-op average milliseconds (hipEvents on the launch stream, eager launches)"""
        out = (C.c_float * max(len(self.meta), 1))()
        L.check(L.load().mvldm_plan_profile(self._h, self._cur(), iters, out))
        return [float(v) for v in out[:len(self.meta)]]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.load().mvldm_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass
