"""Training step of the multi-view denoiser on the HIP path: the counterpart of `DiffusionWrapper.training_step`
(src/model/diffusion_wrapper.py:324-411), `configure_optimizers` (:1111-1122: AdamW lr 2e-5 + LinearLR warm-up,
config/experiment/baseline.yaml:62-73) and of what Lightning's Trainer does around them in the reference run
(src/main.py:119-136, config/main.yaml:82-84: `accumulate_grad_batches=2`, `gradient_clip_val=0.1`,
`strategy=ddp_find_unused_parameters_true`).

There is no autograd here: `TrainBuilder` records the forward walk of mvunet.py:90-208 op by op together with a closure
that emits each op's backward kernels, `finalize()` replays the tape in reverse, and the whole micro-batch -- input
assembly (add_noise, masks, ray grid), forward, MSE loss, backward -- becomes ONE C-side plan (`plan.Plan`).
Gradient kernels (include/mvldm.h, "Training"):
    data gradients   the forward implicit GEMM on transposed / flipped packs of the weights (`mvldm_pack_weight(transpose)`)
    weight gradients `mvldm_igemm_wgrad` (pixel-reduction GEMM, deterministic split-K), fp32, straight into the flat buffer
    norms            `mvldm_groupnorm_bwd`, `mvldm_layernorm_bwd`;  attention `mvldm_attention_bwd` (recompute from lse)
    elementwise      SiLU / GEGLU backward, column sums for biases and the time-embedding rows
Master weights, gradients and AdamW moments are fp32 in flat buffers (`FlatParams`); activations and activation
gradients are in the compute dtype (bf16: the bench dtype; f32: the parity mode; the reference trains under fp16
autocast WITH a loss scaler -- f16 activations gradients would need one here as well, so f16 is refused for training).
Parameters that never enter the graph (the SD up-block transformers when `pretrained_from` is set, mvunet.py:178) are
excluded statically instead of discovered per step (`find_unused_parameters`); parameters that enter it with an exactly
zero gradient (cross-attention to the all-zero context, mvunet.py:124-128) stay in the optimizer -- AdamW's decoupled
weight decay still moves them, as in the reference.
Multi-GPU: `DistributedOptimizer` shards the flat buffers ZeRO-1 style -- bucketed reduce-scatter of the gradients in
reverse-layer order on a side stream while the backward plan is still running, AdamW on the owned slices, all-gather of
the updated weights (RCCL over xGMI via torch.distributed; gloo in the CPU tests).
"""
from __future__ import annotations

import contextlib

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib as L
from . import ops
from .ops import PackedWeight, dt, ptr
from .plan import Builder, Plan


# ================================================================================================ flat fp32 state
class FlatParams:
    """fp32 master parameters and gradients of the TRAINED part of a module in two flat device buffers; every
    `nn.Parameter` becomes a view (`p.data`, `p.grad`), so state dicts / checkpoints keep working.  Order = the stacked `groups`
    first, then module registration order (what `parameters()` yields); offsets aligned to 16 bytes."""

    def __init__(self, module: nn.Module, exclude: Sequence[nn.Parameter] = (), groups: Sequence[Sequence[nn.Parameter]] = ()):
        """`groups`: parameter lists that one kernel treats as ONE stacked matrix (the 22 `time_emb_proj` weights / biases of the UNet's
        resnets): laid out contiguously, in the given order, in front of everything else -- their stacked forward / gradient views are then
        views of the flat buffers (no concatenated copy to refresh after every optimizer step, one weight-gradient launch for the stack)."""
        skip = {id(p) for p in exclude}
        self.module = module
        front, seen = [], set()
        for grp in groups:
            for p in grp:
                if id(p) not in skip and p.requires_grad and id(p) not in seen:
                    front.append(p)
                    seen.add(id(p))
        self.params = front + [p for p in module.parameters() if id(p) not in skip and p.requires_grad and id(p) not in seen]
        self.excluded = [p for p in module.parameters() if id(p) in skip]
        assert self.params and all(p.dtype == torch.float32 for p in self.params), "FlatParams needs fp32 master parameters"
        self.offset: Dict[int, int] = {}
        off = 0
        for p in self.params:
            self.offset[id(p)] = off
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        dev = self.params[0].device
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p in self.params:
                o, n = self.offset[id(p)], p.numel()
                self.flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)
        self.epoch = 0
        self.dirty: set = set()          # (offset, numel) ranges of `grad` written since the last zero_grad()

    def zero_grad(self):
        self.grad.zero_()
        self.dirty.clear()

    def begin_window(self, stored: set):
        """instead of `zero_grad()` in front of a plan whose FIRST write to every range in `stored` is a store (`TrainPlan.stored`):
        only what earlier plans wrote and this one will not overwrite is zeroed -- nothing at all in the steady state of one window
        shape (saves the 3.7 GB fill and, in the plan, the read half of 926 M read-modify-writes).  Ranges are compared as whole
        (offset, numel) tuples: a range written under a different fusion (one QKV gradient / three) is zeroed AND stored, which is
        merely redundant."""
        for off, n in sorted(self.dirty - stored):
            self.grad[off:off + n].zero_()
        self.dirty &= stored

    def wait_readers(self):
        """call before WRITING `flat` in place on the current stream: a weight re-pack launched ahead on the trainer's side stream
        (`MVLDMTrainer._repack_ahead`) may still be READING it.  The optimizer step, `EMAWeights.applied` (entry and exit) and
        `MVLDMTrainer.load_denoiser_state_dict` go through here; without it the ~5 ms packing launch could pick up a mix of live and averaged / updated
        weights, and because the plan is already marked current the bad packs would never be redone."""
        ev = self.__dict__.get("_read_event")
        if ev is not None and torch.cuda.is_available():
            torch.cuda.current_stream().wait_event(ev)

    def bump(self):
        """in-place kernel updates do not move torch's version counters: recorded INFERENCE plans / packs of THIS module key
        on its own epoch instead (a frozen VAE next to it keeps its plans: modules.bump_weights_epoch(module))"""
        from .modules import bump_weights_epoch
        self.epoch = bump_weights_epoch(self.module)

    def contiguous(self, ps: Sequence[nn.Parameter]) -> bool:
        o = [self.offset[id(p)] for p in ps]
        return all(o[i] + ps[i].numel() == o[i + 1] for i in range(len(ps) - 1))


# ================================================================================================ tape builder
def _key(t: torch.Tensor):
    # (storage position, size): a [n, h, w, c] activation and its [n*h*w, c] token view are the same tensor
    return (t.data_ptr(), t.numel())


class TrainBuilder(Builder):
    """records forward ops like `Builder` plus, per op, a closure that emits its backward ops"""

    def __init__(self, device, dtype, flat: FlatParams, ws_bytes: int = 512 << 20, store_first: bool = False):
        assert dtype in (torch.float32, torch.bfloat16), "training runs in bf16 (fp32 accumulation / master weights) or f32"
        super().__init__(device, dtype, record=True)
        self.skinny = False             # the training packs are refreshed in place after every optimizer step: no fragment-order copies
        self.flat = flat
        # plans that run ONCE per accumulation window (training_window): the first write of every gradient range is a store (grad = ...),
        # later ones accumulate; the caller then needs no zeroed gradient buffer (FlatParams.begin_window)
        self.store_first = bool(store_first)
        self.grad_stored: List[bool] = []                      # per entry of grad_writes: was it emitted as a store
        self._fw = False
        # the weight gradient, the data gradient and the bias sums of ONE layer only share their input dY: emitted as three parallel
        # lanes (MVLDM_OP_PAR_*: side streams) they overlap each other's tails and launch gaps (MVLDM_TRAIN_PAR=0: one after the other)
        self.par_bwd = os.environ.get("MVLDM_TRAIN_PAR", "0") != "0"
        self.par_rows = int(os.environ.get("MVLDM_TRAIN_PAR_ROWS", "0"))      # > 0: lanes only for layers with at most this many rows
        self._wws_lane: Dict[int, torch.Tensor] = {}
        self.tape: List[Callable[[], None]] = []
        self.grads: Dict[tuple, torch.Tensor] = {}
        self.repack: List[Callable[[], None]] = []
        self.touched: set = set()
        self.grad_writes: List[Tuple[int, int, int]] = []      # (op index, flat offset, numel) of every parameter-gradient write
        self.pack_jobs: List["L.PackJob"] = []                 # one per packed weight (forward and data-gradient packs)
        self._wws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)      # wgrad slabs / colsum / norm partials
        self.keep.append(self._wws)                            # (the plan outlives this builder: it must own the workspace)

    def small_launch(self, rows: int, n_out: int) -> bool:
        """training plans are run in segments between collectives and replay their tape in reverse: no parallel lanes"""
        return False

    def free(self, t):      # every activation is kept for the backward pass
        return

    # ---- gradient bookkeeping ----------------------------------------------------------------------
    def grad_of(self, t):
        return self.grads.get(_key(t))

    def pop_grad(self, t):
        g = self.grads.pop(_key(t), None)
        return None if g is None else g.view(t.shape[:-1] + (g.numel() // max(math.prod(t.shape[:-1]), 1),))

    def add_grad(self, t: Optional[torch.Tensor], g: torch.Tensor):
        if t is None:
            return
        cur = self.grads.get(_key(t))
        if cur is None:
            self.grads[_key(t)] = g
        else:
            self._elt(L.TE_ADD, g, None, cur, 1, g.numel(), name="grad+=")

    def _range(self, p: nn.Parameter, through: Optional[nn.Parameter] = None) -> Tuple[int, int]:
        off = self.flat.offset[id(p)]
        last = p if through is None else through
        end = self.flat.offset[id(last)] + last.numel()
        assert end > off
        return off, end - off

    def _unwritten(self, p: nn.Parameter, through: Optional[nn.Parameter] = None) -> bool:
        """may the op about to be emitted STORE into this gradient range (store_first plans: nothing emitted so far writes any of it)"""
        off, n = self._range(p, through)
        return self.store_first and not any(o < off + n and off < o + m for (_, o, m) in self.grad_writes)

    def _pgrad(self, p: nn.Parameter, through: Optional[nn.Parameter] = None, store: Optional[bool] = None) -> torch.Tensor:
        """gradient view of `p` for the op ABOUT to be emitted (its plan index is len(self.ops)); the write is recorded as
        the flat range [offset(p), offset(p) + numel) -- up to the end of `through` when one kernel fills several
        parameters that are contiguous in the flat buffer (the fused QKV weight gradient).  `self._fw` tells the caller whether
        the op is to store (first write of the range in a store_first plan) or accumulate; `store` overrides the decision for ops
        that fill two ranges with one flag (norm gamma / beta)."""
        self.touched.add(id(p))
        self.touched.add(id(p if through is None else through))
        off, n = self._range(p, through)
        self._fw = self._unwritten(p, through) if store is None else bool(store)
        self.grad_writes.append((len(self.ops), off, n))
        self.grad_stored.append(self._fw)
        return p.grad

    def touch(self, *ps):
        """parameters that are in the graph but provably receive a zero gradient"""
        for p in ps:
            if p is not None:
                self.touched.add(id(p))

    # ---- raw op emitters (no tape) -------------------------------------------------------------------
    def _elt(self, code, a, b, out, rows, d, name):
        op = L.Op()
        op.kind = L.OP_TRAIN_ELTWISE
        e = op.u.train_eltwise
        e.a, e.b, e.out, e.rows, e.op, e.d, e.a_dtype, e.dtype = ptr(a), ptr(b), ptr(out), rows, code, d, dt(a), dt(out)
        self._emit(op, name, 0.0, (a.numel() + out.numel()) * out.element_size(), (a, b, out))
        return out

    def fill_zero(self, t, name="zero"):
        op = L.Op()
        op.kind = L.OP_FILL_ZERO
        op.u.fill.dst, op.u.fill.bytes = ptr(t), t.numel() * t.element_size()
        self._emit(op, name, 0.0, t.numel() * t.element_size(), (t,))

    class _NoLanes:
        def __enter__(self):
            return self

        def lane(self):
            pass

        def __exit__(self, *a):
            return False

    def _bwd_lanes(self, rows: int = 0):
        on = self.par_bwd and (self.par_rows <= 0 or rows <= self.par_rows)
        return self.parallel() if on else TrainBuilder._NoLanes()

    def _tws(self) -> torch.Tensor:
        """scratch of the training ops (weight-gradient slabs, column-sum / norm partials): the plan's big one on the main lane, a
        64 MB one per side lane (the lanes of a parallel group run at the same time)"""
        lane = getattr(self, "_lane", 0)
        if not lane:
            return self._wws
        ws = self._wws_lane.get(lane)
        if ws is None:
            ws = torch.empty(64 << 20, dtype=torch.uint8, device=self.device)
            self._wws_lane[lane] = ws
            self.keep.append(ws)
        return ws

    def _colsum(self, x2d, n, dst, ld_dst, rows_per_seg, per_seg, accumulate, name):
        rows = x2d.shape[0]
        op = L.Op()
        op.kind = L.OP_COLSUM
        c = op.u.colsum
        tws = self._tws()
        c.x, c.dst, c.workspace, c.workspace_bytes = ptr(x2d), ptr(dst), ptr(tws), tws.numel()
        c.n_seg, c.rows_per_seg, c.n, c.ld, c.ld_dst = rows // rows_per_seg, rows_per_seg, n, x2d.stride(0), ld_dst
        c.per_seg, c.accumulate, c.dtype = int(per_seg), int(accumulate), dt(x2d)
        self._emit(op, name, 0.0, x2d.numel() * x2d.element_size(), (x2d, dst))

    def _wgrad(self, x, x2, dy2d, n_out, grad, geom, c_in, name):
        """`grad` comes from `_pgrad` in the caller's argument list: `self._fw` is that range's store / accumulate decision"""
        n, h, w, c0 = x.shape
        op = L.Op()
        op.kind = L.OP_WGRAD
        d = op.u.wgrad
        d.src0, d.src1, d.dy, d.grad = ptr(x), ptr(x2), ptr(dy2d), ptr(grad)
        assert not getattr(self, "_lane", 0), "weight gradients run on the main lane (they own the big slab workspace)"
        d.workspace, d.workspace_bytes = ptr(self._wws), self._wws.numel()
        d.c0, d.c1, d.c_in = c0, 0 if x2 is None else x2.shape[-1], c_in
        d.n_img, d.h_in, d.w_in, d.h_out, d.w_out = n, h, w, geom["ho"], geom["wo"]
        d.ksize, d.stride, d.pad, d.upsample = geom["ksize"], geom["stride"], geom["pad"], int(geom["upsample"])
        d.n_out, d.dy_ld, d.act_dtype, d.accumulate = n_out, dy2d.stride(0), dt(x), 0 if self._fw else 1
        m = n * geom["ho"] * geom["wo"]
        k = geom["ksize"] ** 2 * (c0 + d.c1)
        self._emit(op, name, 2.0 * m * n_out * k, (m * (n_out + k / geom["ksize"] ** 2)) * x.element_size() + n_out * k * 4.0, (x, x2, dy2d, grad))

    def _as_act(self, g, name="cast"):
        if g.dtype == self.dtype:
            return g
        return Builder.eltwise(self, g, L.ELT_COPY, out_dtype=self.dtype, name=name)

    # ---- weights -----------------------------------------------------------------------------------
    def _weight2d(self, weights: Sequence[nn.Parameter]):
        """fp32 [sum n_i, k] view (or persistent copy) of one or several stacked weights + its refresh closure"""
        ws = list(weights)
        for w in ws:
            self.touched.add(id(w))
        if len(ws) == 1:
            return ws[0].data, None
        if self.flat.contiguous(ws) and all(w.numel() % 4 == 0 for w in ws):
            o = self.flat.offset[id(ws[0])]
            tot = sum(w.numel() for w in ws)
            return self.flat.flat[o:o + tot].view(sum(w.shape[0] for w in ws), *ws[0].shape[1:]), None
        cat = torch.cat([w.detach() for w in ws], 0).contiguous()
        rows = np.cumsum([0] + [w.shape[0] for w in ws])

        def refresh():
            for i, w in enumerate(ws):
                cat[rows[i]:rows[i + 1]].copy_(w.detach())
        return cat, refresh

    def _packs(self, wsrc, refresh, need_t: bool, c_pad=None, c_split=None, t_splits=None):
        """forward pack and (for the data gradient) transposed pack(s), refreshed in place when the weights change"""
        pw = ops.pack_weight(wsrc, self.dtype, c_pad=c_pad, c_split=c_split)
        pts = []
        if need_t:
            for (c_off, n_rows) in (t_splits or [(0, wsrc.shape[1])]):
                pts.append((ops.pack_weight_t(wsrc, self.dtype, c_off, n_rows), c_off, n_rows))

        # after an optimizer step: the fp32 sources that are copies first (`repack`), then every pack -- one job each of the
        # plan's single batched launch (`pack_jobs`; TrainPlan.refresh_weights)
        if refresh is not None:
            self.repack.append(refresh)
        self.pack_jobs.append(ops.pack_job(wsrc, pw))
        for pt, c_off, n_rows in pts:
            self.pack_jobs.append(ops.pack_job(wsrc, pt, transpose=True, c_off=c_off))
        self.keep.extend([wsrc, pw.data] + [p[0].data for p in pts])
        return pw, [p[0] for p in pts]

    def _bias(self, segments: Optional[Sequence[Sequence[nn.Parameter]]]):
        """bias vector = concatenation over segments of the SUM of each segment's parameters (fp32)"""
        if not segments:
            return None
        for sg in segments:
            for p in sg:
                self.touched.add(id(p))
        if len(segments) == 1 and len(segments[0]) == 1:
            return segments[0][0].data
        if all(len(sg) == 1 for sg in segments) and self.flat.contiguous([sg[0] for sg in segments]) \
                and all(sg[0].numel() % 4 == 0 for sg in segments):
            o = self.flat.offset[id(segments[0][0])]        # stacked in the flat buffer (FlatParams groups): a view, nothing to refresh
            return self.flat.flat[o:o + sum(sg[0].numel() for sg in segments)]
        sizes = [sg[0].numel() for sg in segments]
        buf = torch.zeros(sum(sizes), dtype=torch.float32, device=self.device)
        offs = np.cumsum([0] + sizes)

        def refresh():
            for i, sg in enumerate(segments):
                sl = buf[offs[i]:offs[i + 1]]
                sl.copy_(sg[0].detach())
                for extra in sg[1:]:       # fp32 add by the HIP elementwise kernel
                    ops.train_eltwise(L.TE_ADD, extra.detach(), None, sl, 1, sl.numel())
        refresh()
        self.repack.append(refresh)
        self.keep.append(buf)
        return buf

    # ---- differentiable ops ------------------------------------------------------------------------------
    def t_linear(self, x, weights, biases=None, *, residual=None, out_dtype=None, x_grad=True, name="linear"):
        """x [rows, c] -> [rows, sum n_i]; `weights`: parameters stacked along the output dim (fused QKV, all time_emb_proj);
        `biases`: per weight a list of parameters whose SUM is that weight's bias (None: no bias)"""
        rows, c = x.shape
        wsrc, refresh = self._weight2d(weights)
        w2d = wsrc.view(wsrc.shape[0], -1)
        pw, pts = self._packs(w2d, refresh, x_grad)
        bias = self._bias(biases)
        y = Builder.linear(self, x, pw, bias, residual=residual, out_dtype=out_dtype, name=name)
        n_i = [w.shape[0] for w in weights]

        def backward():
            dy = self.pop_grad(y)
            if dy is None:
                return
            dy = self._as_act(dy, name + ".dy_cast")
            with self.scope("bwd/" + name), self._bwd_lanes(rows) as par:
                par.lane()          # ---- the weight gradient(s): main lane
                geom = dict(ho=1, wo=1, ksize=1, stride=1, pad=0, upsample=False)
                if len(weights) == 1 or self.flat.contiguous(list(weights)):
                    self._wgrad(x.view(rows, 1, 1, c), None, dy, sum(n_i), self._pgrad(weights[0], through=weights[-1]), geom, c, "wgrad")
                else:
                    off = 0
                    for i, w in enumerate(weights):
                        self._wgrad(x.view(rows, 1, 1, c), None, dy[:, off:off + n_i[i]], n_i[i], self._pgrad(w), geom, c, f"wgrad.{i}")
                        off += n_i[i]
                if x_grad:
                    par.lane()      # ---- the data gradient
                    cur = self.pop_grad(x)
                    dx = Builder.linear(self, dy, pts[0], None, residual=cur, name="dgrad")
                    self.grads[_key(x)] = dx
                if biases or residual is not None:
                    par.lane()      # ---- bias sums, the residual branch's share of dY
                    off = 0
                    if biases and len(weights) > 1 and all(len(sg) == 1 for sg in biases) and self.flat.contiguous([sg[0] for sg in biases]):
                        # the stacked biases are one flat range: one column-sum launch for all of them
                        self._colsum(dy, sum(n_i), self._pgrad(biases[0][0], through=biases[-1][0]), sum(n_i), rows, False, not self._fw, "dbias")
                    else:
                        for i, w in enumerate(weights):
                            if biases:
                                for p in biases[i]:
                                    self._colsum(dy[:, off:off + n_i[i]], n_i[i], self._pgrad(p), n_i[i], rows, False, not self._fw, "dbias")
                            off += n_i[i]
                    if residual is not None:
                        self.add_grad(residual, dy)
        self.tape.append(backward)
        return y

    def t_conv(self, x, mod, *, x2=None, row_bias=None, residual=None, upsample=False, out_dtype=None, x_grad=True, name="conv"):
        """mod: modules.Conv2d.  row_bias: (fp32 [n_img, C_out] view of the time-embedding projection, matching gradient view)"""
        n, h, w, c0 = x.shape
        c1 = 0 if x2 is None else x2.shape[-1]
        weight, bias_p = mod.weight, mod.bias
        ks, stride, pad = mod.kernel_size[0], mod.stride[0], mod.padding[0]
        assert not (upsample and stride != 1)
        c_in = weight.shape[1]
        self.touched.add(id(weight))
        t_splits = [(0, c0)] if x2 is None else [(0, c0), (c0, c1)]
        if c0 + c1 != c_in:     # conv_in: zero-padded input channels (no data gradient needed there)
            assert not x_grad and x2 is None
        pw, pts = self._packs(weight.data, None, x_grad, c_pad=c0 + c1, c_split=None if x2 is None else c0, t_splits=t_splits)
        bias = self._bias([[bias_p]]) if bias_p is not None else None
        rb = None if row_bias is None else row_bias[0]
        y = Builder.conv(self, x, pw, bias, x2=x2, stride=stride, pad=pad, upsample=upsample, row_bias=rb, residual=residual,
                         out_dtype=out_dtype, name=name)
        ho, wo, co = y.shape[1], y.shape[2], weight.shape[0]
        geom = dict(ho=ho, wo=wo, ksize=ks, stride=stride, pad=pad, upsample=upsample)

        def backward():
            dy = self.pop_grad(y)
            if dy is None:
                return
            dy = self._as_act(dy, name + ".dy_cast")
            dy2d = dy.view(n * ho * wo, dy.shape[-1])
            with self.scope("bwd/" + name), self._bwd_lanes(n * ho * wo) as par:
                par.lane()          # ---- the weight gradient: main lane
                self._wgrad(x, x2, dy2d, co, self._pgrad(weight), geom, c_in, "wgrad")
                if bias_p is not None or row_bias is not None or residual is not None:
                    par.lane()      # ---- bias / time-embedding-row sums, the residual branch's share of dY
                    if row_bias is not None:      # gradient of the per-image time-embedding row: column sums per image
                        self._colsum(dy2d, co, row_bias[1], row_bias[1].stride(0), ho * wo, True, False, "d_temb_row")
                    if bias_p is not None:
                        if row_bias is not None and row_bias[1].dtype == torch.float32:
                            # the bias gradient is the sum of those per-image rows: 32 rows instead of a second pass over dY
                            self._colsum(row_bias[1], co, self._pgrad(bias_p), co, n, False, not self._fw, "dbias")
                        else:
                            self._colsum(dy2d, co, self._pgrad(bias_p), co, n * ho * wo, False, not self._fw, "dbias")
                    if residual is not None:
                        self.add_grad(residual, dy)
                if x_grad:
                    par.lane()      # ---- the data gradient
                    src = dy
                    if stride == 2:          # zero insertion turns the strided conv's data gradient into a stride-1 conv
                        z = self.empty(n, 2 * ho, 2 * wo, dy.shape[-1])
                        self._resample(L.OP_ZERO_INSERT, dy, z, n, ho, wo, dy.shape[-1], "zero_insert")
                        src = z
                        assert (2 * ho, 2 * wo) == (h, w), "stride-2 data gradient: even input sizes only"
                    for (xs, pt) in zip((x, x2), pts):
                        if xs is None:
                            continue
                        if upsample:         # gradient at the upsampled size, then 2x2 sums (backward of nearest-2x)
                            du = Builder.conv(self, src, pt, None, name="dgrad_up")
                            dx = self.empty(n, h, w, xs.shape[-1])
                            self._resample(L.OP_POOL2X2, du, dx, n, h, w, xs.shape[-1], "pool2x2")
                            self.add_grad(xs, dx)
                        else:
                            cur = self.pop_grad(xs)
                            dx = Builder.conv(self, src, pt, None, residual=cur, name="dgrad")
                            self.grads[_key(xs)] = dx
        self.tape.append(backward)
        return y

    def _resample(self, kind, src, dst, n, h, w, c, name):
        op = L.Op()
        op.kind = kind
        r = op.u.resample
        r.src, r.dst, r.n_img, r.h, r.w, r.c, r.dtype = ptr(src), ptr(dst), n, h, w, c, dt(src)
        self._emit(op, name, 0.0, (src.numel() + dst.numel()) * src.element_size(), (src, dst))

    def t_groupnorm(self, x, mod, silu, x2=None, name="groupnorm"):
        n, c0 = x.shape[0], x.shape[-1]
        c1 = 0 if x2 is None else x2.shape[-1]
        hw = x.numel() // (n * c0)
        gamma, beta = mod.weight, mod.bias
        self.touch(gamma, beta)
        stats = torch.zeros(n, mod.num_groups, 2, dtype=torch.float32, device=self.device)
        self.keep.append(stats)
        y = Builder.groupnorm(self, x, gamma.data, beta.data, mod.num_groups, mod.eps, silu, x2=x2, name=name, stats_out=stats)

        def backward():
            dy = self.pop_grad(y)
            if dy is None:
                return
            dx, dx2 = self.empty(*x.shape), (None if x2 is None else self.empty(*x2.shape))
            op = L.Op()
            op.kind = L.OP_GROUPNORM_BWD
            g = op.u.groupnorm_bwd
            g.x0, g.x1, g.dy, g.dx0, g.dx1 = ptr(x), ptr(x2), ptr(dy), ptr(dx), ptr(dx2)
            st = self._unwritten(gamma) and self._unwritten(beta)      # one flag for both outputs (MVLDM_NORM_BWD_STORE rides in `silu`)
            g.gamma, g.beta, g.stats, g.dgamma, g.dbeta = (ptr(gamma.data), ptr(beta.data), ptr(stats), ptr(self._pgrad(gamma, store=st)),
                                                           ptr(self._pgrad(beta, store=st)))
            g.workspace, g.workspace_bytes = ptr(self._wws), self._wws.numel()
            g.n_img, g.hw, g.c0, g.c1, g.groups, g.silu, g.dtype = n, hw, c0, c1, mod.num_groups, int(silu) | (L.NORM_BWD_STORE if st else 0), dt(x)
            self._emit(op, "bwd/" + name, 0.0, 5.0 * dy.numel() * dy.element_size(), (x, x2, dy, dx, dx2, stats))
            self.add_grad(x, dx)
            self.add_grad(x2, dx2)
        self.tape.append(backward)
        return y

    def t_layernorm(self, x, mod, name="layernorm"):
        c = x.shape[-1]
        gamma, beta = mod.weight, mod.bias
        self.touch(gamma, beta)
        y = Builder.layernorm(self, x, gamma.data, beta.data, mod.eps, name=name)

        def backward():
            dy = self.pop_grad(y)
            if dy is None:
                return
            dx = self.empty(*x.shape)
            op = L.Op()
            op.kind = L.OP_LAYERNORM_BWD
            l = op.u.layernorm_bwd
            st = self._unwritten(gamma) and self._unwritten(beta)      # (MVLDM_NORM_BWD_STORE rides in `dtype`)
            l.x, l.dy, l.dx, l.gamma, l.dgamma, l.dbeta = (ptr(x), ptr(dy), ptr(dx), ptr(gamma.data), ptr(self._pgrad(gamma, store=st)),
                                                           ptr(self._pgrad(beta, store=st)))
            l.workspace, l.workspace_bytes = ptr(self._wws), self._wws.numel()
            l.rows, l.c, l.dtype, l.eps = x.numel() // c, c, dt(x) | (L.NORM_BWD_STORE if st else 0), mod.eps
            self._emit(op, "bwd/" + name, 0.0, 3.0 * dy.numel() * dy.element_size(), (x, dy, dx))
            self.add_grad(x, dx)
        self.tape.append(backward)
        return y

    def t_attention(self, qkv, heads, head_dim, seg, lens, name="sdpa"):
        """self-attention on a fused projection qkv [M, 3C]"""
        Cw = heads * head_dim
        M = qkv.shape[0]
        lse = torch.zeros(heads, M, dtype=torch.float32, device=self.device)
        delta = torch.zeros(heads, M, dtype=torch.float32, device=self.device)
        self.keep.extend([lse, delta])
        out = Builder.attention(self, qkv[:, :Cw], qkv[:, Cw:2 * Cw], qkv[:, 2 * Cw:], heads, head_dim, seg, lens, lens, name=name, lse=lse)

        def backward():
            do = self.pop_grad(out)
            if do is None:
                return
            dqkv = self.empty(M, 3 * Cw)
            op = L.Op()
            op.kind = L.OP_ATTENTION_BWD
            d = op.u.attention_bwd
            d.q, d.k, d.v, d.out, d.dout = ptr(qkv[:, :Cw]), ptr(qkv[:, Cw:2 * Cw]), ptr(qkv[:, 2 * Cw:]), ptr(out), ptr(do)
            d.dq, d.dk, d.dv = ptr(dqkv[:, :Cw]), ptr(dqkv[:, Cw:2 * Cw]), ptr(dqkv[:, 2 * Cw:])
            d.lse, d.delta, d.seg = ptr(lse), ptr(delta), ptr(seg)
            d.ld_q = d.ld_k = d.ld_v = d.ld_dq = d.ld_dk = d.ld_dv = 3 * Cw
            d.ld_o, d.ld_do = out.stride(0), do.stride(0)
            d.heads, d.head_dim, d.n_seg, d.max_q_len, d.max_kv_len = heads, head_dim, seg.shape[0], max(lens), max(lens)
            d.total_q_rows, d.stat_ld, d.dtype, d.scale = M, M, dt(qkv), head_dim ** -0.5
            pairs = sum(l * l for l in lens)
            self._emit(op, "bwd/" + name, 14.0 * pairs * Cw, 8.0 * M * Cw * qkv.element_size(), (qkv, out, do, dqkv, lse, delta, seg))
            self.add_grad(qkv, dqkv)
        self.tape.append(backward)
        return out

    def t_silu(self, x, out_dtype=None, name="silu"):
        y = Builder.eltwise(self, x, L.ELT_SILU, out_dtype=out_dtype, name=name)

        def backward():
            dy = self.pop_grad(y)
            if dy is None:
                return
            dy = self._as_act(dy)
            dx = self.empty(*x.shape)
            self._elt(L.TE_SILU_BWD, x, dy, dx, 1, x.numel(), "bwd/" + name)
            self.add_grad(x, dx)
        self.tape.append(backward)
        return y

    def t_gelu(self, x, name="gelu"):
        """exact GELU as its own op (the inference path fuses it into the GEMM epilogue; training keeps the pre-activation)"""
        y = self.empty(*x.shape)
        op = L.Op()
        op.kind = L.OP_ELTWISE
        e = op.u.eltwise
        e.x, e.y, e.n, e.op, e.src_dtype, e.dst_dtype = ptr(x), ptr(y), x.numel(), L.ELT_GELU, dt(x), dt(y)
        self._emit(op, name, 0.0, 2.0 * x.numel() * x.element_size(), (x, y))

        def backward():
            dy = self.pop_grad(y)
            if dy is None:
                return
            dx = self.empty(*x.shape)
            self._elt(L.TE_GELU_BWD, x, dy, dx, 1, x.numel(), "bwd/" + name)
            self.add_grad(x, dx)
        self.tape.append(backward)
        return y

    def t_geglu(self, ag, name="geglu"):
        rows, d2 = ag.shape
        h = self.empty(rows, d2 // 2)
        self._elt(L.TE_GEGLU_FWD, ag, None, h, rows, d2 // 2, name)

        def backward():
            dh = self.pop_grad(h)
            if dh is None:
                return
            dag = self.empty(rows, d2)
            self._elt(L.TE_GEGLU_BWD, ag, dh, dag, rows, d2 // 2, "bwd/" + name)
            self.add_grad(ag, dag)
        self.tape.append(backward)
        return h

    def emit_backward(self):
        for fn in reversed(self.tape):
            fn()
        self.tape = []


# ================================================================================================ the network walk
def _segments(b: Builder, lens):
    from .modules import _segments as seg_of
    return seg_of(b, lens)


def _t_resnet(b: TrainBuilder, r, x, skip, tproj, name):
    with b.scope(name):
        g1 = b.t_groupnorm(x, r.norm1, True, x2=skip, name="norm1+silu")
        h = b.t_conv(g1, r.conv1, row_bias=tproj, name="conv1")
        g2 = b.t_groupnorm(h, r.norm2, True, name="norm2+silu")
        if r.conv_shortcut is not None:
            sc = b.t_conv(x, r.conv_shortcut, x2=skip, name="conv_shortcut")
        else:
            assert skip is None
            sc = x
        return b.t_conv(g2, r.conv2, residual=sc, name="conv2")


def _t_ff(b: TrainBuilder, ff, xn, residual, name="ff"):
    p = ff.net[0].proj
    ag = b.t_linear(xn, [p.weight], [[p.bias]], name=name + ".proj")
    h = b.t_geglu(ag, name=name + ".geglu")
    o = ff.net[2]
    return b.t_linear(h, [o.weight], [[o.bias]], residual=residual, name=name + ".out")


def _t_self_attn(b: TrainBuilder, attn, xn, residual, lens, extra_bias=None, name="attn"):
    qkv = b.t_linear(xn, [attn.to_q.weight, attn.to_k.weight, attn.to_v.weight], None if attn.to_q.bias is None else
                     [[attn.to_q.bias], [attn.to_k.bias], [attn.to_v.bias]], name=name + ".to_qkv")
    a = b.t_attention(qkv, attn.heads, attn.dim_head, _segments(b, lens), lens, name=name + ".sdpa")
    o = attn.to_out[0]
    biases = [o.bias] + ([extra_bias] if extra_bias is not None else [])
    return b.t_linear(a, [o.weight], [biases], residual=residual, name=name + ".to_out")


def _proj(b: TrainBuilder, mod, x2d, residual=None, name="proj"):
    """Linear, or a 1x1 Conv2d applied to token rows"""
    return b.t_linear(x2d, [mod.weight], [[mod.bias]] if mod.bias is not None else None, residual=residual, name=name)


def _t_transformer2d(b: TrainBuilder, t, x, name):
    """diffusers Transformer2DModel with the all-zero context of mvunet.py:124-128: cross-attention == its output bias"""
    n, h, w, c = x.shape
    with b.scope(name):
        g = b.t_groupnorm(x, t.norm, False, name="norm")
        hs = _proj(b, t.proj_in, g.view(n * h * w, c), name="proj_in")
        for i, blk in enumerate(t.transformer_blocks):
            with b.scope(f"transformer_blocks.{i}"):
                lens = [h * w] * n
                n1 = b.t_layernorm(hs, blk.norm1, name="norm1")
                a2 = blk.attn2
                h1 = _t_self_attn(b, blk.attn1, n1, hs, lens, extra_bias=a2.to_out[0].bias, name="attn1")
                # in the graph with an exactly zero gradient (k = v = 0): decayed by AdamW like in the reference
                b.touch(a2.to_q.weight, a2.to_k.weight, a2.to_v.weight, a2.to_out[0].weight, blk.norm2.weight, blk.norm2.bias,
                        a2.to_q.bias, a2.to_k.bias, a2.to_v.bias)
                n3 = b.t_layernorm(h1, blk.norm3, name="norm3")
                hs = _t_ff(b, blk.ff, n3, h1)
        out = _proj(b, t.proj_out, hs, residual=x.view(n * h * w, c), name="proj_out")
        return out.view(n, h, w, c)


def _t_standard_block(b: TrainBuilder, m, x, groups, name):
    """StandardTransformer (standard/transformer.py:45-136): pre-norm ViT layers over all views' tokens of a scene"""
    n, h, w, c = x.shape
    lens = [g * h * w for g in groups]
    hs = x.view(n * h * w, c)
    with b.scope(name):
        for i, (attn, ff) in enumerate(m.transformer.layers):
            with b.scope(f"transformer.layers.{i}"):
                a = attn.fn
                n1 = b.t_layernorm(hs, attn.norm, name="attn.norm")
                qkv = b.t_linear(n1, [a.to_qkv.weight], None, name="attn.to_qkv")
                o = b.t_attention(qkv, a.heads, a.dim_head, _segments(b, lens), lens, name="attn.sdpa")
                h1 = b.t_linear(o, [a.to_out[0].weight], [[a.to_out[0].bias]], residual=hs, name="attn.to_out")
                n2 = b.t_layernorm(h1, ff.norm, name="ff.norm")
                l0, l3 = ff.fn.net[0], ff.fn.net[3]
                pre = b.t_linear(n2, [l0.weight], [[l0.bias]], name="ff.net.0")
                g = b.t_gelu(pre, name="ff.gelu")
                hs = b.t_linear(g, [l3.weight], [[l3.bias]], residual=h1, name="ff.net.3")
    return hs.view(n, h, w, c)


def _t_mv_block(b: TrainBuilder, m, x, groups, name):
    """SpatialTransformer3D (mvdream/attention.py:371-439)"""
    from .mvunet import StandardTransformer
    if isinstance(m, StandardTransformer):
        return _t_standard_block(b, m, x, groups, name)
    n, h, w, c = x.shape
    tokens = h * w
    with b.scope(name):
        g = b.t_groupnorm(x, m.norm, False, name="norm")
        hs = _proj(b, m.proj_in, g.view(n * tokens, c), name="proj_in")
        for i, blk in enumerate(m.transformer_blocks):
            with b.scope(f"transformer_blocks.{i}"):
                scene_lens, view_lens = [v * tokens for v in groups], [tokens] * sum(groups)
                n1 = b.t_layernorm(hs, blk.norm1, name="norm1")
                h1 = _t_self_attn(b, blk.attn1, n1, hs, scene_lens, name="attn1_3d")
                n2 = b.t_layernorm(h1, blk.norm2, name="norm2")
                h2 = _t_self_attn(b, blk.attn2, n2, h1, view_lens, name="attn2_view")
                n3 = b.t_layernorm(h2, blk.norm3, name="norm3")
                hs = _t_ff(b, blk.ff, n3, h2)
        out = _proj(b, m.proj_out, hs, residual=x.view(n * tokens, c), name="proj_out")
        return out.view(n, h, w, c)


def emit_unet_train(b: TrainBuilder, den, x_in, timesteps, groups):
    """the walk of MultiViewUNet.forward (mvunet.py:90-208) on the tape builder; returns eps fp32 NHWC [n_img, h, w, out]"""
    u = den.unet
    n_img = x_in.shape[0]
    with b.scope("time"):
        t_emb = u.time_proj.emit(b, timesteps, dtype=b.dtype)
        te = u.time_embedding
        l1 = b.t_linear(t_emb, [te.linear_1.weight], [[te.linear_1.bias]], x_grad=False, name="time_embedding.linear_1")
        a1 = b.t_silu(l1, name="time_embedding.act")
        emb = b.t_linear(a1, [te.linear_2.weight], [[te.linear_2.bias]], name="time_embedding.linear_2")
        emb_act = b.t_silu(emb, name="temb_silu")
        rs = den._resnets_in_order()
        allp = b.t_linear(emb_act, [r.time_emb_proj.weight for r in rs], [[r.time_emb_proj.bias] for r in rs], out_dtype=torch.float32,
                          name="time_emb_proj[all]")
        d_allp = torch.zeros_like(allp)
        b.keep.append(d_allp)
        b.grads[_key(allp)] = d_allp
        tproj, off = {}, 0
        for r in rs:
            tproj[id(r)] = (allp[:, off:off + r.out_channels], d_allp[:, off:off + r.out_channels])
            off += r.out_channels
    h = b.t_conv(x_in, u.conv_in, x_grad=False, name="conv_in")
    skips = [h]
    for lvl, blk in enumerate(u.down_blocks):
        has_attn = getattr(blk, "has_cross_attention", False)
        for i, r in enumerate(blk.resnets):
            h = _t_resnet(b, r, h, None, tproj[id(r)], f"down{lvl}.resnets.{i}")
            if has_attn:
                h = _t_transformer2d(b, blk.attentions[i], h, f"down{lvl}.attentions.{i}")
            skips.append(h)
        if h.shape[1] <= 32 and h.shape[2] <= 32 and den.cfg.encoder_conditioning:
            h = _t_mv_block(b, den.cross_attn_blocks_encoder[lvl], h, groups, f"mv_encoder.{lvl}")
        if blk.downsamplers is not None:
            for d in blk.downsamplers:
                h = b.t_conv(h, d.conv, name=f"down{lvl}.downsample")
            skips.append(h)
    mid = u.mid_block
    h = _t_resnet(b, mid.resnets[0], h, None, tproj[id(mid.resnets[0])], "mid.resnets.0")
    for i, (attn, r) in enumerate(zip(mid.attentions, mid.resnets[1:])):
        h = _t_transformer2d(b, attn, h, f"mid.attentions.{i}")
        h = _t_resnet(b, r, h, None, tproj[id(r)], f"mid.resnets.{i + 1}")
    if den.cfg.mid_conditioning:
        h = _t_mv_block(b, den.cross_attn_blocks_mid[0], h, groups, "mv_mid")
    for lvl, blk in enumerate(u.up_blocks):
        has_attn = getattr(blk, "has_cross_attention", False) and den.pretrained_from is None
        for i, r in enumerate(blk.resnets):
            h = _t_resnet(b, r, h, skips.pop(), tproj[id(r)], f"up{lvl}.resnets.{i}")
            if has_attn:
                h = _t_transformer2d(b, blk.attentions[i], h, f"up{lvl}.attentions.{i}")
        if h.shape[1] <= 32 and h.shape[2] <= 32 and den.cfg.decoder_conditioning:
            h = _t_mv_block(b, den.cross_attn_blocks_decoder[lvl], h, groups, f"mv_decoder.{lvl}")
        if blk.upsamplers is not None:
            for up in blk.upsamplers:
                h = b.t_conv(h, up.conv, upsample=True, name=f"up{lvl}.upsample")
    g = b.t_groupnorm(h, u.conv_norm_out, True, name="conv_norm_out+silu")
    return b.t_conv(g, u.conv_out, out_dtype=torch.float32, name="conv_out")


def never_trained(den) -> List[nn.Parameter]:
    """parameters that never enter the training graph: the SD up-block transformers when the UNet comes from a
    pretrained SD checkpoint (mvunet.py:178: `self.pretrained_from is None` gates them)"""
    out = []
    if den.pretrained_from is not None:
        for blk in den.unet.up_blocks:
            if getattr(blk, "has_cross_attention", False):
                out += list(blk.attentions.parameters())
    return out


# ================================================================================================ one micro-batch plan
class TrainPlan:
    """input assembly + forward + loss + backward as a C-side plan, for ONE micro-batch shape (b scenes, v_c context views
    [0 = unconditional], v_t target views) or for a whole accumulation window: `parts` = one (b, v_c, v_t) per micro-batch, the
    scenes of all parts concatenated into one forward / backward over view groups.  Gradients of the parts sum in the same
    kernels (the loss of part i is the mean over ITS target elements times `loss_scale`, exactly what accumulating the
    micro-batches one by one adds up to), the weights are read once, and every launch serves the whole window."""

    def __init__(self, den, flat: FlatParams, b, v_c=None, v_t=None, hl: int = 0, wl: int = 0, dtype=torch.bfloat16, loss_scale: float = 1.0,
                 grad_scale: float = 1.0, graph: bool = False, rays=None, tune: Optional[bool] = None, store_first: bool = False):
        """`store_first`: the plan runs once per accumulation window and its first write of every gradient range is a store -- the
        caller replaces `flat.zero_grad()` by `flat.begin_window(plan.stored)`"""
        dev = flat.flat.device
        if tune is None:        # plan-time tile selection of the forward / data-gradient implicit GEMMs (MVLDM_TRAIN_AUTOTUNE=0: rules only)
            tune = os.environ.get("MVLDM_TRAIN_AUTOTUNE", "1") != "0" and os.environ.get("MVLDM_AUTOTUNE", "1") != "0"
        parts = [(int(b), int(v_c), int(v_t))] if v_c is not None else [tuple(int(q) for q in part) for part in b]
        self.parts = parts
        self.shape = parts[0] + (hl, wl) if len(parts) == 1 else (tuple(parts), hl, wl)
        lc = den.out_channels
        e = ops.epc(dtype)
        c_pad = (den.in_channels + e - 1) // e * e
        # image / target rows of every part inside the concatenated buffers
        self.img0, self.tgt0, groups, tgt_rows = [], [], [], []
        n_img = n_tgt = 0
        for (pb, pc, pt) in parts:
            self.img0.append(n_img)
            self.tgt0.append(n_tgt)
            v = pc + pt
            groups += [v] * pb
            tgt_rows += [n_img + s_ * v + pc + j for s_ in range(pb) for j in range(pt)]
            n_img += pb * v
            n_tgt += pb * pt
        self.n_img, self.n_tgt = n_img, n_tgt
        bld = TrainBuilder(dev, dtype, flat, store_first=store_first)
        self.flat, self.store_first = flat, bool(store_first)
        z = lambda *s_, dt_=torch.float32: torch.zeros(*s_, dtype=dt_, device=dev)
        # ---- inputs staged by the host (copies), assembled by HIP kernels (diffusion_wrapper.py:362-398) ----
        self.latents = z(n_img, lc, hl, wl)            # first_stage_encode of [context | target] views, per scene
        self.noise = z(n_tgt, lc, hl, wl)              # target_noise
        self.coef = z(n_img, 2)                        # (sqrt(a_t), sqrt(1 - a_t)); context rows: (1, 0)
        self.timesteps = torch.zeros(n_img, dtype=torch.int64, device=dev)
        self.extr, self.intr = z(n_img, 4, 4), z(n_img, 3, 3)
        self.loss = z(len(parts))                      # one accumulator per part (micro-batch)
        ones = torch.ones(n_tgt, 1, hl, wl, dtype=torch.float32, device=dev)
        unet_in = z(n_img, hl, wl, c_pad, dt_=dtype)
        tgt_img = torch.tensor(tgt_rows, dtype=torch.int32, device=dev)
        # the noise buffer holds target views only; add_noise walks all images with a per-image coefficient pair, so the
        # context rows read a zero "noise" with coefficient (1, 0): give it a full-size view
        self.noise_all = z(n_img, lc, hl, wl)
        with bld.scope("inputs"):
            op = L.Op()
            op.kind = L.OP_ADD_NOISE
            a = op.u.add_noise
            a.x0, a.noise, a.coef, a.dst, a.img_map = ptr(self.latents), ptr(self.noise_all), ptr(self.coef), ptr(unet_in), None
            a.n, a.c, a.hw, a.dst_c, a.dst_c_off, a.dst_dtype = n_img, lc, hl * wl, c_pad, 0, dt(unet_in)
            bld._emit(op, "add_noise -> latent channels", 0.0, 3.0 * self.latents.numel() * 4, (self.latents, self.noise_all, self.coef, unet_in))
            bld.nchw_to_nhwc(ones, unet_in, lc, img_map=tgt_img, name="target mask")
            bld.ray_encode(self.extr, self.intr, hl, wl, unet_in, lc + 1, name="ray grid", **({} if rays is None else rays.kernel_args()))
        with bld.scope("unet"):
            eps = emit_unet_train(bld, den, unet_in, self.timesteps, groups)
        dc = (lc + e - 1) // e * e
        d_eps = z(n_img, hl, wl, dc, dt_=dtype)
        ws = torch.zeros(256, dtype=torch.float64, device=dev)
        bld.fill_zero(d_eps, "d_eps = 0")
        for i, (pb, pc, pt) in enumerate(parts):        # F.mse_loss per micro-batch: the mean runs over that part's target elements
            t0, nt = self.tgt0[i], pb * pt
            op = L.Op()
            op.kind = L.OP_MSE_LOSS
            m = op.u.mse
            m.pred, m.noise, m.tgt_img, m.loss, m.dpred, m.workspace = ptr(eps), ptr(self.noise[t0:t0 + nt]), ptr(tgt_img[t0:t0 + nt]), ptr(self.loss[i:i + 1]), ptr(d_eps), ptr(ws)
            m.n_tgt, m.hw, m.c, m.accumulate, m.dpred_c, m.dpred_dtype = nt, hl * wl, lc, 1, dc, dt(d_eps)
            m.loss_scale, m.grad_scale = loss_scale, grad_scale
            bld._emit(op, "mse_loss" if len(parts) == 1 else f"mse_loss.{i}", 0.0, nt * hl * wl * lc * 8.0, (eps, self.noise, tgt_img, self.loss, d_eps, ws))
        self.n_forward_ops = len(bld.ops)
        bld.grads[_key(eps)] = d_eps
        with bld.scope("backward"):
            bld.emit_backward()
        assert not bld.tape
        self.eps, self.unet_in, self.tgt_img = eps, unet_in, tgt_img
        self.touched, self.grad_writes, self.repack = bld.touched, bld.grad_writes, bld.repack
        self.written = {(o, n_) for (_, o, n_) in bld.grad_writes}                                        # gradient ranges this plan writes
        self.stored = {(o, n_) for (_, o, n_), st in zip(bld.grad_writes, bld.grad_stored) if st}        # ... whose first write is a store
        self.pack_jobs, self._pack_dtype = bld.pack_jobs, dtype
        self._pack_batch = ops.PackBatch(bld.pack_jobs, dtype, dev) if os.environ.get("MVLDM_TRAIN_PACK_BATCH", "1") != "0" else None
        self.plan: Plan = bld.finalize(autotune=tune)
        self.graph = graph
        if graph:       # (capturing runs the plan once eagerly: the caller restores the gradient / loss accumulators)
            self.plan.capture()

    def refresh_weights(self):
        """after an optimizer step: bring the packed (forward + data-gradient) weights and the summed / concatenated fp32 copies up
        to date.  The ~380 packs are ONE launch (`ops.PackBatch`: 7.2 -> ... ms against one launch per pack, which is latency-bound);
        MVLDM_TRAIN_PACK_BATCH=0 keeps the per-pack launches (A/B; same bytes)."""
        for fn in self.repack:
            fn()
        if self._pack_batch is not None:
            self._pack_batch.run()
        else:
            for j in self.pack_jobs:
                L.check(L.load().mvldm_pack_weight(j.src, j.dst, j.n_out, j.c_in, j.ksize, j.c_pad, j.n_pad, j.k_pad, j.geglu, j.k_order,
                                                   dt(self._pack_dtype), j.transpose, j.c_off, j.n_rows, ops.stream()))

    def run(self, first: int = 0, last: Optional[int] = None):
        self.flat.dirty |= self.written
        if self.graph and first == 0 and last is None:
            self.plan.replay()
        else:
            self.plan.run(first, last)


# ================================================================================================ optimizer
@dataclass
class OptimizerCfg:
    """src/model/diffusion_wrapper.py:44-56 + config/experiment/baseline.yaml:62-73"""
    name: str = "AdamW"
    lr: float = 2.0e-5
    scale_lr: bool = False
    kwargs: Optional[dict] = None                 # forwarded like `getattr(optim, name)(params, lr=lr, **kwargs)`: betas, eps, weight_decay
    scheduler: Optional[dict] = field(default_factory=lambda: {"name": "LinearLR", "frequency": 1, "interval": "step",
                                                               "kwargs": {"start_factor": 5e-4, "total_iters": 200}})


def linear_lr_factor(step: int, start_factor: float = 1.0 / 3, end_factor: float = 1.0, total_iters: int = 5) -> float:
    """torch.optim.lr_scheduler.LinearLR in closed form: the factor in force after `step` scheduler steps"""
    return start_factor + (end_factor - start_factor) * min(step, total_iters) / total_iters


class DistributedOptimizer:
    """AdamW over the flat buffers, sharded ZeRO-1 style over `world` ranks.

    The flat gradient is cut into buckets (contiguous ranges, sizes a multiple of `world` x 4 floats); bucket k is
    reduce-scattered as soon as the backward plan has produced it (buckets complete from the END of the buffer: the
    backward pass visits layers in reverse), rank r keeps the r-th slice of every bucket, updates the matching slice of
    the master weights with its slice of the moments, and the slices are all-gathered back.  world = 1 without a process
    group: no collectives; world = 1 WITH a group (`collective=True`): the same reduce-scatter / all-gather calls on a one-rank
    communicator -- the RCCL branch end to end on a single GPU (tests/test_hip_train.py), bit-identical to the plain step.
    `update` / `sumsq` default to the HIP kernels; the CPU tests inject torch implementations (there is no CPU product path)."""

    def __init__(self, flat: FlatParams, cfg: OptimizerCfg = None, world: int = 1, rank: int = 0, group=None,
                 bucket_bytes: int = 256 << 20, max_norm: float = 0.1, update=None, sumsq=None, clip=None,
                 collective: Optional[bool] = None, effective_batch_size: Optional[int] = None, gather_dtype: Optional[torch.dtype] = None):
        cfg = cfg or OptimizerCfg()
        if cfg.name != "AdamW":
            raise NotImplementedError(f"optimizer {cfg.name}: the released config trains with AdamW (baseline.yaml:63)")
        kw = dict(cfg.kwargs or {})
        self.lr0, self.betas = cfg.lr, tuple(kw.get("betas", (0.9, 0.999)))
        if cfg.scale_lr:        # diffusion_wrapper.py:157-166: lr *= accumulate_grad_batches x devices x nodes x per-device batch size
            if not effective_batch_size:
                raise ValueError("OptimizerCfg.scale_lr=True needs effective_batch_size = accumulate_grad_batches x world x per-device "
                                 "batch size (what the reference multiplies lr by, diffusion_wrapper.py:157-166)")
            self.lr0 = cfg.lr * effective_batch_size
        self.eps, self.weight_decay = kw.get("eps", 1e-8), kw.get("weight_decay", 1e-2)
        self.sched = cfg.scheduler
        if self.sched is not None and self.sched.get("name") != "LinearLR":
            raise NotImplementedError(f"lr scheduler {self.sched.get('name')}: the released config uses LinearLR (baseline.yaml:68)")
        self.flat, self.world, self.rank, self.group, self.max_norm = flat, world, rank, group, max_norm
        self.collective = (world > 1) if collective is None else bool(collective)
        assert self.collective or world == 1, "world > 1 needs the collectives"
        self.buckets = make_buckets(flat.numel, world, bucket_bytes // 4)
        self.step_count = 0          # optimizer steps taken == scheduler steps taken
        dev = flat.flat.device
        self.owned = [(a + rank * (b - a) // world, a + (rank + 1) * (b - a) // world) for a, b in self.buckets]
        n_own = sum(b - a for a, b in self.owned)
        self.exp_avg = torch.zeros(n_own, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n_own, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(4, dtype=torch.float32, device=dev)
        self._update, self._sumsq, self._clip = update or _hip_adamw, sumsq or _hip_sumsq, clip or _hip_clip
        self._pending = []
        # ---- 16-bit parameter gather (round 6).  What the next forward needs from the other ranks is their slice of the 16-bit weight
        # PACKS -- a per-layer permutation of RNE_16(master) -- not their fp32 masters (ZeRO-1: a master is only ever updated by its
        # owner).  With `gather_dtype` (the trainer passes its compute dtype when that is 16-bit) a rank rounds its updated slice to
        # that type, all-gathers THOSE bytes (half of the fp32 gather: the exposed part of an 8-GPU step) and widens
        # the other ranks' slices back into its fp32 buffer: RNE_16(float(RNE_16(x))) == RNE_16(x), so every pack comes out bit-identical
        # to the one the fp32 gather gives.  Parameters the kernels read in fp32 (biases, norm affines: every <= 1-D parameter, ~0.1 %
        # of the elements) travel exactly, as integers through one small all-reduce.  The price: a non-owner's copy of a weight
        # master is 16-bit precise until `sync_masters()` gathers the fp32 slices (checkpoints; an EMA needs exact masters every
        # step, so the trainer leaves the 16-bit gather off when it keeps one).  MVLDM_TRAIN_GATHER16=0 keeps the fp32 gather.
        self.gather_dtype = gather_dtype if (self.collective and gather_dtype in (torch.bfloat16, torch.float16)
                                             and os.environ.get("MVLDM_TRAIN_GATHER16", "1") != "0") else None
        self.masters_exact = True
        self._p16 = self._direct_idx = self._direct_own = None
        # communication accounting (bench.py --train prints it per rank; tests read the events): bytes handed to reduce-scatter /
        # all-gather, the device time the compute stream spends WAITING for outstanding collectives (`exposed`), and per bucket the
        # compute-stream event at which its reduce was enqueued (to check that it went out before the backward pass had finished)
        self.bytes_reduced = 0
        self.bytes_gathered = 0
        self._exposed = []            # (start event, end event) pairs around every wait for collectives on the compute stream
        self.bucket_events = {}       # bucket -> event recorded on the compute stream when its reduce was enqueued (last step)
        # The timing events are an OPT-IN (bench.py --train and the tests set `account_comm = True`): a production run would otherwise
        # create ~ (buckets + 4) events per optimizer step and keep the exposed-wait pairs until somebody called comm_stats() -- nobody
        # does in a training loop.  The byte counters are plain integers and always on.
        self.account_comm = False

    # ---- learning rate (LinearLR: factor after `step_count` scheduler steps) ----
    def lr(self) -> float:
        if self.sched is None:
            return self.lr0
        return self.lr0 * linear_lr_factor(self.step_count, **self.sched.get("kwargs", {}))

    # ---- gradient exchange ----
    def reduce_bucket(self, k: int, stream=None):
        """reduce-scatter bucket k of the flat gradient (sum over ranks; the loss scale already carries 1/world)"""
        if not self.collective:
            return
        import torch.distributed as dist
        a, b = self.buckets[k]
        oa, ob = self.owned[k]
        g = self.flat.grad
        self.bytes_reduced += (b - a) * 4
        if g.is_cuda and self.account_comm:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.bucket_events[k] = ev
        if dist.get_backend(self.group) == "gloo":      # CPU tests: gloo has no reduce-scatter; the owned slice of an all-reduce is the same thing
            self._pending.append(dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:                                           # RCCL, in place: recvbuff == sendbuff + rank * recvcount
            self._pending.append(dist.reduce_scatter_tensor(g[oa:ob], g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        timed = self.account_comm and bool(self._pending) and self.flat.grad.is_cuda
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._pending:
            w.wait()
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._exposed.append((e0, e1))
        self._pending = []

    def comm_stats(self, reset: bool = True) -> dict:
        """bytes reduced / gathered and the exposed communication time since the last call (synchronises the device: call it outside
        the timed region).  `exposed_comm_ms` = device time between reaching a wait for outstanding collectives on the compute stream
        and getting past it: what overlap did NOT hide."""
        if self._exposed:
            torch.cuda.synchronize()
        out = {"bytes_reduced": int(self.bytes_reduced), "bytes_gathered": int(self.bytes_gathered),
               "exposed_comm_ms": round(sum(a.elapsed_time(b) for a, b in self._exposed), 3), "waits": len(self._exposed)}
        if reset:
            self.bytes_reduced = self.bytes_gathered = 0
            self._exposed = []
        return out

    def _init_gather16(self):
        flat = self.flat
        dev = flat.flat.device
        self._p16 = torch.zeros(flat.numel, dtype=self.gather_dtype, device=dev)
        self._p16.copy_(flat.flat)
        idx = [torch.arange(flat.offset[id(q)], flat.offset[id(q)] + q.numel(), dtype=torch.int64) for q in flat.params if q.ndim <= 1]
        idx = torch.cat(idx) if idx else torch.zeros(0, dtype=torch.int64)
        own = torch.zeros(idx.numel(), dtype=torch.int32)
        for oa, ob in self.owned:
            own |= ((idx >= oa) & (idx < ob)).to(torch.int32)
        self._direct_idx, self._direct_own = idx.to(dev), own.to(dev)

    def sync_masters(self):
        """after steps with the 16-bit gather: bring every rank's fp32 copy of the OTHER ranks' weight masters back to the exact values
        (the owners' slices, gathered in fp32 like the plain step does).  Call before anything reads the masters of the whole model --
        a checkpoint, a state dict, an EMA switched on later.  A no-op while the masters are exact."""
        if self.masters_exact or not self.collective:
            return
        import torch.distributed as dist
        p = self.flat.flat
        self.flat.wait_readers()
        if dist.get_backend(self.group) == "gloo":
            for (a, b) in self.buckets:
                n = (b - a) // self.world
                dist.all_gather([p[a + r * n:a + (r + 1) * n] for r in range(self.world)], p[a + self.rank * n:a + (self.rank + 1) * n].clone(),
                                group=self.group)
        else:
            for (a, b), (oa, ob) in zip(self.buckets, self.owned):
                dist.all_gather_into_tensor(p[a:b], p[oa:ob], group=self.group)
        self.masters_exact = True

    def step(self):
        """clip (global norm over ALL ranks' slices) + AdamW on the owned slices + all-gather of the weights"""
        self.wait()
        self.flat.wait_readers()          # a re-pack running ahead on the side stream reads the parameters this step overwrites
        g, p = self.flat.grad, self.flat.flat
        own_sq = torch.zeros(1, dtype=torch.float32, device=g.device)
        for oa, ob in self.owned:
            own_sq += self._sumsq(g[oa:ob])                 # (a handful of scalars: bookkeeping, not the data path)
        if self.collective:
            import torch.distributed as dist
            dist.all_reduce(own_sq, op=dist.ReduceOp.SUM, group=self.group)
        self._clip(own_sq, self.max_norm or 0.0, self.norm)     # norm[0] = total norm, norm[1] = clip coefficient
        total = self.norm[0]
        self.step_count += 1
        lr, off = self._lr_for_step(), 0
        dist = gloo = None
        if self.collective:
            import torch.distributed as dist
            gloo = dist.get_backend(self.group) == "gloo"
        works = []
        g16 = self.gather_dtype is not None
        if g16 and self._p16 is None:
            self._init_gather16()
        for (a, b), (oa, ob) in zip(self.buckets, self.owned):
            n = ob - oa
            self._update(p[oa:ob], g[oa:ob], self.exp_avg[off:off + n], self.exp_avg_sq[off:off + n], lr, self.betas, self.eps,
                         self.weight_decay, self.step_count, self.norm)
            off += n
            if g16:
                self._p16[oa:ob].copy_(p[oa:ob])             # RNE to the pack type
            if self.collective and not gloo:
                # in place (sendbuff == recvbuff + rank * sendcount), issued as soon as THIS bucket's slice is updated: RCCL's
                # stream picks up behind the AdamW kernel just enqueued, so the gather of bucket k runs under the update of k+1
                src = self._p16 if g16 else p
                works.append(dist.all_gather_into_tensor(src[a:b], src[oa:ob], group=self.group, async_op=True))
                self.bytes_gathered += (b - a) * (2 if g16 else 4)
        if self.collective and gloo:
            src, es = (self._p16.view(torch.uint8), 2) if g16 else (p, 1)      # (gloo moves bytes: the 16-bit slices travel as uint8)
            for (a, b) in self.buckets:
                n = (b - a) // self.world
                dist.all_gather([src[es * (a + r * n):es * (a + (r + 1) * n)] for r in range(self.world)],
                                src[es * (a + self.rank * n):es * (a + (self.rank + 1) * n)].clone(), group=self.group)
                self.bytes_gathered += (b - a) * (2 if g16 else 4)
        small = None
        if g16 and self._direct_idx.numel():
            # the fp32-consumed parameters, exactly: every rank contributes the elements it owns, zeros elsewhere; integer sum
            small = p.index_select(0, self._direct_idx).view(torch.int32) * self._direct_own
            dist.all_reduce(small, op=dist.ReduceOp.SUM, group=self.group)
            self.bytes_gathered += small.numel() * 4
        timed = self.account_comm and bool(works) and p.is_cuda
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in works:
            w.wait()
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._exposed.append((e0, e1))
        if g16:
            for (a, b), (oa, ob) in zip(self.buckets, self.owned):     # the other ranks' slices, widened (exact: 16-bit -> fp32)
                if oa > a:
                    p[a:oa].copy_(self._p16[a:oa])
                if b > ob:
                    p[ob:b].copy_(self._p16[ob:b])
            if small is not None:
                p.index_copy_(0, self._direct_idx, small.view(torch.float32))
            self.masters_exact = self.world == 1
        self.flat.bump()
        return float(total)

    def _lr_for_step(self) -> float:
        # the factor in force for optimizer step t (1-based) is the one after t-1 scheduler steps
        if self.sched is None:
            return self.lr0
        return self.lr0 * linear_lr_factor(self.step_count - 1, **self.sched.get("kwargs", {}))


def make_buckets(numel: int, world: int, bucket_elems: int) -> List[Tuple[int, int]]:
    """contiguous [a, b) ranges covering [0, numel), every length a multiple of world*4 (numel is padded by the caller's
    buffers: FlatParams aligns to 4; the last bucket absorbs the remainder and is padded virtually up to the quantum)"""
    q = world * 4
    bucket_elems = max(q, bucket_elems // q * q)
    out, a = [], 0
    while a < numel:
        b = min(numel, a + bucket_elems)
        out.append((a, b))
        a = b
    if out and (out[-1][1] - out[-1][0]) % q:
        raise ValueError(f"flat buffer of {numel} elements is not a multiple of world*4 = {q}: pad FlatParams (pad_to=world*4)")
    return out


def _hip_sumsq(g: torch.Tensor) -> torch.Tensor:
    n = ops.grad_norm(g, 0.0)
    return n[2:3].clone()


def _hip_clip(sumsq: torch.Tensor, max_norm: float, norm_out: torch.Tensor):
    """total norm + clip coefficient from an (all-reduced) sum of squares: the HIP kernel with an empty buffer"""
    ws = ops.workspace(1024 * 8, sumsq.device, "norm")
    L.check(L.load().mvldm_grad_norm(sumsq.data_ptr(), 0, sumsq.data_ptr(), max_norm, norm_out.data_ptr(), ws.data_ptr(), ops.stream()))


def _hip_adamw(p, g, m, v, lr, betas, eps, wd, step, norm):
    ops.adamw_step(p, g, m, v, lr, betas, eps, wd, step, 1.0, norm)


# ================================================================================================ the training wrapper
@dataclass
class TrainCfg:
    cfg_train: bool = True               # config/main.yaml:33,67: 10 % unconditional steps
    accumulate_grad_batches: int = 2     # config/main.yaml:84
    gradient_clip_val: float = 0.1       # config/main.yaml:82
    num_train_timesteps: int = 1000      # config/model/scheduler/ddim.yaml:4


class EMAWeights:
    """`self.ema = AveragedModel(self.denoiser, multi_avg_fn=get_ema_multi_avg_fn(0.995))` + `self.ema.update_parameters(self.denoiser)`
    (diffusion_wrapper.py:138-142,152-154; `model.ema`, off in the released config) on the flat parameter buffer: ONE fused HIP
    kernel (`mvldm_ema_update`: avg.lerp_(p, 1 - decay), torch's arithmetic) per update instead of a foreach over ~1500 tensors.
    Like AveragedModel the first update copies the parameters.  Parameters outside the flat buffer (never trained here) are
    constant, so their average is themselves.  `state_dict()` uses AveragedModel's key layout (`module.<name>`, `n_averaged`), i.e.
    what a DiffusionWrapper checkpoint stores under `ema.`."""

    def __init__(self, flat: FlatParams, decay: float = 0.995):
        self.flat, self.decay = flat, float(decay)
        self.avg = flat.flat.clone()                 # AveragedModel deep-copies the model at construction
        self.n_averaged = 0

    def update(self):
        from . import ops
        if self.n_averaged == 0:
            self.avg.copy_(self.flat.flat)
        else:
            ops.ema_update(self.avg, self.flat.flat, 1.0 - self.decay)
        self.n_averaged += 1

    def _names(self):
        by_id = {id(p): n for n, p in self.flat.module.named_parameters()}
        return [(by_id[id(p)], p) for p in self.flat.params], [(by_id[id(p)], p) for p in self.flat.excluded]

    def state_dict(self) -> Dict[str, torch.Tensor]:
        trained, _ = self._names()
        sd = {"module." + k: v.detach().clone() for k, v in self.flat.module.state_dict().items()}     # untrained parameters, buffers
        for n, p in trained:
            o = self.flat.offset[id(p)]
            sd["module." + n] = self.avg[o:o + p.numel()].view(p.shape).detach().clone()
        sd["n_averaged"] = torch.tensor(self.n_averaged, dtype=torch.long)
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        trained, _ = self._names()
        with torch.no_grad():
            for n, p in trained:
                o = self.flat.offset[id(p)]
                self.avg[o:o + p.numel()].copy_(sd["module." + n].reshape(-1))
        self.n_averaged = int(sd.get("n_averaged", torch.tensor(1)))

    @contextlib.contextmanager
    def applied(self):
        """`model = self.ema` (use_ema_sampling, diffusion_wrapper.py:460-463): the averaged weights in place of the live ones
        for the duration of the block (recorded inference plans re-pack on entry and on exit)"""
        self.flat.wait_readers()          # (typically called right after a training window: its re-pack may still be running ahead)
        live = self.flat.flat.clone()
        self.flat.flat.copy_(self.avg)
        self.flat.bump()
        try:
            yield self.flat.module
        finally:
            self.flat.wait_readers()
            self.flat.flat.copy_(live)
            self.flat.bump()


class MVLDMTrainer:
    """`DiffusionWrapper` + the Lightning loop for the training path: `training_step(batch)` per micro-batch, an optimizer
    step every `accumulate_grad_batches` micro-batches."""

    def __init__(self, denoiser, autoencoder, scheduler, optimizer_cfg: OptimizerCfg = None, train_cfg: TrainCfg = None,
                 dtype=torch.bfloat16, world: int = 1, rank: int = 0, group=None, graph: bool = False, bucket_bytes: int = 256 << 20,
                 rays=None, collective: Optional[bool] = None, effective_batch_size: Optional[int] = None, ema_decay: Optional[float] = None):
        self.denoiser, self.autoencoder, self.scheduler = denoiser, autoencoder, scheduler
        self.cfg = train_cfg or TrainCfg()
        self.dtype, self.world, self.rank, self.graph, self.rays = dtype, world, rank, graph, rays
        for p in autoencoder.parameters():           # freeze.autoencoder = true (config/main.yaml:20)
            p.requires_grad_(False)
        self.flat = _flat_padded(denoiser, world)
        # (the 16-bit parameter gather needs no exact replica of the other ranks' masters -- an EMA over the whole model does)
        self.opt = DistributedOptimizer(self.flat, optimizer_cfg, world, rank, group, bucket_bytes, self.cfg.gradient_clip_val,
                                        collective=collective, effective_batch_size=effective_batch_size,
                                        gather_dtype=None if ema_decay is not None else self.dtype)
        self.plans: Dict[tuple, TrainPlan] = {}          # insertion order = least recently used first (plan_for_parts)
        self.max_plans = max(1, int(os.environ.get("MVLDM_TRAIN_MAX_PLANS", "4")))
        self.micro = 0
        self.global_step = 0
        self._weights_gen = 0          # bumped by every optimizer step; a TrainPlan re-packs lazily when it is about to run
        # `model.ema` (diffusion_wrapper.py:138-142): Lightning calls on_before_zero_grad -> ema.update_parameters at the START of
        # every accumulation window (before its zero_grad), so the average sees theta_0 first and lags the optimizer by one step
        self.ema = EMAWeights(self.flat, ema_decay) if ema_decay is not None else None

    # ---- plans -------------------------------------------------------------------------------------------
    def plan_for(self, b, v_c, v_t, hl, wl) -> TrainPlan:
        return self.plan_for_parts([(b, v_c, v_t)], hl, wl)

    def plan_for_parts(self, parts, hl, wl) -> TrainPlan:
        """the recorded plan of one micro-batch shape (one part) or of a whole accumulation window (one part per micro-batch)"""
        parts = [tuple(int(q) for q in part) for part in parts]
        key = parts[0] + (hl, wl) if len(parts) == 1 else (tuple(parts), hl, wl)
        tp = self.plans.get(key)
        if tp is not None:
            self.plans[key] = self.plans.pop(key)           # most recently used last
        if tp is None:
            # The reference draws the context count and the CFG drop per micro-batch (diffusion_wrapper.py:336,381), so a window of
            # `acc` micro-batches has S^acc shape combinations, and every recorded plan keeps its own activation arena and its own
            # forward + transposed packs of the 926 M parameters.  Keep the `max_plans` most recently used (MVLDM_TRAIN_MAX_PLANS,
            # default 4 -- a handful of GB each at configs[3] size); an evicted shape is recorded again when it comes back.
            while len(self.plans) >= self.max_plans:
                old_key = next(iter(self.plans))
                old = self.plans.pop(old_key)
                ev = old.__dict__.pop("_repack_event", None)
                if ev is not None and torch.cuda.is_available():
                    torch.cuda.current_stream().wait_event(ev)      # its re-pack may still be in flight on the side stream
                del old
            acc = self.cfg.accumulate_grad_batches
            use_graph = self.graph and not self.opt.collective
            saved = self.flat.grad.clone() if use_graph else None       # a plan recorded mid-accumulation must not disturb it
            # a window plan (one part per micro-batch) runs once per optimizer step: its first write of every gradient range is a store
            # and `training_window` skips the zero_grad (MVLDM_TRAIN_STORE_FIRST=0: accumulate into a zeroed buffer like the
            # micro-batch plans, A/B knob)
            store = len(parts) > 1 and os.environ.get("MVLDM_TRAIN_STORE_FIRST", "1") != "0"
            tp = TrainPlan(self.denoiser, self.flat, parts, None, None, hl, wl, self.dtype, loss_scale=1.0 / acc,
                           grad_scale=1.0 / (acc * self.world), graph=use_graph, rays=self.rays, store_first=store)
            if saved is not None:
                self.flat.grad.copy_(saved)
                tp.loss.zero_()
            used = {id(p) for p in self.flat.params}
            missing = used - tp.touched
            assert not missing, f"{len(missing)} trained parameter(s) never entered the training graph (static exclusion list is incomplete)"
            tp.weights_gen = self._weights_gen      # packed at record time from the current weights
            self.plans[key] = tp
        return tp

    # ---- the reference's training_step, host part (diffusion_wrapper.py:324-400) ---------------------------
    def _host_part(self, batch, index=None, second=None, relative_coin=None, unconditional=None, noise=None, timestep=None, encode_noise=None):
        """the host side of one micro-batch up to (not including) the VAE encode.  The random choices of the reference (context
        count :336, relative vs absolute poses :346, CFG drop :381, noise :362, timesteps :363) are drawn here the same way and in
        the same order unless given explicitly (tests / reproducible runs): `second` = the second torch.randint of sample_indices
        (relative index, or which context view is kept), `relative_coin` / `unconditional` = what the two np.random.choice calls
        returned.  (noise / timesteps are drawn after the encode, like the reference: `_finish_part`.)"""
        from .pipeline import absolute_to_relative_camera
        ctx, tgt = batch["context"], batch["target"]
        v_c0 = ctx["image"].shape[1]
        if index is None:
            index = int(torch.randint(1, v_c0 + 1, size=(1,)).item())
        c_img, c_ext, c_int, t_img, t_ext, t_int, rel_index = sample_indices(ctx, tgt, index, random=True, second=second)
        b, v_c = c_img.shape[:2]
        v_t = t_img.shape[1]
        ext = torch.cat([c_ext, t_ext], dim=1)
        if relative_coin is None:
            relative_coin = bool(np.random.choice([False, True], 1, p=[0.50, 0.50])[0])
        if not relative_coin:          # diffusion_wrapper.py:347-350: `if relative_pose == 0` converts to RELATIVE poses
            ext = absolute_to_relative_camera(ext.float(), index=rel_index).float()
        intr = torch.cat([c_int, t_int], dim=1)
        images = torch.cat([c_img, t_img], dim=1)
        dev = self.flat.flat.device
        x = images.reshape(b * (v_c + v_t), *images.shape[2:]).to(dev, torch.float32).contiguous()
        return dict(b=b, v_c=v_c, v_t=v_t, x=x, ext=ext, intr=intr, unconditional=unconditional, noise=noise, timestep=timestep,
                    encode_noise=encode_noise)

    def _encode(self, xs: Sequence[torch.Tensor], encode_noises: Sequence[Optional[torch.Tensor]]) -> List[torch.Tensor]:
        """first_stage_encode (diffusion_wrapper.py:278-287) of the images of one or several micro-batches in ONE encoder call"""
        from .pipeline import VAE_SCALE
        with torch.no_grad():
            post = self.autoencoder.encode(torch.cat(list(xs), 0) if len(xs) > 1 else xs[0], dtype=self.dtype, pre_scale=2.0, pre_shift=-1.0).latent_dist
            en = None
            if any(n is not None for n in encode_noises):
                assert all(n is not None for n in encode_noises), "encode_noise: give it for every micro-batch of the window or for none"
                en = torch.cat([n.reshape(x.shape[0], -1, *n.shape[-2:]) for n, x in zip(encode_noises, xs)], 0)
            lat = post.sample(noise=en, scale=VAE_SCALE)               # [sum b*v, 4, hl, wl]
        return list(torch.split(lat, [x.shape[0] for x in xs], 0))

    def _finish_part(self, part: dict, lat: torch.Tensor) -> dict:
        """the draws that follow the encode in the reference (:362-381), and the effective shape of the part"""
        dev = self.flat.flat.device
        b, v_c, v_t = part["b"], part["v_c"], part["v_t"]
        hl, wl = lat.shape[-2:]
        part["lat"] = lat.view(b, v_c + v_t, -1, hl, wl)
        if part["unconditional"] is None:
            part["unconditional"] = bool(np.random.choice([False, True], 1, p=[0.90, 0.10])[0]) if self.cfg.cfg_train else True
        if part["noise"] is None:
            part["noise"] = torch.randn((b, v_t, lat.shape[1], hl, wl), device=dev)
        if part["timestep"] is None:
            part["timestep"] = torch.randint(0, self.cfg.num_train_timesteps, size=(b,), device=dev, dtype=torch.long)
        part["vc_eff"] = 0 if part["unconditional"] else v_c
        return part

    def _stage_part(self, tp: TrainPlan, i: int, part: dict):
        """copy part i's inputs into its slice of the plan's fixed buffers (copies only: the assembly is HIP kernels of the plan)"""
        dev = self.flat.flat.device
        b, v_c, v_t, vc_eff = part["b"], part["v_c"], part["v_t"], part["vc_eff"]
        assert tp.parts[i] == (b, vc_eff, v_t)
        v = vc_eff + v_t
        i0, t0 = tp.img0[i], tp.tgt0[i]
        lat, noise = part["lat"], part["noise"].to(dev)
        keep = slice(v_c, None) if part["unconditional"] else slice(None)
        tp.latents[i0:i0 + b * v].copy_(lat[:, keep].reshape(b * v, *lat.shape[2:]))
        tp.noise[t0:t0 + b * v_t].copy_(noise.reshape(b * v_t, *noise.shape[2:]))
        na = tp.noise_all[i0:i0 + b * v].view(b, v, *tp.noise_all.shape[1:])
        na[:, vc_eff:].copy_(noise)
        ac = self.scheduler.alphas_cumprod.to(dev)
        t_dev = part["timestep"].to(dev)
        coef = torch.zeros(b, v, 2, device=dev)
        coef[:, :vc_eff, 0] = 1.0
        coef[:, vc_eff:, 0] = (ac[t_dev] ** 0.5)[:, None]
        coef[:, vc_eff:, 1] = ((1 - ac[t_dev]) ** 0.5)[:, None]
        tp.coef[i0:i0 + b * v].copy_(coef.view(b * v, 2))
        ts = torch.zeros(b, v, dtype=torch.int64, device=dev)
        ts[:, vc_eff:] = t_dev[:, None]
        tp.timesteps[i0:i0 + b * v].copy_(ts.view(-1))
        tp.extr[i0:i0 + b * v].copy_(part["ext"][:, keep].reshape(b * v, 4, 4))
        tp.intr[i0:i0 + b * v].copy_(part["intr"][:, keep].reshape(b * v, 3, 3))

    def prepare(self, batch, **choices) -> TrainPlan:
        """stage ONE micro-batch into the plan of its shape (recorded on first use); returns the plan"""
        self._take_prefetched(())           # a window encoded ahead (`_start_prefetch`) may still be using the encoder on its side stream: wait, drop it
        part = self._host_part(batch, **choices)
        part = self._finish_part(part, self._encode([part["x"]], [part["encode_noise"]])[0])
        lat = part["lat"]
        tp = self.plan_for(part["b"], part["vc_eff"], part["v_t"], lat.shape[-2], lat.shape[-1])
        self._stage_part(tp, 0, part)
        return tp

    def sync_masters(self):
        """multi-rank runs with the 16-bit parameter gather (DistributedOptimizer): make every rank's fp32 masters exact again -- call
        before `denoiser.state_dict()` / a checkpoint.  (`denoiser_state_dict()` does.)"""
        self.opt.sync_masters()

    def denoiser_state_dict(self):
        self.sync_masters()
        return self.denoiser.state_dict()

    def load_denoiser_state_dict(self, state_dict, strict: bool = True):
        """`denoiser.load_state_dict` for a trainer that is alive: the flat parameter views are written in place, so a re-pack still
        reading them on the side stream is waited for first, and every recorded plan re-packs before its next run (ADVICE round 4:
        a plain `denoiser.load_state_dict` right after a window could leave mixed packs that were never redone)."""
        self.flat.wait_readers()
        out = self.denoiser.load_state_dict(state_dict, strict=strict)
        self.opt.masters_exact = True             # (every rank has just loaded the same exact values)
        self._weights_gen += 1
        self.flat.bump()
        return out

    def _fresh(self, tp: TrainPlan):
        ev = tp.__dict__.pop("_repack_event", None)
        if ev is not None:                           # re-packed ahead of time on the side stream (`_repack_ahead`)
            torch.cuda.current_stream().wait_event(ev)
        if tp.weights_gen != self._weights_gen:      # only the plan about to run re-packs (cond / uncond / other shapes wait their turn)
            tp.refresh_weights()
            tp.weights_gen = self._weights_gen

    def _repack_ahead(self, tp: TrainPlan):
        """right after an optimizer step: re-pack the weights of the plan that just ran on a SIDE stream, so that the ~380 small
        memory-bound packing launches (5 ms) run under the next window's VAE encode and input staging (compute-bound, independent of
        the denoiser weights) instead of in front of its forward pass.  The next `_fresh(tp)` waits for the event; a different plan
        shape next time re-packs lazily as before.  MVLDM_TRAIN_REPACK_AHEAD=0 disables it."""
        if os.environ.get("MVLDM_TRAIN_REPACK_AHEAD", "1") == "0" or not torch.cuda.is_available():
            return
        side = self.__dict__.get("_side_stream")
        if side is None:
            side = self._side_stream = torch.cuda.Stream(device=self.flat.flat.device)
        side.wait_stream(torch.cuda.current_stream())        # the updated (and, under ZeRO-1, gathered) parameters
        with torch.cuda.stream(side):
            tp.refresh_weights()
            tp._repack_event = side.record_event()
        self.flat._read_event = tp._repack_event     # every in-place writer of the flat parameters waits for it (FlatParams.wait_readers)
        tp.weights_gen = self._weights_gen

    def training_step(self, batch, **choices) -> torch.Tensor:
        """one micro-batch: forward + loss + backward (gradients accumulate in the flat buffer); every
        `accumulate_grad_batches`-th call also clips, steps AdamW and advances the LR schedule.  Returns the
        micro-batch's (unscaled) loss as a device scalar."""
        acc = self.cfg.accumulate_grad_batches
        if self.micro % acc == 0:
            if self.ema is not None:
                self.ema.update()
            self.flat.zero_grad()
            for tp in self.plans.values():
                tp.loss.zero_()
        tp = self.prepare(batch, **choices)
        self._fresh(tp)
        before = tp.loss[0:1].clone()
        last_micro = (self.micro + 1) % acc == 0
        if self.opt.collective and last_micro:
            self._run_overlapped(tp)
        else:
            tp.run()
        loss = (tp.loss[0:1] - before) * acc
        self.micro += 1
        if last_micro:
            self.opt.step()
            self.global_step += 1
            self._weights_gen += 1
            self._repack_ahead(tp)
        return loss.squeeze(0)

    # ---- the frozen VAE encoder, one window ahead ----------------------------------------------------------------------
    def _prepare_window(self, batches, choices):
        """host part of every micro-batch (the reference's pre-encode random choices, in its order) and ONE encoder call"""
        choices = list(choices) if choices is not None else [{} for _ in batches]
        parts = [self._host_part(bt, **ch) for bt, ch in zip(batches, choices)]
        same_res = all(p_["x"].shape[1:] == parts[0]["x"].shape[1:] for p_ in parts)
        if same_res:
            lats = self._encode([p_["x"] for p_ in parts], [p_["encode_noise"] for p_ in parts])
        else:
            lats = [self._encode([p_["x"]], [p_["encode_noise"]])[0] for p_ in parts]
        return parts, lats

    def _start_prefetch(self, batches, choices=None):
        """`_prepare_window` of the NEXT window on a side stream, launched right after this window's plan: the encoder does not
        depend on the weights being trained, and its large, regular kernels fill the CUs the backward pass's few-hundred-row
        launches leave idle.  The random draws keep the reference's order: nothing is drawn between a window's plan launch and the
        next window's pre-encode choices (`_finish_part` of THIS window ran before, the next window's post-encode draws follow its
        encode, at its own `training_window` call)."""
        main = torch.cuda.current_stream()
        if self.__dict__.get("_enc_stream") is None:
            self._enc_stream = torch.cuda.Stream(device=self.flat.flat.device)
        # The encoder has ONE set of plan buffers (vae.py: source copy, arena, output) and `_busy_event` only orders side-stream use
        # before LATER main-stream use.  The other direction: a main-stream encode of THIS window (`_prepare_window` after a prefetch
        # mismatch, the first window, `prepare()`), and any main-stream copies that filled the next batches' tensors, must have finished
        # before the side stream touches the same buffers -- the event `training_window` records right after staging its parts (before
        # the plan launch: the overlap with the backward pass stays).  (with a graph-replayed plan the host is far enough
        # ahead for the side-stream encode to overwrite a main-stream encode still in flight.)
        staged = self.__dict__.get("_staged_event")
        if staged is not None:
            self._enc_stream.wait_event(staged)
        else:
            self._enc_stream.wait_stream(main)
        with torch.cuda.stream(self._enc_stream):
            parts, lats = self._prepare_window(batches, choices)
            for t in lats:
                t.record_stream(main)
            ev = torch.cuda.Event()
            ev.record(self._enc_stream)
        self.autoencoder.__dict__["_busy_event"] = ev          # any other user of the encoder's plans (a validation sample on the main stream) waits for it
        self._prefetched = dict(ids=self._batches_key(batches), choices=self._choices_key(choices), parts=parts, lats=lats, event=ev)

    @staticmethod
    def _batches_key(batches):
        """identity AND state of what was encoded ahead: the batch objects, their image tensors' storage and in-place version counters
        (images, poses, ...: a loader that refills the same tensors in place must not get the previous contents' latents)"""
        key = []
        for bt in batches:
            k = [id(bt)]
            for side in ("context", "target"):
                d_ = bt.get(side, {}) if isinstance(bt, dict) else {}
                for name in sorted(d_):
                    t = d_[name]
                    if torch.is_tensor(t):
                        k += [name, id(t), t.data_ptr(), t._version, tuple(t.shape)]
            key.append(tuple(k))
        return tuple(key)

    def _prefetch_ok(self) -> bool:
        if not self.opt.collective:
            return True
        import torch.distributed as dist
        return dist.get_backend(self.opt.group) != "gloo"

    @staticmethod
    def _choices_key(choices):
        """the explicit draws a window was prepared with (index, second, relative_coin, unconditional, noise, timestep, encode_noise per
        micro-batch): a prefetched window is only taken when the caller passes the same ones again"""
        if choices is None:
            return None
        key = []
        for ch in choices:
            k = []
            for name in sorted(ch or {}):
                v = ch[name]
                k.append((name, ("t", id(v), v.data_ptr(), v._version, tuple(v.shape)) if torch.is_tensor(v) else repr(v)))
            key.append(tuple(k))
        return tuple(key)

    def _take_prefetched(self, batches, choices=None):
        pre = self.__dict__.pop("_prefetched", None)
        if pre is None:
            return None
        torch.cuda.current_stream().wait_event(pre["event"])
        if pre["ids"] != self._batches_key(batches) or pre["choices"] != self._choices_key(choices):
            return None               # prepared for other batches / other explicit draws (or the same objects, modified since): encode what came
        return pre["parts"], pre["lats"]

    def training_window(self, batches: Sequence[dict], choices: Optional[Sequence[dict]] = None, prefetch=None) -> torch.Tensor:
        """a whole accumulation window -- `accumulate_grad_batches` micro-batches -- as ONE forward / loss / backward plan over the
        concatenated scenes, then the optimizer step: the same gradients as `training_step` called once per micro-batch (each
        part's loss is the mean over its own target elements / accumulate_grad_batches; sums differ in rounding order only),
        with the weights read once, half the launches and one VAE-encoder call.  Returns the per-micro-batch (unscaled) losses
        `[accumulate_grad_batches]`.  Draw order: the pre-encode choices of every micro-batch, ONE encoder call (one posterior draw
        for all views), then each micro-batch's post-encode choices.
        `prefetch=(next_batches, next_choices)`: encode the NEXT window (the same objects must then be passed to the next call) on a
        side stream while this window's backward runs -- same draws in the same order, same result, the encoder's 12 ms off the
        critical path (`_start_prefetch`)."""
        acc = self.cfg.accumulate_grad_batches
        assert len(batches) == acc and self.micro % acc == 0, "training_window takes one full accumulation window at its start"
        parts, lats = self._take_prefetched(batches, choices) or self._prepare_window(batches, choices)
        parts = [self._finish_part(p_, lat) for p_, lat in zip(parts, lats)]
        hw = {tuple(p_["lat"].shape[-2:]) for p_ in parts}
        assert len(hw) == 1, "one accumulation window, one latent resolution"
        hl, wl = next(iter(hw))
        if self.ema is not None:
            self.ema.update()
        tp = self.plan_for_parts([(p_["b"], p_["vc_eff"], p_["v_t"]) for p_ in parts], hl, wl)
        if tp.store_first:
            self.flat.begin_window(tp.stored)       # (nothing to zero in the steady state: every range's first write is a store)
        else:
            self.flat.zero_grad()
        for tp_ in self.plans.values():
            tp_.loss.zero_()
        for i, p_ in enumerate(parts):
            self._stage_part(tp, i, p_)
        if torch.cuda.is_available():
            self._staged_event = torch.cuda.current_stream().record_event()      # (what `_start_prefetch` orders its side stream after)
        self._fresh(tp)
        if self.opt.collective:
            self._run_overlapped(tp)
        else:
            tp.run()
        if prefetch is not None and self._prefetch_ok():
            # the NEXT window's host part + VAE encode, on a side stream, under this window's backward (and, on RCCL, beside its bucket
            # reduce-scatters, which run on RCCL's own stream).  Not on the gloo test backend: two CPU-staged ranks sharing one GPU made
            # it pathological (2.6 -> 57 s per optimizer step); there the window encodes at its own call.
            self._start_prefetch(*prefetch)
        losses = tp.loss * acc
        self.micro += acc
        self.opt.step()
        self.global_step += 1
        self._weights_gen += 1
        self._repack_ahead(tp)
        return losses

    def _run_overlapped(self, tp: TrainPlan):
        """backward in segments: as soon as the last write into a bucket has been issued, its reduce-scatter starts on
        RCCL's stream while the remaining (earlier-layer) backward kernels keep the compute stream busy"""
        n_ops = len(tp.plan)
        cuts = bucket_cut_points(tp.grad_writes, self.opt.buckets, n_ops)
        cuts = [(k, par_safe_cut(tp.plan.ops, end)) for k, end in cuts]       # never inside a parallel group (plan_run_range refuses that)
        done = 0
        self.opt.bucket_events = {}
        for k, end in cuts:
            if end > done:
                tp.run(done, end)
                done = end
            self.opt.reduce_bucket(k)
        if done < n_ops:
            tp.run(done, n_ops)
        if torch.cuda.is_available() and self.opt.account_comm:
            self._bwd_done_event = torch.cuda.Event(enable_timing=True)      # (tests: every bucket but the last went out before this)
            self._bwd_done_event.record()


def gradient_drift_vs_f32(trainer: "MVLDMTrainer", batch, **choices) -> dict:
    """ONE micro-batch of `trainer`'s compute dtype against the exact-f32 HIP plan of the same shape, from the SAME staged
    inputs (VAE latents, noise, timesteps, cameras) and the same master weights: relative L2 error of the whole flat gradient,
    ratio of the global norms, relative loss difference, and the worst per-parameter relative L2 among parameters that carry at
    least 1e-3 of the global gradient norm.  The f32 plan is the witness that tests/test_hip_train.py proves against the
    oracle + autograd; this is the on-GPU statement of what the 16-bit path costs at production size (bench `training.grad_rel_err`).
    Leaves the gradient / loss accumulators zeroed; takes no optimizer step."""
    flat = trainer.flat
    tp = trainer.prepare(batch, **choices)
    trainer._fresh(tp)
    b, v_c, v_t, hl, wl = tp.shape
    acc = trainer.cfg.accumulate_grad_batches
    ref = TrainPlan(trainer.denoiser, flat, b, v_c, v_t, hl, wl, torch.float32, loss_scale=1.0 / acc,
                    grad_scale=1.0 / (acc * trainer.world), graph=False, rays=trainer.rays, tune=False)
    for name in ("latents", "noise", "noise_all", "coef", "timesteps", "extr", "intr"):
        getattr(ref, name).copy_(getattr(tp, name))
    out = []
    for plan in (tp, ref):
        flat.zero_grad()
        plan.loss.zero_()
        plan.run()
        torch.cuda.synchronize()
        out.append((flat.grad.clone(), float(plan.loss[0]) * acc))
    flat.zero_grad()
    tp.loss.zero_()
    (g_lo, l_lo), (g_hi, l_hi) = out
    n_hi = float(g_hi.double().norm())
    worst, worst_name = 0.0, ""
    names = {id(p): n for n, p in trainer.denoiser.named_parameters()}
    for p in flat.params:
        o, n = flat.offset[id(p)], p.numel()
        pn = float(g_hi[o:o + n].double().norm())
        if pn >= 1e-3 * n_hi:
            e = float((g_lo[o:o + n] - g_hi[o:o + n]).double().norm()) / pn
            if e > worst:
                worst, worst_name = e, names.get(id(p), "?")
    del ref
    return {"dtype": str(trainer.dtype).replace("torch.", ""), "reference": "f32 HIP plan, same inputs and weights",
            "micro_batch": f"{b} scenes x ({v_c} ctx + {v_t} tgt) @ {hl * 8}x{wl * 8}",
            "grad_rel_l2": round(float((g_lo - g_hi).double().norm()) / n_hi, 5),
            "grad_norm_ratio": round(float(g_lo.double().norm()) / n_hi, 5), "loss_rel": round(abs(l_lo - l_hi) / abs(l_hi), 6),
            "worst_large_param_rel_l2": round(worst, 4), "worst_large_param": worst_name}


def bucket_cut_points(grad_writes: Sequence[Tuple[int, int, int]], buckets: Sequence[Tuple[int, int]], n_ops: int) -> List[Tuple[int, int]]:
    """for every bucket, the plan index right after the LAST op that writes a parameter gradient anywhere inside it;
    returned as [(bucket, end index)] sorted by end index (the order the buckets become ready).  A write is the flat range
    [off, off + numel): it holds back EVERY bucket it overlaps -- a parameter may straddle a bucket boundary, and a bucket
    may lie wholly inside one large parameter.  A bucket nothing writes to (padding, zero-gradient parameters) is ready
    at once."""
    last = {k: 0 for k in range(len(buckets))}
    starts = [a for a, _ in buckets]
    import bisect
    for op_idx, off, numel in grad_writes:
        assert numel > 0
        k = max(bisect.bisect_right(starts, off) - 1, 0)
        while k < len(buckets) and buckets[k][0] < off + numel:
            if buckets[k][1] > off:
                last[k] = max(last[k], op_idx + 1)
            k += 1
    return sorted(((k, min(e, n_ops)) for k, e in last.items()), key=lambda t: (t[1], -t[0]))


def par_safe_cut(ops, end: int) -> int:
    """the first index >= `end` that does not lie inside a parallel group (PAR_BEGIN .. PAR_END) of the plan's op list"""
    open_at = None
    for i in range(end):
        k = ops[i].kind
        if k == L.OP_PAR_BEGIN:
            open_at = i
        elif k == L.OP_PAR_END:
            open_at = None
    if open_at is None:
        return end
    i = end
    while ops[i].kind != L.OP_PAR_END:
        i += 1
    return i + 1


def _flat_padded(denoiser, world: int) -> FlatParams:
    groups = ()
    if hasattr(denoiser, "_resnets_in_order"):      # the stack `emit_unet_train` projects the time embedding with in ONE GEMM
        rs = denoiser._resnets_in_order()
        groups = ([r.time_emb_proj.weight for r in rs], [r.time_emb_proj.bias for r in rs])
    flat = FlatParams(denoiser, exclude=never_trained(denoiser), groups=groups)
    q = world * 4
    if flat.numel % q:       # pad the buffers so that every bucket splits evenly over the ranks
        pad = q - flat.numel % q
        dev = flat.flat.device
        new_p, new_g = torch.zeros(flat.numel + pad, device=dev), torch.zeros(flat.numel + pad, device=dev)
        new_p[:flat.numel].copy_(flat.flat)
        flat.flat, flat.grad, flat.numel = new_p, new_g, flat.numel + pad
        with torch.no_grad():
            for p in flat.params:
                o, n = flat.offset[id(p)], p.numel()
                p.data = flat.flat[o:o + n].view(p.shape)
                p.grad = flat.grad[o:o + n].view(p.shape)
    return flat


def sample_indices(ctx: dict, tgt: dict, index: int, random: bool = True, second: Optional[int] = None):
    """`DiffusionWrapper.sample_indices` (diffusion_wrapper.py:213-276) on the tensors the training path uses:
    index > 1: the first `index` context views condition, relative index drawn in [0, index); else ONE context view
    (random or the first) conditions and the others join the targets.  Returns
    (ctx image, extrinsics, intrinsics, tgt image, extrinsics, intrinsics, rel_index)."""
    v_c = ctx["image"].shape[1]
    if index > 1:
        rel_index = int(torch.randint(0, index, size=(1,), dtype=torch.long).item()) if second is None else int(second)
        return (ctx["image"][:, :index], ctx["extrinsics"][:, :index], ctx["intrinsics"][:, :index],
                tgt["image"], tgt["extrinsics"], tgt["intrinsics"], rel_index)
    idx = (int(torch.randint(0, v_c, size=(1,), dtype=torch.long).item()) if random else 0) if second is None else int(second)
    mask = torch.zeros(v_c, dtype=torch.bool)
    mask[idx] = True
    cat = lambda k: torch.cat([tgt[k], ctx[k][:, ~mask]], dim=1)
    return (ctx["image"][:, mask], ctx["extrinsics"][:, mask], ctx["intrinsics"][:, mask],
            cat("image"), cat("extrinsics"), cat("intrinsics"), idx)
