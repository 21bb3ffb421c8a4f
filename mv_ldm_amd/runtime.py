"""Process-wide settings of the HIP path and the NCHW<->NHWC boundary helpers.

The reference runs its modules under `torch.autocast(fp16)` (`trainer.precision: 16-mixed`,
config/experiment/baseline.yaml:60); here the equivalent knob is the *activation dtype* of the
kernels: bf16 (default, BASELINE.json config 2), f16, or f32 (exact-f32 MFMA, the parity mode).
"""
from __future__ import annotations

import contextlib

import torch

from . import ops

_compute_dtype = torch.bfloat16


def set_compute_dtype(dtype: torch.dtype):
    global _compute_dtype
    assert dtype in (torch.float32, torch.bfloat16, torch.float16)
    _compute_dtype = dtype


def get_compute_dtype() -> torch.dtype:
    return _compute_dtype


@contextlib.contextmanager
def compute_dtype(dtype: torch.dtype):
    prev = get_compute_dtype()
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        set_compute_dtype(prev)


def require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("mv_ldm_amd modules run only on a HIP device (no CPU fallback): move the module and its "
                           "inputs to 'cuda'")


def to_nhwc(x: torch.Tensor, dtype: torch.dtype, pad_to: int = 1) -> torch.Tensor:
    """NCHW-shaped tensor (any strides / float dtype) -> contiguous NHWC `[n, h, w, c_pad]` in `dtype`.
    Free when `x` is already channels_last in `dtype` with a channel count that needs no padding."""
    require_gpu(x)
    n, c, h, w = x.shape
    c_pad = (c + pad_to - 1) // pad_to * pad_to
    v = x.permute(0, 2, 3, 1)
    if c_pad == c and v.is_contiguous() and x.dtype == dtype:
        return v
    if x.dtype == torch.float32 and x.is_contiguous():
        return ops.nchw_to_nhwc(x, dtype, dst_c=c_pad)          # one HIP pass: transpose + cast + pad
    if c_pad == c and v.is_contiguous():
        return ops.convert(v, dtype)                             # HIP cast
    # odd strides / dtypes: let torch lay the tensor out (plumbing), then cast/pad with the HIP pass
    return ops.nchw_to_nhwc(x.float().contiguous(), dtype, dst_c=c_pad)


def from_nhwc(y: torch.Tensor) -> torch.Tensor:
    """NHWC `[n, h, w, c]` -> NCHW-shaped view (channels_last strides, no copy)."""
    return y.permute(0, 3, 1, 2)
