"""Deterministic weight / input recipes shared by make_golden.py and the tests (test infrastructure).

Fixtures store *seeds* for weights (the tensors are regenerated on the CPU generator, which is
bit-stable for a given torch build; the GPU box runs the same image) plus a checksum that the
tests verify, and store inputs / expected outputs explicitly.
"""
from __future__ import annotations

import math

import torch


def seeded_state(module: torch.nn.Module, seed: int) -> dict:
    """Return a state_dict for `module` filled from a CPU generator, iterating keys in sorted order.
    Norm scales ~ 1 +- 0.1, norm/linear/conv biases ~ 0.05, matrices ~ N(0, 1/fan_in) so that
    activations stay O(1) through depth (and the multi-view `proj_out` is NOT left at zero)."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    out = {}
    for k in sorted(sd.keys()):
        ref = sd[k]
        if not ref.dtype.is_floating_point:
            out[k] = ref.clone()
            continue
        t = torch.randn(ref.shape, generator=g, dtype=torch.float32)
        leaf = k.split(".")
        is_norm = any(s.startswith("norm") or s in ("group_norm", "conv_norm_out") for s in leaf[:-1])
        if leaf[-1] == "bias":
            t = 0.05 * t
        elif is_norm:
            t = 1.0 + 0.1 * t
        else:
            fan_in = ref[0].numel() if ref.ndim > 1 else ref.numel()
            t = t / math.sqrt(fan_in)
        out[k] = t.to(ref.dtype)
    return out


def checksum(sd: dict) -> float:
    return float(sum(v.double().abs().sum() for k, v in sorted(sd.items()) if v.dtype.is_floating_point))


def load_seeded(module: torch.nn.Module, seed: int) -> float:
    sd = seeded_state(module, seed)
    module.load_state_dict(sd)
    return checksum(sd)


def random_cameras(b: int, v: int, seed: int):
    """Synthetic RE10K-shaped cameras: first view identity, others a small random
    SE(3) (translation N(0, 0.1^2)/axis, axis-angle N(0, 0.05^2)); normalised intrinsics
    fx=fy=0.9, cx=cy=0.5.  Returns extrinsics [b,v,4,4] (camera-to-world), intrinsics [b,v,3,3]."""
    g = torch.Generator().manual_seed(seed)
    extr = torch.eye(4).repeat(b, v, 1, 1)
    for bi in range(b):
        for vi in range(1, v):
            aa = 0.05 * torch.randn(3, generator=g)
            th = aa.norm()
            k = aa / th
            K = torch.tensor([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            R = torch.eye(3) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)
            extr[bi, vi, :3, :3] = R
            extr[bi, vi, :3, 3] = 0.1 * torch.randn(3, generator=g)
    intr = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]]).repeat(b, v, 1, 1)
    return extr, intr
