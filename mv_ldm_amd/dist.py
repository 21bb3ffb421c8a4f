"""Multi-GPU plumbing for the sampling path: one process per GPU (`torch.distributed`, backend "nccl" = RCCL
on ROCm; "gloo" in the CPU tests).  Scenes are independent: rank r owns scenes r, r+G, ...,
weights are replicated, the data path has NO collective.  The only exchanges are the timing reduction
(MAX over ranks) and, optionally, a gather of per-rank results on rank 0."""
from __future__ import annotations

import os
from typing import List, Sequence

import torch


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_scenes(n_scenes: int, rank: int, world: int) -> List[int]:
    """round-robin ownership: scene i -> rank i mod world"""
    return list(range(rank, n_scenes, world))


def max_over_ranks(value: float, device=None) -> float:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local: Sequence[int], device=None) -> List[List[int]]:
    """every rank's list of finished scene ids, on every rank (bookkeeping / result collection)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [list(local)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(local))
    return out
