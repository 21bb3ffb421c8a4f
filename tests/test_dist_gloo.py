"""CPU, world_size 2, gloo: the N>1 path of the sampler -- scene sharding with no data-path collective,
MAX-over-ranks timing, result bookkeeping (bench.py uses the same helpers over RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mv_ldm_amd.dist import gather_counts, max_over_ranks, shard_scenes


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_scenes, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_scenes(n_scenes, rank, world)
    # each rank "samples" its own scenes: no collective is needed for that
    elapsed = 1.0 + 0.5 * rank
    dist.barrier()
    worst = max_over_ranks(elapsed)
    everyone = gather_counts(mine)
    q.put((rank, mine, worst, everyone))
    dist.barrier()
    dist.destroy_process_group()


def test_scene_sharding_and_timing_reduction_world2():
    world, n_scenes = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_scenes, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    assert got[0][1] == [0, 2, 4, 6] and got[1][1] == [1, 3, 5]
    assert sorted(got[0][1] + got[1][1]) == list(range(n_scenes))          # every scene exactly once
    assert got[0][2] == got[1][2] == 1.5                                    # MAX over ranks, same on all ranks
    assert got[0][3] == got[1][3] == [[0, 2, 4, 6], [1, 3, 5]]


def test_single_process_helpers_are_identity():
    assert shard_scenes(5, 0, 1) == [0, 1, 2, 3, 4]
    assert max_over_ranks(2.5) == 2.5
    assert gather_counts([3, 4]) == [[3, 4]]
