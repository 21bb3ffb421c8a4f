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


# ---- one scene's schedule spread over ranks (broadcast the anchors once, round-robin the groups) ----------
class _StubPipeline:
    """a CPU stand-in for MVLDMPipeline.sample with the same signature: the 'generated' views are a deterministic function
    of everything a real call consumes (context images, all poses, x_T, posterior noise), so any mix-up of provenance,
    pose or noise between ranks changes the result.  Test infrastructure only."""
    device = torch.device("cpu")
    latent_downscale = 8

    def sample(self, batch, x_T=None, encode_noise=None):
        c, t = batch["context"], batch["target"]
        b, v_t = t["extrinsics"].shape[:2]
        H, W = c["image"].shape[-2:]
        ctx = c["image"].mean(dim=(1, 2, 3, 4)).view(b, 1, 1, 1, 1) + c["extrinsics"].sum(dim=(1, 2, 3)).view(b, 1, 1, 1, 1)
        pose = t["extrinsics"].reshape(b, v_t, 16).sum(-1).view(b, v_t, 1, 1, 1) + t["intrinsics"].reshape(b, v_t, 9).sum(-1).view(b, v_t, 1, 1, 1)
        n = x_T.mean(dim=(2, 3, 4)).view(b, v_t, 1, 1, 1) + encode_noise.view(b, -1).mean(dim=1).view(b, 1, 1, 1, 1)
        base = torch.linspace(0, 1, H * W).view(1, 1, 1, H, W)
        return torch.sigmoid(base + ctx + 0.1 * pose + n).expand(b, v_t, 3, H, W).contiguous(), None


def _scene(n_frames=41, seed=3):
    g = torch.Generator().manual_seed(seed)
    extr = torch.eye(4).repeat(n_frames, 1, 1)
    extr[:, :3, 3] = torch.randn(n_frames, 3, generator=g) * 0.1
    intr = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]]).repeat(n_frames, 1, 1)
    img = {0: torch.rand(3, 16, 16, generator=g)}
    return extr, intr, img


def _sharded_worker(rank, world, port, q):
    from mv_ldm_amd.schedules import anchored_schedule, run_schedule_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    extr, intr, img = _scene()
    calls = anchored_schedule([0], extr[:1], list(range(1, 41)), extr[1:], ctx_intrinsics=intr[:1], tgt_intrinsics=intr[1:])
    out = run_schedule_sharded(_StubPipeline(), calls, img, rank, world, noise_seed=11)
    q.put((rank, {f: v.numpy().copy() for f, v in out.items()}))      # by value (no shared-memory handles)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_together_produce_exactly_the_frames_of_one():
    from mv_ldm_amd.schedules import anchored_schedule, producer_calls, run_schedule
    extr, intr, img = _scene()
    calls = anchored_schedule([0], extr[:1], list(range(1, 41)), extr[1:], ctx_intrinsics=intr[:1], tgt_intrinsics=intr[1:])
    assert producer_calls(calls) == [0] and len(calls) == 13
    want = run_schedule(_StubPipeline(), calls, img, noise_seed=11)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    anchors = set(calls[0].tgt_index)
    assert anchors <= set(got[0]) and not (anchors & set(got[1]))                       # rank 0 made (and keeps) the anchors
    rest0, rest1 = set(got[0]) - anchors, set(got[1])
    assert not (rest0 & rest1) and rest0 | rest1 | anchors == set(want)                  # every frame exactly once
    assert abs(len(rest0) - len(rest1)) <= 3                                              # groups of 3, round-robin
    for r in range(world):
        for f, im in got[r].items():
            assert torch.equal(torch.from_numpy(im), want[f]), (r, f)
    # the same schedule with a different seed gives different frames (the per-call noise is really used)
    other = run_schedule(_StubPipeline(), calls, img, noise_seed=12)
    assert not torch.equal(other[1], want[1])
