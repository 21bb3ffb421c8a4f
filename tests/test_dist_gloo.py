"""CPU, world_size 2, gloo: the N>1 path of the sampler -- scene sharding with no data-path collective,
MAX-over-ranks timing, result bookkeeping (bench.py uses the same helpers over RCCL)."""
import os

import numpy as np
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
    out = run_schedule_sharded(_StubPipeline(), calls, img, rank, world, noise_seed=11, leaf_batch=4)     # (a rank's 6 groups: 3 + 3 per sample())
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


# ---- DDP training: bucketed reduce-scatter / sharded AdamW / all-gather (training row) ---------------------
def _torch_update(p, g, m, v, lr, betas, eps, wd, step, norm):
    """test stand-in for mvldm_adamw_step (same arithmetic in torch; there is no CPU product path)"""
    gi = g * norm[1]
    p.mul_(1 - lr * wd)
    m.mul_(betas[0]).add_(gi, alpha=1 - betas[0])
    v.mul_(betas[1]).addcmul_(gi, gi, value=1 - betas[1])
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    p.addcdiv_(m, v.sqrt() / bc2 ** 0.5 + eps, value=-lr / bc1)


def _torch_clip(sumsq, max_norm, norm_out):
    total = sumsq.sqrt()
    norm_out[0:1] = total
    norm_out[1:2] = torch.clamp(max_norm / (total + 1e-6), max=1.0) if max_norm > 0 else 1.0


def _grad_mask(flat):
    """1 at the positions parameters occupy, 0 in the alignment gaps / tail padding (real gradients never touch those)"""
    m = torch.zeros(flat.numel)
    for q in flat.params:
        m[flat.offset[id(q)]:flat.offset[id(q)] + q.numel()] = 1.0
    return m


def _toy_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(37, 64), torch.nn.Linear(64, 51, bias=False), torch.nn.LayerNorm(51), torch.nn.Linear(51, 10))


def _ddp_worker(rank, world, port, q):
    from mv_ldm_amd.train import DistributedOptimizer, OptimizerCfg, _flat_padded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _toy_model()
    model.pretrained_from = None
    flat = _flat_padded(model, world)
    opt = DistributedOptimizer(flat, OptimizerCfg(lr=1e-2, scheduler={"name": "LinearLR", "kwargs": {"start_factor": 0.5, "total_iters": 2}}),
                               world, rank, bucket_bytes=4096, max_norm=0.1, update=_torch_update,
                               sumsq=lambda g: (g.double() ** 2).sum().float().reshape(1), clip=_torch_clip)
    assert len(opt.buckets) > 3 and all((b - a) % (world * 4) == 0 for a, b in opt.buckets)
    for step in range(3):
        g = torch.Generator().manual_seed(100 * step + rank)
        flat.grad.copy_(torch.randn(flat.numel, generator=g) * 0.01 * _grad_mask(flat))      # this rank's gradients (the loss carries 1/world)
        for k in reversed(range(len(opt.buckets))):                         # the order the backward pass completes them
            opt.reduce_bucket(k)
        opt.step()
    q.put((rank, flat.flat.numpy().copy(), float(opt.norm[0]), opt.step_count))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_optimizer_equals_single_process_adamw_world2():
    from mv_ldm_amd.train import _flat_padded, bucket_cut_points, make_buckets
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process: the same three steps on the SUMMED gradients with torch.optim.AdamW + clip_grad_norm_ + LinearLR
    model = _toy_model()
    model.pretrained_from = None
    flat = _flat_padded(model, world)
    params = list(model.parameters())
    opt = torch.optim.AdamW(params, lr=1e-2)
    sch = torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.5, total_iters=2)
    for step in range(3):
        tot = sum(torch.randn(flat.numel, generator=torch.Generator().manual_seed(100 * step + r)) * 0.01 for r in range(world))
        flat.grad.copy_(tot * _grad_mask(flat))
        norm = torch.nn.utils.clip_grad_norm_(params, 0.1)
        opt.step()
        sch.step()
    n_real = sum(p.numel() for p in params)
    for rank, w, total, steps in got:
        assert steps == 3 and abs(total - float(norm)) < 1e-5 * float(norm)
        assert np.allclose(w, got[0][1])                                             # all-gather: every rank holds the same weights
        live = flat.flat.detach().numpy()
        assert np.abs(w - live).max() < 2e-6, np.abs(w - live).max()                 # == the unsharded optimizer
    assert n_real <= flat.numel < n_real + 64
    # bucket readiness: a bucket is cut right after the last plan op that writes a gradient inside it
    buckets = make_buckets(64, 2, 16)
    assert buckets == [(0, 16), (16, 32), (32, 48), (48, 64)]
    cuts = bucket_cut_points([(10, 50, 4), (11, 40, 8), (20, 20, 4), (21, 17, 3), (30, 3, 5)], buckets, 40)
    assert cuts == [(3, 11), (2, 12), (1, 22), (0, 31)]


def _g16_worker(rank, world, port, q):
    from mv_ldm_amd.train import DistributedOptimizer, OptimizerCfg, _flat_padded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    for mode, gd in (("fp32", None), ("g16", torch.bfloat16)):
        model = _toy_model()
        model.pretrained_from = None
        flat = _flat_padded(model, world)
        opt = DistributedOptimizer(flat, OptimizerCfg(lr=1e-2, scheduler=None), world, rank, bucket_bytes=4096, max_norm=0.1, update=_torch_update,
                                   sumsq=lambda g: (g.double() ** 2).sum().float().reshape(1), clip=_torch_clip, gather_dtype=gd)
        assert (opt.gather_dtype is not None) == (gd is not None)
        for step in range(3):
            g = torch.Generator().manual_seed(100 * step + rank)
            flat.grad.copy_(torch.randn(flat.numel, generator=g) * 0.01 * _grad_mask(flat))
            for k in reversed(range(len(opt.buckets))):
                opt.reduce_bucket(k)
            opt.step()
        out[mode] = flat.flat.numpy().copy()
        if gd is not None:
            assert not opt.masters_exact
            out["direct"] = opt._direct_idx.numpy().copy()
            out["owned"] = list(opt.owned)
            out["bytes"] = opt.bytes_gathered
            opt.sync_masters()
            assert opt.masters_exact
            out["synced"] = flat.flat.numpy().copy()
        else:
            out["bytes32"] = opt.bytes_gathered
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_16bit_parameter_gather_world2():
    """round 6: the ranks exchange their updated slices rounded to the PACK type (half the bytes of the fp32 all-gather).  What must hold
    on every rank: the 16-bit rounding of every parameter -- i.e. every weight pack, a permutation of it -- is bit-identical to the
    fp32 gather's; the fp32-consumed (<= 1-D) parameters and the rank's own slices are exact; `sync_masters()` restores exact
    masters everywhere."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_g16_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        o = got[rank]
        ref, g16 = torch.from_numpy(o["fp32"]), torch.from_numpy(o["g16"])
        assert torch.equal(ref.bfloat16(), g16.bfloat16())                                # the packs come out bit-identical
        assert not torch.equal(ref, g16)                                                  # ... although the other rank's masters are 16-bit precise
        d = torch.from_numpy(o["direct"])
        assert d.numel() > 0 and torch.equal(ref[d], g16[d])                              # biases / norm affines: exact
        for oa, ob in o["owned"]:
            assert torch.equal(ref[oa:ob], g16[oa:ob])                                    # the rank's own slices: exact
        assert torch.equal(torch.from_numpy(o["synced"]), ref)                            # sync_masters(): exact everywhere
        assert np.array_equal(o["fp32"], got[0]["fp32"])
        assert o["bytes"] < 0.6 * o["bytes32"]                                            # half the gathered bytes (+ the small exact exchange)


def test_bucket_cut_points_hold_back_every_bucket_a_write_overlaps():
    """a parameter that straddles a bucket boundary, and a bucket lying wholly inside one parameter, must wait for that
    parameter's gradient write (a write is the flat range [off, off + numel), not its start offset)"""
    from mv_ldm_amd.train import bucket_cut_points, make_buckets
    buckets = make_buckets(64, 2, 16)
    # op 7 writes [12, 36): starts in bucket 0, covers all of bucket 1, ends inside bucket 2; op 3 writes [60, 64)
    cuts = dict(bucket_cut_points([(7, 12, 24), (3, 60, 4)], buckets, 20))
    assert cuts == {0: 8, 1: 8, 2: 8, 3: 4}
    # the tail element alone decides: [15, 17) touches buckets 0 and 1; [31, 32) is bucket 1 only; [32, 33) bucket 2 only
    cuts = dict(bucket_cut_points([(5, 15, 2), (9, 31, 1), (2, 32, 1)], buckets, 20))
    assert cuts == {0: 6, 1: 10, 2: 3, 3: 0}
    # a bias gradient emitted BEFORE its weight's gradient must not release the bucket the weight's tail lies in
    # (the layout of cross_attn_blocks_encoder.3.proj_out at the released widths with 256 MB buckets)
    cuts = bucket_cut_points([(4, 40, 2), (6, 20, 20)], buckets, 20)         # bias at [40, 42), weight [20, 40)
    assert dict(cuts)[1] == 7 and dict(cuts)[2] == 7            # (start-offset bookkeeping released bucket 2 after op 4)
    # ends are clamped to the plan length
    assert dict(bucket_cut_points([(19, 0, 64)], buckets, 20)) == {0: 20, 1: 20, 2: 20, 3: 20}


def test_recorded_gradient_writes_cover_every_trained_parameter_exactly():
    """the tape's (op, offset, numel) records, taken from a real (meta-free, CPU-recordable) layout: every trained parameter's
    flat range is covered by the writes the builder recorded for it -- incl. the fused QKV weight gradient, one kernel that
    fills three contiguous parameters"""
    from mv_ldm_amd.train import FlatParams, TrainBuilder
    import torch.nn as nn

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.to_q, self.to_k, self.to_v = nn.Linear(8, 8, bias=False), nn.Linear(8, 8, bias=False), nn.Linear(8, 8, bias=False)
            self.n = nn.LayerNorm(8)
    m = Tiny()
    flat = FlatParams(m)
    tb = TrainBuilder.__new__(TrainBuilder)            # bookkeeping only: no device, no ops
    tb.flat, tb.touched, tb.grad_writes, tb.ops = flat, set(), [], [None] * 5
    tb.store_first, tb.grad_stored, tb._fw = False, [], False
    g = tb._pgrad(m.to_q.weight, through=m.to_v.weight)
    assert g is m.to_q.weight.grad
    assert tb.grad_writes == [(5, flat.offset[id(m.to_q.weight)], 3 * 64)]
    assert {id(m.to_q.weight), id(m.to_v.weight)} <= tb.touched
    tb.ops.append(None)
    tb._pgrad(m.n.bias)
    assert tb.grad_writes[-1] == (6, flat.offset[id(m.n.bias)], 8)
    assert tb.grad_stored == [False, False] and not tb._fw           # an accumulating plan never stores


def test_store_first_bookkeeping_and_begin_window():
    """window plans (TrainBuilder(store_first=True)): the FIRST recorded write of a gradient range is a store, any later write that
    overlaps it -- the same parameter, or one parameter of a fused range -- accumulates; `FlatParams.begin_window(stored)` zeroes
    exactly the ranges earlier plans wrote that this plan does not store, `zero_grad()` forgets them all"""
    from mv_ldm_amd.train import FlatParams, TrainBuilder
    import torch.nn as nn

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.to_q, self.to_k, self.to_v = nn.Linear(8, 8, bias=False), nn.Linear(8, 8, bias=False), nn.Linear(8, 8, bias=False)
            self.n = nn.LayerNorm(8)
    m = Tiny()
    flat = FlatParams(m)
    tb = TrainBuilder.__new__(TrainBuilder)
    tb.flat, tb.touched, tb.grad_writes, tb.ops = flat, set(), [], []
    tb.store_first, tb.grad_stored, tb._fw = True, [], False
    tb._pgrad(m.to_q.weight, through=m.to_v.weight)
    assert tb._fw                                                       # the fused QKV range: first write, a store
    tb._pgrad(m.to_k.weight)
    assert not tb._fw                                                   # inside the fused range: accumulates
    st = tb._unwritten(m.n.weight) and tb._unwritten(m.n.bias)
    tb._pgrad(m.n.weight, store=st)
    tb._pgrad(m.n.bias, store=st)
    assert st and tb.grad_stored == [True, False, True, True]
    tb._pgrad(m.n.weight)
    assert not tb._fw
    written = {(o, n) for (_, o, n) in tb.grad_writes}
    stored = {(o, n) for (_, o, n), s_ in zip(tb.grad_writes, tb.grad_stored) if s_}
    assert stored < written
    # begin_window: a stale range of another plan is zeroed, stored ranges are left for the plan to overwrite
    flat.grad.fill_(3.0)
    oq, ok_ = flat.offset[id(m.to_q.weight)], flat.offset[id(m.to_k.weight)]
    flat.dirty = {(ok_, 64), (oq, 3 * 64)}                              # an earlier plan wrote K's gradient on its own, and the fused range
    flat.begin_window(stored)
    assert flat.dirty == {(oq, 3 * 64)}                                 # still to be overwritten by this plan's store
    assert float(flat.grad[ok_:ok_ + 64].abs().max()) == 0.0           # zeroed (redundantly: the fused store covers it)
    assert float(flat.grad[oq:oq + 64].min()) == 3.0                    # not touched: the plan stores it
    flat.zero_grad()
    assert not flat.dirty and float(flat.grad.abs().max()) == 0.0


def _bench(args, env_extra, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_launches_its_own_ranks_when_started_without_a_launcher():
    """`python bench.py --gpus 2` as a plain process (the form the driver uses for N = 1) must become a 2-rank job, not a mislabelled
    1-GPU run: it starts torch.distributed.run itself (before touching the GPU), forwards the single JSON line of rank 0 and the
    exit code.  MVLDM_BENCH_DRYRUN=1 stops each rank after the rendezvous + barrier (gloo), so the contract is testable without a GPU."""
    import json
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"MVLDM_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["dryrun"] is True


def test_bench_refuses_a_rank_count_that_is_not_the_one_asked_for():
    """WORLD_SIZE=1 in the environment but --gpus 2 (a launcher that started the wrong number of ranks): no line, non-zero exit"""
    r = _bench(["--gpus", "2"], {"MVLDM_BENCH_DRYRUN": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_bench_self_launch_reports_a_failing_rank():
    """a rank that dies (here: no GPU in the test container, no dry-run knob) makes the launcher exit non-zero -- no silent success"""
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {})
    assert r.returncode != 0


def _tune_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mv_ldm_amd import plan as P
    P._TUNE_CACHE.clear(); P._WGRAD_CACHE.clear()
    # every rank "measured" something different; rank 0's choices must win everywhere
    P._TUNE_CACHE[(64, 32, 32, 32, 32, 320, 0, 1, 1, 0, 0, 960, 960, 320, 0, 1, 1, 1, 0, False, False, True, 1, 0)] = 13 if rank == 0 else 10
    if rank == 1:
        P._TUNE_CACHE[(1, 2, 3)] = 7
    P._WGRAD_CACHE[(320, 0, 32)] = 2 - rank
    P.broadcast_tune_cache(dist, src=0)
    q.put((rank, dict(P._TUNE_CACHE), dict(P._WGRAD_CACHE)))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_choices_are_rank_zeros_on_every_rank_and_survive_a_file(tmp_path):
    """plan-time tile selection times candidates per process: the ranks of one job take rank 0's choices (`broadcast_tune_cache`),
    and MVLDM_TUNE_CACHE=<file> carries them to another process (a profiler pass then records its plans without one trial launch)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tune_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] and got[0][2] == got[1][2]
    assert 13 in got[1][1].values() and (1, 2, 3) not in got[1][1] and list(got[1][2].values()) == [2]
    from mv_ldm_amd import plan as P
    keep = (dict(P._TUNE_CACHE), dict(P._WGRAD_CACHE))
    try:
        P._TUNE_CACHE.clear(); P._WGRAD_CACHE.clear()
        key = (64, 32, 32, 32, 32, 320, 0, 1, 1, 0, 0, 960, 960, 320, 0, 1, 1, 1, 0, False, False, True, 1, 0)
        P._TUNE_CACHE[key] = 13
        P._WGRAD_CACHE[(320, 0, 32)] = 2
        f = str(tmp_path / "tune.json")
        assert P.save_tune_cache(f) == f
        P._TUNE_CACHE.clear(); P._WGRAD_CACHE.clear()
        assert P.load_tune_cache(f) == 2 and P._TUNE_CACHE == {key: 13} and P._WGRAD_CACHE == {(320, 0, 32): 2}
        assert all(type(a) is type(b) for a, b in zip(next(iter(P._TUNE_CACHE)), key))       # bools stay bools: the lookup key is the tuple
    finally:
        P._TUNE_CACHE.clear(); P._WGRAD_CACHE.clear()
        P._TUNE_CACHE.update(keep[0]); P._WGRAD_CACHE.update(keep[1])


def test_bucket_cuts_never_land_inside_a_parallel_group():
    """the backward plan emits the weight gradient, the data gradient and the bias sums of a layer as parallel lanes; the segments
    between two reduce-scatters must not cut such a group (mvldm_plan_run_range refuses a range that starts or ends inside one)"""
    from types import SimpleNamespace as NS
    from mv_ldm_amd import _lib as L
    from mv_ldm_amd.train import par_safe_cut
    K = [1, L.OP_PAR_BEGIN, 14, L.OP_PAR_NEXT, 1, L.OP_PAR_NEXT, 18, L.OP_PAR_END, 2, L.OP_PAR_BEGIN, 14, L.OP_PAR_END, 3]
    ops = [NS(kind=k) for k in K]
    assert [par_safe_cut(ops, e) for e in range(len(K) + 1)] == [0, 1, 8, 8, 8, 8, 8, 8, 8, 9, 12, 12, 12, 13]


def _harness_worker(rank, world, port, q):
    from mv_ldm_amd import generate as G
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = G.merge_config(G.DEFAULT_CONFIG, {"test": {"sampling_mode": "anchored", "num_anchors_views": 4}, "seed": 7})
    ex = [G.synthetic_example(0, 40, 64, 7)]
    r = G.evaluate(cfg, ex, pipe=_StubPipeline(), rank=rank, world=world)
    q.put((rank, r["sharded_calls"], r["views_per_rank"], {f: v.numpy().copy() for f, v in r["frames"][ex[0]["scene"][0]].items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_harness_shards_the_calls_of_one_scene_when_there_are_fewer_scenes_than_ranks():
    """`generate.evaluate` with ONE 40-frame scene on two ranks (SURVEY.md §8e "within one 80-frame scene"): the anchors come from rank 0
    (broadcast once), the independent groups are split over the ranks, together the ranks hold exactly the frames -- bit for bit -- of the
    one-rank walk with the same seed, and the per-rank view counts add up"""
    from mv_ldm_amd import generate as G
    cfg = G.merge_config(G.DEFAULT_CONFIG, {"test": {"sampling_mode": "anchored", "num_anchors_views": 4}, "seed": 7})
    ex = [G.synthetic_example(0, 40, 64, 7)]
    one = G.evaluate(cfg, ex, pipe=_StubPipeline(), leaf_batch=1)
    want = one["frames"][ex[0]["scene"][0]]
    assert not one.get("sharded_calls") and len(want) == one["views"]
    # the single-rank walk seeds scene i's noise with seed * 65537 + i: the sharded walk uses the same streams
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_harness_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(g[1] for g in got) and got[0][2] == got[1][2] and sum(got[0][2]) == len(want)
    f0, f1 = got[0][3], got[1][3]
    assert not (set(f0) & set(f1)) and set(f0) | set(f1) == set(want)
    for fr in (f0, f1):
        for f, im in fr.items():
            assert torch.equal(torch.from_numpy(im), want[f].cpu()), f
