"""CPU: the outer generation schedules against G7 -- the call sequences recorded
from the reference's own `test_video_anchored` / `test_video_autoregressive` (tests/golden/make_golden_schedules.py).
Frame indices and image provenance are compared exactly (INT), the relative poses to fp32 round-off."""
import numpy as np
import torch

from mv_ldm_amd.schedules import anchored_schedule, autoregressive_schedule


def _check(g, i):
    p = f"c{i}_"
    mode = str(g[p + "mode"])
    ctx_index, tgt_index = [int(v) for v in g[p + "ctx_index"]], [int(v) for v in g[p + "tgt_index"]]
    limit = int(g[p + "limit_frames"])
    limit = None if limit < 0 else limit
    extr = torch.from_numpy(g[p + "abs_extr"])
    n_c = len(ctx_index)
    fn = anchored_schedule if mode == "anchored" else autoregressive_schedule
    kw = {}
    if mode == "anchored" and p + "num_anchors_views" in g.files:
        kw["num_anchors_views"] = int(g[p + "num_anchors_views"])
    calls = fn(ctx_index, extr[:n_c], tgt_index, extr[n_c:], limit_frames=limit, **kw)
    prov = g[p + "calls_ctx_prov"]
    want_ctx, want_tgt, want_tag = g[p + "calls_ctx_idx"], g[p + "calls_tgt_idx"], g[p + "calls_ctx_tag"]
    assert len(calls) == want_ctx.shape[0], (mode, len(calls), want_ctx.shape[0])
    for k, c in enumerate(calls):
        assert c.ctx_index == [int(v) for v in want_ctx[k] if v >= 0], (k, c.ctx_index, want_ctx[k])
        assert c.tgt_index == [int(v) for v in want_tgt[k] if v >= 0], (k, c.tgt_index, want_tgt[k])
        assert c.ctx_source == [int(v) for v in want_tag[k] if v >= 0], (k, c.ctx_source, want_tag[k])
        # provenance of every context image: 0 = original, else call * 8 + slot + 1 (make_golden_schedules.rec_sample)
        got = [0 if src is None else src[0] * 8 + src[1] + 1 for src in c.ctx_from]
        assert got == [int(v) for v in prov[k] if v >= 0], (k, got, prov[k])
        ce, te = g[p + "calls_ctx_extr"][k, :len(c.ctx_index)], g[p + "calls_tgt_extr"][k, :len(c.tgt_index)]
        assert np.abs(c.ctx_extrinsics.numpy() - ce).max() < 2e-6, (mode, k)
        assert np.abs(c.tgt_extrinsics.numpy() - te).max() < 2e-6, (mode, k)
    return calls


def test_schedules_match_reference(golden):
    g = golden("g7_schedules")
    for i in range(int(g["n"])):
        _check(g, i)


def test_known_call_counts(golden):
    """anchored N=278 -> 92 calls / 277 views; limit 80 -> 26 calls / 78 views;
    autoregressive N=80 -> 26 calls / 79 views"""
    g = golden("g7_schedules")
    c0, c1, c2 = _check(g, 0), _check(g, 1), _check(g, 2)
    assert (len(c0), sum(len(c.tgt_index) for c in c0)) == (92, 277)
    assert (len(c1), sum(len(c.tgt_index) for c in c1)) == (26, 78)
    assert (len(c2), sum(len(c.tgt_index) for c in c2)) == (26, 79)
    assert c0[0].tgt_index == [70, 139, 208, 277] and len(c0[0].ctx_index) == 1
    assert all(len(c.ctx_index) == 2 and len(c.tgt_index) == 3 for c in c0[1:])
    assert c1[0].tgt_index == [21, 41, 61]


def test_chained_anchor_calls(golden):
    """num_anchors_views = 7 / 10 (diffusion_wrapper.py:744-792): 1 + 1 (+1) anchor calls chained through the last
    anchor, labels 4 anchors ahead of the poses, the frame generated twice; configurations the reference itself cannot
    run are refused"""
    import pytest
    g = golden("g7_schedules")
    c6, c7 = _check(g, 6), _check(g, 7)
    assert len(c6[0].tgt_index) == 4 and len(c6[1].tgt_index) == 3 and c6[1].ctx_from[-1] == (0, 3)
    assert c6[1].tgt_index[0] == c6[0].tgt_index[-1]              # the 4th anchor's label is generated again
    assert len(c7[2].tgt_index) == 3 and c7[2].ctx_from[-1] == (1, 2)
    extr = torch.eye(4).repeat(131, 1, 1)
    with pytest.raises(ValueError):
        anchored_schedule([0], extr[:1], list(range(1, 131)), extr[1:], num_anchors_views=10)
    with pytest.raises(ValueError):
        anchored_schedule([0], extr[:1], list(range(1, 101)), extr[1:], num_anchors_views=5)


# ---- independent calls of one scene sharing a sample() (schedules.run_schedule_batched / run_schedule_sharded, leaf_batch > 1) -------
class _CountingStub:
    """CPU stand-in for MVLDMPipeline.sample (same signature): every generated view is a deterministic function of what ITS batch row
    consumes -- context images, poses, x_T, posterior noise -- so a mix-up of rows, provenance or noise changes the frames.  Without
    explicit noise it draws x_T from the global CPU generator like the product (`pipeline.py` sample()).  Test infrastructure only."""
    device = torch.device("cpu")
    latent_downscale = 8

    def __init__(self):
        self.batches = []

    def sample(self, batch, x_T=None, encode_noise=None):
        c, t = batch["context"], batch["target"]
        b, v_t = t["extrinsics"].shape[:2]
        v_c = c["extrinsics"].shape[1]
        H, W = c["image"].shape[-2:]
        self.batches.append((b, v_c, v_t))
        if x_T is None:
            x_T = torch.randn((b, v_t, 4, H // 8, W // 8))
        if encode_noise is None:
            encode_noise = torch.zeros((b * v_c, 4, H // 8, W // 8))
        ctx = c["image"].mean(dim=(1, 2, 3, 4)).view(b, 1, 1, 1, 1) + c["extrinsics"].sum(dim=(1, 2, 3)).view(b, 1, 1, 1, 1)
        pose = t["extrinsics"].reshape(b, v_t, 16).sum(-1).view(b, v_t, 1, 1, 1) + t["intrinsics"].reshape(b, v_t, 9).sum(-1).view(b, v_t, 1, 1, 1)
        n = x_T.mean(dim=(2, 3, 4)).view(b, v_t, 1, 1, 1) + encode_noise.view(b, -1).mean(dim=1).view(b, 1, 1, 1, 1)
        base = torch.linspace(0, 1, H * W).view(1, 1, 1, H, W)
        return torch.sigmoid(base + ctx + 0.1 * pose + n).expand(b, v_t, 3, H, W).contiguous(), None


def _stub_scene(n_frames, seed):
    g = torch.Generator().manual_seed(seed)
    extr = torch.eye(4).repeat(n_frames, 1, 1)
    extr[:, :3, 3] = torch.randn(n_frames, 3, generator=g) * 0.1
    intr = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]]).repeat(n_frames, 1, 1)
    return extr, intr, {0: torch.rand(3, 16, 16, generator=g)}


def test_leaf_chunks_are_balanced():
    from mv_ldm_amd.schedules import _leaf_chunks
    assert _leaf_chunks(25, 64) == [25] and _leaf_chunks(91, 64) == [46, 45] and _leaf_chunks(12, 5) == [4, 4, 4]
    assert _leaf_chunks(0, 8) == [] and _leaf_chunks(3, 1) == [1, 1, 1] and _leaf_chunks(7, 0) == [1] * 7
    for n in range(1, 70):
        for cap in (1, 2, 7, 64):
            ch = _leaf_chunks(n, cap)
            assert sum(ch) == n and max(ch) <= cap and max(ch) - min(ch) <= 1 and len(ch) == -(-n // cap)


def test_independent_calls_batched_give_the_frames_of_the_call_by_call_walk():
    """configs[2]'s shape (80 target frames, 4 anchors: 1 anchor call + 25 independent groups) and the 278-frame walk with chained
    anchor calls (a frame label generated twice keeps the later call's image), two scenes: leaf_batch in {1, 4, 64} -- identical
    frames (seeded per (scene, call) noise), and 26 sample() calls become 1 + ceil(25 / leaf_batch)"""
    from mv_ldm_amd.schedules import anchored_schedule, producer_calls, run_schedule, run_schedule_batched
    for n_frames, limit in ((81, 80), (279, None)):
        scenes = [_stub_scene(n_frames, 5 + s) for s in range(2)]
        calls = [anchored_schedule([0], e[:1], list(range(1, n_frames)), e[1:], num_anchors_views=4, limit_frames=limit,
                                   ctx_intrinsics=k[:1], tgt_intrinsics=k[1:]) for e, k, _ in scenes]
        imgs = [im for _, _, im in scenes]
        n_prod = len(producer_calls(calls[0]))
        n_leaf = len(calls[0]) - n_prod
        pipe = _CountingStub()
        want = run_schedule_batched(pipe, calls, imgs, noise_seeds=[11, 12])
        assert len(pipe.batches) == len(calls[0]) and all(b == 2 for b, _, _ in pipe.batches)
        single = run_schedule(_CountingStub(), calls[1], imgs[1], noise_seed=12)       # (and the one-scene walk)
        assert sorted(single) == sorted(want[1]) and all(torch.equal(single[f], want[1][f]) for f in single)
        for lb in (4, 64):
            pipe = _CountingStub()
            got = run_schedule_batched(pipe, calls, imgs, noise_seeds=[11, 12], leaf_batch=lb)
            assert len(pipe.batches) == n_prod + -(-n_leaf // lb), (lb, pipe.batches)
            assert max(b for b, _, _ in pipe.batches) <= 2 * lb
            for s in range(2):
                assert sorted(got[s]) == sorted(want[s])
                assert all(torch.equal(got[s][f], want[s][f]) for f in want[s]), (n_frames, lb, s)
    assert (len(calls[0]), n_prod) == (92, 1) or n_prod >= 1


def test_unseeded_batched_walk_consumes_the_cpu_generator_in_call_order():
    """without seeds x_T comes from the global CPU generator inside sample(): rows are ordered call-major, scene-minor, so one draw
    for a batch of calls yields the numbers the call-by-call walk draws one call at a time"""
    from mv_ldm_amd.schedules import anchored_schedule, run_schedule_batched
    scenes = [_stub_scene(41, 21 + s) for s in range(3)]
    calls = [anchored_schedule([0], e[:1], list(range(1, 41)), e[1:], ctx_intrinsics=k[:1], tgt_intrinsics=k[1:]) for e, k, _ in scenes]
    imgs = [im for _, _, im in scenes]
    torch.manual_seed(3)
    want = run_schedule_batched(_CountingStub(), calls, imgs)
    for lb in (5, 12):
        torch.manual_seed(3)
        got = run_schedule_batched(_CountingStub(), calls, imgs, leaf_batch=lb)
        for s in range(3):
            assert all(torch.equal(got[s][f], want[s][f]) for f in want[s]), (lb, s)


def test_sharded_walk_with_batched_leaves_single_rank():
    from mv_ldm_amd.schedules import anchored_schedule, run_schedule, run_schedule_sharded
    e, k, img = _stub_scene(41, 3)
    calls = anchored_schedule([0], e[:1], list(range(1, 41)), e[1:], ctx_intrinsics=k[:1], tgt_intrinsics=k[1:])
    want = run_schedule(_CountingStub(), calls, img, noise_seed=11)
    pipe = _CountingStub()
    got = run_schedule_sharded(pipe, calls, img, 0, 1, noise_seed=11, leaf_batch=5, broadcast=lambda t, shape, src: t)
    assert sorted(got) == sorted(want) and all(torch.equal(got[f], want[f]) for f in want)
    assert len(pipe.batches) == 1 + 3 and sorted(b for b, _, _ in pipe.batches[1:]) == [4, 4, 4]
