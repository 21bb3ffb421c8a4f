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
