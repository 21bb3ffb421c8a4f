"""CPU: host logic of the plan builder.  The whole UNet / VAE / sampler graph is *recorded* (no
kernel is launched: weight packing is stubbed with shape-only buffers) and checked structurally:
op counts, algorithmic FLOPs against SURVEY.md §8d, fused-op properties (no standalone concat,
zero-context cross-attention folded away, one batched time_emb_proj GEMM)."""
import pytest
import torch

import mv_ldm_amd
from mv_ldm_amd import _lib as L
from mv_ldm_amd import modules, mvunet, ops, pipeline, plan, runtime, vae


@pytest.fixture()
def cpu_record(monkeypatch):
    def fake_pack(w, dtype, c_pad=None, geglu=False, c_split=None):
        n_out, c_in = w.shape[0], w.shape[1]
        k = w.shape[2] if w.ndim == 4 else 1
        e = ops.epc(dtype)
        c_pad = (c_in + e - 1) // e * e if c_pad is None else c_pad
        bk = 32 if dtype == torch.float32 else 64
        k_pad = (k * k * c_pad + bk - 1) // bk * bk
        n_pad = (n_out + 63) // 64 * 64
        k_order = int(c_pad % bk == 0 and (c_split is None or c_split % bk == 0))
        return ops.PackedWeight(torch.empty(0), n_out, n_pad, k_pad, c_pad, k, geglu, k_order)
    monkeypatch.setattr(ops, "pack_weight", fake_pack)
    for mod in (modules, mvunet, runtime, vae):
        monkeypatch.setattr(mod, "require_gpu", lambda t: None, raising=False)
    L.load()


_MODELS = {}


def build_unet_plan(views, groups, h, dtype=torch.bfloat16, widths=None):
    with torch.device("meta"):
        pass
    if widths is None:
        cfg = mvunet.MultiViewUNetCfg(pretrained_from="sd21")
    else:
        over = dict(block_out_channels=widths, attention_head_dim=tuple(max(1, c // 64) for c in widths))
        cfg = mvunet.MultiViewUNetCfg(autoencoder=mvunet.UNet2DModelCfg(block_out_channels=widths), pretrained_from="sd21",
                                      pretrained_overrides=over)
    m = _MODELS.get(widths)          # (building the 1.07 B-parameter module takes ~20 s: once per width set and test session)
    if m is None:
        m = _MODELS[widths] = mvunet.MultiViewUNet(cfg, 11, 4)
    b = plan.Builder("cpu", dtype, record=True, splitk_ws_bytes=1 << 20)
    x = torch.zeros(views, h, h, 16, dtype=dtype)
    ts = torch.zeros(views, dtype=torch.int64)
    eps = m.emit(b, x, ts, groups)
    return m, b, eps


def test_unet_plan_structure_small(cpu_record):
    m, b, eps = build_unet_plan(5, [3, 2], 8, widths=(64, 128, 256, 256))
    assert eps.shape == (5, 8, 8, 4) and eps.dtype == torch.float32
    kinds = [mm.kind for mm in b.meta]
    names = [mm.name for mm in b.meta]
    assert kinds.count(L.OP_TIMESTEP_EMBED) == 1
    assert sum("time_emb_proj" in n for n in names) == 1, "all 22 time_emb_proj must be ONE batched GEMM"
    assert not any("attn2" in n and "attentions" in n for n in names), "zero-context SD cross-attention must be folded away"
    assert sum(n.endswith("attn1_3d.sdpa") for n in names) == 9, "9 multi-view blocks"
    assert sum(n.endswith("attn2_view.sdpa") for n in names) == 9
    assert sum(n.endswith("attn1.sdpa") for n in names) == 7, "7 SD transformer blocks (down0-2 x2 + mid)"
    assert kinds.count(L.OP_MEMCPY) == 0
    p = b.finalize()
    assert len(p) == len(names)


def reference_flops(meta):
    """FLOPs of the reference's formulation.  The plan counts what it EXECUTES: a nearest-2x upsampling conv runs as four
    2x2 phase convs on the low-resolution image (ops `upsample.p0..p3`), 4/9 of the reference's 3x3-on-upsampled work."""
    return sum(mm.flops * (9.0 / 4.0 if ".p" in mm.name.rsplit("/", 1)[-1] and "upsample" in mm.name else 1.0) for mm in meta)


def test_unet_plan_flops_match_survey(cpu_record):
    """conditional forward V=5 @32x32 = 0.918 TFLOP, V=4: 0.722, V=3: 0.532 (2*MAC)"""
    for v, want in ((5, 0.918e12), (4, 0.722e12), (3, 0.532e12)):
        _, b, _ = build_unet_plan(v, [v], 32)
        assert abs(reference_flops(b.meta) / want - 1) < 0.03, (v, reference_flops(b.meta))
        executed = sum(mm.flops for mm in b.meta)
        assert 0.92 < executed / reference_flops(b.meta) < 0.95     # the three upsampling convs are 12 % of the reference count
    # cond + uncond batched as groups [5, 4]: one pass, 1.64 TFLOP, the weights are touched once
    _, b, _ = build_unet_plan(9, [5, 4], 32)
    assert abs(reference_flops(b.meta) / 1.64e12 - 1) < 0.03


def test_level_gate_skips_multiview_blocks_above_32(cpu_record):
    """mvunet.py:137,190: 3-D attention only where the feature map is <= 32x32"""
    _, b, _ = build_unet_plan(2, [2], 64, widths=(64, 128, 256, 256))
    names = [mm.name for mm in b.meta]
    assert sum(n.endswith("attn1_3d.sdpa") for n in names) == 7      # level 0 (64x64) skipped on both sides
    assert not any(n.startswith("mv_encoder.0/") or n.startswith("mv_decoder.3/") for n in names)


def test_vae_decoder_plan(cpu_record):
    v = vae.AutoencoderKL.from_pretrained("x")
    b = plan.Builder("cpu", torch.bfloat16, record=True, splitk_ws_bytes=1 << 20)
    y = v.decoder.emit(b, torch.zeros(1, 32, 32, 8, dtype=torch.bfloat16))
    assert y.shape == (1, 256, 256, 3)
    flops = reference_flops(b.meta)
    assert abs(flops / 0.622e12 - 1) < 0.05, flops       # 0.62 TFLOP / view
    b2 = plan.Builder("cpu", torch.bfloat16, record=True, splitk_ws_bytes=1 << 20)
    z = v.encoder.emit(b2, torch.zeros(1, 256, 256, 8, dtype=torch.bfloat16))
    assert z.shape == (1, 32, 32, 8)
    assert abs(sum(mm.flops for mm in b2.meta) / 0.273e12 - 1) < 0.06


def test_parallel_lanes_only_at_small_batches(cpu_record, monkeypatch):
    """MVLDM_OP_PAR_* markers (opt-in, MVLDM_PAR_ROWS): at one scene (9 images, 32x32 latents) the four phase convs of every upsampler and the 1x1
    shortcut of a resnet beside its norm1 -> conv1 -> norm2 chain are emitted as parallel lanes -- well-formed groups, the
    shortcut's lane holds exactly one op, lanes own their split-K workspaces -- and at 64 scenes no marker is emitted at all"""
    _, b_off, _ = build_unet_plan(9, [5, 4], 32)
    assert not any(mm.kind == L.OP_PAR_BEGIN for mm in b_off.meta)      # default: off (measured slower, plan.Builder.small_launch)
    monkeypatch.setenv("MVLDM_PAR_ROWS", "16384")
    _, b, _ = build_unet_plan(9, [5, 4], 32)
    kinds = [mm.kind for mm in b.meta]
    names = [mm.name for mm in b.meta]
    n_begin, n_end = kinds.count(L.OP_PAR_BEGIN), kinds.count(L.OP_PAR_END)
    assert n_begin == n_end == 3 + 14, (n_begin, n_end)       # 3 upsamplers + the 14 resnets that have a conv_shortcut
    depth, lanes, groups = 0, 0, []
    for k, n in zip(kinds, names):
        if k == L.OP_PAR_BEGIN:
            assert depth == 0
            depth, lanes, cur = 1, 1, [[]]
        elif k == L.OP_PAR_NEXT:
            assert depth == 1
            lanes += 1
            cur.append([])
        elif k == L.OP_PAR_END:
            assert depth == 1
            depth = 0
            groups.append(cur)
        elif depth:
            cur[-1].append(n)
    assert depth == 0
    up = [g for g in groups if len(g) == 4]
    sc = [g for g in groups if len(g) == 2]
    assert len(up) == 3 and all(len(lane) == 1 and f".p{i}" in lane[0] for g in up for i, lane in enumerate(g))
    assert len(sc) == 14 and all(len(g[1]) == 1 and g[1][0].endswith("conv_shortcut") for g in sc)
    assert all([n.rsplit("/", 1)[-1] for n in g[0] if "reduce" not in n][:3] == ["norm1+silu", "conv1", "norm2+silu"] for g in sc)
    # lanes > 0 use their own split-K workspace
    ws = {}
    lane = 0
    for op, k in zip(b.ops, kinds):
        if k == L.OP_PAR_BEGIN:
            lane = 0
        elif k == L.OP_PAR_NEXT:
            lane += 1
        elif k == L.OP_PAR_END:
            lane = 0
        elif k == L.OP_IGEMM and op.u.igemm.workspace:
            ws.setdefault(lane, set()).add(op.u.igemm.workspace)
    assert all(len(v) == 1 for v in ws.values()) and len({next(iter(v)) for v in ws.values()}) == len(ws) >= 4
    p = b.finalize()
    assert len(p) == len(names)
    # 8 scenes: only the 8x8 / 4x4 levels are still small; 64 scenes: nothing is
    _, b8, _ = build_unet_plan(9 * 8, [5] * 8 + [4] * 8, 32)
    in_group = False
    for mm in b8.meta:
        in_group = mm.kind == L.OP_PAR_BEGIN or (in_group and mm.kind != L.OP_PAR_END)
        if in_group and mm.kind == L.OP_IGEMM:
            assert any(lvl in mm.name for lvl in ("down2", "down3", "mid", "up0", "up1")), mm.name
    # (64 scenes: 9216 rows x 1280 columns at the 4x4 level is above the threshold too -- arithmetic, not another full-width build)
    assert not plan.Builder("cpu", torch.bfloat16, record=True).small_launch(576 * 16, 1280)
    monkeypatch.setenv("MVLDM_PAR_ROWS", "0")
    _, b0, _ = build_unet_plan(9, [5, 4], 32)
    assert not any(mm.kind == L.OP_PAR_BEGIN for mm in b0.meta)


def test_temporaries_freed_inside_a_parallel_group_wait_for_the_join(cpu_record):
    b = plan.Builder("cpu", torch.bfloat16, record=True, splitk_ws_bytes=1 << 20)
    a = b.empty(4, 8)
    with b.parallel() as par:
        par.lane()
        b.free(a)
        par.lane()
        c = b.empty(4, 8)                  # must NOT be handed the buffer lane 0 just released
        assert c.data_ptr() != a.data_ptr()
    d = b.empty(4, 8)                      # after the join it is free again
    assert d.data_ptr() == a.data_ptr()


def test_shared_cfg_prefix_structure(cpu_record):
    """`MultiViewUNet.emit(dup=(n_src, src_rows))`: conv_in and the level-0 down block (2 resnets + 2 SD transformer blocks, all
    per-image work) run on the conditional images only, three row gathers fill the unconditional images' skip tensors right
    before the first multi-view block, everything after walks the full batch; executed FLOPs drop by the prefix's share"""
    m, _, _ = build_unet_plan(1, [1], 8)

    def build(dup):
        b = plan.Builder("cpu", torch.bfloat16, record=True, splitk_ws_bytes=1 << 20)
        x = torch.zeros(18, 32, 32, 16, dtype=torch.bfloat16)
        idx = torch.tensor([1, 2, 3, 4, 6, 7, 8, 9], dtype=torch.int32)
        eps = m.emit(b, x, torch.zeros(18, dtype=torch.int64), [5, 5, 4, 4], dup=(10, idx) if dup else None)
        assert eps.shape == (18, 32, 32, 4)
        return b

    full, shared = build(False), build(True)
    assert len(shared.meta) == len(full.meta) + 3 + 3          # 3 gathers; the first 3-D attention becomes 3 launches + a merge
    att = [(op, mm) for op, mm in zip(shared.ops, shared.meta) if "mv_encoder.0" in mm.name and "attn1_3d" in mm.name and op.kind in (L.OP_ATTENTION, L.OP_ATTN_MERGE)]
    assert [mm.name.split("/")[-1] for _, mm in att] == ["attn1_3d.sdpa", "attn1_3d.sdpa.ctx_queries", "attn1_3d.sdpa.ctx_keys", "attn1_3d.merge"]
    full_att = next(mm for mm in full.meta if "mv_encoder.0" in mm.name and mm.name.endswith("attn1_3d.sdpa"))
    assert abs(sum(mm.flops for _, mm in att) / full_att.flops - 25.0 / 41.0) < 1e-6          # view x view score blocks: 25 of 41
    gathers = [(op, mm) for op, mm in zip(shared.ops, shared.meta) if op.kind == L.OP_GATHER_ROWS]
    assert [mm.name for _, mm in gathers] == ["unet/cfg_share/skip0", "unet/cfg_share/skip1", "unet/cfg_share/skip2"] or \
        [mm.name.split("/")[-1] for _, mm in gathers] == ["skip0", "skip1", "skip2"]
    assert all(op.u.gather.n_rows == 8 and op.u.gather.row_bytes == 32 * 32 * 320 * 2 for op, _ in gathers)
    first_mv = next(i for i, mm in enumerate(shared.meta) if "mv_encoder.0" in mm.name)
    for i, (op, mm) in enumerate(zip(shared.ops, shared.meta)):
        if op.kind != L.OP_IGEMM or "time" in mm.name:
            continue
        rows = op.u.igemm.n_img * op.u.igemm.h_out * op.u.igemm.w_out
        if i < first_mv:
            assert rows == 10 * 1024, (mm.name, rows)          # the shared prefix: conditional images only
        elif "mv_encoder.0" in mm.name or "up3" in mm.name:
            assert rows == 18 * 1024, (mm.name, rows)          # full batch from the first multi-view block on
    # second form: the context views (rows 0 and 5) come from a per-sample store, the shared layers run on the duplicate block
    cb = plan.Builder("cpu", torch.bfloat16, record=True, splitk_ws_bytes=1 << 20)
    store = m.emit(cb, torch.zeros(2, 32, 32, 16, dtype=torch.bfloat16), torch.zeros(2, dtype=torch.int64), [1, 1], prefix_only=True)
    assert [tuple(t.shape) for t in store] == [(2, 32, 32, 320)] * 3
    assert not any("mv_" in mm.name for mm in cb.meta) and sum(mm.kind == L.OP_IGEMM for mm in cb.meta) == 3 + 1 + 4 + 2 * 6
    b2 = plan.Builder("cpu", torch.bfloat16, record=True, splitk_ws_bytes=1 << 20)
    idx = torch.tensor([1, 2, 3, 4, 6, 7, 8, 9], dtype=torch.int32)
    ctx = torch.tensor([0, 5], dtype=torch.int32)
    m.emit(b2, torch.zeros(18, 32, 32, 16, dtype=torch.bfloat16), torch.zeros(18, dtype=torch.int64), [5, 5, 4, 4], dup=(10, idx, ctx, store))
    g2 = [(op, mm) for op, mm in zip(b2.ops, b2.meta) if op.kind == L.OP_GATHER_ROWS]
    assert len(g2) == 6 and [op.u.gather.n_rows for op, _ in g2] == [8, 2] * 3
    first_mv2 = next(i for i, mm in enumerate(b2.meta) if "mv_encoder.0" in mm.name)
    for i, (op, mm) in enumerate(zip(b2.ops, b2.meta)):
        if op.kind == L.OP_IGEMM and "time" not in mm.name and i < first_mv2:
            assert op.u.igemm.n_img * op.u.igemm.h_out * op.u.igemm.w_out == 8 * 1024, mm.name      # one copy of every target view
    f_full, f_shared = sum(mm.flops for mm in full.meta), sum(mm.flops for mm in shared.meta)
    prefix = sum(mm.flops for mm in full.meta[:next(i for i, mm in enumerate(full.meta) if "mv_encoder.0" in mm.name)])
    # (the time-embedding GEMMs in front are not shared; the first 3-D attention drops 16 of its 41 view x view blocks)
    assert abs((f_full - f_shared) / (prefix * 8 / 18 + full_att.flops * 16 / 41) - 1) < 1e-2


def test_tail_drop_structure(cpu_record):
    """`MultiViewUNet.emit(tail=(keep_rows, drops))`: in the last multi-view block the 3-D attention attends the kept views' queries
    only, everything behind it and the output stage run on the kept views (16 of 18 images), eps is scattered back"""
    m, _, _ = build_unet_plan(1, [1], 8)
    keep = torch.tensor([1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17], dtype=torch.int32)

    def build(tail):
        b = plan.Builder("cpu", torch.bfloat16, record=True, splitk_ws_bytes=1 << 20)
        eps = m.emit(b, torch.zeros(18, 32, 32, 16, dtype=torch.bfloat16), torch.zeros(18, dtype=torch.int64), [5, 5, 4, 4],
                     tail=(keep, [1, 1, 0, 0]) if tail else None)
        assert eps.shape == (18, 32, 32, 4)
        return b

    full, cut = build(False), build(True)
    names = [mm.name for mm in cut.meta]
    i0 = next(i for i, n in enumerate(names) if "mv_decoder.3" in n and "keep_views" in n)
    gathers = [mm.name.split("/")[-1] for op, mm in zip(cut.ops, cut.meta) if op.kind == L.OP_GATHER_ROWS]
    assert gathers == ["keep_views", "keep_views.residual", "eps.scatter"], gathers
    for i, (op, mm) in enumerate(zip(cut.ops, cut.meta)):
        if op.kind == L.OP_IGEMM and "time" not in mm.name:
            rows = op.u.igemm.n_img * op.u.igemm.h_out * op.u.igemm.w_out
            if i > i0:
                assert rows == 16 * 1024, (mm.name, rows)
            elif "up3" in mm.name or "mv_decoder.3" in mm.name:
                assert rows == 18 * 1024, (mm.name, rows)
    att = [op for op, mm in zip(cut.ops, cut.meta) if op.kind == L.OP_ATTENTION and "mv_decoder.3" in mm.name and "attn1_3d" in mm.name]
    assert len(att) == 1 and att[0].u.attention.max_q_len == 4 * 1024 and att[0].u.attention.n_seg == 4
    f_full, f_cut = sum(mm.flops for mm in full.meta), sum(mm.flops for mm in cut.meta)
    assert 0.985 < f_cut / f_full < 0.999
