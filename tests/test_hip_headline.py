"""GPU parity at the HEADLINE shape (BASELINE.json configs[1] / configs[2]): the full-width SD-2.1 topology
(320/640/1280/1280, 1.07 B parameters), 32x32 latents, the production plan -- ONE UNet forward over the
fused CFG view groups ([5, 4] for 1 context + 4 targets; [5, 3] for 2 contexts + 3 targets) followed by the
fused CFG + DDIM kernel -- against `oracle.pipeline.step` (the restated DiffusionWrapper.step,
diffusion_wrapper.py:413-453) on the same seeded weights and inputs.

Tolerances (relative L2 of the updated latents x_{t-1} against the fp32 CPU oracle), one DDIM step:
    f32   1e-3  (the north-star tolerance; measured ~1e-5)
    f16   1e-2
    bf16  3e-2
and, through all 50 steps against the f32 HIP path (which the step test pins to the oracle):
    f16   2e-3   (measured 7.2e-4: inside the 1e-3 north-star tolerance)
    bf16  1.5e-2 (measured 5.8e-3; the bench dtype of BASELINE.json configs[1])
"""
import json
import os

import pytest
import torch

from conftest import record_err, rel_err
from seeded import random_cameras, seeded_state

pytestmark = pytest.mark.gpu
GRAD_ENABLED = False      # tests/conftest.py::_grad_mode: no autograd graphs in this module

# f32: the north-star tolerance (measured 9e-7); 16-bit: 2 x the maximum measured on MI355X (tests/golden/measured_errors_r04.json: 3.9e-4 / 3.1e-3)
STEP_TOL = {torch.float32: 1e-3, torch.float16: 8e-4, torch.bfloat16: 6.5e-3}
# 50 DDIM steps, CFG 3.0, random-init weights: the end-to-end drift of the 16-bit paths against the f32 HIP path.
DRIFT_TOL = {torch.float16: 2e-3, torch.bfloat16: 1.5e-2}      # measured on MI355X: 7.2e-4 / 5.8e-3 (profiles/r02_drift.json)
CASES = {"ctx1_tgt4": (1, 4), "ctx2_tgt3": (2, 3)}


@pytest.fixture(scope="module")
def models():
    """the HIP denoiser and the CPU oracle with the same seeded weights"""
    import mv_ldm_amd
    from mv_ldm_amd import _lib
    from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg
    from oracle import multiview as OMV
    _lib.load()
    o = OMV.MultiViewUNet(OMV.MVUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1"), 11, 4).eval()
    sd = seeded_state(o, 21)
    o.load_state_dict(sd)
    m = MultiViewUNet(MultiViewUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1", allow_random_init=True), 11, 4)
    m.load_state_dict(sd)
    del sd
    return mv_ldm_amd, m.cuda(), o


def _inputs(v_c, v_t, b=1, hl=32, seed=0):
    g = torch.Generator().manual_seed(100 + seed)
    ctx_lat = torch.randn(b, v_c, 4, hl, hl, generator=g) * 0.8
    x_t = torch.randn(b, v_t, 4, hl, hl, generator=g)
    extr, intr = random_cameras(b, v_c + v_t, seed=30 + seed)
    return ctx_lat, x_t, extr, intr


@pytest.fixture(scope="module")
def oracle_steps(models):
    """one oracle DDIM step (conditional V = v_c + v_t and unconditional V = v_t forward, CFG 3.0, t = 960) per case"""
    from oracle import pipeline as OPL
    from oracle.scheduler import DDIMScheduler as ODDIM
    _, _, o = models
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    out = {}
    for name, (v_c, v_t) in CASES.items():
        ctx_lat, x_t, extr, intr = _inputs(v_c, v_t)
        sch = ODDIM(clip_sample=False)
        sch.set_timesteps(50)
        rays = OPL.ray_encode(extr[:, :v_c], intr[:, :v_c], extr[:, v_c:], intr[:, v_c:], 32, 32)
        ctx_in = torch.cat([ctx_lat, torch.zeros_like(ctx_lat[:, :, :1])], dim=2)
        out[name] = OPL.step(o, sch, x_t, sch.timesteps[1], ctx_in, rays, torch.ones_like(x_t[:, :, :1]), use_cfg=True, cfg_scale=3.0)
    return out


def _pipe(m, n_steps=50):
    from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
    from mv_ldm_amd.scheduler import DDIMScheduler
    pipe = MVLDMPipeline(m, None, DDIMScheduler(clip_sample=False), SamplerCfg(True, 3.0, n_steps))
    pipe.set_timesteps(n_steps)
    return pipe


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_fused_cfg_step_at_headline_shape_vs_oracle(models, oracle_steps, dtype, case):
    M, m, _ = models
    v_c, v_t = CASES[case]
    ctx_lat, x_t, extr, intr = _inputs(v_c, v_t)
    pipe = _pipe(m)
    with M.compute_dtype(dtype):
        st = pipe._compile(1, v_c, v_t, 32, 32, dtype, 50)
        pipe.load_inputs(st, ctx_lat, x_t, (extr[:, :v_c], intr[:, :v_c]), (extr[:, v_c:], intr[:, v_c:]))
        # the plan's step counter starts at 0 (t = 980); the oracle step was taken at timesteps[1] = 960: advance once
        # WITHOUT touching the state (the advance kernel only moves the counter and the timestep rows)
        from mv_ldm_amd import ops as O
        from mv_ldm_amd import _lib as L
        L.check(L.load().mvldm_ddim_advance(st["step_ptr"].data_ptr(), st["t_table"].data_ptr(), 50, st["timesteps"].data_ptr(),
                                            st["tgt_rows"].data_ptr(), st["tgt_rows"].numel(), O.stream()))
        st["plan"].replay()
        got = pipe._read_state(st, 1, v_t)
    torch.cuda.synchronize()
    assert int(st["step_ptr"].item()) == 2
    e = record_err(f"headline_step/{str(dtype)[6:]}", rel_err(got.cpu(), oracle_steps[case]))
    print(f"headline step [{case}, {dtype}]: rel-err {e:.3e} (tol {STEP_TOL[dtype]})")
    assert e < STEP_TOL[dtype], (case, dtype, e)
    pipe._plans.clear()


def test_50_step_drift_of_the_16_bit_paths_vs_f32(models):
    """configs[1] for one scene: the full 50-step CFG sampler in f32 (pinned to the oracle by the step test above and
    by the G5 goldens), bf16 (the bench dtype) and f16 (the reference's own 16-mixed) from the same x_T.  The measured
    end-to-end relative errors are written to gpurun_out/r02_drift.json (committed under profiles/)."""
    M, m, _ = models
    v_c, v_t = 1, 4
    out = {}
    lat = {}
    for b in (1, 2):
        ctx_lat, x_T, extr, intr = _inputs(v_c, v_t, b=b, seed=7)
        for dtype in (torch.float32, torch.bfloat16, torch.float16):
            pipe = _pipe(m)
            with M.compute_dtype(dtype):
                x0 = pipe.denoise(ctx_lat, x_T, (extr[:, :v_c], intr[:, :v_c]), (extr[:, v_c:], intr[:, v_c:]))
            lat[(b, dtype)] = x0.cpu()
            assert torch.isfinite(x0).all()
            pipe._plans.clear()
        for dtype in (torch.bfloat16, torch.float16):
            out[f"b{b}_{str(dtype).split('.')[-1]}_vs_f32_50step_latent_rel_err"] = rel_err(lat[(b, dtype)], lat[(b, torch.float32)])
    # batch-position independence of the f32 path: scene 0 of the 2-scene run equals the 1-scene run to round-off
    print("50-step drift:", json.dumps(out))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r02_drift.json", "w") as f:
        json.dump({"workload": "configs[1]: 1 ctx + 4 tgt @ 32x32 latents, 50 DDIM steps, CFG 3.0, seeded random-init SD-2.1-width weights",
                   "reference": "f32 HIP path (exact-f32 MFMA), itself within 1e-3 of the CPU oracle per step", **out}, f, indent=1)
    for b in (1, 2):
        assert out[f"b{b}_float16_vs_f32_50step_latent_rel_err"] < DRIFT_TOL[torch.float16], out
        assert out[f"b{b}_bfloat16_vs_f32_50step_latent_rel_err"] < DRIFT_TOL[torch.bfloat16], out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_rule_based_tiles_match_the_tuned_plan(models, monkeypatch, dtype):
    """plan-time tile selection (timed per box) only changes WHICH tile / split-K computes a product, never the product: the
    fused CFG step recorded with `MVLDM_AUTOTUNE=0` (the rules of `choose_config`) equals the tuned plan's up to the summation
    order of split-K slabs (f32: 1e-5; bf16: one 16-bit rounding per layer, 1e-2) -- the fallback configuration multi-rank runs
    use for bit-identical results across ranks is the same arithmetic."""
    M, m, _ = models
    from mv_ldm_amd import plan as P
    v_c, v_t, b = 1, 4, 4
    ctx_lat, x_t, extr, intr = _inputs(v_c, v_t, b=b, seed=11)
    outs, tiles = {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MVLDM_AUTOTUNE", mode)
        pipe = _pipe(m)
        with M.compute_dtype(dtype):
            st = pipe._compile(b, v_c, v_t, 32, 32, dtype, 50)
            pipe.load_inputs(st, ctx_lat, x_t, (extr[:, :v_c], intr[:, :v_c]), (extr[:, v_c:], intr[:, v_c:]))
            st["plan"].replay()
            outs[mode] = pipe._read_state(st, b, v_t).cpu()
        tiles[mode] = [t for t in st["plan"].tiles if t is not None]
        pipe._plans.clear()
    assert all(t in (0, 15) for t in tiles["0"])                    # rules only (15: the shape rule of the skinny weight-streaming kernel, plan.skinny_rule)
    if dtype != torch.float32:
        assert any(t != 0 for t in tiles["1"]) or not P._TUNE_CACHE  # the tuned plan froze at least one tile (f32 is never tuned)
    e = record_err(f"tuned_vs_rules/{str(dtype)[6:]}", rel_err(outs["1"], outs["0"]))
    print(f"tuned vs rule-based plan [{dtype}]: rel-err {e:.3e}; frozen tiles {sorted(set(tiles['1']))}")
    assert torch.isfinite(outs["0"]).all() and e < (1e-5 if dtype == torch.float32 else 1e-2), e


def test_fused_step_with_the_wide_halo_tile_pinned(models, monkeypatch):
    """`MVLDM_IGEMM_TILE=17` sends EVERY 16-bit conv / Linear of the fused CFG step to tile 17: the one-source 3x3 convs of the 16x16 and
    smaller levels run the wide pixel-halo kernel, everything else (1x1, Linears, 32x32 maps, two sources, phase convs) must fall back
    to a streaming tile with the same values -- the step equals the plan pinned to tile 7 bit for bit (same K order, one K pass in both)"""
    M, m, _ = models
    v_c, v_t, b = 1, 4, 2
    ctx_lat, x_t, extr, intr = _inputs(v_c, v_t, b=b, seed=17)
    dtype = torch.bfloat16
    outs = {}
    for tile in ("17", "7"):
        monkeypatch.setenv("MVLDM_IGEMM_TILE", tile)
        pipe = _pipe(m)
        with M.compute_dtype(dtype):
            st = pipe._compile(b, v_c, v_t, 32, 32, dtype, 50)
            pipe.load_inputs(st, ctx_lat, x_t, (extr[:, :v_c], intr[:, :v_c]), (extr[:, v_c:], intr[:, v_c:]))
            st["plan"].replay()
            outs[tile] = pipe._read_state(st, b, v_t).cpu()
        if tile == "17":
            assert any(t == 17 for t in st["plan"].tiles if t is not None)
        pipe._plans.clear()
    assert torch.isfinite(outs["17"]).all() and torch.equal(outs["17"], outs["7"])


def test_cold_cache_tuning_path(models, monkeypatch):
    """the product times launches of <= 9216 rows behind a cache-flushing fill (plan._TUNE_COLD; the suite default is hot trials, see
    conftest): the one-scene plan recorded that way must equal the rule-based plan like any tuned plan does, and the fill must really
    have run (the fill counter moved) and its 640 MB buffer must be gone again once the plan is tuned"""
    M, m, _ = models
    from mv_ldm_amd import plan as P
    v_c, v_t, b = 1, 4, 1
    ctx_lat, x_t, extr, intr = _inputs(v_c, v_t, b=b, seed=13)
    dtype = torch.bfloat16
    outs = {}
    saved = dict(P._TUNE_CACHE)
    for mode in ("cold", "rules"):
        monkeypatch.setenv("MVLDM_AUTOTUNE", "1" if mode == "cold" else "0")
        monkeypatch.setattr(P, "_TUNE_COLD", 9216 if mode == "cold" else 0)
        if mode == "cold":
            P._TUNE_CACHE.clear()                      # every problem of this plan is timed again, cold
            P._THRASH.clear()
            fills0 = P._THRASH_FILLS[0]
        pipe = _pipe(m)
        with M.compute_dtype(dtype):
            st = pipe._compile(b, v_c, v_t, 32, 32, dtype, 50)
            pipe.load_inputs(st, ctx_lat, x_t, (extr[:, :v_c], intr[:, :v_c]), (extr[:, v_c:], intr[:, v_c:]))
            st["plan"].replay()
            outs[mode] = pipe._read_state(st, b, v_t).cpu()
        if mode == "cold":
            assert P._THRASH_FILLS[0] > fills0 and not P._THRASH      # the fill ran, and its 640 MB buffer is gone again
        pipe._plans.clear()
    P._TUNE_CACHE.clear()
    P._TUNE_CACHE.update(saved)
    e = rel_err(outs["cold"], outs["rules"])
    assert torch.isfinite(outs["cold"]).all() and e < 1e-2, e


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_shared_cfg_prefix_equals_the_two_pass_walk(models, monkeypatch, dtype):
    """the unconditional pass of classifier-free guidance re-submits the target views of the conditional pass; the layers in front
    of the first multi-view block work per image, so the fused plan evaluates them once (`MultiViewUNet.emit(dup=...)`) and copies
    the feature maps: x_{t-1} must equal the plan that walks all images (`MVLDM_CFG_SHARE=0`) -- f32 to accumulation-order
    round-off of a differently tiled GEMM, bf16 to one rounding per layer -- and the oracle (the step test above runs with it on)"""
    M, m, _ = models
    v_c, v_t, b = 1, 4, 2
    ctx_lat, x_t, extr, intr = _inputs(v_c, v_t, b=b, seed=13)
    outs, n_ops = {}, {}
    # 1: targets once per step + context once per sample; 1a: conditional images once per step; 0: all.  "+m": with the first multi-view
    # block's shared attention scores (the default), else MVLDM_CFG_SHARE_ATTN=0
    for mode in ("1+m", "1", "1a", "0"):
        monkeypatch.setenv("MVLDM_CFG_SHARE", mode.rstrip("+m"))
        monkeypatch.setenv("MVLDM_CFG_SHARE_ATTN", "1" if mode.endswith("+m") else "0")
        monkeypatch.setenv("MVLDM_AUTOTUNE", "0")
        # ONE tile and ONE K pass for every implicit GEMM of all plans: the shared layers see 10 (or 8) images instead of 18, the size
        # rules would pick other tiles / split counts and the K sums would round differently; pinned, a row's dot product is the same
        # instruction sequence whatever the row count
        monkeypatch.setenv("MVLDM_IGEMM_TILE", "2")
        pipe = _pipe(m)
        with M.compute_dtype(dtype):
            st = pipe._compile(b, v_c, v_t, 32, 32, dtype, 50)
            pipe.load_inputs(st, ctx_lat, x_t, (extr[:, :v_c], intr[:, :v_c]), (extr[:, v_c:], intr[:, v_c:]))
            st["plan"].replay()
            st["plan"].replay()
            outs[mode] = pipe._read_state(st, b, v_t).cpu()
        names = [mm.name for mm in st["plan"].meta]
        n_ops[mode] = sum("cfg_share" in n for n in names)
        pipe._plans.clear()
    assert n_ops == {"1+m": 6, "1": 6, "1a": 3, "0": 0}, n_ops      # conv_in + the two level-0 skips are gathered, nothing else is copied
    e_gemm = record_err(f"cfg_share_vs_full_walk_pinned/{str(dtype)[6:]}", max(rel_err(outs["1"], outs["0"]), rel_err(outs["1a"], outs["0"])))
    e = record_err(f"cfg_share_vs_full_walk/{str(dtype)[6:]}", rel_err(outs["1+m"], outs["0"]))
    print(f"shared CFG prefix vs full walk [{dtype}], two DDIM steps: rel-err {e_gemm:.3e} (pinned tiles, separate attention), {e:.3e} (merged attention)")
    # The shared LAYERS are the same arithmetic on the same values: with tile and split pinned the bf16 plans are bit-identical.
    # The shared attention SCORES are a different evaluation order of the softmax (two partial softmaxes combined by their log-sum-exp, each
    # partial output rounded to bf16): 2 x the measured 4.6e-3 after two DDIM steps
    assert all(torch.isfinite(o).all() for o in outs.values())
    assert e_gemm < (2e-6 if dtype == torch.float32 else 1e-5), e_gemm           # (bf16: measured 0.0 -- bit-identical)
    assert e < (2e-6 if dtype == torch.float32 else 9.5e-3), e


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_tail_drop_equals_the_full_walk(models, monkeypatch, dtype):
    """the context views' eps is never read by the sampler: the last multi-view block skips their queries in its 3-D attention and
    runs the rest of the block + the output stage on the target views only (`MultiViewUNet.emit(tail=...)`).  The target views'
    x_{t-1} must equal the plan that computes every view (`MVLDM_TAIL_DROP=0`)."""
    M, m, _ = models
    v_c, v_t, b = 1, 4, 2
    ctx_lat, x_t, extr, intr = _inputs(v_c, v_t, b=b, seed=17)
    outs, n_ops = {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MVLDM_TAIL_DROP", mode)
        monkeypatch.setenv("MVLDM_AUTOTUNE", "0")
        pipe = _pipe(m)
        with M.compute_dtype(dtype):
            st = pipe._compile(b, v_c, v_t, 32, 32, dtype, 50)
            pipe.load_inputs(st, ctx_lat, x_t, (extr[:, :v_c], intr[:, :v_c]), (extr[:, v_c:], intr[:, v_c:]))
            st["plan"].replay()
            st["plan"].replay()
            outs[mode] = pipe._read_state(st, b, v_t).cpu()
        n_ops[mode] = sum("keep_views" in mm.name or "eps.scatter" in mm.name for mm in st["plan"].meta)
        pipe._plans.clear()
    assert n_ops == {"1": 3, "0": 0}, n_ops
    e = record_err(f"tail_drop_vs_full_walk/{str(dtype)[6:]}", rel_err(outs["1"], outs["0"]))
    print(f"tail drop vs full walk [{dtype}], two DDIM steps: rel-err {e:.3e}")
    # (rules only, same row counts in front of the drop: measured 0.0 in both dtypes)
    assert all(torch.isfinite(o).all() for o in outs.values()) and e < (2e-6 if dtype == torch.float32 else 1e-4), e


def test_plans_follow_weight_changes(models):
    """recorded plans hold pointers to PACKED copies of the weights: after `load_state_dict` (Lightning's
    load_from_checkpoint), an in-place copy or an optimizer step the next forward must use the new weights"""
    M, m, _ = models
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 2, 11, 8, 8, generator=g).cuda()
    t = torch.tensor([[0, 500]]).cuda()
    with M.compute_dtype(torch.bfloat16):
        y0 = m(x, t).clone()
        assert torch.equal(m(x, t), y0)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        new = {k: (v * 1.25 if k.endswith("conv_in.weight") else v) for k, v in sd.items()}
        m.load_state_dict(new)
        y1 = m(x, t).clone()
        assert not torch.equal(y1, y0)
        with torch.no_grad():
            m.unet.conv_out.bias.add_(0.5)          # what an optimizer step does
        y2 = m(x, t).clone()
        assert rel_err((y2 - y1).cpu(), torch.full_like(y1, 0.5).cpu()) < 1e-2
        m.load_state_dict(sd)
        assert torch.equal(m(x, t), y0)
