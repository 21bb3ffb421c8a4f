"""CPU: host-side logic of the product path that needs no GPU -- camera / ray geometry vs the reference's
own outputs (G3), the scheduler's integer and fp32 tables vs G6 (bit-exact) -- and the "fail loudly" rule."""
import numpy as np
import pytest
import torch

from conftest import rel_err


def test_rays_and_relative_poses_vs_reference(golden):
    from mv_ldm_amd.pipeline import absolute_to_relative_camera, get_world_rays, ray_encode_host as ray_encode, sample_image_grid
    g = golden("g3_rays")
    extr, intr = torch.from_numpy(g["extrinsics"]), torch.from_numpy(g["intrinsics"])
    for (h, w) in [(8, 8), (4, 6)]:
        xy, ij = sample_image_grid((h, w))
        assert torch.equal(xy, torch.from_numpy(g[f"xy_{h}x{w}"]))
        assert ij.shape == (h, w, 2) and ij[1, 2].tolist() == [1, 2]
        o, d = get_world_rays(xy.reshape(h * w, 2), extr[:, :, None], intr[:, :, None])
        assert rel_err(o, g[f"origins_{h}x{w}"]) < 1e-7 and rel_err(d, g[f"directions_{h}x{w}"]) < 1e-6
    enc = ray_encode(extr[:, :1], intr[:, :1], extr[:, 1:], intr[:, 1:], 8, 8)
    assert enc.shape == (2, 4, 6, 8, 8)
    assert rel_err(enc[:, :, 3:].permute(0, 1, 3, 4, 2).reshape(2, 4, 64, 3), g["directions_8x8"]) < 1e-6
    for idx in (0, 1, 3):
        assert rel_err(absolute_to_relative_camera(extr, idx), g[f"relative_{idx}"]) < 1e-6


def test_scheduler_tables_bit_exact(golden):
    from mv_ldm_amd.scheduler import DDIMScheduler, DDIMSchedulerCfg, SchedulerCfg, get_scheduler
    g = golden("g6_ddim")
    s = get_scheduler(SchedulerCfg(name="ddim", kwargs=DDIMSchedulerCfg(clip_sample=False)))
    assert isinstance(s, DDIMScheduler) and s.init_noise_sigma == 1.0
    assert np.array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])
    for n in (5, 25, 50, 70):
        s.set_timesteps(n)
        assert s.timesteps.dtype == torch.int64 and np.array_equal(s.timesteps.numpy(), g[f"timesteps_{n}"])
    s.set_timesteps(50)
    tab = s.coefficient_table()
    assert tab.shape == (50, 4) and tab.dtype == torch.float32
    ac = s.alphas_cumprod
    assert torch.equal(tab[0], torch.stack([(1 - ac[980]) ** 0.5, ac[980] ** 0.5, ac[960] ** 0.5, (1 - ac[960]) ** 0.5]))
    assert float(tab[-1, 2]) == 1.0 and float(tab[-1, 3]) == 0.0          # final step: alpha_prev = 1 (set_alpha_to_one)
    x, e = torch.from_numpy(g["kat_x"]), torch.from_numpy(g["kat_eps"])
    assert np.array_equal(s.add_noise(x, e, torch.tensor([10, 900])).numpy(), g["kat_add_noise"])
    # the mirrored dataclass defaults are usable as they are (diffusers' default clip_sample=True clamps x0 in the kernel)
    d = get_scheduler(SchedulerCfg())
    assert d.clip_range == 1.0 and DDIMScheduler(clip_sample=False).clip_range == 0.0 and SchedulerCfg().num_inference_steps == 70
    from mv_ldm_amd.scheduler import SCHEDULER
    assert set(SCHEDULER) == {"ddim", "ddpm"}                    # src/model/scheduler/__init__.py:19-22
    p = SCHEDULER["ddpm"](clip_sample=False)
    assert np.array_equal(p.add_noise(x, e, torch.tensor([10, 900])).numpy(), g["kat_add_noise"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):    # the ancestral step is a HIP kernel (tests/test_hip_ops.py)
        p.step(e, 10, x)
    with pytest.raises(NotImplementedError):
        SCHEDULER["ddpm"](variance_type="learned")


def test_scheduler_from_pretrained_reads_a_local_snapshot(tmp_path):
    """`SCHEDULER[name].from_pretrained(path, subfolder="scheduler")` (src/model/scheduler/__init__.py:37)"""
    import json
    from mv_ldm_amd.scheduler import DDIMScheduler, SchedulerCfg, get_scheduler
    (tmp_path / "scheduler").mkdir()
    (tmp_path / "scheduler" / "scheduler_config.json").write_text(json.dumps(
        {"_class_name": "DDIMScheduler", "beta_schedule": "scaled_linear", "beta_start": 0.00085, "beta_end": 0.012,
         "clip_sample": False, "set_alpha_to_one": False, "steps_offset": 1, "num_train_timesteps": 1000, "skip_prk_steps": True}))
    s = get_scheduler(SchedulerCfg(name="ddim", pretrained_from=str(tmp_path)))
    assert isinstance(s, DDIMScheduler) and s.config.steps_offset == 1 and s.clip_range == 0.0
    assert abs(float(s.betas[0]) - 0.00085) < 1e-9 and float(s.final_alpha_cumprod) == float(s.alphas_cumprod[0])
    with pytest.raises(FileNotFoundError):
        get_scheduler(SchedulerCfg(name="ddim", pretrained_from="stabilityai/stable-diffusion-2-1"))


def test_no_cpu_fallback():
    """the product path must fail loudly without a GPU, never compute on the CPU"""
    from mv_ldm_amd.modules import Conv2d, GroupNorm
    from mv_ldm_amd.scheduler import DDIMScheduler
    with pytest.raises(RuntimeError, match="no CPU fallback|only on a HIP device"):
        Conv2d(8, 8, 3, padding=1)(torch.zeros(1, 8, 4, 4))
    with pytest.raises(RuntimeError):
        GroupNorm(4, 8)(torch.zeros(1, 8, 4, 4))
    s = DDIMScheduler(clip_sample=False)
    s.set_timesteps(5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.step(torch.zeros(1, 4), 800, torch.zeros(1, 4))


def test_oracle_is_not_imported_by_the_product():
    import pathlib
    import re
    root = pathlib.Path(__file__).resolve().parent.parent / "mv_ldm_amd"
    for f in root.rglob("*.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle\b", f.read_text(), re.M), f


def test_scale_lr_multiplies_by_the_effective_batch_size_or_refuses():
    """diffusion_wrapper.py:157-166: `lr = effective_batch_size * lr if scale_lr else lr`"""
    import pytest
    import torch
    from torch import nn
    from mv_ldm_amd.train import DistributedOptimizer, FlatParams, OptimizerCfg
    flat = FlatParams(nn.Linear(4, 4))
    noop = lambda *a, **k: None
    with pytest.raises(ValueError, match="effective_batch_size"):
        DistributedOptimizer(flat, OptimizerCfg(lr=2e-5, scale_lr=True), update=noop, sumsq=noop, clip=noop)
    opt = DistributedOptimizer(flat, OptimizerCfg(lr=2e-5, scale_lr=True, scheduler=None), update=noop, sumsq=noop, clip=noop,
                               effective_batch_size=2 * 8 * 4)
    assert abs(opt.lr() - 2e-5 * 64) < 1e-12
    assert abs(DistributedOptimizer(flat, OptimizerCfg(lr=2e-5, scheduler=None), update=noop, sumsq=noop, clip=noop).lr() - 2e-5) < 1e-15


def test_weight_epochs_are_per_module():
    """an in-place optimizer step on one module must not invalidate the packs / plans of another (the frozen VAE)"""
    from torch import nn
    from mv_ldm_amd.modules import bump_weights_epoch, weights_version
    a, b = nn.Linear(3, 3), nn.Linear(3, 3)
    va, vb = weights_version(a), weights_version(b)
    bump_weights_epoch(a)
    assert weights_version(a) != va and weights_version(b) == vb
    bump_weights_epoch()            # no module: everything
    assert weights_version(b) != vb


def test_committed_traffic_table_answers_the_default_bench_workload():
    """`roofline.traffic` regressed to null in round 2 when the committed table lost its workload key: the default line's
    lookup (bf16, 64 scenes, 256x256) must find the igemm and the attention family"""
    import importlib.util
    import os
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = types.SimpleNamespace(dtype="bf16", res=256)
    ig, at = bench.pmc_traffic(args, 64), bench.pmc_traffic(args, 64, "attention_kernel")
    assert ig and ig > 1e8 and at and at > 1e8
    assert bench.pmc_traffic(args, 7) is None          # no pass for that workload: null, not a wrong number
