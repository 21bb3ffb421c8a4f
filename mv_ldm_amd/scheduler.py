"""DDIM scheduler with the diffusers `DDIMScheduler` surface the reference uses
(src/model/scheduler/__init__.py:19-40; src/model/diffusion_wrapper.py:198,370,417,451,474,486):
`set_timesteps`, `.timesteps`, `.init_noise_sigma`, `scale_model_input`, `step(...).prev_sample`,
`add_noise`.

Host logic only: the beta / alpha tables and the integer timestep grid are built with the same fp32
torch ops diffusers uses (bit-exact tables), the per-step coefficients are four
fp32 scalars, and the elementwise update runs in the fused HIP kernel `mvldm_ddim_cfg_step` (which
evaluates the reference's expression with separately rounded fp32 operations, bit-identically).
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import ops


@dataclass
class DDIMSchedulerCfg:
    """src/model/scheduler/ddim.py:10-18"""
    num_train_timesteps: int = 1000
    beta_start: float = 0.0001
    beta_end: float = 0.02
    beta_schedule: str = "linear"
    trained_betas: Optional[object] = None
    clip_sample: bool = True
    set_alpha_to_one: bool = True
    steps_offset: int = 0


@dataclass
class SchedulerCfg:
    """src/model/scheduler/__init__.py:11-17"""
    name: str = "ddim"
    num_train_timesteps: int = 1000
    num_inference_steps: int = 70       # config/model/scheduler/ddim.yaml, config/experiment/baseline.yaml:36
    pretrained_from: Optional[str] = None
    kwargs: Optional[DDIMSchedulerCfg] = None


class DDIMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", trained_betas=None, clip_sample: bool = True, set_alpha_to_one: bool = True,
                 steps_offset: int = 0, prediction_type: str = "epsilon", timestep_spacing: str = "leading",
                 clip_sample_range: float = 1.0):
        if trained_betas is not None:
            self.betas = torch.tensor(np.asarray(trained_betas), dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule}")
        if prediction_type != "epsilon" or timestep_spacing != "leading":
            raise NotImplementedError("only epsilon prediction with leading spacing is on the reference's path")
        # diffusers' default `clip_sample=True` clamps the predicted x0 to +-clip_sample_range inside the fused kernel
        # (the released config sets it False, config/model/scheduler/ddim.yaml:9)
        self.clip_range = float(clip_sample_range) if clip_sample else 0.0
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, clip_sample=clip_sample,
                                      clip_sample_range=clip_sample_range,
                                      steps_offset=steps_offset, prediction_type=prediction_type,
                                      set_alpha_to_one=set_alpha_to_one)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self._dev = {}

    @classmethod
    def from_pretrained(cls, path, subfolder: str = "scheduler"):
        """`SCHEDULER[name].from_pretrained(path, subfolder="scheduler")` (src/model/scheduler/__init__.py:37): reads
        `<path>/<subfolder>/scheduler_config.json` from a LOCAL diffusers snapshot (no hub access offline)."""
        import inspect
        import json
        import os
        f = os.path.join(str(path), subfolder, "scheduler_config.json")
        if not os.path.isfile(f):
            raise FileNotFoundError(f"{f}: scheduler.from_pretrained needs a local diffusers snapshot (no hub access); "
                                    "or pass kwargs like the released config (config/model/scheduler/ddim.yaml)")
        with open(f) as fh:
            cfg = json.load(fh)
        ok = set(inspect.signature(cls.__init__).parameters) - {"self"}
        return cls(**{k: v for k, v in cfg.items() if k in ok})

    # ---- host-side tables ------------------------------------------------------------------------
    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)       # CPU int64, like diffusers with device=None
        self._dev = {}

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step_coefficients(self, timestep: int) -> torch.Tensor:
        """fp32 [4] = sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev)  (0-d torch ops, as diffusers)"""
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return torch.stack([(1 - a_t) ** 0.5, a_t ** 0.5, a_p ** 0.5, (1 - a_p) ** 0.5]).float()

    def coefficient_table(self) -> torch.Tensor:
        """fp32 [n_steps, 4] for the current timestep grid"""
        return torch.stack([self.step_coefficients(int(t)) for t in self.timesteps]).contiguous()

    # ---- device-side update ----------------------------------------------------------------------
    def step(self, model_output, timestep, sample, eta: float = 0.0, **_unused):
        """`sample`/`model_output`: any shape (treated as flat fp32); returns `.prev_sample`."""
        if eta != 0.0:
            raise NotImplementedError("eta != 0")
        if not sample.is_cuda:
            raise RuntimeError("DDIMScheduler.step runs the HIP kernel: tensors must be on the GPU (no CPU fallback)")
        dev = sample.device
        coef = self.step_coefficients(int(timestep)).reshape(1, 4).to(dev)
        step0 = self._dev.get(("zero", str(dev)))
        if step0 is None:
            step0 = torch.zeros(1, dtype=torch.int32, device=dev)
            self._dev[("zero", str(dev))] = step0
        x = sample.float().contiguous().view(1, 1, -1, 1)
        e = model_output.float().contiguous().view(1, 1, -1, 1)
        idx = self._dev.setdefault(("idx", str(dev)), torch.zeros(1, dtype=torch.int32, device=dev))
        out = ops.ddim_cfg_step(e, x, idx, None, 0.0, coef, step0, None, clip_range=self.clip_range)
        return SimpleNamespace(prev_sample=out.view(sample.shape))

    def add_noise(self, original_samples, noise, timesteps):
        """sqrt(a_t) x0 + sqrt(1-a_t) n  (training path; host tables, torch elementwise -- the
        training step is a "next" row)"""
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = (ac[timesteps] ** 0.5).flatten()
        so = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.ndim < original_samples.ndim:
            sa, so = sa.unsqueeze(-1), so.unsqueeze(-1)
        return sa * original_samples + so * noise


class DDPMScheduler(DDIMScheduler):
    """`SCHEDULER["ddpm"]` (src/model/scheduler/__init__.py:19-22): diffusers' `DDPMScheduler` for epsilon prediction with
    `variance_type="fixed_small"` and leading timestep spacing (its defaults; `clip_sample=True` by default, like diffusers).
    The beta tables, `set_timesteps` and `add_noise` are the DDIM class's (identical in diffusers); `step` is ancestral sampling:
        x0 = (x_t - sqrt(1-a_t) eps) / sqrt(a_t) [clipped];  x_prev = c0 x0 + c1 x_t + sqrt(var_t) z,  z ~ N(0, 1) for t > 0
        c0 = sqrt(a_prev) b_t / (1-a_t),  c1 = sqrt(alpha_t) (1-a_prev) / (1-a_t),  alpha_t = a_t / a_prev,  b_t = 1 - alpha_t,
        var_t = clamp((1-a_prev) / (1-a_t) b_t, 1e-20)
    with the scalars from 0-d fp32 torch ops (as diffusers computes them) and the elementwise update in the HIP kernel
    `mvldm_ddpm_cfg_step`.  The released config samples with DDIM (config/model/scheduler/ddim.yaml); SURVEY.md §2 #6."""

    def __init__(self, *a, variance_type: str = "fixed_small", **kw):
        super().__init__(*a, **kw)
        if variance_type != "fixed_small":
            raise NotImplementedError(f"variance_type {variance_type!r}: only diffusers' default 'fixed_small'")
        self.config.variance_type = variance_type
        self.one = torch.tensor(1.0)

    def previous_timestep(self, timestep: int) -> int:
        n = self.num_inference_steps or self.config.num_train_timesteps
        return int(timestep) - self.config.num_train_timesteps // n

    def step_coefficients(self, timestep: int) -> torch.Tensor:
        """fp32 [5] = sqrt(1-a_t), sqrt(a_t), c0, c1, sigma (sigma = 0 at t = 0: diffusers adds no noise there)"""
        t = int(timestep)
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_alpha = a_t / a_p
        cur_beta = 1 - cur_alpha
        c0 = (a_p ** 0.5 * cur_beta) / b_t
        c1 = cur_alpha ** 0.5 * b_p / b_t
        var = torch.clamp((1 - a_p) / (1 - a_t) * cur_beta, min=1e-20)
        sigma = var ** 0.5 if t > 0 else torch.tensor(0.0)
        return torch.stack([b_t ** 0.5, a_t ** 0.5, c0, c1, sigma]).float()

    def coefficient_table(self) -> torch.Tensor:
        return torch.stack([self.step_coefficients(int(t)) for t in self.timesteps]).contiguous()

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, model_output_uncond=None,
             cfg_scale: float = 0.0, **_unused):
        """`.prev_sample` of diffusers' `DDPMScheduler.step(model_output, timestep, sample, generator)`.  `variance_noise` (same
        shape) replaces the draw; otherwise z is drawn on the sample's device with `generator` / the global generator, as
        diffusers' `randn_tensor` does.  `model_output_uncond` + `cfg_scale`: fuse the CFG compose of diffusion_wrapper.py:444."""
        if not sample.is_cuda:
            raise RuntimeError("DDPMScheduler.step runs the HIP kernel: tensors must be on the GPU (no CPU fallback)")
        t = int(timestep)
        coef = self.step_coefficients(t).to(sample.device)
        z = None
        if t > 0:
            if variance_noise is None:
                gdev = generator.device if generator is not None else sample.device
                variance_noise = torch.randn(model_output.shape, generator=generator, device=gdev, dtype=torch.float32)
            z = variance_noise.to(sample.device, torch.float32).contiguous().view(-1)
        x = sample.float().contiguous().view(-1)
        e = model_output.float().contiguous().view(-1)
        eu = None if model_output_uncond is None else model_output_uncond.float().contiguous().view(-1)
        out = ops.ddpm_cfg_step(e, eu, x, z, float(cfg_scale), coef, clip_range=self.clip_range)
        return SimpleNamespace(prev_sample=out.view(sample.shape))


SCHEDULER = {"ddim": DDIMScheduler, "ddpm": DDPMScheduler}


def get_scheduler(cfg: SchedulerCfg) -> DDIMScheduler:
    """src/model/scheduler/__init__.py:30-40"""
    if cfg.pretrained_from is not None:
        return SCHEDULER[cfg.name].from_pretrained(cfg.pretrained_from, subfolder="scheduler")
    kw = cfg.kwargs if cfg.kwargs is not None else {}
    kw = asdict(kw) if hasattr(kw, "__dataclass_fields__") else dict(kw)
    return SCHEDULER[cfg.name](**kw)
