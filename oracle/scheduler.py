"""Oracle (test infrastructure): diffusers==0.27.2 `DDIMScheduler` restated.

Used by the reference through `src/model/scheduler/__init__.py:19-40` (ctor kwargs from
`DDIMSchedulerCfg`, `src/model/scheduler/ddim.py:10-18`) and at
`src/model/diffusion_wrapper.py:198,370,417,451,474,486`.  The package is not in /root/reference:
parity for this file is unpinned by the reference; the integer timestep tables are checked against
closed forms and the fp32 tables against an independent fp64 recomputation (tests/test_scheduler.py).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch


class DDIMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", trained_betas=None, clip_sample: bool = True,
                 set_alpha_to_one: bool = True, steps_offset: int = 0, prediction_type: str = "epsilon",
                 clip_sample_range: float = 1.0, timestep_spacing: str = "leading"):
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        assert prediction_type == "epsilon" and timestep_spacing == "leading"
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, clip_sample=clip_sample,
                                      clip_sample_range=clip_sample_range, steps_offset=steps_offset,
                                      prediction_type=prediction_type, set_alpha_to_one=set_alpha_to_one)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    def step(self, model_output, timestep, sample, eta: float = 0.0):
        assert eta == 0.0
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_t = 1 - a_t
        x0 = (sample - beta_t ** 0.5 * model_output) / a_t ** 0.5
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        direction = (1 - a_prev) ** 0.5 * model_output
        prev = a_prev ** 0.5 * x0 + direction
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = (ac[timesteps] ** 0.5).flatten()
        so = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.ndim < original_samples.ndim:
            sa, so = sa.unsqueeze(-1), so.unsqueeze(-1)
        return sa * original_samples + so * noise


class DDPMScheduler(DDIMScheduler):
    """diffusers==0.27.2 `DDPMScheduler` restated for what `SCHEDULER["ddpm"]` (src/model/scheduler/__init__.py:19-22) can reach
    through `DiffusionWrapper.step` (diffusion_wrapper.py:451): epsilon prediction, `variance_type="fixed_small"`, leading
    spacing; tables / `set_timesteps` / `add_noise` as the DDIM class.  Published algorithm (Ho et al. 2020, eq. 7 + the
    posterior variance), operations in diffusers' order.  Parity unpinned by the reference (package absent, never stepped by the
    released config); cross-checked against an independent fp64 evaluation in tests/test_scheduler.py."""

    def __init__(self, *a, variance_type: str = "fixed_small", **kw):
        super().__init__(*a, **kw)
        assert variance_type == "fixed_small"
        self.one = torch.tensor(1.0)

    def previous_timestep(self, timestep):
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // n

    def _get_variance(self, t):
        prev_t = self.previous_timestep(t)
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        current_beta_t = 1 - alpha_prod_t / alpha_prod_t_prev
        variance = (1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * current_beta_t
        return torch.clamp(variance, min=1e-20)

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None):
        t = int(timestep)
        prev_t = self.previous_timestep(t)
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        current_alpha_t = alpha_prod_t / alpha_prod_t_prev
        current_beta_t = 1 - current_alpha_t
        pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
        if self.config.clip_sample:
            pred_original_sample = pred_original_sample.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        pred_original_sample_coeff = (alpha_prod_t_prev ** 0.5 * current_beta_t) / beta_prod_t
        current_sample_coeff = current_alpha_t ** 0.5 * beta_prod_t_prev / beta_prod_t
        pred_prev_sample = pred_original_sample_coeff * pred_original_sample + current_sample_coeff * sample
        variance = 0
        if t > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            variance = (self._get_variance(t) ** 0.5) * variance_noise
        pred_prev_sample = pred_prev_sample + variance
        return SimpleNamespace(prev_sample=pred_prev_sample, pred_original_sample=pred_original_sample)
