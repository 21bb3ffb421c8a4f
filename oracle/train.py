"""Oracle (test infrastructure): restatement of the reference's training step and of the optimizer recipe around it.

Follows
* `src/model/diffusion_wrapper.py:213-276` (`sample_indices`), `:324-411` (`training_step`: context-count draw, view
  split, relative-pose coin, VAE encode of all 5 views, noise / timesteps, `add_noise`, masks, 10 % context drop, forward,
  MSE on the target views),
* `:1092-1122` (`configure_optimizers`: `getattr(optim, name)(params, lr=lr, **kwargs)` + `LinearLR`),
  `config/experiment/baseline.yaml:62-73` (AdamW, lr 2e-5, LinearLR start_factor 5e-4 over 200 steps),
* `src/main.py:119-136` + `config/main.yaml:82-84` (Lightning: `gradient_clip_val=0.1` by norm, `accumulate_grad_batches=2`:
  the loss of each micro-batch is divided by 2 before backward).

Gradients come from torch.autograd on the CPU.  Every random draw of the reference is an explicit argument here.
PINNED: G9 (tests/golden/g9_training_step.npz) holds the loss and the gradients of the reference's own
`training_step` (imported in the build container) for the same inputs; tests/test_oracle_train.py compares.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import pipeline as PL


def sample_indices(ctx: dict, tgt: dict, index: int, second: int):
    """diffusion_wrapper.py:213-276 with its draw passed in (`second` = rel_index if index > 1 else the kept context view)"""
    v_c = ctx["image"].shape[1]
    if index > 1:
        keys = ("image", "extrinsics", "intrinsics")
        return {k: ctx[k][:, :index] for k in keys}, {k: tgt[k] for k in keys}, second
    mask = torch.zeros(v_c, dtype=torch.bool)
    mask[second] = True
    c = {k: ctx[k][:, mask] for k in ("image", "extrinsics", "intrinsics")}
    t = {k: torch.cat([tgt[k], ctx[k][:, ~mask]], dim=1) for k in ("image", "extrinsics", "intrinsics")}
    return c, t, second


def training_step(denoiser, vae, scheduler, batch: dict, *, index: int, second: int, relative_coin: bool, unconditional: bool,
                  noise: torch.Tensor, timesteps: torch.Tensor, encode_noise: torch.Tensor) -> torch.Tensor:
    """diffusion_wrapper.py:324-411.  `relative_coin` is the value `np.random.choice([False, True])` returned: the
    reference converts to RELATIVE poses when it is False (`if relative_pose == 0`, :347)."""
    ctx, tgt, rel_index = sample_indices(batch["context"], batch["target"], index, second)
    b, v_c = ctx["image"].shape[:2]
    v_t = tgt["image"].shape[1]
    extr = torch.cat([ctx["extrinsics"], tgt["extrinsics"]], dim=1)
    if not relative_coin:
        extr = PL.absolute_to_relative_camera(extr, rel_index).float()
    c_ext, t_ext = extr[:, :v_c], extr[:, v_c:]
    latents = PL.first_stage_encode(vae, torch.cat([ctx["image"], tgt["image"]], dim=1), noise=encode_noise)
    ctx_lat, tgt_lat = latents[:, :v_c], latents[:, v_c:]
    noisy = scheduler.add_noise(tgt_lat, noise, timesteps)
    hl, wl = tgt_lat.shape[-2:]
    rays = PL.ray_encode(c_ext, ctx["intrinsics"], t_ext, tgt["intrinsics"], hl, wl)
    tgt_in = torch.cat([noisy, torch.ones(b, v_t, 1, hl, wl)], dim=2)
    if unconditional:
        inputs = torch.cat([tgt_in, rays[:, v_c:]], dim=2)
        ts = timesteps[:, None].expand(b, v_t)
    else:
        ctx_in = torch.cat([ctx_lat, torch.zeros(b, v_c, 1, hl, wl)], dim=2)
        inputs = torch.cat([torch.cat([ctx_in, tgt_in], dim=1), rays], dim=2)
        ts = torch.cat([torch.zeros(b, v_c, dtype=torch.long), timesteps[:, None].expand(b, v_t)], dim=1)
    pred = denoiser.forward(inputs, ts)
    pred = pred if unconditional else pred[:, v_c:]
    return F.mse_loss(pred.float(), noise.float(), reduction="mean")


def make_optimizer(params, lr: float = 2.0e-5, start_factor: float = 5.0e-4, total_iters: int = 200, **kwargs):
    """configure_optimizers with the released values (baseline.yaml:62-73)"""
    opt = torch.optim.AdamW(params, lr=lr, **kwargs)
    return opt, torch.optim.lr_scheduler.LinearLR(opt, start_factor=start_factor, total_iters=total_iters)


def optimizer_step(params, opt, sched, clip: Optional[float] = 0.1) -> float:
    """what Lightning does once the accumulated gradients are complete: clip by global norm, step, advance the schedule"""
    params = [p for p in params if p.grad is not None]
    total = float(torch.nn.utils.clip_grad_norm_(params, clip)) if clip else float(torch.norm(torch.stack([p.grad.norm() for p in params])))
    opt.step()
    sched.step()
    return total
