"""Oracle (test infrastructure): restatement of the sampling harness around the denoiser.

Follows
* `src/geometry/projection.py:117-138` (`sample_image_grid`), `:74-88` (`unproject`),
  `:91-114` (`get_world_rays`), `src/misc/camera_utils.py:7-25` (`absolute_to_relative_camera`),
* `src/model/diffusion_wrapper.py:169-190` (`generate_image_rays`), `:301-322` (`ray_encode`, raw
  3+3 channels: `use_ray_encoding=False`, `srt_ray_encoding=False`, `use_plucker=False`),
* `:278-287` / `:289-298` (`first_stage_encode` / `last_stage_decode`, factor 0.18215),
* `:413-453` (`step`: context timestep 0, CFG compose, scheduler step),
* `:455-490` (`sample`).

PINNED: G3 (rays / relative poses) and G5 (step / sample) in tests/golden were produced by the
reference's own functions imported in the build container.
"""
from __future__ import annotations

from typing import Optional

import torch

VAE_SCALE = 0.18215  # hard-coded in the reference (diffusion_wrapper.py:283,293)


# ----------------------------------------------------------------------------- geometry
def sample_image_grid(h: int, w: int, dtype=torch.float32):
    """Pixel centres in (0,1), last dim ordered (x, y).  projection.py:117-138."""
    ys = (torch.arange(h) + 0.5) / h
    xs = (torch.arange(w) + 0.5) / w
    gx, gy = torch.meshgrid(xs.to(dtype), ys.to(dtype), indexing="xy")
    return torch.stack([gx, gy], dim=-1)  # [h, w, 2]


def get_world_rays(xy, extrinsics, intrinsics):
    """xy [..., 2]; extrinsics [..., 4, 4] (camera-to-world); intrinsics [..., 3, 3] (normalised).
    projection.py:74-114."""
    ones = torch.ones_like(xy[..., :1])
    pix = torch.cat([xy, ones], dim=-1)
    d = torch.einsum("...ij,...j->...i", intrinsics.inverse(), pix)
    d = d * torch.ones_like(xy[..., 0])[..., None]          # z = 1
    d = d / d.norm(dim=-1, keepdim=True)
    d4 = torch.cat([d, torch.zeros_like(d[..., :1])], dim=-1)
    d_world = torch.einsum("...ij,...j->...i", extrinsics, d4)[..., :3]
    origins = extrinsics[..., :3, 3].broadcast_to(d_world.shape)
    return origins, d_world


def absolute_to_relative_camera(tform, index: int):
    """camera_utils.py:7-25: inv(T[index]) @ T."""
    ref = tform[:, index:index + 1].expand(-1, tform.shape[1], -1, -1)
    return torch.linalg.inv(ref) @ tform


def image_rays(h: int, w: int, extrinsics, intrinsics):
    """diffusion_wrapper.py:169-190 -> origins, directions  [b, v, h*w, 3]."""
    xy = sample_image_grid(h, w, extrinsics.dtype).reshape(h * w, 2)
    return get_world_rays(xy, extrinsics[:, :, None], intrinsics[:, :, None])


def ray_encode(ctx_extr, ctx_intr, tgt_extr, tgt_intr, h: int, w: int):
    """diffusion_wrapper.py:301-322 with raw rays -> [b, v_c+v_t, 6, h, w] (origins then directions)."""
    oc, dc = image_rays(h, w, ctx_extr, ctx_intr)
    ot, dt = image_rays(h, w, tgt_extr, tgt_intr)
    enc = torch.cat([torch.cat([oc, ot], dim=1), torch.cat([dc, dt], dim=1)], dim=-1)  # [b, v, hw, 6]
    b, v = enc.shape[:2]
    return enc.reshape(b, v, h, w, 6).permute(0, 1, 4, 2, 3).contiguous()


# ----------------------------------------------------------------------------- VAE wrappers
def first_stage_encode(vae, images, noise=None, generator=None):
    """images [b, v, 3, H, W] in [0,1] -> latents [b, v, 4, H/8, W/8].  diffusion_wrapper.py:278-287."""
    b, v = images.shape[:2]
    x = images.reshape(b * v, *images.shape[2:]) * 2.0 - 1.0
    with torch.no_grad():
        z = vae.encode(x).latent_dist.sample(generator=generator, noise=noise) * VAE_SCALE
    return z.reshape(b, v, *z.shape[1:])


def last_stage_decode(vae, latents):
    """diffusion_wrapper.py:289-298."""
    b, v = latents.shape[:2]
    z = (1 / VAE_SCALE) * latents.reshape(b * v, *latents.shape[2:])
    with torch.no_grad():
        img = vae.decode(z).sample
    img = img.reshape(b, v, *img.shape[1:])
    return (img / 2 + 0.5).clamp(0, 1)


# ----------------------------------------------------------------------------- step / sample
def step(model, scheduler, x_t, ts, context_inputs, ray_encodings, target_mask,
         use_cfg: bool = True, cfg_scale: float = 3.0):
    """diffusion_wrapper.py:413-453.  `context_inputs` = [ctx latents | ctx mask] ([b, v_c, 5, h, w])."""
    b, v_c = context_inputs.shape[:2]
    v_t = x_t.shape[1]
    x_in = scheduler.scale_model_input(x_t, ts)
    t_tgt = torch.as_tensor(ts).to(torch.long).reshape(1).expand(b)
    timesteps = torch.cat([torch.zeros(b, v_c, dtype=torch.long), t_tgt[:, None].expand(b, v_t)], dim=1)
    target_inputs = torch.cat([x_in, target_mask], dim=2)
    inputs = torch.cat([torch.cat([context_inputs, target_inputs], dim=1), ray_encodings], dim=2)
    pred_c = model.forward(inputs, timesteps)
    if use_cfg:
        inputs_u = torch.cat([target_inputs, ray_encodings[:, v_c:]], dim=2)
        pred_u = model.forward(inputs_u, t_tgt[:, None].expand(b, v_t))
        pred = pred_u + cfg_scale * (pred_c[:, v_c:] - pred_u)
    else:
        pred = pred_c[:, v_c:]
    return scheduler.step(pred, ts, x_t).prev_sample


def sample(model, vae, scheduler, ctx_images, ctx_extr, ctx_intr, tgt_extr, tgt_intr,
           x_T: Optional[torch.Tensor] = None, encode_noise=None, use_cfg: bool = True, cfg_scale: float = 3.0,
           decode: bool = True):
    """diffusion_wrapper.py:455-490.  `x_T` (the CPU-generated initial noise, :473) and the VAE
    posterior noise are explicit inputs so that runs are reproducible across devices."""
    ctx_lat = first_stage_encode(vae, ctx_images, noise=encode_noise)
    b, v_c, c, hl, wl = ctx_lat.shape
    v_t = tgt_extr.shape[1]
    if x_T is None:
        x_T = torch.randn(b, v_t, c, hl, wl)
    x_t = x_T * scheduler.init_noise_sigma
    target_mask = torch.ones(b, v_t, 1, hl, wl, dtype=x_t.dtype)
    context_inputs = torch.cat([ctx_lat, torch.zeros(b, v_c, 1, hl, wl, dtype=ctx_lat.dtype)], dim=2)
    rays = ray_encode(ctx_extr, ctx_intr, tgt_extr, tgt_intr, hl, wl).to(x_t.dtype)
    for ts in scheduler.timesteps:
        x_t = step(model, scheduler, x_t, ts, context_inputs, rays, target_mask, use_cfg, cfg_scale)
    return (last_stage_decode(vae, x_t) if decode else None), x_t
