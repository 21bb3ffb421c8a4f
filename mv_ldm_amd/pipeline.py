"""Sampling harness: the counterpart of `DiffusionWrapper`'s inference methods
(src/model/diffusion_wrapper.py) on the HIP path.

  geometry      `sample_image_grid`, `get_world_rays` (src/geometry/projection.py:117-138,91-114),
                `absolute_to_relative_camera` (src/misc/camera_utils.py:7-25): the reference's host-side
                functions on 4x4 / 3x3 camera matrices, kept as the readable specification
                (`ray_encode_host`); the per-pixel ray grid the sampler consumes is evaluated by the HIP
                kernel `mvldm_ray_encode` (`ray_encode`), straight into the UNet input buffer
  MVLDMPipeline `first_stage_encode` (:278-287), `last_stage_decode` (:289-298), `ray_encode`
                (:301-322), `step` (:413-453), `sample` (:455-490)

`step()` is the literal reference sequence (two denoiser forwards, CFG compose, scheduler step) on
the drop-in module objects.  `sample()` is the production path: ONE recorded plan per shape holds the
conditional and the unconditional pass as a single UNet forward over `groups = [v_c+v_t]*b + [v_t]*b`
(weights are read once for both), the fused CFG+DDIM kernel and the step bookkeeping; it is captured
into a hipGraph and replayed N times with no Python or host synchronisation in the loop.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Optional

import torch

from . import ops
from .mvunet import MultiViewUNet
from .plan import Builder
from .runtime import compute_dtype, get_compute_dtype
from .scheduler import DDIMScheduler
from .vae import AutoencoderKL

VAE_SCALE = 0.18215  # diffusion_wrapper.py:283,293


# ------------------------------------------------------------------------------------------ geometry
def sample_image_grid(shape, device="cpu", dtype=torch.float32):
    """pixel centres in (0,1), last dim (x, y); integer (row, col) indices.  projection.py:117-138"""
    h, w = shape
    ys = ((torch.arange(h, device=device) + 0.5) / h).to(dtype)
    xs = ((torch.arange(w, device=device) + 0.5) / w).to(dtype)
    gx, gy = torch.meshgrid(xs, ys, indexing="xy")
    ii, jj = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    return torch.stack([gx, gy], dim=-1), torch.stack([ii, jj], dim=-1)


def get_world_rays(coordinates, extrinsics, intrinsics):
    """coordinates [..., 2]; extrinsics [..., 4, 4] camera-to-world; intrinsics [..., 3, 3] normalised.
    Returns (origins, directions) [..., 3].  projection.py:74-114"""
    pix = torch.cat([coordinates, torch.ones_like(coordinates[..., :1])], dim=-1)
    d = torch.einsum("...ij,...j->...i", intrinsics.inverse(), pix)
    d = d * torch.ones_like(coordinates[..., 0])[..., None]
    d = d / d.norm(dim=-1, keepdim=True)
    d = torch.cat([d, torch.zeros_like(d[..., :1])], dim=-1)
    d = torch.einsum("...ij,...j->...i", extrinsics, d)[..., :3]
    return extrinsics[..., :3, 3].broadcast_to(d.shape), d


def absolute_to_relative_camera(tform, index: int):
    """inv(T[index]) @ T.  camera_utils.py:7-25"""
    ref = tform[:, index:index + 1].expand(-1, tform.shape[1], -1, -1)
    return torch.linalg.inv(ref) @ tform


@dataclass
class RayEncodingCfg:
    """the ray-encoding switches of `ModelCfg` (diffusion_wrapper.py:98-127; config/main.yaml:26-34,
    config/experiment/baseline.yaml:46-52): released values = raw 3 + 3 channels"""
    use_ray_encoding: bool = False        # PositionalEncoding (src/model/encodings/positional_encoding.py) on origins / directions
    srt_ray_encoding: bool = False        # SRT RayEncoder (src/model/srt/layers.py:35-58)
    use_plucker: bool = False             # origins <- origins x directions
    num_origin_octaves: int = 15
    num_direction_octaves: int = 15

    @property
    def mode(self) -> int:
        from . import _lib as L
        return L.RAYS_SRT if self.srt_ray_encoding else (L.RAYS_POSITIONAL if self.use_ray_encoding else L.RAYS_RAW)

    @property
    def octaves(self):
        return (self.num_origin_octaves, self.num_direction_octaves) if (self.srt_ray_encoding or self.use_ray_encoding) else (0, 0)

    @property
    def channels(self) -> int:
        no, nd = self.octaves
        if self.srt_ray_encoding:
            return 2 * (3 * no + 3 * nd)
        if self.use_ray_encoding:
            return (6 * no if no > 0 else 3) + (6 * nd if nd > 0 else 3)
        return 6

    def kernel_args(self) -> dict:
        no, nd = self.octaves
        return dict(mode=self.mode, n_origin_octaves=no, n_dir_octaves=nd, plucker=self.use_plucker)

    def denoiser_in_channels(self, latent_channels: int = 4) -> int:
        """latent + ray channels + mask (diffusion_wrapper.py:98-127)"""
        return latent_channels + self.channels + 1


def ray_encode(ctx_extr, ctx_intr, tgt_extr, tgt_intr, hl: int, wl: int, device=None, cfg: Optional[RayEncodingCfg] = None):
    """[b, v_c+v_t, C, hl, wl] fp32 on the GPU by the HIP kernel: ray origins then directions, raw (C = 6: the released
    config) or through the positional / SRT encodings of `cfg`.  diffusion_wrapper.py:169-190,301-322"""
    cfg = cfg or RayEncodingCfg()
    dev = torch.device(device) if device is not None else (ctx_extr.device if ctx_extr.is_cuda else torch.device("cuda"))
    extr = torch.cat([ctx_extr, tgt_extr], dim=1).to(dev, torch.float32).contiguous()
    intr = torch.cat([ctx_intr, tgt_intr], dim=1).to(dev, torch.float32).contiguous()
    b, v = extr.shape[:2]
    return ops.ray_encode(extr.view(b * v, 4, 4), intr.view(b * v, 3, 3), hl, wl, **cfg.kernel_args()).view(b, v, cfg.channels, hl, wl)


def ray_encode_host(ctx_extr, ctx_intr, tgt_extr, tgt_intr, hl: int, wl: int):
    """the same tensor by the reference's own host functions (specification; CPU tests pin it to G3)"""
    def rays(extr, intr):
        xy, _ = sample_image_grid((hl, wl), device=extr.device, dtype=extr.dtype)
        return get_world_rays(xy.reshape(hl * wl, 2), extr[:, :, None], intr[:, :, None])
    oc, dc = rays(ctx_extr, ctx_intr)
    ot, dt_ = rays(tgt_extr, tgt_intr)
    enc = torch.cat([torch.cat([oc, ot], dim=1), torch.cat([dc, dt_], dim=1)], dim=-1)
    b, v = enc.shape[:2]
    return enc.reshape(b, v, hl, wl, 6).permute(0, 1, 4, 2, 3).contiguous()


# ------------------------------------------------------------------------------------------ pipeline
@dataclass
class SamplerCfg:
    use_cfg: bool = True       # config/main.yaml:30
    cfg_scale: float = 3.0     # config/main.yaml:31
    num_inference_steps: int = 50


class MVLDMPipeline:
    def __init__(self, denoiser: MultiViewUNet, autoencoder: AutoencoderKL, scheduler: DDIMScheduler,
                 cfg: Optional[SamplerCfg] = None, rays: Optional[RayEncodingCfg] = None):
        self.denoiser, self.autoencoder, self.scheduler = denoiser, autoencoder, scheduler
        self.cfg = cfg or SamplerCfg()
        self.rays = rays or RayEncodingCfg()
        need = self.rays.denoiser_in_channels(denoiser.out_channels)
        if denoiser.in_channels != need:
            raise ValueError(f"denoiser built for {denoiser.in_channels} input channels, the ray encoding needs {need} "
                             "(latent + rays + mask, diffusion_wrapper.py:98-127)")
        self._plans = {}

    @property
    def latent_downscale(self) -> int:
        return 2 ** (len(self.autoencoder.config.block_out_channels) - 1)

    @property
    def device(self):
        return next(self.denoiser.parameters()).device

    def set_timesteps(self, num: Optional[int] = None):
        self.scheduler.set_timesteps(self.cfg.num_inference_steps if num is None else num)

    # ---- VAE wrappers (diffusion_wrapper.py:278-298) -----------------------------------------------
    def first_stage_encode(self, images, noise=None, generator=None):
        """`inputs * 2 - 1` rides in the encoder plan's layout kernel, `latent_dist.sample() * 0.18215` is one HIP kernel"""
        b, v = images.shape[:2]
        x = images.reshape(b * v, *images.shape[2:]).to(self.device, torch.float32).contiguous()
        z = self.autoencoder.encode(x, pre_scale=2.0, pre_shift=-1.0).latent_dist.sample(generator=generator, noise=noise,
                                                                                         scale=VAE_SCALE)
        return z.reshape(b, v, *z.shape[1:])

    def last_stage_decode(self, latents):
        """`(1 / 0.18215) * latents` and `(image / 2 + 0.5).clamp(0, 1)` ride in the decoder plan's layout kernels"""
        b, v = latents.shape[:2]
        z = latents.reshape(b * v, *latents.shape[2:]).to(self.device, torch.float32).contiguous()
        img = self.autoencoder.decode(z, pre_scale=1 / VAE_SCALE, post_scale=0.5, post_shift=0.5, clamp01=True).sample
        return img.reshape(b, v, *img.shape[1:])

    # ---- the reference's step, literally (diffusion_wrapper.py:413-453) ----------------------------
    def step(self, model, x_t, ts, context_inputs, ray_encodings, target_mask, step_generator=None, step_noise=None):
        """`step_generator` / `step_noise`: the per-step variance noise of an ancestral (DDPM) scheduler; ignored by DDIM"""
        b, v_c = context_inputs.shape[:2]
        v_t = x_t.shape[1]
        dev = x_t.device
        x_in = self.scheduler.scale_model_input(x_t, ts)
        t_tgt = torch.as_tensor(ts).to(torch.long).reshape(1).expand(b).to(dev)
        timesteps = torch.cat([torch.zeros(b, v_c, dtype=torch.long, device=dev), t_tgt[:, None].expand(b, v_t)], dim=1)
        target_inputs = torch.cat([x_in, target_mask], dim=2)
        inputs = torch.cat([torch.cat([context_inputs, target_inputs], dim=1), ray_encodings], dim=2)
        pred_c = model.forward(inputs, timesteps)
        if self.cfg.use_cfg:
            inputs_u = torch.cat([target_inputs, ray_encodings[:, v_c:]], dim=2)
            pred_u = model.forward(inputs_u, t_tgt[:, None].expand(b, v_t))
            if self._ancestral():        # CFG compose + DDPM update (+ fresh noise): one fused HIP kernel
                return self.scheduler.step(pred_c[:, v_c:].contiguous(), ts, x_t, model_output_uncond=pred_u.contiguous(),
                                           cfg_scale=self.cfg.cfg_scale, generator=step_generator, variance_noise=step_noise).prev_sample
            # CFG compose + DDIM update: one fused HIP kernel
            return self._cfg_ddim(pred_c[:, v_c:].contiguous(), pred_u, x_t, ts)
        if self._ancestral():
            return self.scheduler.step(pred_c[:, v_c:].contiguous(), ts, x_t, generator=step_generator, variance_noise=step_noise).prev_sample
        return self.scheduler.step(pred_c[:, v_c:].contiguous(), ts, x_t).prev_sample

    def _ancestral(self) -> bool:
        from .scheduler import DDPMScheduler
        return isinstance(self.scheduler, DDPMScheduler)

    def _cfg_ddim(self, pred_c, pred_u, x_t, ts):
        dev = x_t.device
        coef = self.scheduler.step_coefficients(int(ts)).reshape(1, 4).to(dev)
        zero = torch.zeros(1, dtype=torch.int32, device=dev)
        eps = torch.stack([pred_c.float().reshape(-1), pred_u.float().reshape(-1)]).view(2, 1, -1, 1).contiguous()
        out = ops.ddim_cfg_step(eps, x_t.float().contiguous().view(1, 1, -1, 1), zero, torch.ones(1, dtype=torch.int32, device=dev),
                                self.cfg.cfg_scale, coef, zero, None, clip_range=self.scheduler.clip_range)
        return out.view(x_t.shape)

    # ---- production sampler ------------------------------------------------------------------------
    def _compile(self, b: int, v_c: int, v_t: int, hl: int, wl: int, dtype, n_steps: int):
        sch = self.scheduler
        key = (b, v_c, v_t, hl, wl, dtype, n_steps, self.cfg.use_cfg, self.cfg.cfg_scale, str(self.device),
               tuple(int(t) for t in sch.timesteps), sch.clip_range, tuple(sorted(self.rays.kernel_args().items())))
        st = self._plans.get(key)
        if st is not None and st["weights_version"] == self.denoiser.weights_version():
            return st
        dev, den = self.device, self.denoiser
        lc = den.out_channels
        use_cfg = self.cfg.use_cfg
        v = v_c + v_t
        n_cond, n_unc = b * v, (b * v_t if use_cfg else 0)
        n_img = n_cond + n_unc
        e = ops.epc(dtype)
        c_pad = (den.in_channels + e - 1) // e * e
        unet_in = torch.zeros(n_img, hl, wl, c_pad, dtype=dtype, device=dev)
        x_state = torch.zeros(b * v_t, hl, wl, lc, dtype=torch.float32, device=dev)
        eps = torch.zeros(n_img, hl, wl, lc, dtype=torch.float32, device=dev)
        timesteps = torch.zeros(n_img, dtype=torch.int64, device=dev)
        cond_img = torch.tensor([s * v + v_c + j for s in range(b) for j in range(v_t)], dtype=torch.int32, device=dev)
        unc_img = (torch.tensor([n_cond + s * v_t + j for s in range(b) for j in range(v_t)], dtype=torch.int32, device=dev)
                   if use_cfg else None)
        tgt_rows = cond_img if not use_cfg else torch.cat([cond_img, unc_img])
        t_table = sch.timesteps.to(dev, torch.int64).contiguous()
        coef = sch.coefficient_table().to(dev)
        step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
        groups = [v] * b + ([v_t] * b if use_cfg else [])
        ctx_rows = torch.tensor([s * v + j for s in range(b) for j in range(v_c)], dtype=torch.int32, device=dev)
        # Shared layers of the fused CFG forward (mvunet.MultiViewUNet.emit, `dup`): the unconditional images [n_cond, n_img) re-submit
        # the target views cond_img of the conditional pass, and the context views' inputs (latents, mask 0, rays, timestep 0) do not
        # change during sampling -- the layers in front of the first multi-view block run once per STEP on one copy of every target
        # view and once per SAMPLE on the context views (`const_plan`, run by load_inputs after the loader plan).
        # (v_c == 0, a context-free sample: both passes are the same images -- nothing to share, and the pairing of the first multi-view
        #  block assumes context keys; the plain two-group walk handles it)
        dup, const_plan = None, None
        if use_cfg and v_c > 0 and os.environ.get("MVLDM_CFG_SHARE", "1") != "0":
            dup = (n_cond, cond_img)
            if os.environ.get("MVLDM_CFG_SHARE", "1") != "1a":
                cb = Builder(dev, dtype, record=True)
                x_ctx = torch.zeros(b * v_c, hl, wl, c_pad, dtype=dtype, device=dev)
                t_ctx = torch.zeros(b * v_c, dtype=torch.int64, device=dev)          # diffusion_wrapper.py:419: context timestep 0
                cb.gather_rows(unet_in, x_ctx, src_index=ctx_rows, name="context rows")
                with cb.scope("unet_ctx"):
                    const_skips = den.emit(cb, x_ctx, t_ctx, [1] * (b * v_c), prefix_only=True)
                cb.keep.extend([x_ctx, t_ctx, *const_skips])
                const_plan = cb.finalize()
                dup = (n_cond, cond_img, ctx_rows, const_skips)
        bld = Builder(dev, dtype, record=True)
        with bld.scope("unet"):
            # (`tail`: only the target views' eps is read -- the last multi-view block and the output stage drop the context views)
            tail = (tgt_rows, [v_c] * b + ([0] * b if use_cfg else [])) if v_c > 0 else None
            den.emit(bld, unet_in, timesteps, groups, out=eps, dup=dup, tail=tail)
        bld.ddim_step(eps, x_state, x_state, cond_img, unc_img, self.cfg.cfg_scale, coef, step_ptr, unet_in,
                      clip_range=sch.clip_range)
        bld.ddim_advance(step_ptr, t_table, timesteps, tgt_rows)
        plan = bld.finalize()
        plan.capture()
        # ---- loader plan: everything `sample()` writes before the DDIM loop (diffusion_wrapper.py:476-481, 429-432):
        # [latent 0..3 | mask 4 | rays 5..10] of every UNet input row, the fp32 DDIM state, step counter / timesteps.
        # Padding channels, context-view masks and context-view timesteps are zero from the allocation above.
        ctx_lat = torch.zeros(b * v_c, lc, hl, wl, dtype=torch.float32, device=dev)
        x_T = torch.zeros(b * v_t, lc, hl, wl, dtype=torch.float32, device=dev)
        ones = torch.ones(b * v_t, 1, hl, wl, dtype=torch.float32, device=dev)
        extr = torch.zeros(n_img, 4, 4, dtype=torch.float32, device=dev)
        intr = torch.zeros(n_img, 3, 3, dtype=torch.float32, device=dev)
        minus_one = torch.full((1,), -1, dtype=torch.int32, device=dev)
        ld = Builder(dev, dtype, record=True)
        ld.memcpy(step_ptr, minus_one, name="step_ptr=-1")
        ld.ddim_advance(step_ptr, t_table, timesteps, tgt_rows, name="step 0 / timesteps[targets] = t_0")
        ld.nchw_to_nhwc(ctx_lat, unet_in, 0, img_map=ctx_rows, name="context latents")
        for rows in ((cond_img, unc_img) if use_cfg else (cond_img,)):
            ld.nchw_to_nhwc(x_T, unet_in, 0, img_map=rows, name="x_T")
            ld.nchw_to_nhwc(ones, unet_in, lc, img_map=rows, name="target mask")
        ld.ray_encode(extr, intr, hl, wl, unet_in, lc + 1, name="ray grid", **self.rays.kernel_args())
        ld.nchw_to_nhwc(x_T, x_state, 0, name="x_T -> fp32 state")
        loader = ld.finalize(autotune=False)
        st = dict(plan=plan, loader=loader, const_plan=const_plan, unet_in=unet_in, x_state=x_state, eps=eps, timesteps=timesteps, cond_img=cond_img,
                  unc_img=unc_img, tgt_rows=tgt_rows, t_table=t_table, step_ptr=step_ptr, n_cond=n_cond,
                  ctx_lat=ctx_lat, x_T=x_T, extr=extr, intr=intr, weights_version=den.weights_version())
        self._plans[key] = st
        return st

    def load_inputs(self, st, ctx_latents, x_T, ctx_cams, tgt_cams):
        """stage the per-sample inputs (context latents [b,v_c,c,h,w], x_T [b,v_t,c,h,w], cameras = (extrinsics
        [b,v,4,4], intrinsics [b,v,3,3])) into the plan's fixed buffers -- copies only -- and run the loader plan
        (HIP layout / ray kernels) that assembles the UNet input and re-arms the step counter."""
        b, v_t = x_T.shape[:2]
        st["ctx_lat"].copy_(ctx_latents.reshape(st["ctx_lat"].shape))
        st["x_T"].copy_(x_T.reshape(st["x_T"].shape))
        n_cond = st["n_cond"]
        for buf, ci, ti, k in ((st["extr"], ctx_cams[0], tgt_cams[0], 4), (st["intr"], ctx_cams[1], tgt_cams[1], 3)):
            allc = torch.cat([ci.reshape(b, -1, k, k), ti.reshape(b, -1, k, k)], dim=1).to(torch.float32)
            buf[:n_cond].copy_(allc.reshape(n_cond, k, k))
            if st["unc_img"] is not None:
                buf[n_cond:].copy_(ti.reshape(b * v_t, k, k))
        st["loader"].run()
        if st.get("const_plan") is not None:      # the context views' shared layers: once per sample
            st["const_plan"].run()

    def _scaled_noise(self, x_T):
        sigma = self.scheduler.init_noise_sigma        # 1.0 for DDIM (diffusion_wrapper.py:474)
        return x_T if sigma == 1.0 else x_T * sigma

    def _read_state(self, st, b, v_t):
        hl, wl, lc = st["x_state"].shape[1:]
        return ops.nhwc_to_nchw(st["x_state"]).view(b, v_t, lc, hl, wl)

    def denoise(self, ctx_latents, x_T, ctx_cams, tgt_cams, dtype=None):
        """the DDIM loop of `sample()` on latents: returns x_0 [b, v_t, c, hl, wl] fp32.  `*_cams` = (extrinsics, intrinsics)"""
        dtype = dtype or get_compute_dtype()
        b, v_c = ctx_latents.shape[:2]
        v_t, hl, wl = x_T.shape[1], x_T.shape[3], x_T.shape[4]
        n_steps = len(self.scheduler.timesteps)
        st = self._compile(b, v_c, v_t, hl, wl, dtype, n_steps)
        self.load_inputs(st, ctx_latents, self._scaled_noise(x_T), ctx_cams, tgt_cams)
        for _ in range(n_steps):
            st["plan"].replay()
        return self._read_state(st, b, v_t)

    def prepare(self, batch, x_T=None, encode_noise=None, dtype=None):
        """everything of `sample()` before the DDIM loop: encode the context views, draw x_T, evaluate the ray
        grid, record (or fetch) the plan and load its input buffers.  Returns the plan state at step 0."""
        dtype = dtype or get_compute_dtype()
        ctx, tgt = batch["context"], batch["target"]
        ctx_lat = self.first_stage_encode(ctx["image"], noise=encode_noise)
        b, v_c, c, hl, wl = ctx_lat.shape
        v_t = tgt["extrinsics"].shape[1]
        if x_T is None:
            x_T = torch.randn((b, v_t, c, hl, wl))
        st = self._compile(b, v_c, v_t, hl, wl, dtype, len(self.scheduler.timesteps))
        self.load_inputs(st, ctx_lat, self._scaled_noise(x_T), (ctx["extrinsics"], ctx["intrinsics"]),
                         (tgt["extrinsics"], tgt["intrinsics"]))
        return st

    def sample(self, batch, x_T=None, encode_noise=None, decode: bool = True, dtype=None, step_generator=None, step_noise=None):
        """diffusion_wrapper.py:455-490.  batch: {"context": {image [b,v_c,3,H,W], extrinsics, intrinsics},
        "target": {extrinsics [b,v_t,4,4], intrinsics}}.  `x_T` / `encode_noise`: explicit noise (the
        reference draws x_T on the CPU generator, :473)."""
        if self._ancestral():
            return self._sample_ancestral(batch, x_T, encode_noise, decode, dtype, step_generator, step_noise)
        st = self.prepare(batch, x_T, encode_noise, dtype)
        for _ in range(len(self.scheduler.timesteps)):
            st["plan"].replay()
        b = batch["context"]["image"].shape[0]
        v_t = batch["target"]["extrinsics"].shape[1]
        x0 = self._read_state(st, b, v_t)
        return (self.last_stage_decode(x0) if decode else None), x0

    def _sample_ancestral(self, batch, x_T, encode_noise, decode, dtype, step_generator, step_noise):
        """`sample()` with `SCHEDULER["ddpm"]`: the reference's loop as written -- not captured into a single graph: every step
        draws new noise.  `step_noise`: [n_steps, b, v_t, c, hl, wl] explicit draws."""
        return self.sample_literal(batch, x_T, encode_noise, decode, dtype, step_generator, step_noise)

    def sample_literal(self, batch, x_T=None, encode_noise=None, decode: bool = True, dtype=None, step_generator=None, step_noise=None):
        """The reference's `sample` AS WRITTEN (diffusion_wrapper.py:455-490): a Python loop over the timesteps calling `step`
        (:413-453) -- per step the two denoiser forwards (conditional over [v_c + v_t] views, unconditional over [v_t]; each a
        recorded plan behind `MultiViewUNet.forward`), then CFG compose + scheduler update in one HIP kernel.  This is what
        registering the denoiser / scheduler / autoencoder in the reference's registries (INTEGRATION.md level A) executes; the
        fused `sample()` is the level-B replacement of this loop.  Same result as `sample()` up to rounding (the fused sampler shares
        the CFG prefix exactly: tests/test_hip_model.py)."""
        dtype = dtype or get_compute_dtype()
        ctx, tgt = batch["context"], batch["target"]
        dev = self.device
        ctx_lat = self.first_stage_encode(ctx["image"], noise=encode_noise)
        b, v_c, c, hl, wl = ctx_lat.shape
        v_t = tgt["extrinsics"].shape[1]
        if x_T is None:
            x_T = torch.randn((b, v_t, c, hl, wl))
        x_t = self._scaled_noise(x_T).to(dev, torch.float32)
        rays = ray_encode(ctx["extrinsics"], ctx["intrinsics"], tgt["extrinsics"], tgt["intrinsics"], hl, wl, device=dev, cfg=self.rays)
        ctx_in = torch.cat([ctx_lat.to(dev, torch.float32), torch.zeros(b, v_c, 1, hl, wl, device=dev)], dim=2)
        mask = torch.ones(b, v_t, 1, hl, wl, device=dev)
        with compute_dtype(dtype):
            for i, t in enumerate(self.scheduler.timesteps):
                z = None if step_noise is None else step_noise[i]
                x_t = self.step(self.denoiser, x_t, t, ctx_in, rays, mask, step_generator=step_generator, step_noise=z)
        return (self.last_stage_decode(x_t) if decode else None), x_t


def install_fused_sampler(wrapper, *, num_inference_steps: Optional[int] = None):
    """One call on a constructed reference `DiffusionWrapper` (level A of INTEGRATION.md: its denoiser / scheduler / autoencoder came
    out of the registries as this package's classes) that makes `wrapper.sample(batch)` (`diffusion_wrapper.py:455-490`) run the FUSED
    sampler -- both CFG passes in one forward, the DDIM update inside the captured graph -- without editing `DiffusionWrapper`:

        wrapper = DiffusionWrapper(...)                      # unchanged reference code
        mv_ldm_amd.pipeline.install_fused_sampler(wrapper)   # once, after construction / checkpoint load

    `sample` keeps the reference's contract: it returns `(images [b, v_t, 3, H, W] in [0, 1], batch)`, reads `model_cfg.use_cfg`,
    `model_cfg.cfg_scale`, `model_cfg.use_ema_sampling` (then `wrapper.ema.module` is the denoiser) and the scheduler's current
    `timesteps` at every call, and draws x_T on the CPU generator like `:473`.  `wrapper.step` and everything else stay the reference's.
    Returns the pipeline object that now backs `sample` (its recorded plans live there; `wrapper._mvldm_pipeline`)."""
    mc = wrapper.model_cfg

    def _rays() -> RayEncodingCfg:
        # the wrapper's own ray-encoding switches (diffusion_wrapper.py:98-127 builds the encoders from them, :301-320 applies them)
        re_ = getattr(mc, "ray_encodings", None)
        d = RayEncodingCfg()
        return RayEncodingCfg(use_ray_encoding=bool(getattr(mc, "use_ray_encoding", False)),
                              srt_ray_encoding=bool(getattr(mc, "srt_ray_encoding", False)),
                              use_plucker=bool(getattr(mc, "use_plucker", False)),
                              num_origin_octaves=int(getattr(re_, "num_origin_octaves", d.num_origin_octaves)),
                              num_direction_octaves=int(getattr(re_, "num_direction_octaves", d.num_direction_octaves)))

    def _denoiser():
        if getattr(mc, "use_ema_sampling", False) and getattr(wrapper, "ema", None) is not None:
            return getattr(wrapper.ema, "module", wrapper.ema)
        return wrapper.denoiser

    if not isinstance(_denoiser(), MultiViewUNet):
        raise TypeError("install_fused_sampler: wrapper.denoiser is not mv_ldm_amd.mvunet.MultiViewUNet -- register the HIP classes in the "
                        "reference's DENOISER / SCHEDULER / AUTOENCODERS registries first (INTEGRATION.md level A)")
    pipe = MVLDMPipeline(_denoiser(), wrapper.autoencoder, wrapper.scheduler,
                         SamplerCfg(bool(mc.use_cfg), float(mc.cfg_scale), int(num_inference_steps or len(wrapper.scheduler.timesteps) or 50)),
                         rays=_rays())

    def sample(batch):
        den = _denoiser()
        if den is not pipe.denoiser:                       # EMA switched on / off since the last call
            pipe.denoiser = den
            pipe._plans.clear()
        rays = _rays()
        if rays != pipe.rays:                              # re-read like cfg_scale (a changed channel count fails in the constructor's check)
            need = rays.denoiser_in_channels(den.out_channels)
            if den.in_channels != need:
                raise ValueError(f"install_fused_sampler: the ray encoding now needs {need} denoiser input channels, the denoiser has {den.in_channels}")
            pipe.rays = rays
            pipe._plans.clear()
        cfg = (bool(mc.use_cfg), float(mc.cfg_scale))
        if cfg != (pipe.cfg.use_cfg, pipe.cfg.cfg_scale):
            pipe.cfg = SamplerCfg(cfg[0], cfg[1], pipe.cfg.num_inference_steps)
            pipe._plans.clear()
        images, _ = pipe.sample({"context": batch["context"], "target": batch["target"]})
        return images, batch

    wrapper.sample = sample
    wrapper._mvldm_pipeline = pipe
    return pipe
