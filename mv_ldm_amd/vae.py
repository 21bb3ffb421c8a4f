"""`AutoencoderKL` (diffusers surface) on the HIP kernels: the SD-2.1 VAE the reference reaches through
`AUTOENCODERS` (src/model/autoencoder/__init__.py:15-43) and `first_stage_encode` /
`last_stage_decode` (src/model/diffusion_wrapper.py:278-298).  Topology: SURVEY.md App. A.8; state-dict
keys: App. A.9 (`encoder.*`, `decoder.*`, `quant_conv`, `post_quant_conv`,
`mid_block.attentions.0.{group_norm,to_q,to_k,to_v,to_out.0}`).

`decode(z).sample` / `encode(x).latent_dist` run the whole conv stack as one recorded plan per input
shape (same kernels as the UNet: implicit-GEMM 3x3 conv with fused nearest-upsample / asymmetric
stride-2, GroupNorm+SiLU, the single-head mid attention).
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Sequence

import torch
from torch import nn

from . import ops
import warnings

from .modules import Builder, Conv2d, Downsample2D, GroupNorm, ResnetBlock2D, UNetMidBlock2D, Upsample2D, weights_version
from .runtime import get_compute_dtype, require_gpu

SD21_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                       latent_channels=4, norm_num_groups=32, scaling_factor=0.18215)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, None, groups, 1e-6)
                                      for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, out_channels, padding=0)]) if add_downsample else None


class UpDecoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, groups, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, None, groups, 1e-6)
                                      for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, out_channels)]) if add_upsample else None


def _chain(b: Builder, h, mods, prefix, free_first=True):
    """run modules with .emit(b, h) in sequence, freeing intermediates"""
    first = True
    for name, m in mods:
        with b.scope(f"{prefix}.{name}"):
            nh = m.emit(b, h)
        if not first or free_first:
            b.free(h)
        first = False
        h = nh
    return h


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, boc: Sequence[int], layers_per_block, groups):
        super().__init__()
        self.conv_in = Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i in range(len(boc)):
            in_c, out_c = out_c, boc[i]
            self.down_blocks.append(DownEncoderBlock2D(in_c, out_c, layers_per_block, groups, i != len(boc) - 1))
        self.mid_block = UNetMidBlock2D(boc[-1], None, num_layers=1, resnet_eps=1e-6, resnet_groups=groups, add_attention=True)
        self.conv_norm_out = GroupNorm(groups, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(boc[-1], 2 * latent_channels, 3, padding=1)

    def emit(self, b: Builder, x):
        h = self.conv_in.emit(b, x, name="encoder.conv_in")
        for i, blk in enumerate(self.down_blocks):
            mods = [(f"resnets.{j}", r) for j, r in enumerate(blk.resnets)]
            if blk.downsamplers is not None:
                mods += [("downsamplers.0", blk.downsamplers[0])]
            h = _chain(b, h, mods, f"encoder.down_blocks.{i}")
        with b.scope("encoder.mid_block"):
            h2 = self.mid_block.emit(b, h)
        b.free(h)
        g = self.conv_norm_out.emit(b, h2, silu=True, name="encoder.conv_norm_out+silu")
        b.free(h2)
        out = self.conv_out.emit(b, g, name="encoder.conv_out")
        b.free(g)
        return out


class Decoder(nn.Module):
    def __init__(self, latent_channels, out_channels, boc: Sequence[int], layers_per_block, groups):
        super().__init__()
        self.conv_in = Conv2d(latent_channels, boc[-1], 3, padding=1)
        self.mid_block = UNetMidBlock2D(boc[-1], None, num_layers=1, resnet_eps=1e-6, resnet_groups=groups, add_attention=True)
        self.up_blocks = nn.ModuleList()
        rboc = list(reversed(boc))
        out_c = rboc[0]
        for i in range(len(rboc)):
            prev, out_c = out_c, rboc[i]
            self.up_blocks.append(UpDecoderBlock2D(prev, out_c, layers_per_block + 1, groups, i != len(rboc) - 1))
        self.conv_norm_out = GroupNorm(groups, boc[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(boc[0], out_channels, 3, padding=1)

    def emit(self, b: Builder, z, out_dtype=None):
        h = self.conv_in.emit(b, z, name="decoder.conv_in")
        with b.scope("decoder.mid_block"):
            h2 = self.mid_block.emit(b, h)
        b.free(h)
        h = h2
        for i, blk in enumerate(self.up_blocks):
            mods = [(f"resnets.{j}", r) for j, r in enumerate(blk.resnets)]
            if blk.upsamplers is not None:
                mods += [("upsamplers.0", blk.upsamplers[0])]
            h = _chain(b, h, mods, f"decoder.up_blocks.{i}")
        g = self.conv_norm_out.emit(b, h, silu=True, name="decoder.conv_norm_out+silu")
        b.free(h)
        out = self.conv_out.emit(b, g, out_dtype=out_dtype, name="decoder.conv_out")
        b.free(g)
        return out


class DiagonalGaussianDistribution:
    """diffusers `DiagonalGaussianDistribution` over the encoder's `moments` ([N, 2c, h, w] fp32 = mean | logvar).
    `sample()` is the HIP kernel `mvldm_posterior_sample` (clamp, exp, reparameterisation, optional scale); the RNG is
    the caller's (SURVEY.md §7 'RNG placement').  `.mean` / `.logvar` / `.std` are views / lazily evaluated accessors
    for inspection (tests), not used on the sampling path."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, self._raw_logvar = torch.chunk(parameters, 2, dim=1)

    @property
    def logvar(self):
        return torch.clamp(self._raw_logvar, -30.0, 20.0)

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, noise=None, scale: float = 1.0):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=torch.float32)
        noise = noise.to(self.mean.device, torch.float32).contiguous()
        return ops.posterior_sample(self.parameters.contiguous(), noise, scale)

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(64,), layers_per_block=1, latent_channels=4,
                 norm_num_groups=32, scaling_factor=0.18215, **_ignored):
        super().__init__()
        boc = tuple(block_out_channels)
        self.config = SimpleNamespace(scaling_factor=scaling_factor, latent_channels=latent_channels, block_out_channels=boc,
                                      in_channels=in_channels, out_channels=out_channels)
        self.encoder = Encoder(in_channels, latent_channels, boc, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, boc, layers_per_block, norm_num_groups)
        self.quant_conv = Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = Conv2d(latent_channels, latent_channels, 1)
        self._plans = {}

    @classmethod
    def from_pretrained(cls, path, subfolder="vae", config_overrides=None, state_dict=None, allow_random_init=False):
        """SD-2.1 VAE topology.  Weights: `state_dict` (diffusers key layout), or a LOCAL snapshot / checkpoint file at
        `path` (`<path>/<subfolder>/diffusion_pytorch_model.safetensors|.bin`, or a .safetensors/.ckpt file); there is no
        hub access offline, so anything else leaves torch's random init -- with a warning unless `allow_random_init`."""
        cfg = dict(SD21_VAE_CONFIG)
        cfg.update(config_overrides or {})
        m = cls(**cfg)
        if state_dict is not None:
            m.load_state_dict(state_dict)
            return m
        from .checkpoint import find_local_weights, load_vae_checkpoint
        f = find_local_weights(path, subfolder)
        if f is not None:
            load_vae_checkpoint(m, f)
        elif not allow_random_init:
            warnings.warn(f"AutoencoderKL.from_pretrained({path!r}): no local weights found and no state_dict given -- the "
                          "module keeps RANDOM initial weights (pass allow_random_init=True to silence)", stacklevel=2)
        return m

    # ---- plans -----------------------------------------------------------------------------------
    def _compile(self, kind: str, n: int, h: int, w: int, dtype, io=(1.0, 0.0, 1.0, 0.0, False)):
        """`io` = (pre_scale, pre_shift, post_scale, post_shift, clamp01): affine maps folded into the plan's layout kernels"""
        dev = next(self.parameters()).device
        key = (kind, n, h, w, dtype, str(dev), io)
        st = self._plans.get(key)
        if st is not None and st["weights_version"] == weights_version(self):
            return st
        pre_scale, pre_shift, post_scale, post_shift, clamp01 = io
        e = ops.epc(dtype)
        lc = self.config.latent_channels
        bld = Builder(dev, dtype, record=True)
        if kind == "decode":
            c_pad = (lc + e - 1) // e * e
            src = torch.zeros(n, lc, h, w, dtype=torch.float32, device=dev)
            z = torch.zeros(n, h, w, c_pad, dtype=dtype, device=dev)
            bld.nchw_to_nhwc(src, z, scale=pre_scale, shift=pre_shift)
            pq = self.post_quant_conv
            # the 1x1 post_quant conv writes its `latent_channels` columns into a zeroed c_pad-wide NHWC
            # buffer (dst_ld = c_pad) so that decoder.conv_in reads whole 16-byte chunks
            z2 = torch.zeros(n, h, w, c_pad, dtype=dtype, device=dev)
            bld.conv(z, pq.packed(dtype, c_pad), pq._f32("bias"), out=z2, name="post_quant_conv")
            y = self.decoder.emit(bld, z2)
            out = torch.zeros(n, self.config.out_channels, y.shape[1], y.shape[2], dtype=torch.float32, device=dev)
            bld.nhwc_to_nchw(y, out, scale=post_scale, shift=post_shift, clamp01=clamp01)
        else:
            ic = self.config.in_channels
            c_pad = (ic + e - 1) // e * e
            src = torch.zeros(n, ic, h, w, dtype=torch.float32, device=dev)
            x = torch.zeros(n, h, w, c_pad, dtype=dtype, device=dev)
            bld.nchw_to_nhwc(src, x, scale=pre_scale, shift=pre_shift)
            hh = self.encoder.emit(bld, x)
            qc = self.quant_conv
            m = bld.conv(hh, qc.packed(dtype, hh.shape[-1]), qc._f32("bias"), out_dtype=torch.float32, name="quant_conv")
            out = torch.zeros(n, 2 * lc, m.shape[1], m.shape[2], dtype=torch.float32, device=dev)
            bld.nhwc_to_nchw(m, out)
        plan = bld.finalize()
        st = dict(plan=plan, src=src, out=out, weights_version=weights_version(self))
        self._plans[key] = st
        return st

    # ---- diffusers surface -----------------------------------------------------------------------
    def _chunks(self, n: int, h_full: int, w_full: int, dtype) -> list:
        """split a batch so that the largest NHWC activation of one launch (full resolution x 2*block_out_channels[0]
        channels) stays below ~3.5 GB: the lean implicit-GEMM loop addresses its sources with 32-bit buffer
        offsets and larger tensors fall back to the (3-5x slower) 64-bit-pointer loop.  Equal chunks when n
        divides, so one recorded plan serves them all."""
        es = 4 if dtype == torch.float32 else 2
        per_img = h_full * w_full * 2 * self.config.block_out_channels[0] * es
        cap = max(1, int(3.5e9 // per_img))
        if n <= cap:
            return [n]
        k = -(-n // cap)
        while n % k and k < n:
            k += 1
        size = n // k if n % k == 0 else cap
        out = [size] * (n // size)
        if n % size:
            out.append(n % size)
        return out

    def _wait_side_user(self):
        """the recorded plans own ONE set of buffers: a caller that ran this module on a side stream (the trainer encoding the next
        window ahead, `MVLDMTrainer._start_prefetch`) leaves its completion event here, and every later call -- on whatever stream --
        waits for it first"""
        ev = self.__dict__.get("_busy_event")
        if ev is not None and torch.cuda.is_available():
            torch.cuda.current_stream().wait_event(ev)

    def decode(self, z: torch.Tensor, dtype=None, pre_scale: float = 1.0, post_scale: float = 1.0, post_shift: float = 0.0,
               clamp01: bool = False):
        """diffusers `decode(z).sample`; the optional affine maps (`z * pre_scale`, `img * post_scale + post_shift`,
        clamp to [0,1]) ride in the boundary layout kernels (last_stage_decode, diffusion_wrapper.py:289-298)"""
        require_gpu(z)
        self._wait_side_user()
        n, c, h, w = z.shape
        dtype = dtype or get_compute_dtype()
        outs, i0 = [], 0
        up = 2 ** (len(self.config.block_out_channels) - 1)
        for m in self._chunks(n, up * h, up * w, dtype):
            st = self._compile("decode", m, h, w, dtype, (pre_scale, 0.0, post_scale, post_shift, bool(clamp01)))
            st["src"].copy_(z[i0:i0 + m])
            st["plan"].run()
            outs.append(st["out"].clone())
            i0 += m
        return SimpleNamespace(sample=outs[0] if len(outs) == 1 else torch.cat(outs))

    def encode(self, x: torch.Tensor, dtype=None, pre_scale: float = 1.0, pre_shift: float = 0.0):
        """diffusers `encode(x).latent_dist`; `x * pre_scale + pre_shift` rides in the boundary layout kernel
        (first_stage_encode's `inputs * 2 - 1`, diffusion_wrapper.py:281)"""
        require_gpu(x)
        self._wait_side_user()
        n, c, h, w = x.shape
        dtype = dtype or get_compute_dtype()
        outs, i0 = [], 0
        for m in self._chunks(n, h, w, dtype):
            st = self._compile("encode", m, h, w, dtype, (pre_scale, pre_shift, 1.0, 0.0, False))
            st["src"].copy_(x[i0:i0 + m])
            st["plan"].run()
            outs.append(st["out"].clone())
            i0 += m
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(outs[0] if len(outs) == 1 else torch.cat(outs)))


@dataclass
class AutoencoderCfg:
    """src/model/autoencoder/__init__.py:9-13"""
    name: str = "kl"
    pretrained_from: Optional[str] = "stabilityai/stable-diffusion-2-1"
    kwargs: Optional[object] = None
    pretrained_overrides: Optional[dict] = None   # not in the reference: reduced widths for tests
    state_dict: Optional[dict] = None             # not in the reference: weights in diffusers layout (no hub access offline)
    allow_random_init: bool = False


AUTOENCODERS = {"kl": AutoencoderKL}


def get_autoencoder(cfg: AutoencoderCfg) -> AutoencoderKL:
    """src/model/autoencoder/__init__.py:33-43.  Like the reference, only the `from_pretrained`
    branch is functional (the from-config branch of the reference raises NameError)."""
    if cfg.pretrained_from is None:
        raise NotImplementedError("autoencoder from config: the reference's own branch is broken "
                                  "(autoencoder/__init__.py:28,40); use pretrained_from")
    return AUTOENCODERS[cfg.name].from_pretrained(cfg.pretrained_from, subfolder="vae",
                                                  config_overrides=getattr(cfg, "pretrained_overrides", None),
                                                  state_dict=getattr(cfg, "state_dict", None),
                                                  allow_random_init=getattr(cfg, "allow_random_init", False))
