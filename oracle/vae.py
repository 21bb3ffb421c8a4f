"""Oracle (test infrastructure): diffusers==0.27.2 `AutoencoderKL` restated.

Reached from the reference via `src/model/autoencoder/__init__.py:15-43` and
`src/model/diffusion_wrapper.py:278-298` (`encode(2x-1).latent_dist.sample() * 0.18215`,
`decode(z / 0.18215).sample`).  Parity unpinned by the reference (package absent, no tests);
cross-checked against torch-primitive compositions in tests/test_oracle_blocks.py.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Sequence

import torch
from torch import nn

from .blocks import Downsample2D, ResnetBlock2D, UNetMidBlock2D, Upsample2D

SD21_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, latent_channels=4, norm_num_groups=32, scaling_factor=0.18215)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, None, groups, 1e-6)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, out_channels, padding=0)]) if add_downsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d(x)
        return x


class UpDecoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, groups, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, None, groups, 1e-6)
            for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, out_channels)]) if add_upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u(x)
        return x


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, boc: Sequence[int], layers_per_block, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i in range(len(boc)):
            in_c, out_c = out_c, boc[i]
            self.down_blocks.append(DownEncoderBlock2D(in_c, out_c, layers_per_block, groups, i != len(boc) - 1))
        self.mid_block = UNetMidBlock2D(boc[-1], None, num_layers=1, resnet_eps=1e-6, resnet_groups=groups,
                                        add_attention=True, attention_head_dim=boc[-1])
        self.conv_norm_out = nn.GroupNorm(groups, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[-1], 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, latent_channels, out_channels, boc: Sequence[int], layers_per_block, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(latent_channels, boc[-1], 3, padding=1)
        self.mid_block = UNetMidBlock2D(boc[-1], None, num_layers=1, resnet_eps=1e-6, resnet_groups=groups,
                                        add_attention=True, attention_head_dim=boc[-1])
        self.up_blocks = nn.ModuleList()
        rboc = list(reversed(boc))
        out_c = rboc[0]
        for i in range(len(rboc)):
            prev, out_c = out_c, rboc[i]
            self.up_blocks.append(UpDecoderBlock2D(prev, out_c, layers_per_block + 1, groups, i != len(rboc) - 1))
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.conv_in(z)
        x = self.mid_block(x)
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class DiagonalGaussianDistribution:
    def __init__(self, parameters):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(64,), layers_per_block=1,
                 latent_channels=4, norm_num_groups=32, scaling_factor=0.18215, **_ignored):
        super().__init__()
        boc = tuple(block_out_channels)
        self.config = SimpleNamespace(scaling_factor=scaling_factor, latent_channels=latent_channels,
                                      block_out_channels=boc)
        self.encoder = Encoder(in_channels, latent_channels, boc, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, boc, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    @classmethod
    def from_pretrained(cls, path, subfolder="vae", config_overrides=None):
        cfg = dict(SD21_VAE_CONFIG)
        cfg.update(config_overrides or {})
        return cls(**cfg)

    def encode(self, x):
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(z)))
