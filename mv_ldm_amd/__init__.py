"""mv_ldm_amd -- MI355X-native (gfx950) multi-view latent-diffusion denoising path.

Host side of the drop-in: Python modules that mirror the reference's operator / plug-in interface
for this path (`DENOISER`, `SCHEDULER`, `AUTOENCODERS`, `get_attn_blocks`; diffusers-style
`UNet2DConditionModel`, `DDIMScheduler`, `AutoencoderKL` attribute surface) and run every bit of
arithmetic in hand-written HIP kernels through the C ABI of `csrc/libmvldm_hip.so`
(`include/mvldm.h`).  There is no CPU or PyTorch-math fallback.
"""
__version__ = "0.1.0"
