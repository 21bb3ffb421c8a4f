"""mv_ldm_amd -- MI355X-native (gfx950) multi-view latent-diffusion denoising path.

Host side of the drop-in: Python modules that mirror the reference's operator / plug-in interface
for this path (`DENOISER`, `SCHEDULER`, `AUTOENCODERS`, `get_attn_blocks`; diffusers-style
`UNet2DConditionModel`, `DDIMScheduler`, `AutoencoderKL` attribute surface) and run every bit of
arithmetic in hand-written HIP kernels through the C ABI of `csrc/libmvldm_hip.so`
(`include/mvldm.h`).  There is no CPU or PyTorch-math fallback: without the shared library (or
without a GPU) the compute entry points raise.
"""
__version__ = "0.1.0"

from .runtime import compute_dtype, get_compute_dtype, set_compute_dtype  # noqa: F401


def __getattr__(name):
    # lazy: importing the package must not require the shared library (CPU tooling, build step)
    import importlib
    table = {
        "UNet2DConditionModel": "modules", "Transformer2DModel": "modules", "ResnetBlock2D": "modules",
        "MultiViewUNet": "mvunet", "MultiViewUNetCfg": "mvunet", "UNet2DModelCfg": "mvunet",
        "SpatialTransformer3D": "mvunet", "SpatialTransformer3DCfg": "mvunet", "get_attn_blocks": "mvunet",
        "StandardTransformer": "mvunet", "CrossAttentionCfg": "mvunet", "RayEncodingCfg": "pipeline",
        "DENOISER": "mvunet", "get_denoiser": "mvunet",
        "DDIMScheduler": "scheduler", "DDIMSchedulerCfg": "scheduler", "SchedulerCfg": "scheduler",
        "SCHEDULER": "scheduler", "get_scheduler": "scheduler",
        "AutoencoderKL": "vae", "AutoencoderCfg": "vae", "AUTOENCODERS": "vae", "get_autoencoder": "vae",
        "MVLDMPipeline": "pipeline", "SamplerCfg": "pipeline",
    }
    if name in table:
        return getattr(importlib.import_module(f".{table[name]}", __name__), name)
    raise AttributeError(name)
