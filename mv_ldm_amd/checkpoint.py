"""Checkpoint loading for the drop-in modules.

The reference trains with Lightning and saves `checkpoints/*.ckpt` whose `state_dict` holds the whole
`DiffusionWrapper`: `denoiser.*` (the multi-view UNet, `src/model/denoiser/mvunet.py`) and `autoencoder.*` (the frozen
SD-2.1 VAE, `src/model/autoencoder/__init__.py`); `src/main.py:119-139` resumes from it and
`src/scripts/generate_mvldm.py:29-87` loads it for sampling.  The modules of this package keep the diffusers /
reference parameter names, so loading is key-for-key; this file only reads the container formats, strips the
wrapper prefixes, maps the pre-0.15 diffusers VAE attention names and reports what did not match.

Kernel-layout copies of the parameters are version-checked (modules._PackCache) and refresh themselves after a
load; recorded plans hold raw pointers to the OLD copies and are dropped here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional

import torch

_OLD_VAE_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def read_state_dict(path) -> Dict[str, torch.Tensor]:
    """`.safetensors` (diffusers layout) or a torch pickle (`.ckpt` / `.pt` / `.pth` / `.bin`); a Lightning
    checkpoint's tensors sit under its `state_dict` entry."""
    path = Path(path)
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file
        return dict(load_file(str(path), device="cpu"))
    obj = torch.load(str(path), map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and isinstance(obj.get("state_dict"), dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict) or not all(isinstance(v, torch.Tensor) for v in obj.values()):
        raise ValueError(f"{path}: not a state dict")
    return obj


def find_local_weights(path, subfolder: str) -> Optional[Path]:
    """the weight file `from_pretrained(path, subfolder=...)` would read from a LOCAL diffusers snapshot:
    `<path>/<subfolder>/diffusion_pytorch_model.{safetensors,bin}`, or `path` itself when it is a weight file.
    None when `path` is a hub id / does not exist (there is no hub access offline)."""
    if path is None:
        return None
    p = Path(str(path))
    if p.is_file() and p.suffix in (".safetensors", ".bin", ".ckpt", ".pt", ".pth"):
        return p
    for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"):
        f = p / subfolder / name
        if f.is_file():
            return f
    return None


def load_unet_checkpoint(unet, path, strict: bool = True) -> "LoadReport":
    """a stand-alone diffusers UNet file (`unet/diffusion_pytorch_model.safetensors` / `.bin`).  Like the reference
    (mvunet.py:66-72) the caller replaces conv_in / conv_out AFTER this load, so their SD shapes (4 in / 4 out) must fit
    the module as built by `from_pretrained` -- mismatching conv_in/conv_out entries are skipped, not an error."""
    sd = read_state_dict(path)
    own = unet.state_dict()
    sd = {k: v for k, v in sd.items() if not (k.startswith(("conv_in.", "conv_out.")) and k in own and own[k].shape != v.shape)}
    return load_module_state(unet, sd, strict, "unet")


def split_wrapper_state(sd: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """`denoiser.* / autoencoder.* / ema.*` of a DiffusionWrapper checkpoint -> {"denoiser": {...}, "autoencoder": {...},
    "ema": {...}, "other": {...}} with the prefixes removed (`ema.` = the AveragedModel of diffusion_wrapper.py:138-142: keys
    `module.<denoiser key>` and `n_averaged`).  A state dict without those prefixes is returned under "other"."""
    out = {"denoiser": {}, "autoencoder": {}, "ema": {}, "other": {}}
    for k, v in sd.items():
        for part in ("denoiser", "autoencoder", "ema"):
            if k.startswith(part + "."):
                out[part][k[len(part) + 1:]] = v
                break
        else:
            out["other"][k] = v
    return out


def _modernise_vae_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """diffusers < 0.15 named the VAE mid-block attention `query/key/value/proj_attn` (weights `[C, C]`)"""
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if len(parts) >= 2 and "attentions" in parts and parts[-2] in _OLD_VAE_ATTN:
            parts[-2:-1] = _OLD_VAE_ATTN[parts[-2]].split(".")
            k = ".".join(parts)
        out[k] = v
    return out


@dataclass
class LoadReport:
    loaded: int = 0
    missing: List[str] = field(default_factory=list)
    unexpected: List[str] = field(default_factory=list)
    reshaped: List[str] = field(default_factory=list)

    def ok(self) -> bool:
        return not self.missing and not self.unexpected


def load_module_state(module: torch.nn.Module, sd: Dict[str, torch.Tensor], strict: bool = True, what: str = "module") -> LoadReport:
    """copy `sd` into `module` key for key (fp32 parameters keep their dtype/device).  1x1-conv weights stored as
    Linear `[out, in]` (or the reverse) are reshaped: the multi-view blocks' proj_in/proj_out are 1x1 convs in the
    reference (`mvdream/attention.py:398-414`) and Linears in `use_linear_projection` checkpoints."""
    own = module.state_dict()
    rep = LoadReport()
    with torch.no_grad():
        for k, dst in own.items():
            src = sd.get(k)
            if src is None:
                rep.missing.append(k)
                continue
            if src.shape != dst.shape:
                if src.numel() == dst.numel() and {src.dim(), dst.dim()} == {2, 4}:
                    src = src.reshape(dst.shape)
                    rep.reshaped.append(k)
                else:
                    raise ValueError(f"{what}: {k} has shape {tuple(src.shape)}, expected {tuple(dst.shape)}")
            dst.copy_(src.to(dst.dtype))
            rep.loaded += 1
    rep.unexpected = [k for k in sd if k not in own]
    if strict and not rep.ok():
        raise KeyError(f"{what}: {len(rep.missing)} missing key(s) (first: {rep.missing[:3]}), "
                       f"{len(rep.unexpected)} unexpected key(s) (first: {rep.unexpected[:3]})")
    return rep


def load_pipeline_checkpoint(pipe, path, strict: bool = True, load_autoencoder: Optional[bool] = None,
                             use_ema: bool = False) -> Dict[str, LoadReport]:
    """load a DiffusionWrapper checkpoint (Lightning `.ckpt` or the same keys as `.safetensors`) into an
    `MVLDMPipeline`.  The VAE part is optional (`load_autoencoder=None`: load it if the file has one -- released
    checkpoints carry the frozen SD-2.1 VAE, a denoiser-only export does not).  `use_ema`: load the `ema.module.*` copy into
    the denoiser instead of `denoiser.*`."""
    parts = split_wrapper_state(read_state_dict(path))
    den_sd = parts["denoiser"] or parts["other"]
    if use_ema:      # `model.use_ema_sampling` (diffusion_wrapper.py:460-463): sample with the averaged weights
        den_sd = {k[len("module."):]: v for k, v in parts["ema"].items() if k.startswith("module.")}
        if not den_sd:
            raise KeyError(f"{path}: use_ema_sampling needs the `ema.module.*` tensors of a run trained with model.ema = true")
    reports = {"denoiser": load_module_state(pipe.denoiser, den_sd, strict, "denoiser")}
    has_vae = bool(parts["autoencoder"])
    if load_autoencoder or (load_autoencoder is None and has_vae):
        if not has_vae:
            raise KeyError(f"{path}: no autoencoder.* tensors")
        reports["autoencoder"] = load_module_state(pipe.autoencoder, _modernise_vae_keys(parts["autoencoder"]), strict, "autoencoder")
    invalidate_plans(pipe)
    return reports


def load_vae_checkpoint(vae, path, strict: bool = True) -> LoadReport:
    """a stand-alone diffusers VAE file (`vae/diffusion_pytorch_model.safetensors` / `.bin`)"""
    rep = load_module_state(vae, _modernise_vae_keys(read_state_dict(path)), strict, "autoencoder")
    if hasattr(vae, "_plans"):
        vae._plans.clear()
    return rep


def invalidate_plans(pipe) -> None:
    """recorded plans point at the packed copies of the OLD weights"""
    for obj in (pipe, getattr(pipe, "autoencoder", None), getattr(pipe, "denoiser", None)):
        plans = getattr(obj, "_plans", None)
        if isinstance(plans, dict):
            plans.clear()


def wrapper_state_dict(pipe) -> Dict[str, torch.Tensor]:
    """the DiffusionWrapper-style state dict of a pipeline (for round trips and for exporting random-init models)"""
    sd = {"denoiser." + k: v.detach().cpu() for k, v in pipe.denoiser.state_dict().items()}
    sd.update({"autoencoder." + k: v.detach().cpu() for k, v in pipe.autoencoder.state_dict().items()})
    return sd
