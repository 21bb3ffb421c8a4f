"""Import the reference's own Python modules from /root/reference (build container only).

Used ONLY by tests/golden/make_golden.py to produce the committed vectors.  Nothing here travels to
the GPU box and nothing in `tests/test_*.py`, `bench.py` or `__graft_entry__.py` imports it.

The reference depends on packages that are absent here; they are replaced by inert
stubs so that the reference-owned arithmetic imports unmodified:

  jaxtyping, dacite, hydra, moviepy, pytorch_lightning, torchvision, wandb, colorspacious, skimage
        -> attribute-swallowing dummies (annotations / logging / image IO only)
  diffusers
        -> re-exports the oracle's restatements (`oracle.blocks`, `oracle.vae`, `oracle.scheduler`);
           this is what SURVEY.md §8c prescribes: diffusers-owned arithmetic is "parity unpinned",
           the reference-owned walk / attention / step / sample / rays / schedules are pinned.

`torch.Tensor.cuda` / `torch.zeros(...).cuda()` are patched to identity while reference code runs
(the reference hard-codes `.cuda()`).  Bytecode writing is disabled so the
read-only reference tree is left untouched.
"""
from __future__ import annotations

import contextlib
import importlib
import sys
import types

import torch
from torch import nn

sys.dont_write_bytecode = True
REF_ROOT = "/root/reference"


class _Dummy:
    """Subscriptable, callable, attribute-swallowing placeholder (type annotations, decorators)."""

    def __init__(self, name="dummy"):
        self._n = name

    def __getitem__(self, item):
        return self

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Dummy(f"{self._n}.{item}")

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k and not isinstance(a[0], _Dummy):
            return a[0]  # used as a decorator
        return self

    def __or__(self, other):
        return self

    def __ror__(self, other):
        return self

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Dummy(f"{self.__name__}.{item}")


def _stub(name: str, **attrs):
    mod = _StubModule(name)
    mod.__path__ = []  # behave like a package so sub-imports resolve through the finder
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


class _StubFinder:
    """meta-path finder: any submodule of a stubbed top-level package becomes a stub too."""

    ROOTS = ("jaxtyping", "dacite", "hydra", "moviepy", "pytorch_lightning", "torchvision", "wandb",
             "colorspacious", "skimage", "lpips", "omegaconf", "beartype", "DISTS_pytorch", "cleanfid")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        mod = _StubModule(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        return None


class _LightningModule(nn.Module):
    """The handful of LightningModule members `DiffusionWrapper` touches on the sampling path."""
    global_rank = 0

    @property
    def device(self):
        return torch.device("cpu")

    def log(self, *a, **k):
        return None

    @property
    def global_step(self):
        return 0


_INSTALLED = False


def install():
    global _INSTALLED
    if _INSTALLED:
        return
    import os
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import blocks, scheduler, vae

    sys.meta_path.insert(0, _StubFinder())
    import pytorch_lightning as pl  # resolved by the finder
    pl.LightningModule = _LightningModule
    pl_util = importlib.import_module("pytorch_lightning.utilities")
    pl_util.rank_zero_only = lambda f: f
    jt = importlib.import_module("jaxtyping")
    jt.install_import_hook = lambda *a, **k: contextlib.nullcontext()

    d = _stub("diffusers",
              UNet2DConditionModel=blocks.UNet2DConditionModel,
              AutoencoderKL=vae.AutoencoderKL,
              DDIMScheduler=scheduler.DDIMScheduler,
              DDPMScheduler=scheduler.DDIMScheduler)  # DDPM is never stepped on the sampling path
    _stub("diffusers.utils")
    iu = _stub("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    d.utils = sys.modules["diffusers.utils"]
    d.utils.import_utils = iu

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _INSTALLED = True


@contextlib.contextmanager
def cpu_cuda():
    """Make `.cuda()` an identity (the reference hard-codes it on its hot path)."""
    orig_t, orig_m = torch.Tensor.cuda, nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda, nn.Module.cuda = orig_t, orig_m


def ref(module: str):
    """import a reference module, e.g. ref('src.model.denoiser.mvunet')."""
    install()
    return importlib.import_module(module)
