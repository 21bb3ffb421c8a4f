"""Checkpoint loader: Lightning-style `.ckpt` and `.safetensors` round trips,
key-layout checks, old diffusers VAE attention names, strictness.  CPU only (parameters, no kernels) + one GPU test
that a load invalidates recorded plans."""
import pytest
import torch

from mv_ldm_amd import checkpoint as CK
from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg, UNet2DModelCfg
from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
from mv_ldm_amd.scheduler import DDIMScheduler
from mv_ldm_amd.vae import AutoencoderKL

GRAD_ENABLED = False      # tests/conftest.py::_grad_mode: no autograd graphs in this module
WIDTHS = (64, 64, 128, 128)


def small_pipeline(seed, device="cpu"):
    over = dict(block_out_channels=WIDTHS, attention_head_dim=tuple(max(1, c // 64) for c in WIDTHS))
    cfg = MultiViewUNetCfg(autoencoder=UNet2DModelCfg(block_out_channels=WIDTHS),
                           pretrained_from="stabilityai/stable-diffusion-2-1", pretrained_overrides=over)
    with torch.device(device):
        den = MultiViewUNet(cfg, 11, 4)
        vae = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1", config_overrides=dict(block_out_channels=(32, 32, 64, 64)))
    g = torch.Generator().manual_seed(seed)
    for m in (den, vae):
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return MVLDMPipeline(den, vae, DDIMScheduler(clip_sample=False), SamplerCfg(True, 3.0, 5))


def test_state_dict_layout_matches_app_a9():
    pipe = small_pipeline(0)
    sd = CK.wrapper_state_dict(pipe)
    for k in ("denoiser.unet.conv_in.weight", "denoiser.unet.time_embedding.linear_1.weight",
              "denoiser.unet.down_blocks.0.resnets.0.time_emb_proj.weight",
              "denoiser.unet.down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.bias",
              "denoiser.unet.down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
              "denoiser.unet.down_blocks.0.downsamplers.0.conv.weight", "denoiser.unet.mid_block.attentions.0.proj_in.weight",
              "denoiser.unet.up_blocks.3.resnets.2.conv_shortcut.weight", "denoiser.unet.up_blocks.0.upsamplers.0.conv.bias",
              "denoiser.unet.conv_norm_out.weight", "denoiser.cross_attn_blocks_encoder.0.proj_in.weight",
              "denoiser.cross_attn_blocks_mid.0.transformer_blocks.0.attn2.to_q.weight",
              "denoiser.cross_attn_blocks_decoder.3.transformer_blocks.0.ff.net.2.weight",
              "autoencoder.encoder.mid_block.attentions.0.to_q.weight", "autoencoder.decoder.up_blocks.3.resnets.2.conv2.bias",
              "autoencoder.quant_conv.weight", "autoencoder.post_quant_conv.bias"):
        assert k in sd, k
    # the multi-view blocks' projections are 1x1 convs in the reference (mvdream/attention.py:398-414)
    assert sd["denoiser.cross_attn_blocks_encoder.0.proj_in.weight"].dim() == 4


@pytest.mark.parametrize("fmt", ["ckpt", "safetensors"])
def test_round_trip(tmp_path, fmt):
    src, dst = small_pipeline(1), small_pipeline(2)
    sd = CK.wrapper_state_dict(src)
    path = tmp_path / f"last.{fmt}"
    if fmt == "ckpt":     # Lightning container: tensors under "state_dict" next to trainer bookkeeping
        torch.save({"state_dict": sd, "global_step": 1679000, "epoch": 3}, path)
    else:
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, str(path))
    dst._plans["stale"] = object()
    rep = CK.load_pipeline_checkpoint(dst, path)
    assert rep["denoiser"].ok() and rep["autoencoder"].ok() and not dst._plans
    assert rep["denoiser"].loaded == len(src.denoiser.state_dict()) and rep["autoencoder"].loaded == len(src.autoencoder.state_dict())
    for a, b in ((src.denoiser, dst.denoiser), (src.autoencoder, dst.autoencoder)):
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb and torch.equal(va, vb), ka


def test_use_ema_sampling_loads_the_averaged_copy(tmp_path):
    """`model.use_ema_sampling` (diffusion_wrapper.py:460-463): a DiffusionWrapper checkpoint of a run with `model.ema = true` holds
    the AveragedModel under `ema.module.*` / `ema.n_averaged`; `use_ema=True` loads THAT copy into the denoiser"""
    src, avg, dst = small_pipeline(1), small_pipeline(5), small_pipeline(2)
    sd = CK.wrapper_state_dict(src)
    sd.update({"ema.module." + k: v.detach().clone() for k, v in avg.denoiser.state_dict().items()})
    sd["ema.n_averaged"] = torch.tensor(1234)
    path = tmp_path / "last.ckpt"
    torch.save({"state_dict": sd}, path)
    parts = CK.split_wrapper_state(sd)
    assert len(parts["ema"]) == len(avg.denoiser.state_dict()) + 1 and not parts["other"]
    rep = CK.load_pipeline_checkpoint(dst, path)
    assert rep["denoiser"].ok()
    assert torch.equal(dst.denoiser.unet.conv_in.weight, src.denoiser.unet.conv_in.weight)
    rep = CK.load_pipeline_checkpoint(dst, path, use_ema=True)
    assert rep["denoiser"].ok() and rep["denoiser"].loaded == len(avg.denoiser.state_dict())
    for (ka, va), (kb, vb) in zip(avg.denoiser.state_dict().items(), dst.denoiser.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    plain = tmp_path / "no_ema.ckpt"
    torch.save({"state_dict": CK.wrapper_state_dict(src)}, plain)
    with pytest.raises(KeyError, match="use_ema_sampling"):
        CK.load_pipeline_checkpoint(dst, plain, use_ema=True)


def test_strictness_old_vae_names_and_linear_conv_reshape(tmp_path):
    src, dst = small_pipeline(3), small_pipeline(4)
    sd = CK.wrapper_state_dict(src)
    # (1) pre-0.15 diffusers VAE attention names
    old = {}
    for k, v in sd.items():
        for new, o in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if k.startswith("autoencoder.") and f"attentions.0.{new}." in k:
                k = k.replace(f"attentions.0.{new}.", f"attentions.0.{o}.")
                break
        old[k] = v
    # (2) 1x1-conv projections exported as Linear matrices
    k_proj = "denoiser.cross_attn_blocks_encoder.0.proj_in.weight"
    old[k_proj] = old[k_proj].reshape(old[k_proj].shape[0], -1)
    torch.save({"state_dict": old}, tmp_path / "old.ckpt")
    rep = CK.load_pipeline_checkpoint(dst, tmp_path / "old.ckpt")
    assert rep["denoiser"].ok() and rep["autoencoder"].ok() and rep["denoiser"].reshaped == [k_proj[len("denoiser."):]]
    assert torch.equal(dst.autoencoder.state_dict()["decoder.mid_block.attentions.0.to_out.0.weight"],
                       src.autoencoder.state_dict()["decoder.mid_block.attentions.0.to_out.0.weight"])
    # (3) a missing and an unexpected key are errors when strict, reported otherwise
    bad = dict(sd)
    del bad["denoiser.unet.conv_out.bias"]
    bad["denoiser.unet.not_a_layer.weight"] = torch.zeros(1)
    torch.save({"state_dict": bad}, tmp_path / "bad.ckpt")
    with pytest.raises(KeyError):
        CK.load_pipeline_checkpoint(dst, tmp_path / "bad.ckpt")
    rep = CK.load_pipeline_checkpoint(dst, tmp_path / "bad.ckpt", strict=False)
    assert rep["denoiser"].missing == ["unet.conv_out.bias"] and rep["denoiser"].unexpected == ["unet.not_a_layer.weight"]
    # (4) a wrong shape is always an error
    bad2 = dict(sd)
    bad2["denoiser.unet.conv_in.weight"] = torch.zeros(3, 3)
    torch.save({"state_dict": bad2}, tmp_path / "bad2.ckpt")
    with pytest.raises(ValueError):
        CK.load_pipeline_checkpoint(dst, tmp_path / "bad2.ckpt", strict=False)
    # (5) a denoiser-only export (no wrapper prefixes)
    torch.save({k[len("denoiser."):]: v for k, v in sd.items() if k.startswith("denoiser.")}, tmp_path / "den.pt")
    rep = CK.load_pipeline_checkpoint(dst, tmp_path / "den.pt")
    assert set(rep) == {"denoiser"} and rep["denoiser"].ok()


@pytest.mark.gpu
def test_loading_a_checkpoint_replaces_the_weights_the_kernels_use(tmp_path):
    import mv_ldm_amd
    from seeded import random_cameras  # noqa: F401  (tests/golden on the path via conftest)
    a, b = small_pipeline(5, "cuda"), small_pipeline(6, "cuda")
    torch.save({"state_dict": CK.wrapper_state_dict(a)}, tmp_path / "a.ckpt")
    g = torch.Generator().manual_seed(0)
    extr = torch.eye(4).repeat(1, 3, 1, 1)
    extr[0, 1:, :3, 3] = 0.1 * torch.randn(2, 3, generator=g)
    intr = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]]).repeat(1, 3, 1, 1)
    batch = {"context": {"image": torch.rand(1, 1, 3, 64, 64, generator=g).cuda(), "extrinsics": extr[:, :1], "intrinsics": intr[:, :1]},
             "target": {"extrinsics": extr[:, 1:], "intrinsics": intr[:, 1:]}}
    x_T, noise = torch.randn(1, 2, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    for p in (a, b):
        p.set_timesteps(5)
    with mv_ldm_amd.compute_dtype(torch.float32):
        ya = a.sample(batch, x_T=x_T, encode_noise=noise)[1]
        yb0 = b.sample(batch, x_T=x_T, encode_noise=noise)[1]
        assert not torch.allclose(ya, yb0)
        CK.load_pipeline_checkpoint(b, tmp_path / "a.ckpt")
        yb1 = b.sample(batch, x_T=x_T, encode_noise=noise)[1]
    assert torch.equal(ya, yb1)
