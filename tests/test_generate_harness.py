"""The generation harness (reference: src/scripts/generate_mvldm.py:29-87 + DiffusionWrapper.test_step,
diffusion_wrapper.py:1057-1067): the object graph is built from the config values the sampling path consumes.
CPU part: config plumbing and construction (no compute); GPU part: the harness end to end on synthetic BatchedExamples."""
import copy
import json

import pytest
import torch

from mv_ldm_amd import generate as G

GRAD_ENABLED = False      # tests/conftest.py::_grad_mode

SMALL = {"denoiser": dict(block_out_channels=(64, 64, 128, 128), attention_head_dim=(1, 1, 2, 2)),
         "autoencoder": dict(block_out_channels=(32, 32, 64, 64))}


def _small_cfg(**test):
    cfg = copy.deepcopy(G.DEFAULT_CONFIG)
    cfg["model"]["denoiser"]["autoencoder"]["block_out_channels"] = [64, 64, 128, 128]
    cfg["model"]["scheduler"]["num_inference_steps"] = 3
    cfg["test"].update(test)
    cfg["seed"] = 4
    return cfg


def test_default_config_holds_the_released_values():
    c = G.DEFAULT_CONFIG
    assert c["model"]["use_cfg"] is True and c["model"]["cfg_scale"] == 3.0                       # config/main.yaml:30-31
    assert c["model"]["scheduler"]["num_inference_steps"] == 70 and c["model"]["scheduler"]["kwargs"]["clip_sample"] is False
    assert c["model"]["ray_encodings"] == {"num_origin_octaves": 15, "num_direction_octaves": 15}
    assert c["model"]["denoiser"]["multi_view_attention"] == {"name": "spatial_transformer_3d", "num_heads": 8, "num_layers": 1,
                                                              "d_mlp_multiplier": 1, "pos_enc": False}
    assert c["test"] == {"sampling_mode": "anchored", "limit_frames": None, "num_anchors_views": 4, "output_dir": ""}
    assert c["trainer"]["precision"] == "16-mixed"


def test_overrides_and_merge():
    cfg = G.apply_overrides(copy.deepcopy(G.DEFAULT_CONFIG), ["test.sampling_mode=autoregressive", "test.limit_frames=80",
                                                               "model.scheduler.num_inference_steps=25", "model.cfg_scale=2.5",
                                                               "checkpointing.load=/x/last.ckpt", "seed=null"])
    assert cfg["test"]["sampling_mode"] == "autoregressive" and cfg["test"]["limit_frames"] == 80
    assert cfg["model"]["scheduler"]["num_inference_steps"] == 25 and cfg["model"]["cfg_scale"] == 2.5
    assert cfg["checkpointing"]["load"] == "/x/last.ckpt" and cfg["seed"] is None
    m = G.merge_config(G.DEFAULT_CONFIG, {"model": {"scheduler": {"kwargs": {"clip_sample": True}}}})
    assert m["model"]["scheduler"]["kwargs"]["clip_sample"] is True and m["model"]["scheduler"]["kwargs"]["beta_end"] == 0.02
    assert G.DEFAULT_CONFIG["model"]["scheduler"]["kwargs"]["clip_sample"] is False


def test_build_pipeline_follows_the_config():
    cfg = _small_cfg()
    cfg["model"]["cfg_scale"] = 2.0
    cfg["model"]["denoiser"]["mid_conditioning"] = False
    pipe, rep = G.build_pipeline(cfg, device="cpu", allow_random_init=True, overrides=SMALL)
    assert rep is None and pipe.cfg.use_cfg and pipe.cfg.cfg_scale == 2.0
    assert pipe.scheduler.timesteps.tolist() == [666, 333, 0] and pipe.scheduler.clip_range == 0.0
    assert pipe.denoiser.in_channels == 11 and pipe.denoiser.out_channels == 4
    assert hasattr(pipe.denoiser, "cross_attn_blocks_encoder") and not hasattr(pipe.denoiser, "cross_attn_blocks_mid")
    assert pipe.denoiser.cross_attn_blocks_encoder[0].transformer_blocks[0].attn1.heads == 8
    # the alternative YAML selections: Pluecker origins keep 6 ray channels, positional encodings widen conv_in,
    # `multi_view_attention: standard` swaps the multi-view blocks
    cfg["model"]["use_plucker"] = True
    p2, _ = G.build_pipeline(cfg, device="cpu", allow_random_init=True, overrides=SMALL)
    assert p2.rays.use_plucker and p2.denoiser.in_channels == 11
    cfg["model"].update(use_ray_encoding=True, ray_encodings={"num_origin_octaves": 10, "num_direction_octaves": 8})
    cfg["model"]["denoiser"]["multi_view_attention"] = {"name": "standard", "num_heads": 8, "d_mlp_multiplier": 1, "pos_enc": False}
    cfg["model"]["denoiser"]["pretrained_from"] = None
    p3, _ = G.build_pipeline(cfg, device="cpu", allow_random_init=True, overrides=SMALL)
    assert p3.denoiser.in_channels == 4 + 6 * 10 + 6 * 8 + 1 and type(p3.denoiser.cross_attn_blocks_encoder[0]).__name__ == "StandardTransformer"


def test_synthetic_example_is_batched_example_shaped():
    ex = G.synthetic_example(3, 12, 64, seed=1)
    c, t = ex["context"], ex["target"]
    assert c["image"].shape == (1, 1, 3, 64, 64) and 0 <= float(c["image"].min()) and float(c["image"].max()) <= 1
    assert c["extrinsics"].shape == (1, 1, 4, 4) and t["extrinsics"].shape == (1, 12, 4, 4) and t["intrinsics"].shape == (1, 12, 3, 3)
    assert c["index"].tolist() == [[0]] and t["index"].tolist() == [list(range(1, 13))] and t["index"].dtype == torch.int64
    assert t["near"].shape == (1, 12) and ex["scene"] == ["synthetic0003"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["anchored", "autoregressive"])
def test_harness_end_to_end_and_rank_sharding(mode, tmp_path):
    """2 ranks' worth of work done by one process in two passes (rank 0 / rank 1 of world 2) equals the one-rank run:
    scenes are independent and, with a seed, every (scene, call) owns its noise stream"""
    import mv_ldm_amd
    from mv_ldm_amd.image_io import decode_png
    cfg = _small_cfg(sampling_mode=mode, limit_frames=13)
    pipe, _ = G.build_pipeline(cfg, device="cuda", allow_random_init=True, overrides=SMALL)
    g = torch.Generator(device="cuda").manual_seed(0)
    for m in (pipe.denoiser, pipe.autoencoder):
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g, device="cuda") * 0.05)
    examples = [G.synthetic_example(i, 14, 64, seed=2) for i in range(3)]
    with mv_ldm_amd.compute_dtype(torch.float32):
        one = G.evaluate(cfg, examples, pipe=pipe, batch_scenes=3, out_dir=str(tmp_path / "one"))
        r0 = G.evaluate(cfg, examples, pipe=pipe, batch_scenes=2, rank=0, world=2)
        r1 = G.evaluate(cfg, examples, pipe=pipe, batch_scenes=2, rank=1, world=2)
    assert one["owned"] == [0, 1, 2] and r0["owned"] == [0, 2] and r1["owned"] == [1]
    assert one["views"] == r0["views"] + r1["views"] and one["views"] > 0
    both = {**r0["frames"], **r1["frames"]}
    assert sorted(both) == sorted(one["frames"]) == ["synthetic0000", "synthetic0001", "synthetic0002"]
    for name, fr in one["frames"].items():
        assert sorted(fr) == sorted(both[name])
        for f, im in fr.items():
            assert im.shape == (3, 64, 64) and torch.isfinite(im).all()
            assert (im - both[name][f]).abs().max() < 2e-4          # f32; only the batch size seen by the kernels differs
    # PNG frames land where the reference puts them: <output_dir>/<scene>/color/<index:0>6>.png
    f0 = sorted(one["frames"]["synthetic0001"])[0]
    png = tmp_path / "one" / "synthetic0001" / "color" / f"{f0:0>6}.png"
    assert png.exists()
    back = torch.from_numpy(decode_png(png.read_bytes())).permute(2, 0, 1).float() / 255.0
    assert (back - one["frames"]["synthetic0001"][f0].cpu()).abs().max() <= 1.0 / 255 + 1e-6
    cfg["test"]["sampling_mode"] = "bogus"
    with pytest.raises(Exception, match="Incorrect Mode"):
        G.evaluate(cfg, examples, pipe=pipe)


@pytest.mark.gpu
def test_cli_runs_the_released_topology(capsys):
    """`python -m mv_ldm_amd.generate` on the full-width model (random init), 2 scenes, 13 frames, 2 DDIM steps"""
    rc = G.main(["model.scheduler.num_inference_steps=2", "test.limit_frames=13", "seed=1", "--scenes", "2", "--frames", "13",
                 "--res", "256", "--dtype", "bf16", "--allow-random-init"])
    assert rc == 0
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["mode"] == "anchored" and out["ddim_steps"] == 2 and out["n_gpus"] == 1 and out["views"] == 2 * 13 and out["views_per_s"] > 0
