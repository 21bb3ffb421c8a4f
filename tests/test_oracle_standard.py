"""CPU: the oracle's "standard" multi-view block and ray encodings (oracle/standard.py) against G10 -- outputs of the
reference's own StandardTransformer / MultiViewUNet walk / DiffusionWrapper.ray_encode (tests/golden/make_golden.py::g10)."""
import torch

from conftest import rel_err
from seeded import load_seeded


def test_standard_transformer_matches_reference(golden):
    from oracle.standard import StandardTransformer, StdAttnCfg
    g = golden("g10_standard_and_encodings")
    for i in range(int(g["n_st"])):
        C, heads, layers, d_dot, mult, b, V, h, seed = (int(v) for v in g[f"st{i}_meta"])
        m = StandardTransformer(StdAttnCfg(num_heads=heads, num_layers=layers, d_dot=None if d_dot < 0 else d_dot, d_mlp_multiplier=mult), C).eval()
        assert len(m.state_dict()) == int(g[f"st{i}_nkeys"])
        assert abs(load_seeded(m, seed) - float(g[f"st{i}_checksum"])) < 1e-6
        with torch.no_grad():
            y = m(torch.from_numpy(g[f"st{i}_x"]))
        assert rel_err(y, g[f"st{i}_y"]) < 2e-6, i


def test_mvunet_walk_with_standard_blocks_matches_reference(golden):
    from oracle import multiview as MV
    from oracle.standard import StdAttnCfg
    g = golden("g10_standard_and_encodings")
    widths = tuple(int(v) for v in g["unet_widths"])
    m = MV.MultiViewUNet(MV.MVUNetCfg(autoencoder=MV.UNetCfg(block_out_channels=widths), multi_view_attention=StdAttnCfg(), pretrained_from=None), 11, 4).eval()
    assert len(m.state_dict()) == int(g["unet_nkeys"])
    assert abs(load_seeded(m, 620) - float(g["unet_checksum"])) < 1e-6
    with torch.no_grad():
        y = m(torch.from_numpy(g["unet_x"]), torch.from_numpy(g["unet_t"]))
    assert rel_err(y, g["unet_y"]) < 2e-5


def test_ray_encodings_match_reference(golden):
    from oracle import pipeline as PL
    from oracle.standard import encode_rays
    g = golden("g10_standard_and_encodings")
    extr, intr = torch.from_numpy(g["rays_extr"]), torch.from_numpy(g["rays_intr"])
    o, d = PL.image_rays(6, 6, extr, intr)
    for name in g["rays_modes"]:
        use_pe, srt, plucker, no, nd = (int(v) for v in g[f"rays_{name}_cfg"])
        enc = encode_rays(o, d, bool(use_pe), bool(srt), bool(plucker), no, nd)
        b, v = enc.shape[:2]
        enc = enc.reshape(b, v, 6, 6, -1).permute(0, 1, 4, 2, 3)
        ref = torch.from_numpy(g[f"rays_{name}"])
        assert enc.shape == ref.shape, name
        assert (enc - ref).abs().max() < 2e-5, (str(name), float((enc - ref).abs().max()))
