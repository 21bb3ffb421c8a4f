"""CPU: the oracle's restated training step (oracle/train.py) against G9 -- loss and gradients of the REFERENCE's own
`DiffusionWrapper.training_step` + autograd (tests/golden/make_golden.py::g9)."""
import numpy as np
import pytest
import torch

from seeded import load_seeded

GRAD_ENABLED = True       # tests/conftest.py::_grad_mode: torch references are differentiated here


def build_oracle(g):
    from oracle import multiview as OMV
    from oracle.scheduler import DDIMScheduler
    from oracle.vae import AutoencoderKL
    widths = tuple(int(v) for v in g["widths"])
    over = dict(block_out_channels=widths, attention_head_dim=tuple(max(1, c // 64) for c in widths))
    den = OMV.MultiViewUNet(OMV.MVUNetCfg(autoencoder=OMV.UNetCfg(block_out_channels=widths), pretrained_from="sd21",
                                          pretrained_overrides=over), 11, 4)
    vae = AutoencoderKL.from_pretrained("x", config_overrides=dict(block_out_channels=tuple(int(v) for v in g["vae_widths"]),
                                                                   layers_per_block=1)).eval()
    return den, vae, DDIMScheduler(clip_sample=False)


def g9_case(g, ci):
    p = f"c{ci}_"
    img, extr, intr = (torch.from_numpy(g[p + k]) for k in ("image", "extr", "intr"))
    view = lambda sl: {"image": img[:, sl], "extrinsics": extr[:, sl], "intrinsics": intr[:, sl]}
    batch = {"context": view(slice(0, 2)), "target": view(slice(2, 5))}
    choices = dict(index=int(g[p + "index"]), second=int(g[p + "second"]), relative_coin=bool(g[p + "relative_coin"]),
                   unconditional=bool(g[p + "unconditional"]), noise=torch.from_numpy(g[p + "noise"]),
                   timesteps=torch.from_numpy(g[p + "timesteps"]), encode_noise=torch.from_numpy(g[p + "enc_noise"]))
    return batch, choices


def test_training_step_loss_and_gradients_match_the_reference(golden):
    from oracle import train as OT
    g = golden("g9_training_step")
    names = [str(n) for n in g["names"]]
    for ci in range(int(g["n"])):
        p = f"c{ci}_"
        den, vae, sch = build_oracle(g)
        assert abs(load_seeded(den, 500) - float(g[p + "checksum_denoiser"])) < 1e-6
        assert abs(load_seeded(vae, 501) - float(g[p + "checksum_vae"])) < 1e-6
        batch, ch = g9_case(g, ci)
        with torch.enable_grad():
            loss = OT.training_step(den, vae, sch, batch, **ch)
            loss.backward()
        assert abs(float(loss) - float(g[p + "loss"])) < 1e-5 * float(g[p + "loss"])
        want = dict(zip(names, g[p + "grad_norms"]))
        own = dict(den.named_parameters())
        assert sorted(own) == sorted(want)
        for n, prm in own.items():
            if want[n] < 0:
                assert prm.grad is None, n          # never in the graph (SD up-block transformers, mvunet.py:178)
            else:
                assert prm.grad is not None and abs(float(prm.grad.double().norm()) - want[n]) <= 2e-4 * want[n] + 1e-9, (n, want[n])
        for k in g.files:
            if k.startswith(p + "grad/"):
                got = own[k[len(p) + 5:]].grad.reshape(-1)
                got = got[::max(1, got.numel() // 2048)][:2048]
                ref = torch.from_numpy(g[k])
                assert float((got - ref).norm()) <= 2e-4 * float(ref.norm()) + 1e-9, k
        # zero-gradient participants: exactly the cross-attention to the all-zero context + its LayerNorm (6 per SD transformer block)
        zero = sorted(n for n in names if want[n] == 0)
        assert len(zero) == 42 and all((".attn2." in n or ".norm2." in n) and n.startswith("unet.") for n in zero)
