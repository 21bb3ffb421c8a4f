"""G11: the shapes of BASELINE.json configs[0] and configs[4] through the REFERENCE's own DiffusionWrapper.sample, imported in place
from /root/reference (tests/golden/ref_import.py) at reduced width -- run HERE, commit the vectors:

    python tests/golden/make_golden_configs.py

  c0  configs[0] literally: 1 scene, 1 context + 1 target view, 64x64 images, a VAE with three downsamples -> 8x8 latents,
      5 DDIM steps, CFG 3.0 (src/scripts/generate_mvldm.py on the CPU reference)
  c1  configs[4]'s geometry: 1 + 8 = 9 views per scene and 64x64 LATENTS (128x128 images through a one-downsample VAE), CFG 3.0, 2 DDIM
      steps: above the `h <= 32` gate of the reference's walk (src/model/denoiser/mvunet.py:137,190) the level-0 multi-view blocks are
      skipped, the 3-D attention of the deeper levels runs over 9 x 32 x 32 tokens
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))
import ref_import as R  # noqa: E402
from make_golden import _ref_wrapper, save  # noqa: E402
from seeded import load_seeded, random_cameras  # noqa: E402


def run(ci, out, b, v_c, v_t, H, vae_widths, n_steps, seed):
    w = _ref_wrapper(True, widths=(64, 128, 256, 256), vae_widths=vae_widths, n_steps=n_steps)
    cs_d = load_seeded(w.denoiser, 400)
    cs_v = load_seeded(w.autoencoder, 401)
    hl = H // 2 ** (len(vae_widths) - 1)
    g = torch.Generator().manual_seed(seed)
    ctx_img = torch.rand(b, v_c, 3, H, H, generator=g)
    extr, intr = random_cameras(b, v_c + v_t, seed=seed + 6)
    enc_noise = torch.randn(b * v_c, 4, hl, hl, generator=g)
    x_T = torch.randn(b, v_t, 4, hl, hl, generator=g)
    batch = {"context": {"image": ctx_img, "extrinsics": extr[:, :v_c], "intrinsics": intr[:, :v_c]},
             "target": {"image": torch.zeros(b, v_t, 3, H, H), "extrinsics": extr[:, v_c:], "intrinsics": intr[:, v_c:]},
             "scene": ["synthetic"] * b}
    w.set_timesteps(n_steps)
    draws = [enc_noise, x_T]
    o_randn = torch.randn

    def fake_randn(*shape, **kw):      # the reference draws from the global CPU generator (diffusion_wrapper.py:283,473)
        t = draws.pop(0)
        shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert tuple(t.shape) == shp, (t.shape, shp)
        return t.clone()
    torch.randn = fake_randn
    try:
        with R.cpu_cuda():
            img, _ = w.sample(batch)
    finally:
        torch.randn = o_randn
    assert not draws
    p = f"c{ci}_"
    out.update({p + "checksum_denoiser": cs_d, p + "checksum_vae": cs_v, p + "ctx_img": ctx_img.numpy(), p + "extr": extr.numpy(),
                p + "intr": intr.numpy(), p + "enc_noise": enc_noise.numpy(), p + "x_T": x_T.numpy(),
                p + "img": img.detach().numpy().astype(np.float16),       # images in [0, 1]: 2^-11 absolute is below every tolerance that reads them
                p + "vae_widths": np.array(vae_widths), p + "n_steps": n_steps})
    print(f"  g11 c{ci}: b={b} v_c={v_c} v_t={v_t} H={H} latents {hl}x{hl} steps={n_steps}: img mean {float(img.mean()):.4f}")


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    out = {}
    run(0, out, 1, 1, 1, 64, (32, 32, 64, 64), 5, 21)
    run(1, out, 1, 1, 8, 128, (32, 64), 2, 22)
    out["n"] = 2
    out["widths"] = np.array([64, 128, 256, 256])
    save("g11_configs", **out)
