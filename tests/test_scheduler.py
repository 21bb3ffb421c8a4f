"""CPU: DDIM tables/timesteps (G6, bit-exact INT + fp32) for the oracle scheduler and the product's
host-side scheduler (`mv_ldm_amd.scheduler.DDIMScheduler`, which builds the same tables with torch
on the host and only runs the elementwise update on the device)."""
import numpy as np
import torch

from oracle.scheduler import DDIMScheduler as OracleDDIM


def test_timesteps_closed_form_and_golden(golden):
    g = golden("g6_ddim")
    s = OracleDDIM(clip_sample=False)
    for n in (5, 25, 50, 70):
        s.set_timesteps(n)
        ratio = 1000 // n
        assert s.timesteps.dtype == torch.int64
        assert s.timesteps.tolist() == [ratio * i for i in range(n - 1, -1, -1)]
        assert np.array_equal(s.timesteps.numpy(), g[f"timesteps_{n}"])
    s.set_timesteps(50)
    assert s.timesteps[:3].tolist() == [980, 960, 940] and int(s.timesteps[-1]) == 0
    s.set_timesteps(5)
    assert s.timesteps.tolist() == [800, 600, 400, 200, 0]


def test_alpha_table_vs_fp64_and_golden(golden):
    g = golden("g6_ddim")
    s = OracleDDIM(clip_sample=False)
    assert np.array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])       # bit-exact fp32 table
    b64 = np.linspace(1e-4, 0.02, 1000, dtype=np.float64)
    ac64 = np.cumprod(1 - b64)
    assert np.abs(s.alphas_cumprod.double().numpy() / ac64 - 1).max() < 5e-5  # fp32 cumprod drift only
    assert s.init_noise_sigma == 1.0 and float(s.final_alpha_cumprod) == 1.0


def test_step_kat(golden):
    g = golden("g6_ddim")
    s = OracleDDIM(clip_sample=False)
    s.set_timesteps(50)
    x, e = torch.from_numpy(g["kat_x"]), torch.from_numpy(g["kat_eps"])
    ac = s.alphas_cumprod.double()
    for t in (980, 500, 0):
        out = s.step(e, torch.tensor(t), x).prev_sample
        assert np.array_equal(out.numpy(), g[f"kat_prev_{t}"])
        a_t, a_p = ac[t], (ac[t - 20] if t >= 20 else torch.tensor(1.0, dtype=torch.float64))
        x0 = (x.double() - (1 - a_t).sqrt() * e.double()) / a_t.sqrt()
        ref = a_p.sqrt() * x0 + (1 - a_p).sqrt() * e.double()
        assert (out.double() - ref).abs().max() < 1e-5
    assert np.array_equal(s.add_noise(x, e, torch.tensor([10, 900])).numpy(), g["kat_add_noise"])


def test_ddpm_step_vs_fp64_and_product_coefficients():
    """the restated DDPMScheduler.step (epsilon, fixed_small) against Ho et al.'s posterior in fp64, and the five per-step scalars
    the product's host-side class hands to the HIP kernel against the oracle's own intermediates (bit-identical fp32)"""
    from mv_ldm_amd.scheduler import DDPMScheduler as HostDDPM
    from oracle.scheduler import DDPMScheduler as OracleDDPM
    g = torch.Generator().manual_seed(3)
    x, e, z = (torch.randn(2, 3, 4, 8, 8, generator=g) for _ in range(3))
    for clip in (False, True):
        s, h = OracleDDPM(clip_sample=clip), HostDDPM(clip_sample=clip)
        s.set_timesteps(50)
        h.set_timesteps(50)
        assert s.timesteps.tolist() == h.timesteps.tolist()
        ac = s.alphas_cumprod.double()
        for t in (980, 500, 20, 0):
            out = s.step(e, torch.tensor(t), x, variance_noise=z).prev_sample
            a_t = ac[t]
            a_p = ac[t - 20] if t >= 20 else torch.tensor(1.0, dtype=torch.float64)
            alpha = a_t / a_p
            beta = 1 - alpha
            x0 = (x.double() - (1 - a_t).sqrt() * e.double()) / a_t.sqrt()
            if clip:
                x0 = x0.clamp(-1, 1)
            mean = a_p.sqrt() * beta / (1 - a_t) * x0 + alpha.sqrt() * (1 - a_p) / (1 - a_t) * x.double()
            var = ((1 - a_p) / (1 - a_t) * beta).clamp(min=1e-20)
            ref = mean + (var.sqrt() * z.double() if t > 0 else 0)
            assert (out.double() - ref).abs().max() < 2e-5, (t, clip)
            c = h.step_coefficients(t)
            a32, p32 = s.alphas_cumprod[t], (s.alphas_cumprod[t - 20] if t >= 20 else s.one)
            want = torch.stack([(1 - a32) ** 0.5, a32 ** 0.5, (p32 ** 0.5 * (1 - a32 / p32)) / (1 - a32),
                                (a32 / p32) ** 0.5 * (1 - p32) / (1 - a32),
                                s._get_variance(t) ** 0.5 if t > 0 else torch.tensor(0.0)])
            assert torch.equal(c, want), (t, c, want)
        assert h.clip_range == (1.0 if clip else 0.0)
    # no-noise last step: t = 0 adds nothing and the posterior collapses onto x0
    s = OracleDDPM(clip_sample=False)
    s.set_timesteps(1000)
    o = s.step(e, 0, x)
    assert torch.allclose(o.prev_sample, o.pred_original_sample, atol=1e-6)
