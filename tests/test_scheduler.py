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
