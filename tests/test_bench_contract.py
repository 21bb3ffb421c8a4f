"""GPU: `bench.py` prints ONE JSON line that honours the driver's contract (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, `roofline` with bound / achieved / peak /
unit / frac / traffic) -- run as the driver runs it (a child process, N = 1), at a reduced size so that it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # exactly one line on stdout
    return json.loads(lines[0])


def test_sampling_line_contract():
    d = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--scenes", "2", "--ddim-steps", "4", "--no-cpu-baseline",
                  "--no-small-batch", "--no-alt-dtype", "--no-train-line", "--no-parity")
    assert d["metric"].startswith("denoised views/sec") and d["unit"] == "views/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and "synthetic" in d["data"] and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]      # views of all scenes / time
    rf = d["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] in ("TFLOP/s", "GB/s") and rf["peak"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    es = d["exact_sharing"]
    assert es["enabled"] is True and es["full_walk"]["value"] > 0 and es["full_walk"]["unit"] == "views/s"


def test_training_line_contract():
    d = run_bench("--gpus", "1", "--train", "--steps", "2", "--warmup", "1", "--scenes", "1")
    assert d["metric"].startswith("training views/sec") and d["unit"] == "views/s" and d["n_gpus"] == 1 and d["steps"] == 2
    assert d["value"] > 0 and d["dtype"] == "bf16" and d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["loss_first_last"][0] > 0 and d["grad_rel_err"]["grad_rel_l2"] < 1e-1      # one scene (4 views) after the randomised leg's 32 optimizer steps: 0.045 - 0.065 measured over two boxes (tile choices differ per box); 64-view windows: 0.013
