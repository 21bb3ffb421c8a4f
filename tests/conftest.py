"""pytest configuration: markers + shared helpers (test infrastructure)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

# The plan-time tuner times small launches behind a 640 MB cache-flushing fill (plan._TUNE_COLD): right for the product, but it doubles
# the wall time of a suite that records hundreds of tiny plans.  The suite runs the hot trials; ONE test turns the cold path on
# (tests/test_hip_headline.py::test_cold_cache_tuning_path).  Must be set before mv_ldm_amd.plan is imported.
os.environ.setdefault("MVLDM_TUNE_COLD", "0")

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
for p in (str(ROOT), str(GOLDEN)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # tests build random-init modules on purpose (no checkpoints offline): the product's "RANDOM initial weights" warning is noise here
    config.addinivalue_line("filterwarnings", "ignore:.*RANDOM initial weights.*:UserWarning")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is present and they were not deselected
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _grad_mode(request):
    """autograd mode is per test MODULE, never a process-wide side effect of importing a test file.  Default off (forward-only
    suites would otherwise keep graphs of billion-parameter models alive); the backward / training suites, which differentiate
    torch references, say `GRAD_ENABLED = True` at module level"""
    with torch.set_grad_enabled(getattr(request.module, "GRAD_ENABLED", False)):
        yield


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    return load


# ---- measured-error report: tests call `record_err(name, value)` next to the assertion that bounds the value; with
# MVLDM_TEST_REPORT=<path> the session writes {name: max value seen} there.  The 16-bit tolerances of the GPU suites are set from
# such a run (tests/golden/measured_errors_r04.json: 2 x the measured maximum), not from round numbers.
_MEASURED = {}


def record_err(name: str, value: float) -> float:
    _MEASURED[name] = max(float(value), _MEASURED.get(name, 0.0))
    return float(value)


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("MVLDM_TEST_REPORT")
    if path and _MEASURED:
        import json
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        for k, v in _MEASURED.items():
            old[k] = max(v, old.get(k, 0.0))
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max())
