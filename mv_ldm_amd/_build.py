"""Build libmvldm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libmvldm_hip.so"
SOURCES = ["api.cpp", "plan.cpp", "igemm.hip", "attention.hip", "norm.hip", "misc.hip",
           "wgrad.hip", "attention_bwd.hip", "norm_bwd.hip", "train_misc.hip", "linear_pp.hip", "linear_pw.hip", "linear_ws.hip", "linear_rs.hip", "skinny.hip"]
HEADERS = [CSRC / "common.h", CSRC / "reduce.h", CSRC.parent.parent / "include" / "mvldm.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
         "-Rpass-analysis=kernel-resource-usage"]          # per-kernel registers / scratch -> csrc/kernel_resources.json
RES = CSRC / "kernel_resources.json"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    flags = list(FLAGS)
    if os.environ.get("MVLDM_EXPERIMENTS") == "1":      # tools/ only: compiles the MVLDM_IGEMM_FAKE roofline knobs in
        flags.append("-DMVLDM_EXPERIMENTS")
    objs, jobs = [], []
    for src in SOURCES:
        s = CSRC / src
        o = CSRC / (s.stem + ".o")
        objs.append(o)
        if force or _stale(o, [s, *HEADERS]):
            cmd = [hipcc, *flags, "-x", "hip", "-c", str(s), "-o", str(o)]
            jobs.append(cmd)
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
            Path(cmd[-1]).with_suffix(".res.json").write_text(json.dumps(_resources(r.stderr), indent=0))
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    parts = [o.with_suffix(".res.json") for o in objs]
    if all(q.exists() for q in parts) and (not RES.exists() or _stale(RES, parts)):
        merged = {}
        for q in parts:
            merged.update(json.loads(q.read_text()))
        RES.write_text(json.dumps(merged, indent=0, sort_keys=True))
    return LIB


_FIELDS = {"VGPRs": "vgpr", "AGPRs": "agpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch", "VGPRs Spill": "vgpr_spill",
           "Occupancy [waves/SIMD]": "waves_per_simd"}


def _resources(stderr: str) -> dict:
    """kernel name -> {vgpr, agpr, sgpr, scratch, vgpr_spill, waves_per_simd} from hipcc's kernel-resource-usage remarks.
    tests/test_build_resources.py keeps the hot kernels at zero scratch: an accumulator array demoted to scratch costs 3-4x
    and no parity test notices."""
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in _FIELDS and v.strip().lstrip("-").isdigit():
                cur[_FIELDS[k.strip()]] = int(v)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
