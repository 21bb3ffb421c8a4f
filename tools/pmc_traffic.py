#!/usr/bin/env python3
"""Summarise HBM traffic per kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/p_counter_collection.csv gpurun_out/pmc_write/p_counter_collection.csv out.json [192] [key] [label]
`out.json` is keyed by workload (`key`, default bf16_b64_res256 = what `bench.py` looks up: "<dtype>_b<scenes>_res<res>"); an existing
file keeps its other keys.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes, so it is doubled; WRITE_SIZE is taken as reported
(uncalibrated per the guide).  Output: per kernel family the launch count and mean bytes per launch.
"""
import csv
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def family(name: str) -> str:
    if "linear_pp_kernel" in name or "linear_pw_kernel" in name or "linear_ws_kernel" in name or "igemm_halo_kernel" in name:      # tiles 12 / 13 / 14 / 11 of the same implicit-GEMM family
        return "igemm"
    for key in ("igemm_bl_kernel", "igemm_kernel", "igemm_splitk_reduce", "attention_kernel",
                "attention_wide_kernel", "attention_dsplit_kernel", "gn_fused_kernel", "gn_apply_kernel", "gn_stats_kernel", "layernorm", "eltwise", "ddim",
                "timestep_embed", "pack_weight", "nchw_to_nhwc", "nhwc_to_nchw"):
        if key in name:
            return "igemm" if key.startswith("igemm_") and key != "igemm_splitk_reduce" else key
    return "other"


def collect(path: str, counter: str, igemm_last: int = 0):
    """igemm_last > 0: keep only the last `igemm_last` igemm dispatches (the final eager UNet pass; the
    dispatches before it are the recording pass and the plan-time tile trials)"""
    rows = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                rows.append((int(row["Dispatch_Id"]), family(row["Kernel_Name"]), float(row["Counter_Value"])))
    rows.sort()
    if igemm_last > 0:
        ig = [r for r in rows if r[1] == "igemm"]
        first_kept = ig[-igemm_last][0] if len(ig) >= igemm_last else 0
        rows = [r for r in rows if r[0] >= first_kept]
    tot, cnt = defaultdict(float), defaultdict(int)
    for _, fam, val in rows:
        tot[fam] += val
        cnt[fam] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    igemm_last = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    key = sys.argv[5] if len(sys.argv) > 5 else "bf16_b64_res256"
    label = sys.argv[6] if len(sys.argv) > 6 else ""
    ft, fc = collect(fetch_csv, "FETCH_SIZE", igemm_last)
    wt, wc = collect(write_csv, "WRITE_SIZE", igemm_last)
    res = {}
    for fam in sorted(set(ft) | set(wt)):
        n = max(fc.get(fam, 0), wc.get(fam, 0))
        fetch_b = ft.get(fam, 0.0) * 1024.0 * 2.0   # KiB -> bytes, gfx950 x2 correction
        write_b = wt.get(fam, 0.0) * 1024.0
        res[fam] = {"launches": n, "fetch_bytes_per_launch": fetch_b / max(fc.get(fam, 1), 1),
                    "write_bytes_per_launch": write_b / max(wc.get(fam, 1), 1),
                    "hbm_bytes_per_launch": fetch_b / max(fc.get(fam, 1), 1) + write_b / max(wc.get(fam, 1), 1),
                    "fetch_bytes_total": fetch_b, "write_bytes_total": write_b}
    try:
        with open(out) as f:
            tab = json.load(f)
        if "families" in tab:       # pre-round-3 un-keyed file
            tab = {}
    except (OSError, ValueError):
        tab = {}
    tab[key] = {"collected": label, "note": "FETCH_SIZE KiB x2 (gfx950 correction), WRITE_SIZE KiB as reported", "families": res}
    with open(out, "w") as f:
        json.dump(tab, f, indent=1)
    for fam, r in res.items():
        print(f"{fam:24s} n={r['launches']:6d} fetch/launch={r['fetch_bytes_per_launch']/1e6:9.2f} MB "
              f"write/launch={r['write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
