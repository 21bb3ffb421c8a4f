#!/usr/bin/env python3
"""HBM traffic of ONE optimizer step of `bench.py --train` from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tf -o p -- python3 bench.py --train --steps 2 --warmup 1 --no-profile --no-parity
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/tw -o p -- python3 bench.py --train --steps 2 --warmup 1 --no-profile --no-parity
  python3 tools/pmc_train_traffic.py <fetch csv> <write csv> profiles/pmc_traffic.json [key=train_bf16_b4_res256] [label]

The LAST optimizer step is the dispatches after the second-to-last AdamW launch up to and including the last one; inside it the
recorded plan (forward + loss + backward: what `training.roofline` times) runs from the `add_noise` launch to the first
gradient-norm launch (`sumsq_kernel`).  Units / corrections as tools/pmc_traffic.py (guide's HBM section): KiB, FETCH_SIZE x2 on gfx950.
"""
import csv
import json
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def family(name: str) -> str:
    for key, fam in (("wgrad_reduce", "wgrad_reduce"), ("wgrad", "wgrad"), ("attention_bwd", "attention_bwd"), ("attention", "attention"),
                     ("igemm_splitk_reduce", "igemm_splitk_reduce"), ("igemm", "igemm"), ("linear_p", "igemm"), ("linear_ws", "igemm"),
                     ("gn_bwd", "norm_bwd"), ("ln_bwd", "norm_bwd"), ("gn_", "groupnorm"), ("layernorm", "layernorm"), ("colsum", "colsum"),
                     ("adamw", "adamw"), ("sumsq", "grad_norm"), ("pack_", "repack")):
        if key in name:
            return fam
    return "other"


def rows_of(path, counter):
    rows = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
    rows.sort()
    return rows


def last_step(rows):
    ad = [i for i, r in enumerate(rows) if "adamw" in r[1]]
    if len(ad) < 2:
        raise SystemExit("need two optimizer steps in the trace")
    # (the vectorised AdamW is one launch per step; a scalar tail launch may follow it directly)
    ends = [i for k, i in enumerate(ad) if k + 1 == len(ad) or ad[k + 1] != i + 1]
    step = rows[ends[-2] + 1: ends[-1] + 1]
    a = next(i for i, r in enumerate(step) if "add_noise" in r[1])
    b = next(i for i, r in enumerate(step) if "sumsq" in r[1])
    return step, step[a:b]


def summarise(rows, scale):
    tot, cnt = defaultdict(float), defaultdict(int)
    for _, name, v in rows:
        tot[family(name)] += v * scale
        cnt[family(name)] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    key = sys.argv[4] if len(sys.argv) > 4 else "train_bf16_b4_res256"
    label = sys.argv[5] if len(sys.argv) > 5 else ""
    fs, fp = last_step(rows_of(fetch_csv, "FETCH_SIZE"))
    ws, wp = last_step(rows_of(write_csv, "WRITE_SIZE"))
    res = {}
    for tag, fr, wr in (("optimizer_step", fs, ws), ("plan", fp, wp)):
        ft, fc = summarise(fr, 2048.0)
        wt, wc = summarise(wr, 1024.0)
        res[tag] = {"dispatches": len(fr), "fetch_bytes": sum(ft.values()), "write_bytes": sum(wt.values()),
                    "hbm_bytes": sum(ft.values()) + sum(wt.values()),
                    "families": {f: {"launches": max(fc.get(f, 0), wc.get(f, 0)), "fetch_bytes": ft.get(f, 0.0), "write_bytes": wt.get(f, 0.0)}
                                 for f in sorted(set(ft) | set(wt))}}
    try:
        with open(out) as f:
            tab = json.load(f)
    except (OSError, ValueError):
        tab = {}
    tab[key] = {"collected": label, "note": "FETCH_SIZE KiB x2 (gfx950 correction), WRITE_SIZE KiB as reported; one optimizer step = the dispatches "
                                            "between two AdamW launches; plan = add_noise .. first grad-norm launch", **res}
    with open(out, "w") as f:
        json.dump(tab, f, indent=1)
    for tag in ("optimizer_step", "plan"):
        r = res[tag]
        print(f"{tag}: {r['dispatches']} dispatches, fetch {r['fetch_bytes'] / 1e9:.2f} GB, write {r['write_bytes'] / 1e9:.2f} GB")
        for f, v in sorted(r["families"].items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"])):
            print(f"   {f:22s} n={v['launches']:5d} fetch {v['fetch_bytes'] / 1e6:9.1f} MB  write {v['write_bytes'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
