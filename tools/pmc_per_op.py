#!/usr/bin/env python3
"""HBM-side traffic PER implicit-GEMM launch of one UNet pass, joined with the op table (names, algorithmic bytes):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python3 bench.py --unet-pass-only      (and WRITE_SIZE in a second pass)
  python tools/pmc_per_op.py fetch.csv write.csv optable.json out.json
The last len(igemm ops) igemm-family dispatches of each pass are the final eager pass, in plan order."""
import csv, json, sys
csv.field_size_limit(1 << 30)


def fam(name):
    return any(k in name for k in ("igemm_bl_kernel", "igemm_kernel", "igemm_halo_kernel", "linear_pp_kernel"))


def load(path, counter):
    rows = []
    for r in csv.DictReader(open(path, newline="")):
        if r["Counter_Name"] == counter and fam(r["Kernel_Name"]):
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r["Kernel_Name"]))
    rows.sort()
    return rows


fetch, write, optable, out = sys.argv[1:5]
ops = [o for o in json.load(open(optable)) if o["kind"] == 1]
f, w = load(fetch, "FETCH_SIZE")[-len(ops):], load(write, "WRITE_SIZE")[-len(ops):]
res = []
for o, (_, fv, kn), (_, wv, _) in zip(ops, f, w):
    hbm = fv * 1024 * 2 + wv * 1024            # KiB; FETCH_SIZE x2 (gfx950 correction)
    res.append({"name": o["name"], "ms": o["ms"], "algorithmic_bytes": o["bytes"], "hbm_bytes": hbm, "ratio": hbm / max(o["bytes"], 1),
                "excess_mb": (hbm - o["bytes"]) / 1e6, "kernel": kn[:80]})
json.dump(res, open(out, "w"), indent=0)
tot_a, tot_h = sum(r["algorithmic_bytes"] for r in res), sum(r["hbm_bytes"] for r in res)
print(f"{len(res)} launches: algorithmic {tot_a / 1e9:.1f} GB, measured {tot_h / 1e9:.1f} GB ({tot_h / tot_a:.2f}x)")
for r in sorted(res, key=lambda r: -r["excess_mb"])[:25]:
    print(f"{r['name'][5:65]:60s} {r['ms']:.3f} ms  alg {r['algorithmic_bytes'] / 1e6:7.0f} MB  hbm {r['hbm_bytes'] / 1e6:7.0f} MB  x{r['ratio']:.2f}")
