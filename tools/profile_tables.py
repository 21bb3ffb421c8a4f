"""turn the rocprofv3 output of tools/rNN_profiles.sh into the tables committed under profiles/:
  rNN_kernel_stats.csv      per kernel template: dispatches, mean / total duration of ONE eager UNet + DDIM pass at 64 scenes (trial-free)
  rNN_mfma_util.json        per kernel template: SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES), VALU-active fraction, mean duration
  rNN_b1_timeline.json      one scene, graph replay: kernels per DDIM step, busy time (sum of kernel durations) against the wall time of a step
python3 tools/profile_tables.py <rocprof output dir> [round, e.g. 05]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
O = sys.argv[1]
RND = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("MVLDM_ROUND", "05")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:150]


def trace(sub):
    rows = []
    for path in glob.glob(f"{O}/{sub}/**/p_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            rows.append((short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda x: x[1])
    # bench.py --unet-pass-only brackets the pass with two `bessel_j0` launches: keep what lies between them
    marks = [i for i, r in enumerate(rows) if "bessel_j0" in r[0]]
    if len(marks) >= 2:
        rows = rows[marks[0] + 1:marks[-1]]
    return rows


# ---- 1. per-template stats of the 64-scene pass
rows = trace("b64")
if rows:
    agg = defaultdict(list)
    for n, s, e in rows:
        agg[n].append((e - s) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    with open(os.path.join(ROOT, "profiles", f"r{RND}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "mean_us", "total_us", "percent"])
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v), round(sum(v) / len(v), 2), round(sum(v), 1), round(100 * sum(v) / tot, 2)])
    print(f"b64: {len(rows)} dispatches, {tot / 1e3:.2f} ms of kernels, {len(agg)} kernel templates")
    # the roofline figure of the bench line, recomputed from this table alone: executed FLOPs of the implicit-GEMM family (bench.py prints
    # `flops_per_pass`) over the summed durations of its kernels in the traced pass
    fam = ("igemm_", "linear_p", "linear_ws", "linear_rs", "skinny_")
    t_fam = sum(sum(v) for n, v in agg.items() if any(k in n for k in fam))
    n_fam = sum(len(v) for n, v in agg.items() if any(k in n for k in fam))
    chk = {"igemm_family_dispatches": n_fam, "igemm_family_total_ms": round(t_fam / 1e3, 3), "all_kernels_ms": round(tot / 1e3, 3), "dispatches": len(rows)}
    bj = os.path.join(ROOT, "gpurun_out", f"r{RND}_bench.json")
    if os.path.exists(bj):
        try:
            line = [x for x in open(bj) if x.startswith("{")][-1]
            fl = json.loads(line)["roofline"]["flops_per_pass"]
            chk.update(flops_per_pass=fl, achieved_tflops=round(fl / (t_fam * 1e-6) / 1e12, 1), frac_of_2500=round(fl / (t_fam * 1e-6) / 2.5e15, 4))
        except Exception as e:      # noqa: BLE001
            chk["bench_line"] = f"unreadable: {e}"
    json.dump(chk, open(os.path.join(ROOT, "profiles", f"r{RND}_roofline_check.json"), "w"), indent=1)
    print("roofline check:", chk)

# ---- 2. MFMA utilisation per template
pm = defaultdict(lambda: defaultdict(list))
for path in glob.glob(f"{O}/pmc/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        pm[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for n, s, e in trace("pmc"):
    dur[n].append((e - s) / 1e3)
if pm:
    out = {}
    for n, cs in pm.items():
        m = {k: sum(v) / len(v) for k, v in cs.items()}
        e = {"dispatches": max(len(v) for v in cs.values()), "mean_duration_us": round(sum(dur[n]) / max(len(dur[n]), 1), 2),
             "total_ms": round(sum(dur[n]) / 1e3, 3)}
        if m.get("SQ_BUSY_CU_CYCLES"):
            e["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * m["SQ_BUSY_CU_CYCLES"]), 4)
            e["valu_active_frac"] = round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / m["SQ_BUSY_CU_CYCLES"], 4)
        out[n] = e
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms"]))
    json.dump({"note": "one eager UNet + DDIM pass at 64 scenes, bf16, plans recorded from the round's tune cache (no tile trial in the "
                       "traced process); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES), mean over the template's dispatches",
               "kernels": out}, open(os.path.join(ROOT, "profiles", f"r{RND}_mfma_util.json"), "w"), indent=1)
    print("pmc:", len(out), "templates")

# ---- 3. one scene: busy vs wall per DDIM step (graph replay)
rows = trace("b1")
if rows:
    # the last 20 replays: find the step period from the repeating DDIM-step kernel
    marks = [i for i, r in enumerate(rows) if "ddim_kernel" in r[0]]
    steps = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a + 1:b + 1]
        if not seg:
            continue
        wall = (seg[-1][2] - rows[a][2]) / 1e3
        busy = sum((e - s) / 1e3 for _, s, e in seg)
        steps.append((len(seg), wall, busy))
    steps = steps[-20:]
    if steps:
        n = len(steps)
        res = {"scenes": 1, "replays_counted": n, "kernels_per_step": round(sum(s[0] for s in steps) / n, 1),
               "wall_us_per_step": round(sum(s[1] for s in steps) / n, 1), "busy_us_per_step": round(sum(s[2] for s in steps) / n, 1)}
        res["gap_us_per_step"] = round(res["wall_us_per_step"] - res["busy_us_per_step"], 1)
        res["mean_gap_us_per_kernel"] = round(res["gap_us_per_step"] / max(res["kernels_per_step"], 1), 2)
        seg = rows[marks[-2] + 1:marks[-1] + 1]
        per = defaultdict(list)
        for nme, s, e in seg:
            per[nme].append((e - s) / 1e3)
        res["by_template_last_step"] = {k: {"n": len(v), "total_us": round(sum(v), 1), "mean_us": round(sum(v) / len(v), 2)}
                                        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:25]}
        json.dump(res, open(os.path.join(ROOT, "profiles", f"r{RND}_b1_timeline.json"), "w"), indent=1)
        print("b1:", {k: v for k, v in res.items() if k != "by_template_last_step"})
