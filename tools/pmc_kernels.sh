#!/bin/bash
# SQ / LDS counters per kernel over any tool command (two rocprofv3 PMC passes; --kernel-trace only, no other trace domains):
#   tools/pmc_kernels.sh <tag> python3 tools/attn_one.py 40 8 64
# writes gpurun_out/pmc_<tag>.json: per kernel name the mean counters of its dispatches (first dispatch of each dropped: warm-up).
# The program itself must follow `--` in the rocprofv3 line (no env / bash -c hops: see the GPU-box rules).
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/a -o p -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $out/b -o p -- "$@" > /dev/null 2>&1
python3 - "$out" "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.json" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
out, dst = sys.argv[1:3]
res = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in sorted(glob.glob(out + "/*")):
    f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    for path in f:
        for r in csv.DictReader(open(path, newline="")):
            name = re.sub(r"\(.*", "", r["Kernel_Name"])[:110]
            res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for path in glob.glob(d + "/**/p_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            dur[re.sub(r"\(.*", "", r["Kernel_Name"])[:110]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
summary = {}
for name, cs in res.items():
    m = {k: (sum(v[1:]) / max(len(v) - 1, 1) if len(v) > 1 else v[0]) for k, v in cs.items()}
    n = max(len(v) for v in cs.values())
    e = {"dispatches": n, **{k: round(v) for k, v in m.items()}}
    if name in dur:
        dd = dur[name]
        e["mean_duration_us"] = round(sum(dd[1:]) / max(len(dd) - 1, 1) / 1e3, 2) if len(dd) > 1 else round(dd[0] / 1e3, 2)
    wc = m.get("SQ_WAVE_CYCLES")          # quad-cycles (MI355X_MICROARCH.md); SQ_VALU_MFMA_BUSY_CYCLES counts cycles
    if m.get("SQ_BUSY_CU_CYCLES") and m.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        # per-CU busy cycles summed over CUs; the matrix pipes are per SIMD (4 per CU)
        e["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_BUSY_CU_CYCLES"]), 4)
        e["valu_active_frac"] = round(4.0 * m.get("SQ_ACTIVE_INST_VALU", 0) / (4.0 * m["SQ_BUSY_CU_CYCLES"]), 4)
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in m:
                e[k.lower() + "_frac_of_wave_cycles"] = round(m[k] / wc, 4)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    if m.get("SQ_INSTS_MFMA"):
        e["valu_insts_per_mfma"] = round(m.get("SQ_INSTS_VALU", 0) / m["SQ_INSTS_MFMA"], 2)
    summary[name] = e
json.dump(summary, open(dst, "w"), indent=1)
for name, e in sorted(summary.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    print(name[:100], {k: v for k, v in e.items() if "frac" in k or k in ("dispatches", "valu_insts_per_mfma", "mean_duration_us")})
PY
