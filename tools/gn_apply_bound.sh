#!/bin/bash
# The bound of "GroupNorm statistics from the producer's epilogue": what the APPLY-ONLY pass costs (statistics given), next to the
# fused single-pass kernel, per shape of the step at 64 scenes.  Needs libmvldm_hip_exp_gn.so (tools/gn_probe.sh: norm.hip with the
# MVLDM_GN_TWOPASS knob).    gpurun -- 'bash tools/gn_apply_bound.sh'   -> gpurun_out/gn_apply_bound.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=/tmp/gnab; rm -rf $O; mkdir -p gpurun_out
python3 tools/gn_time.py 64 --lib-suffix _gn > gpurun_out/gn_apply_bound.txt 2>&1
MVLDM_GN_TWOPASS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o p -- python3 tools/gn_time.py 64 --lib-suffix _gn >> gpurun_out/gn_apply_bound.txt 2>&1
python3 - $O >> gpurun_out/gn_apply_bound.txt <<'PY'
import csv, glob, sys
csv.field_size_limit(1 << 30)
rows = []
for path in glob.glob(sys.argv[1] + "/**/p_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        if "gn_" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
# the script runs 11 GroupNorms per shape, in the shape order of tools/gn_time.py
names = ["L0.320", "L0.640+320", "L1.640", "L1.1280+640", "L2.1280", "L2.1280+1280", "L3.1280"]
shape_bytes = {"L0.320": 576 * 1024 * 320, "L0.640+320": 576 * 1024 * 960, "L1.640": 576 * 256 * 640, "L1.1280+640": 576 * 256 * 1920,
               "L2.1280": 576 * 64 * 1280, "L2.1280+1280": 576 * 64 * 2560, "L3.1280": 576 * 16 * 1280}
kinds = sorted({k for _, k, _ in rows})
print("kernels seen:", kinds)
per = len(rows) // len(names)
for i, nm in enumerate(names):
    seg = rows[i * per:(i + 1) * per]
    by = {}
    for _, k, d in seg:
        by.setdefault(k, []).append(d)
    line = f"{nm:14s}"
    for k, v in by.items():
        v = v[len(v) // 10:]       # (drop the first call of the shape)
        mean = sum(v) / len(v)
        gbs = 2.0 * shape_bytes[nm] * 2 / mean / 1e3 if "apply" in k else 0.0
        line += f"  {k.split('::')[-1][:28]}: {mean:7.1f} us" + (f" ({gbs:.0f} GB/s r+w)" if gbs else "")
    print(line)
PY
