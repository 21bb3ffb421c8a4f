"""post-process a rocprofv3 --kernel-trace CSV of tools/step_trace.py: kernels of the LAST DDIM step in launch order with their durations,
the per-template summary, busy vs wall.   python3 tools/b1_timeline.py <dir with *_kernel_trace.csv> <out.json>"""
import csv, glob, json, re, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
d, out = sys.argv[1], sys.argv[2]
rows = []
for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[1])
marks = [i for i, r in enumerate(rows) if "ddim_kernel" in r[0]]
seg = rows[marks[-2] + 1:marks[-1] + 1]


def short(n):
    n = re.sub(r"^_ZN5mvldm\d+", "", n)
    n = re.sub(r"NS_\d+SkCfgI|NS_\d+IwCfgI", "<", n)
    n = re.sub(r"EvNS_\d+\w+Params?E$", "", n)
    n = n.replace("IDF16b", "<bf16,").replace("Li", ",").replace("ELb", ",b").replace("E", "")
    return n[:90]


steps = []
for a, b in zip(marks[:-1], marks[1:]):
    s = rows[a + 1:b + 1]
    if s:
        steps.append((len(s), (s[-1][2] - rows[a][2]) / 1e3, sum((e - st) / 1e3 for _, st, e in s)))
steps = steps[-20:]
per = defaultdict(list)
for n, s, e in seg:
    per[short(n)].append((e - s) / 1e3)
res = {"kernels_per_step": len(seg), "wall_us_per_step": round(sum(s[1] for s in steps) / len(steps), 1),
       "busy_us_per_step": round(sum(s[2] for s in steps) / len(steps), 1),
       "by_template": {k: {"n": len(v), "total_us": round(sum(v), 1), "mean_us": round(sum(v) / len(v), 2)} for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))},
       "sequence": [[short(n), round((e - s) / 1e3, 2)] for n, s, e in seg]}
json.dump(res, open(out, "w"), indent=0)
print({k: v for k, v in res.items() if k not in ("by_template", "sequence")})
for k, v in list(res["by_template"].items())[:30]:
    print(f"{k:92s} {v}")
