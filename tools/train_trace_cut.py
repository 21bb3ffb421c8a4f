"""per-kernel time of STEADY-STATE training steps from a rocprofv3 kernel trace of `bench.py --train` (tools/train_stats.sh): the trace is
cut into optimizer steps at the AdamW launches and the steps of the fixed-shape timed loop (the last steps of the trace whose dispatch count is within 1 % of
the median: the tuner's trial launches are over by then) are aggregated.  python3 tools/train_trace_cut.py <kernel_trace.csv> <out.csv>"""
import csv
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
rows = []
for r in csv.DictReader(open(sys.argv[1], newline="")):
    n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:120]
    rows.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda x: x[1])
# optimizer steps: maximal runs of adamw launches end a step
steps, cur, in_adam = [], [], False
for r in rows:
    is_adam = "adamw" in r[0]
    if in_adam and not is_adam:
        steps.append(cur)
        cur = []
    cur.append(r)
    in_adam = is_adam
# (the AdamW buckets interleave with the next window's VAE encode on the side stream: fold the short segments into the step that follows)
merged, carry = [], []
for s in steps:
    carry = carry + s
    if len(s) >= 500:
        merged.append(carry)
        carry = []
steps = merged
lens = [len(s) for s in steps]
print("optimizer steps in the trace:", len(steps), "dispatches per step (last 40):", lens[-40:])
# the steady state: the most common step length among the last 40 steps
from collections import Counter
tail = steps[-40:]
common = sorted(len(s) for s in tail)[len(tail) // 2]
sel = [s for s in tail if abs(len(s) - common) <= 0.01 * common]
print(f"aggregating {len(sel)} steps of {common} dispatches")
agg = defaultdict(list)
wall = []
for s in sel:
    wall.append((s[-1][2] - s[0][1]) / 1e6)
    for n, a, b in s:
        agg[n].append((b - a) / 1e3)
k = len(sel)
tot = sum(sum(v) for v in agg.values()) / k
print(f"wall per step {sum(wall) / k:.2f} ms, kernel time per step {tot / 1e3:.2f} ms")
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches_per_step", "mean_us", "total_us_per_step", "percent"])
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([n, round(len(v) / k, 1), round(sum(v) / len(v), 2), round(sum(v) / k, 1), round(100 * sum(v) / k / tot, 2)])
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"{n[:100]:100s} {len(v) / k:7.1f} x {sum(v) / len(v):8.1f} us = {sum(v) / k / 1e3:6.2f} ms")
