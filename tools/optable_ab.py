"""compare two per-op tables of the same plan (bench.py --op-table), batch-split remainders folded into their launch: python3 tools/optable_ab.py old.json new.json [rows]"""
import collections, json, re, sys


def table(path):
    t = collections.OrderedDict()
    for r in json.load(open(path)):
        n = r["name"][:-6] if r["name"].endswith("[rest]") else r["name"]
        t[n] = t.get(n, 0.0) + r["ms"]
    return t


a, b = table(sys.argv[1]), table(sys.argv[2])
assert list(a) == list(b), "different plans"
g = collections.defaultdict(lambda: [0.0, 0.0, 0])
for n in a:
    parts = n.split("/")
    k = re.sub(r"\d+", "#", parts[-1]) + "@" + (parts[1].split(".")[0] if len(parts) > 2 else "")
    g[k][0] += a[n]; g[k][1] += b[n]; g[k][2] += 1
print("total", round(sum(a.values()), 2), round(sum(b.values()), 2))
for k, (o, n, c) in sorted(g.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{k:45s} n={c:3d} old {o:7.3f} new {n:7.3f}  {n / o:.3f}")
