"""compare two per-op tables of the same plan (bench.py --op-table): python3 tools/optable_ab.py old.json new.json"""
import collections, json, re, sys
a = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
g = collections.defaultdict(lambda: [0.0, 0.0, 0])
for x, y in zip(a, b):
    assert x["name"] == y["name"]
    parts = x["name"].split("/")
    k = re.sub(r"\d+", "#", parts[-1]) + "@" + (parts[1].split(".")[0] if len(parts) > 2 else "")
    g[k][0] += x["ms"]; g[k][1] += y["ms"]; g[k][2] += 1
print("total", round(sum(x["ms"] for x in a), 2), round(sum(y["ms"] for y in b), 2))
for k, (o, n, c) in sorted(g.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{k:45s} n={c:3d} old {o:7.3f} new {n:7.3f}  {n / o:.3f}")
