#!/bin/bash
# L2 hit rate and fabric-side fetch bytes per kernel over any tool command (separate PMC passes, --kernel-trace only):
#   tools/pmc_l2.sh <tag> python3 tools/linear_one.py L0.geglu 12
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcl2_$tag
rm -rf $out
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/a -o p -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/b -o p -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/c -o p -- "$@" > /dev/null 2>&1
python3 - "$out" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
res = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        res[re.sub(r"\(.*", "", r["Kernel_Name"])[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in res.items():
    m = {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in cs.items() if len(v) > 1}
    if not m or m.get("TCC_REQ_sum", 0) < 1e5:
        continue
    hit = m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
    # FETCH_SIZE in KiB, gfx950: 128-byte requests tallied as 64 -> doubled (MI355X_MICROARCH.md, HBM)
    print(name, f"L2 hit {hit:.3f}  req {m.get('TCC_REQ_sum', 0) / 1e6:.1f}M  fetch {m.get('FETCH_SIZE', 0) * 2 * 1024 / 1e9:.3f} GB  write {m.get('WRITE_SIZE', 0) * 1024 / 1e9:.3f} GB")
PY
