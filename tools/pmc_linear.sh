#!/bin/bash
# HBM-side traffic (FETCH_SIZE x 2 per the gfx950 correction, WRITE_SIZE; KiB) of ONE Linear shape with a forced tile, against its
# algorithmic bytes:   bash tools/pmc_linear.sh L0.qkv 14      (two PMC passes, --kernel-trace only)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
shape=$1; tile=$2; O=/tmp/pmc_lin_${shape}_${tile}
rm -rf $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o p -- python3 tools/linear_one.py $shape $tile > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o p -- python3 tools/linear_one.py $shape $tile > /dev/null 2>&1
python3 - $O $shape $tile <<'PY'
import csv, glob, sys
csv.field_size_limit(1 << 30)
O, shape, tile = sys.argv[1:4]
res = {}
for sub, cn in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    vals = []
    for path in glob.glob(f"{O}/{sub}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            if r["Counter_Name"] == cn and ("linear_" in r["Kernel_Name"] or "igemm_bl" in r["Kernel_Name"]):
                vals.append(float(r["Counter_Value"]))
    vals = vals[1:] if len(vals) > 1 else vals
    res[cn] = sum(vals) / max(len(vals), 1)
print(f"{shape} tile {tile}: fetched {2 * res['FETCH_SIZE'] / 1024:.0f} MiB (FETCH_SIZE x 2), written {res['WRITE_SIZE'] / 1024:.0f} MiB per launch")
PY
