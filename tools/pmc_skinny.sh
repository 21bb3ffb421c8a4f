#!/bin/bash
# HBM-side traffic of ONE tile-15 problem per launch (FETCH_SIZE x 2 per the gfx950 correction; cold weights):
#   bash tools/pmc_skinny.sh conv4 1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
shape=$1; cfg=$2; O=/tmp/pmc_sk_${shape}_${cfg}
rm -rf $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o p -- python3 tools/sk_one.py $shape $cfg > /dev/null 2>&1
python3 - $O $shape $cfg <<'PY'
import csv, glob, sys
csv.field_size_limit(1 << 30)
O, shape, cfg = sys.argv[1:4]
vals = []
for path in glob.glob(f"{O}/f/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        if r["Counter_Name"] == "FETCH_SIZE" and ("skinny_kernel" in r["Kernel_Name"] or "igemm_bl" in r["Kernel_Name"]):
            vals.append(float(r["Counter_Value"]))
vals = vals[2:] if len(vals) > 3 else vals
print(f"{shape} cfg {cfg}: fetched {2 * sum(vals) / max(len(vals), 1) / 1024:.1f} MiB per launch (FETCH_SIZE x 2, {len(vals)} launches)")
PY
