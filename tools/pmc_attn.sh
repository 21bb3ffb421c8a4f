#!/bin/bash
# PMC passes over one attention shape (tools/attn_one.py d heads scenes); run on the GPU box from the repo root
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/a -o p -- python3 tools/attn_one.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $out/b -o p -- python3 tools/attn_one.py "$@" > /dev/null 2>&1
rm -f $out/a/p_kernel_trace.csv $out/b/p_kernel_trace.csv
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
csv.field_size_limit(1<<30)
for d in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_attn/*')):
    f=d+'/p_counter_collection.csv'
    agg=defaultdict(list)
    for r in csv.DictReader(open(f,newline='')):
        if 'attention_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(os.path.basename(d), {k:round(sum(v[1:])/max(len(v)-1,1)) for k,v in agg.items()})
PY
python3 tools/attn_one.py "$@"
