#!/bin/bash
# Where a Linear's time goes, next to the vendor GEMM on the same operands: kernel names + durations (the hipBLASLt kernel's name spells
# its macro tile / prefetch depth), then separate PMC passes (--kernel-trace only, as gpurun requires): SQ wait / issue / MFMA-busy,
# L2 hit rate + tag stalls, fabric read latency (EA level / requests), L1->L2 read latency, TA stalls.  Every pass is bounded by
# `timeout` and parsed as soon as it ends (a pass that hangs costs its own limit, not the call's).
#   bash tools/gemm_diag.sh L2.to_out 10,13,19 [passes: t sq l2 ea tcp ta sq2 sq3]     -> gpurun_out/gemm_diag_<shape>.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
shape=$1; tiles=$2; shift 2
passes=${@:-t sq l2 ea tcp}
O=/tmp/gemm_diag_$shape; out=gpurun_out/gemm_diag_$shape.txt
rm -rf $O; mkdir -p gpurun_out; : > $out
declare -A PMC
PMC[t]=""
PMC[sq]="--pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
PMC[l2]="--pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
PMC[ea]="--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_CYCLE_sum"
PMC[tcp]="--pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
PMC[ta]="--pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
PMC[sq2]="--pmc SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS"
PMC[sq3]="--pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_INST_LEVEL_VMEM"
for tag in $passes; do
    t0=$(date +%s)
    timeout 240 rocprofv3 --kernel-trace ${PMC[$tag]} --output-format csv -d $O/$tag -o p -- python3 tools/gemm_diag_run.py $shape $tiles > $O.$tag.log 2>&1
    echo "## pass $tag: rc $? in $(( $(date +%s) - t0 )) s" >> $out
    tail -2 $O.$tag.log | cut -c1-300 >> $out
    python3 - $O/$tag >> $out <<'PY'
import csv, glob, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
O = sys.argv[1]
dur, cnt = defaultdict(list), defaultdict(lambda: defaultdict(list))
for path in glob.glob(O + "/**/p_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for path in glob.glob(O + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = [k for k, v in dur.items() if sum(v) / len(v) > 50 and len(v) >= 4]
for k in keep:
    v = dur[k][1:]
    print(f"{sum(v) / len(v):9.1f} us x{len(v)}  {k[:400]}")
    m = {c: sum(x[1:]) / max(len(x) - 1, 1) for c, x in cnt[k].items()}
    for c in sorted(m):
        print(f"        {c:36s} {m[c]:16.0f}")
    g = lambda c: m.get(c, float("nan"))
    if "SQ_WAVE_CYCLES" in m:
        print(f"        -> of wave-cycles: wait_any {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}  wait_inst {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}  active {g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}   mfma-busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * g('SQ_BUSY_CYCLES')):.3f} (per-SE busy basis)")
    if "TCC_HIT_sum" in m:
        print(f"        -> L2 hit {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}   tag-stall cycles per req {g('TCC_TAG_STALL_sum') / g('TCC_REQ_sum'):.3f}")
    if "TCC_EA0_RDREQ_sum" in m:
        print(f"        -> fabric read latency (EA level / req) {g('TCC_EA0_RDREQ_LEVEL_sum') / g('TCC_EA0_RDREQ_sum'):.0f} clk   TCC busy {g('TCC_BUSY_sum') / g('TCC_CYCLE_sum'):.3f}")
    if "TCP_TCC_READ_REQ_sum" in m:
        print(f"        -> L1->L2 read latency {g('TCP_TCC_READ_REQ_LATENCY_sum') / g('TCP_TCC_READ_REQ_sum'):.0f} clk")
PY
done
