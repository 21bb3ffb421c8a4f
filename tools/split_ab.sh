#!/bin/bash
# same-box A/B of the batch split of the big igemm launches (plan.autosplit_igemm): MVLDM_BATCH_SPLIT=0 | 1, alternating
for v in 0 1 0 1; do
  S=$(date +%s)
  MVLDM_BATCH_SPLIT=$v timeout 700 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-small-batch --no-parity --no-train-line --no-full-walk --no-alt-dtype --no-dropin --no-other-configs \
      --op-table gpurun_out/optable_split_$v.json 2>gpurun_out/split_ab_$v.err \
    | grep '^{' | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('split=$v', d['value'], 'views/s', d.get('ddim_step_ms'), 'ms/step', d['roofline']['frac'], 'wall', $(date +%s) - $S)"
done
python3 - <<'PY'
import json
a = json.load(open('gpurun_out/optable_split_0.json')); b = json.load(open('gpurun_out/optable_split_1.json'))
print(len(a), 'ops ->', len(b), 'ops;', sum(1 for r in b if r['name'].endswith('[rest]')), 'launches split')
ta = {r['name']: r['ms'] for r in a}
tb = {}
for r in b:
    n = r['name'][:-6] if r['name'].endswith('[rest]') else r['name']
    tb[n] = tb.get(n, 0.0) + r['ms']
names = [r['name'][:-6] for r in b if r['name'].endswith('[rest]')]
tot_a = sum(ta[n] for n in names); tot_b = sum(tb[n] for n in names)
print(f"split launches: {tot_a:.3f} -> {tot_b:.3f} ms;  all ops {sum(ta.values()):.2f} -> {sum(tb.values()):.2f} ms")
for n in names[:40]:
    print(f"  {n:60s} {ta[n]*1e3:7.1f} -> {tb[n]*1e3:7.1f} us")
PY
