#!/bin/bash
# GPU side of tools/pw_spread.sh: per placement variant, tile 13 with the spread issue (bit 14) against tile 13 without, same library, same box
for v in ${VARIANTS:-333 423 243 033 522 900}; do
  timeout 200 python tools/linear_tiles.py 64 bf16 13,16397,13,16397,13,16397 --lib libmvldm_hip_exp_sp$v.so 2>/dev/null | python3 -c "
import sys, json
out = []
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        out.append('%s %d/%d=%.3f' % (d['shape'], d['t16397'][0], d['t13'][0], d['t16397'][0] / d['t13'][0]))
print('sp$v:', '  '.join(out))
"
done
