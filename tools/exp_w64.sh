#!/bin/bash
# A/B of the block size of the lite kernel's 64-wide instantiations (33-64 pillars, PV + delta): default vs variants_w64.so
cd /root/repo
for i in 1 2; do
  for L in "" $PWD/variants_w64.so; do
    echo "== ${L:-default}"
    ADRATES_HIP_LIB=$L timeout -k 10 400 python tools/bench_many_pillars.py 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['mask'] == 3 or d['pillars'] > 64: print('  ', d['curve'], d['mask'], 'agg_only' if d['aggregate_only'] else '', round(d['ms'], 4), round(d['trades_per_s'] / 1e6, 1), 'M/s')
" || exit 1
  done
done
