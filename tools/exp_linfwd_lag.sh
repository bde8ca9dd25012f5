#!/bin/bash
# Payment lag under LINEAR_FWD_RATES on the lite kernel's rows (PV, PV + delta, the book's ladder): parity tests, then timings
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity_batch.py tests/test_gpu_aggregate_only.py tests/test_gpu_delta_only.py tests/test_gpu_xccy.py tests/test_gpu_many_pillars.py tests/test_gpu_mixed_book.py -m gpu -x -q 2>&1 | tail -4
for m in "lag 3" "lag 7 aggonly" "longlag 3" "longlag 7 aggonly" "lag 7"; do
  set -- $m
  for I in 4 2; do
    ADR_BENCH_INTERP=$I timeout -k 10 200 python tools/bench_long_legs.py $([ $1 = lag ] && echo 200000 || echo 100000) $1 $2 $3 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('interp $I  $m ms', round(d.get('ms', d.get('ms_total', 0.0)), 4), round(d['trades_per_s'] / 1e6, 1), 'M/s')
" || exit 1
  done
done
