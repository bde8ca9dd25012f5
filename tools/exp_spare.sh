#!/bin/bash
# The payment-lag / two-curve rows with the start lookup in the spare lane: parity tests, then the timings of the paths it touches
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_xccy.py tests/test_gpu_parity_batch.py tests/test_gpu_aggregate_only.py tests/test_gpu_delta_only.py tests/test_gpu_mixed_book.py tests/test_gpu_many_pillars.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
  timeout -k 10 200 python tools/bench_xccy.py 100000 3 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: (round(v, 4) if isinstance(v, float) else [round(x, 4) for x in v]) for k, v in d.items() if k in ('ms', 'ms_two_launches', 'ms_foreign_leg_one_launch', 'ms_domestic_foreignrates_foreignflows', 'ms_aggregate_only', 'max_rel_diff_delta_foreign', 'max_rel_diff_delta_basis')})
" || exit 1
  for m in "lag 3" "longlag 3" "lag 7 aggonly" "longlag 7 aggonly"; do
    set -- $m
    timeout -k 10 200 python tools/bench_long_legs.py $([ $1 = lag ] && echo 200000 || echo 100000) $1 $2 $3 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   $m ms', round(d.get('ms', d.get('ms_total', 0.0)), 4))
" || exit 1
  done
done
