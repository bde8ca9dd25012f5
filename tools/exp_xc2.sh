#!/bin/bash
# A/B of the block size of the lite kernel's payment-lag PV + delta instantiation: default build vs variants_xc2.so
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_xccy.py tests/test_gpu_parity_batch.py -m gpu -x -q 2>&1 | tail -2
ADRATES_HIP_LIB=$PWD/variants_xc2.so timeout -k 10 300 python -m pytest tests/test_gpu_xccy.py tests/test_gpu_parity_batch.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
  for L in "" $PWD/variants_xc2.so; do
    ADRATES_HIP_LIB=$L timeout -k 10 200 python tools/bench_xccy.py 100000 3 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${L:-default}'.split('/')[-1], {k: (round(v, 4) if isinstance(v, float) else [round(x, 4) for x in v]) for k, v in d.items() if k in ('ms', 'ms_two_launches', 'ms_foreign_leg_one_launch', 'ms_domestic_foreignrates_foreignflows')})
" || exit 1
    for m in lag longlag; do
      ADRATES_HIP_LIB=$L timeout -k 10 200 python tools/bench_long_legs.py $([ $m = lag ] && echo 200000 || echo 100000) $m 3 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   $m PV+delta ms', round(d.get('ms', d.get('ms_total', 0.0)), 4))
" || exit 1
    done
  done
done
