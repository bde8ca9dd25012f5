#!/bin/bash
# A/B of library builds (variants_*.so in the repo root) on the general kernel's workloads: the cross-currency book
# (PV + deltas, and with gammas) and payment-lag OIS.
for so in variants_*.so; do
  printf "%-32s " "$so"
  for mask in 3 7; do
    ADRATES_HIP_LIB=$PWD/$so python3 tools/bench_xccy.py 100000 $mask 2>/dev/null | python3 -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('xccy', d['mask'], round(d['ms'],3), [round(x,3) for x in d['ms_domestic_foreignrates_foreignflows']], end='  ')"
    ADRATES_HIP_LIB=$PWD/$so python3 tools/bench_long_legs.py 200000 lag $mask 2>/dev/null | python3 -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('lag', round(d['ms'],3), end='  ')"
  done
  echo
done
