#!/bin/bash
# Bench-only A/B of library builds (variants_*.so in the repo root); for ablation builds whose results are wrong.
for so in variants_*.so; do
  printf "%-28s " "$so"
  ADRATES_HIP_LIB=$PWD/$so python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
done
