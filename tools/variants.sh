#!/bin/bash
# A/B different builds of the library on the GPU box: every variants_*.so in the repo root is benchmarked
# (and parity-checked with the GPU test-suite) in turn.
for so in variants_*.so; do
  echo "== $so"
  ADRATES_HIP_LIB=$PWD/$so python -m pytest tests -m gpu -x -q 2>&1 | tail -1
  ADRATES_HIP_LIB=$PWD/$so python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['roofline']['frac'])"
done
