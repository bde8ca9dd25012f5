#!/bin/bash
# round 4: ratio nodes in knot space - parity of the aggregate-only mode on payment-lag / XCCY books, and its timings
cd /root/repo
python -m pytest tests/test_gpu_aggregate_only.py tests/test_gpu_xccy.py tests/test_gpu_mixed_book.py -m gpu -x -q 2>&1 | tail -4
for mode in lag longlag long; do
  python tools/bench_long_legs.py 200000 $mode 7 aggonly 2>/dev/null
  python tools/bench_long_legs.py 200000 $mode 3 aggonly 2>/dev/null
done
timeout -k 10 120 tools/exp_store_stream 2>&1 | grep balance
