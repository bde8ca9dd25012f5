#!/bin/bash
# The two-curve foreign-leg launch: parity tests of the XCCY book, then the bench lines (PV + three delta ladders).
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_xccy.py tests/test_gpu_mixed_book.py tests/test_gpu_aggregate_only.py -m gpu -x -q > gpurun_out/xc_tests.log 2>&1
echo "tests exit $?"; tail -15 gpurun_out/xc_tests.log
grep -q passed gpurun_out/xc_tests.log || exit 1
grep -q failed gpurun_out/xc_tests.log && exit 1
timeout -k 10 300 python tools/bench_xccy.py 100000 3 > gpurun_out/xc_bench3.json 2> gpurun_out/xc_bench3.err || { tail -20 gpurun_out/xc_bench3.err; exit 1; }
python -c "
import json
d = json.load(open('gpurun_out/xc_bench3.json'))
print({k: (round(v, 5) if isinstance(v, float) else v) for k, v in d.items() if k.startswith(('ms', 'host_', 'max_'))})
"
