#!/bin/bash
# XCCY book (BASELINE configs[3]): bench lines for PV + three delta ladders and with the three gammas, the kernel stats of
# the delta run, and the parity tests of the book.
cd /root/repo
python tools/bench_xccy.py 100000 3 > gpurun_out/x_bench3.json 2>/dev/null || exit 1
python tools/bench_xccy.py 100000 7 > gpurun_out/x_bench7.json 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/x_trace -- python3 /root/repo/tools/bench_xccy.py 100000 3 > /root/repo/gpurun_out/x_trace.log 2>&1 || exit 1
cd /root/repo
python -c "
import json
for f in ('gpurun_out/x_bench3.json', 'gpurun_out/x_bench7.json'):
    d = json.load(open(f)); print(d['mask'], round(d['ms'], 4), [round(x, 4) for x in d['ms_domestic_foreignrates_foreignflows']], {k: round(v, 4) for k, v in d.items() if k.startswith('host_')})
"
head -8 $(ls -t gpurun_out/x_trace/*/*kernel_stats.csv | head -1) | cut -c1-200
python -m pytest tests/test_gpu_xccy.py tests/test_gpu_delta_only.py -m gpu -x -q 2>&1 | tail -2
