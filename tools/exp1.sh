#!/bin/bash
# round 4, first call: baseline of the box + where the aggregate's 6-7 % goes (kernel duration vs gaps)
cd /root/repo
python bench.py --cpu-baseline-seconds 0 > gpurun_out/e1_bench.json 2> gpurun_out/e1_bench.err || { tail -5 gpurun_out/e1_bench.err; exit 1; }
python tools/ablate.py > gpurun_out/e1_ablate.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/e1_trace -- python3 /root/repo/tools/ablate.py > /root/repo/gpurun_out/e1_trace.log 2>&1 || exit 1
cd /root/repo
python tools/exp_agg_gaps.py $(ls -t gpurun_out/e1_trace/*/*kernel_trace.csv | head -1) > gpurun_out/e1_gaps.txt
cat gpurun_out/e1_ablate.txt gpurun_out/e1_gaps.txt
python -c "import json;d=json.load(open('gpurun_out/e1_bench.json'));print(d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
