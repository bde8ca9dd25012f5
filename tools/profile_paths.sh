#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of the payment-lag bench and of the XCCY book (with gammas), and the HBM
# traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the payment-lag bench.  Output under gpurun_out/prof_TAG_*.
TAG=${1:-run}
OUT=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_lag/trace -- python3 /root/repo/tools/bench_long_legs.py 200000 lag > $OUT/prof_${TAG}_lag.trace.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_lag/fetch -- python3 /root/repo/tools/bench_long_legs.py 200000 lag > $OUT/prof_${TAG}_lag.fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_lag/write -- python3 /root/repo/tools/bench_long_legs.py 200000 lag > $OUT/prof_${TAG}_lag.write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_xccy/trace -- python3 /root/repo/tools/bench_xccy.py 100000 7 > $OUT/prof_${TAG}_xccy.trace.log 2>&1
