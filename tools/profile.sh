#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile.sh TAG'): kernel-trace stats and, in separate passes as the
# MI355X guide prescribes, the HBM traffic counters of the same bench command.  Summaries land in gpurun_out/.
TAG=${1:-run}
OUT=/root/repo/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
# the bench command itself (default steps / warmup), minus the CPU baseline leg
CMD="python3 /root/repo/bench.py --cpu-baseline-seconds 0 $BENCH_ARGS"
# (100 timed steps in the stats pass: the all-launch average of kernel_stats.csv then is the timed region's to within 1 %;
# with the default 20 it is dominated by the ~90 warm-up launches, the first dozen of them on a cold clock)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD --steps 100 > $OUT.trace.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT.fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT.write.log 2>&1
