#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats and the HBM traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the wide
# route on the 40-pillar curve, 100 000 benchmark trades, PV + delta + gamma with every output stored.
TAG=${1:-run}
OUT=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ABLATE_ONLY=stored
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_wide/trace -- python3 /root/repo/tools/ablate_wide.py 40 100000 offgrid > $OUT/prof_${TAG}_wide.trace.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_wide/fetch -- python3 /root/repo/tools/ablate_wide.py 40 100000 offgrid > $OUT/prof_${TAG}_wide.fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_wide/write -- python3 /root/repo/tools/ablate_wide.py 40 100000 offgrid > $OUT/prof_${TAG}_wide.write.log 2>&1
