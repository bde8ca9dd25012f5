#!/bin/bash
# A/B of library builds on one box: alternate the builds N times, print the per-build median and minimum of bench.py's
# kernel time.  usage: tools/ab.sh N lib1.so lib2.so ... [-- bench args]   ("default" = the in-tree build)
N=$1; shift
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for i in $(seq $N); do
  for L in "${LIBS[@]}"; do
    if [ "$L" = default ]; then P=""; else P=$PWD/$L; fi
    ms=$(ADRATES_HIP_LIB=$P python bench.py --cpu-baseline-seconds 0 "$@" 2>/dev/null | python -c "import json,sys;print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
    echo "$L $ms"
  done
done | python -c "
import sys, collections, statistics
d = collections.defaultdict(list)
for l in sys.stdin:
    k, v = l.split(); d[k].append(float(v))
for k, v in d.items(): print(f'{k:28s} median {statistics.median(v):.4f}  min {min(v):.4f}  n {len(v)}')
"
