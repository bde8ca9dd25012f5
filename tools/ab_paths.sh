#!/bin/bash
# A/B of two library builds on the side paths (payment lag, long legs, XCCY with gammas): alternate the builds N times on one
# box, print every run's ms.  usage: tools/ab_paths.sh N libA.so libB.so   ("default" = the in-tree build)
N=${1:-3}; A=${2:-default}; B=${3:-variants_noasm.so}
cd /root/repo
run() {  # label lib cmd...
  local lib=$1; shift
  if [ "$lib" = default ]; then P=""; else P=$PWD/$lib; fi
  ADRATES_HIP_LIB=$P python "$@" 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d.get('ms', d.get('ms_total', 0.0)), 4), end=' ')
print()"
}
for i in $(seq $N); do
  for L in $A $B; do
    printf "%-22s lag200k: " "$L"; run $L tools/bench_long_legs.py 200000 lag
    printf "%-22s longlag100k: " "$L"; run $L tools/bench_long_legs.py 100000 longlag
    printf "%-22s long200k: " "$L"; run $L tools/bench_long_legs.py 200000 long
  done
done
