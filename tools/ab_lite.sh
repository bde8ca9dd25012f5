#!/bin/bash
# alternate two builds on tools/ab_lite.py: usage tools/ab_lite.sh N libA.so libB.so ("default" = the in-tree build; a variant: make -C adrates_amd/csrc OUT=../../variant.so OBJDIR=build_variant EXTRA=-D...)
N=${1:-3}; A=${2:-default}; B=${3:-default}
cd /root/repo
for i in $(seq $N); do
  for L in $A $B; do
    if [ "$L" = default ]; then P=""; else P=$PWD/$L; fi
    printf "%-20s " "$L"; ADRATES_HIP_LIB=$P timeout -k 10 200 python tools/ab_lite.py 2>/dev/null || exit 1
  done
done
