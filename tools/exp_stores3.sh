#!/bin/bash
# Round 4: the output phase alone (nodes built, not walked) with explicit cache-policy bits on the gamma stores
# (tools/build_store_variants.sh): 1 = nt, 2 = sc1 nt, 3 = sc0 sc1 nt, 4 = sc1, 5 = sc0 sc1, 6 = sc0; the builtin's
# non-temporal store (variants_skipwalk.so) for reference, first and last (drift of the box).
cd /root/repo
for v in variants_skipwalk.so variants_bits1.so variants_bits2.so variants_bits3.so variants_bits4.so variants_bits5.so variants_bits6.so variants_skipwalk.so; do
  printf "%-24s " "$v"
  ADRATES_HIP_LIB=$PWD/$v python tools/ab_calls.py 1000000 4 2>&1 | grep "all outputs, no agg"
done
