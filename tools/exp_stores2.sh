#!/bin/bash
# Round 4: the store stream alone (nodes built, not walked) - does the order of the output addresses or the burst size matter?
cd /root/repo
rm -f variants_plainstore.so variants_skipwalk_plain.so
echo "== output phase alone, by address order"
ADRATES_HIP_LIB=$PWD/variants_skipwalk.so python tools/exp_order.py 2>&1 | grep -v amdgpu
for v in variants_skipwalk_p1.so variants_skipwalk.so variants_skipwalk_p4.so; do
  echo "== $v"
  ADRATES_HIP_LIB=$PWD/$v python tools/ab_calls.py 1000000 4 2>&1 | grep "all outputs, no agg"
done
