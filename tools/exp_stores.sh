#!/bin/bash
# Round 4: what bounds the bench pass - the store stream as the kernel issues it.  The same interleaved call timing
# (tools/ab_calls.py) on the default library and on diagnostic builds: nodes built but not walked (the output phase alone),
# with non-temporal and with plain gamma stores; the full kernel with plain gamma stores.
cd /root/repo
for v in default variants_skipwalk.so variants_skipwalk_plain.so variants_plainstore.so; do
  echo "== $v"
  if [ "$v" = default ]; then P=""; else P=$PWD/$v; fi
  ADRATES_HIP_LIB=$P python tools/ab_calls.py 1000000 5 2>&1 | grep -v amdgpu
done
