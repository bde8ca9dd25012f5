#!/bin/bash
# Diagnostic builds for tools/exp_stores3.sh: the output phase alone (ADR_DEBUG_SKIP_WALK) with the gamma stores carrying
# explicit cache-policy bits.  Run in the build container; the .so files travel to the GPU box with the snapshot.
cd "$(dirname "$0")/../adrates_amd/csrc" || exit 1
i=0
for bits in "nt" "sc1 nt" "sc0 sc1 nt" "sc1" "sc0 sc1" "sc0"; do
  i=$((i + 1))
  make -j8 OUT=../../variants_bits$i.so OBJDIR=build_bits$i EXTRA="-DADR_DEBUG_SKIP_WALK '-DADR_GAMMA_STORE_BITS=\"$bits\"'" 2>&1 | grep -E " error"
  echo "variants_bits$i.so = $bits"
done
