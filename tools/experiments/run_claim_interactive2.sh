#!/bin/bash
set -u
for rnd in 1 2; do
for t in "$@"; do
  if [ "$t" = main ]; then L=""; else L=$PWD/pathtracer_amd/lib/exp/libpt_$t.so; fi
  echo "== interactive $t"
  PT_AMD_LIB=$L timeout -k 10 120 python -u tools/bench_interactive.py 2>&1 | grep "spp/call" || exit 1
done
done
