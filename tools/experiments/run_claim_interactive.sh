#!/bin/bash
# the small-frame call pattern (tools/bench_interactive.py) and the courtyard's per-bounce rates for library variants of the
# short-queue claim scheme (lib/exp/libpt_<tag>.so; "main" = the shipped library)
set -u
for rnd in 1 2; do
for t in "$@"; do
  if [ "$t" = main ]; then L=""; else L=$PWD/pathtracer_amd/lib/exp/libpt_$t.so; fi
  echo "== interactive $t"
  PT_AMD_LIB=$L timeout -k 10 120 python -u tools/bench_interactive.py 2>&1 | grep "1 spp\|16 spp" || exit 1
done
done
bash tools/experiments/run_claim_pb.sh "--workload courtyard --width 3840 --height 2160 --spp-per-step 16 --russian-roulette 1" "$@"
