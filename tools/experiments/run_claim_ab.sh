#!/bin/bash
# A/B of the extend kernel's claim scheme on ONE box: shipped library vs pathtracer_amd/lib/exp/libpt_base.so (the
# previous commit's kernel). Parity subset first, then bench legs alternating, then the small-frame call pattern.
set -u
mkdir -p gpurun_out/claim
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -m gpu -k "bit_identical or schedule or noalloc or hard or courtyard or trace" > gpurun_out/claim/tests.log 2>&1 || { tail -15 gpurun_out/claim/tests.log; exit 1; }
tail -2 gpurun_out/claim/tests.log
bash tools/experiments/ab.sh "--no-configs --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --russian-roulette 1" base main || exit 1
bash tools/experiments/ab.sh "--no-configs" base main || exit 1
bash tools/experiments/ab.sh "--no-configs --workload street" base main || exit 1
for t in base main base main; do
  if [ "$t" = main ]; then L=""; else L=$PWD/pathtracer_amd/lib/exp/libpt_$t.so; fi
  echo "== interactive $t"
  PT_AMD_LIB=$L timeout -k 10 120 python -u tools/bench_interactive.py 2>&1 | grep -v "^\[" || exit 1
done
