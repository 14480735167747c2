#!/bin/bash
# Shared triangle phase (-DPT_SHARED_TRI=1 experiment libraries): parity subset (PARITY=1), then utilisation / rate sweeps
# against main on one box.
set -o pipefail
mkdir -p gpurun_out/shtri
E=$PWD/pathtracer_amd/lib/exp
if [ "${PARITY:-0}" = 1 ]; then
  PT_AMD_LIB=$E/libpt_shtri.so PT_TEST_TUNING="${PT_TEST_TUNING:-}" timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -v --timeout 150 --timeout-method thread \
    -k "bit_identical or soup or pathological or deep_hier or coincident or tmin or schedule or count_kernel or courtyard or street or ao_pipeline or occlusion or instanced" \
    2>&1 | tee gpurun_out/shtri/parity.log | grep -c PASSED || { tail -30 gpurun_out/shtri/parity.log; exit 1; }
  tail -2 gpurun_out/shtri/parity.log
fi
for wl in ${WLS:-atrium street_hard courtyard}; do
  timeout -k 10 300 python -u tools/experiments/exp_tri_util.py $wl 40,25,4,6,73 > gpurun_out/shtri/${wl}_main.txt 2>&1 || { tail gpurun_out/shtri/${wl}_main.txt; exit 1; }
  PT_AMD_LIB=$E/libpt_shtri.so timeout -k 10 300 python -u tools/experiments/exp_tri_util.py $wl ${G256:-40,25,2,6,34 60,25,2,6,34 40,25,3,6,8} > gpurun_out/shtri/${wl}_sh256.txt 2>&1 || { tail gpurun_out/shtri/${wl}_sh256.txt; exit 1; }
  echo "== $wl"; cat gpurun_out/shtri/${wl}_main.txt gpurun_out/shtri/${wl}_sh256.txt
done
if [ -n "${G128:-}" ]; then
  for wl in ${WLS:-atrium street_hard courtyard}; do
    PT_AMD_LIB=$E/libpt_shtri128.so timeout -k 10 300 python -u tools/experiments/exp_tri_util.py $wl $G128 > gpurun_out/shtri/${wl}_sh128.txt 2>&1 || { tail gpurun_out/shtri/${wl}_sh128.txt; exit 1; }
    echo "== $wl (take 2, list 128)"; cat gpurun_out/shtri/${wl}_sh128.txt
  done
fi
