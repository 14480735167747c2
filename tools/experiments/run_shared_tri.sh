#!/bin/bash
# Shared triangle phase against the per-lane phase on one box: lane utilisation per phase and extend rate over the phase's own
# thresholds (rows: postpone,continue,queue,lds_stack,top_nodes,shared). Output: gpurun_out/shtri/<workload>.txt
# (kept as profiles/r03/shared_tri_sweep_<workload>.txt, measured there with the prototype libraries of the same algorithm).
set -o pipefail
mkdir -p gpurun_out/shtri
ROWS=${ROWS:-"40,25,4,6,73,0 25,50,3,6,8,1 40,50,3,6,8,1 25,75,3,6,8,1 40,25,3,6,8,1 40,50,2,6,34,1 40,50,4,5,8,1"}
for wl in ${WLS:-atrium street street_hard courtyard}; do
  timeout -k 10 400 python -u tools/experiments/exp_tri_util.py $wl $ROWS > gpurun_out/shtri/${wl}.txt 2>&1 || { tail gpurun_out/shtri/${wl}.txt; exit 1; }
  echo "== $wl"; cat gpurun_out/shtri/${wl}.txt
done
