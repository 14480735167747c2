#!/bin/bash
# per-bounce extend rates of the shipped library and of lib/exp/libpt_<tag>.so on one workload.  usage: run_claim_pb.sh "<bench args>" tag...
set -u
ARGS=$1; shift
mkdir -p gpurun_out/claim
for rnd in 1 2; do
for t in "$@"; do
  if [ "$t" = main ]; then L=""; else L=$PWD/pathtracer_amd/lib/exp/libpt_$t.so; fi
  PT_AMD_LIB=$L timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-configs $ARGS > gpurun_out/claim/pb_${t}_$rnd.json 2> gpurun_out/claim/pb_${t}_$rnd.err || { echo "$t failed"; exit 1; }
  python - gpurun_out/claim/pb_${t}_$rnd.json $t <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[2]:8s} whole {d['value']:8.1f} sec {r['secondary_mrays_per_s']:8.1f} ", " ".join(f"{x:7.0f}" for x in r["extend_mrays_per_s_by_bounce"]), flush=True)
PY
done
done
