#!/bin/bash
# A/B of library builds on ONE box: alternates bench.py runs and prints whole-job and extend rates.
# usage: tools/ab.sh "<bench args>" tagA tagB ...   (tag "main" = the shipped library, else pathtracer_amd/lib/exp/libpt_<tag>.so)
ARGS=$1; shift
for rnd in 1 2; do
  for t in "$@"; do
    if [ "$t" = main ]; then L=""; else L=$PWD/pathtracer_amd/lib/exp/libpt_$t.so; fi
    PT_AMD_LIB=$L timeout -k 10 200 python -u bench.py --no-cpu-baseline $ARGS > gpurun_out/ab_$t.json 2> gpurun_out/ab_$t.err || { echo "$t failed"; tail -3 gpurun_out/ab_$t.err; exit 1; }
    python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
d = json.load(open(f"gpurun_out/ab_{t}.json")); r = d["roofline"]
print(f"{t:10s} whole {d['value']:8.1f}  ms/step {d['ms_per_step']:7.3f}  extend {r['extend_mrays_per_s']:8.1f}  launch {r['avg_launch_ms']:.4f} ms  share {r['extend_share_of_render']:.3f}", flush=True)
PY
  done
done
