#!/bin/bash
# Insertion-based BVH optimisation (PT_BVH_REINSERT_PASSES / _FRAC) against the plain top-down build, alternating on one box.
mkdir -p gpurun_out/reins
for rnd in 1 2; do
for cfg in "0 0.02" "${CFG:-5 0.02}"; do
  set -- $cfg
  PT_BVH_REINSERT_PASSES=$1 PT_BVH_REINSERT_FRAC=$2 timeout -k 10 400 python -u bench.py --no-cpu-baseline --no-configs $ARGS > gpurun_out/reins/b.json 2> gpurun_out/reins/b.err || { tail -3 gpurun_out/reins/b.err; exit 1; }
  python - "$1 $2" <<'PY'
import json, sys
d = json.load(open("gpurun_out/reins/b.json")); r = d["roofline"]; c = d["config"]
print(f"passes/frac {sys.argv[1]:8s} whole {d['value']:8.1f}  secondary {r['secondary_mrays_per_s']:8.1f}  primaries {(r['primary_kernel'] or {}).get('mrays_per_s')}  nodes {r['nodes_per_ray']} tris {r['tris_per_ray']}  build {c['bvh_build_ms']} ms  bvh {c['bvh_mb']} MB", flush=True)
PY
done
done
