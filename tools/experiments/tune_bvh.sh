#!/bin/bash
# Sweeps the collapse's triangle-cost constant: tools/tune_bvh.sh <workload>
for c in 0.15 0.3 0.5 0.75 1.0 1.5; do
  PT_BVH_CPRIM=$c timeout -k 10 300 python bench.py --no-cpu-baseline --workload $1 --steps 4 > /tmp/b.json || exit 1
  python - $c <<'PY'
import json, sys
d = json.load(open('/tmp/b.json')); r = d['roofline']
print(f"Cprim {sys.argv[1]:5s} value {d['value']:8.1f} extend {r['extend_mrays_per_s']:8.1f} nodes/ray {r['nodes_per_ray']} tris/ray {r['tris_per_ray']} shipped {r['shipped_schedule']['nodes_per_ray']}/{r['shipped_schedule']['tris_per_ray']} bvh_mb {d['config']['bvh_mb']}")
PY
done
