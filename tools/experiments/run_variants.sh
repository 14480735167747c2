#!/bin/bash
# usage: tools/run_variants.sh <bench args...>   runs bench.py once per library variant in pathtracer_amd/lib/variants
set -u
cp pathtracer_amd/lib/libpathtracer_amd.so /tmp/orig.so
for v in pathtracer_amd/lib/variants/*.so; do
  cp $v pathtracer_amd/lib/libpathtracer_amd.so
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > /tmp/v.json || { echo "$v failed"; break; }
  python - "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/v.json')); r = d['roofline']
print(f"{sys.argv[1].split('/')[-1]:14s} value {d['value']:8.1f}  ms/step {d['ms_per_step']:7.3f}  extend {r['extend_mrays_per_s']:8.1f}  nodes/ray {r['nodes_per_ray']} tris/ray {r['tris_per_ray']} build_ms {d['config']['bvh_build_ms']}")
PY
done
cp /tmp/orig.so pathtracer_amd/lib/libpathtracer_amd.so
