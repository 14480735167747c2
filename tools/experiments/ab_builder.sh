#!/bin/bash
# Builder A/B on ONE box: the same bench legs under different PT_BVH_* environment overrides (read by the builder at
# scene creation).  usage (GPU box): bash tools/ab_builder.sh <outdir> "VAR=val VAR=val" "VAR=val" ...
set -u
OUT=$1; shift
mkdir -p $OUT
leg() {  # name, env string, bench args...
  local name=$1 envs=$2; shift 2
  local tag=$(echo "$envs" | tr ' =' '__')
  env $envs timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-configs --steps ${STEPS:-3} --warmup ${WARMUP:-4} "$@" > $OUT/${name}_${tag}.json 2> $OUT/${name}_${tag}.err || { echo "$name $envs FAILED"; tail -3 $OUT/${name}_${tag}.err; return 1; }
  python3 - "$OUT/${name}_${tag}.json" "$name" "$envs" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
r = d["roofline"]
print(f"{sys.argv[2]:12s} {sys.argv[3]:40s} whole {d['value']:8.1f} Mrays/s  secondary {r['secondary_mrays_per_s']:8.1f}  nodes {r['nodes_per_ray']:6.2f} tris {r['tris_per_ray']:6.2f} B/ray {r['bytes_per_ray']:7.1f}  bvh {d['config']['bvh_mb']:6.1f} MB build {d['config']['bvh_build_ms']:.0f} ms")
PY
}
for E in "$@"; do
  if [ -n "${ONLY:-}" ]; then
    case $ONLY in courtyard) leg courtyard "$E" --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --russian-roulette 1 || exit 1;; street_hard) leg street_hard "$E" --workload street_hard || exit 1;; *) leg $ONLY "$E" --workload $ONLY || exit 1;; esac
    continue
  fi
  leg atrium "$E" || exit 1
  leg street "$E" --workload street || exit 1
  leg street_hard "$E" --workload street_hard || exit 1
  leg courtyard "$E" --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --russian-roulette 1 || exit 1
done
