#!/bin/bash
# does PT_OPT_SHARED_TRIANGLE_PHASE = 2 pick the same form for every bench leg in repeated runs on one box?
set -u
mkdir -p gpurun_out/choice
for i in 1 2 3 4 5; do
  timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/choice/run_$i.json 2> gpurun_out/choice/run_$i.err || exit 1
  python - gpurun_out/choice/run_$i.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
t = d["roofline"]["triangle_phase"]
out = [f"C2 {t['choice']} {t['timed_ms_per_lane']:.1f}/{t['timed_ms_shared']:.1f} {d['value']:.0f}"]
for k, v in d["configs"].items():
    t = v["roofline"]["triangle_phase"]
    out.append(f"{k} {t['choice']} {t['timed_ms_per_lane']:.1f}/{t['timed_ms_shared']:.1f} {v['mrays_per_s']:.0f}")
print(" | ".join(out), flush=True)
PY
done
