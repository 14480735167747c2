mkdir -p gpurun_out/round
set -e
T="timeout -k 10 280"
$T python -u bench.py > gpurun_out/round/bench_atrium.json 2> gpurun_out/round/bench_atrium.err
$T python -u bench.py --workload street --no-cpu-baseline > gpurun_out/round/bench_street.json 2> gpurun_out/round/bench_street.err
$T python -u bench.py --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --no-cpu-baseline > gpurun_out/round/bench_courtyard.json 2> gpurun_out/round/bench_courtyard.err
for f in atrium street courtyard; do python - gpurun_out/round/bench_$f.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], d["value"], d.get("msamples_per_s"), d["ms_per_step"], r.get("bound"), r.get("frac"), r.get("extend_mrays_per_s"), r.get("secondary_mrays_per_s"), r.get("avg_launch_ms"), (r.get("valu_issue") or {}).get("frac"), (r.get("fabric") or {}).get("frac_of_hbm_peak"))
PY
done
