# Round numbers of the shipped build (GPU box): the three bench workloads with roofline, variants, AO and interactive benches.
mkdir -p gpurun_out/round
set -e
T="timeout -k 10 280"
$T python -u bench.py > gpurun_out/round/bench_atrium.json 2> gpurun_out/round/bench_atrium.err
$T python -u bench.py --workload street --no-cpu-baseline > gpurun_out/round/bench_street.json 2> gpurun_out/round/bench_street.err
$T python -u bench.py --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --no-cpu-baseline > gpurun_out/round/bench_courtyard.json 2> gpurun_out/round/bench_courtyard.err
$T python -u bench.py --no-cpu-baseline --no-roofline --pipeline 1 > gpurun_out/round/bench_atrium_pipeline.json 2>/dev/null
$T python -u bench.py --no-cpu-baseline --no-roofline --pipeline 1 --sky-mode 1 > gpurun_out/round/bench_atrium_pipeline_skylut.json 2>/dev/null
$T python -u bench.py --workload street --no-cpu-baseline --no-roofline --pipeline 1 --sky-mode 1 > gpurun_out/round/bench_street_pipeline_skylut.json 2>/dev/null
$T python -u bench.py --workload street --no-cpu-baseline --no-roofline --material-ext 1 > gpurun_out/round/bench_street_ext.json 2>/dev/null
$T python -u bench.py --workload street --no-cpu-baseline --no-roofline --material-ext 1 --russian-roulette 1 > gpurun_out/round/bench_street_ext_rr.json 2>/dev/null
$T python -u bench.py --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --no-cpu-baseline --no-roofline --material-ext 1 --russian-roulette 1 > gpurun_out/round/bench_courtyard_ext_rr.json 2>/dev/null
$T python -u tools/bench_ao.py > gpurun_out/round/bench_ao.txt 2>&1
$T python -u tools/bench_interactive.py > gpurun_out/round/bench_interactive.txt 2>&1
for f in gpurun_out/round/*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], d["value"], d.get("msamples_per_s"), d["ms_per_step"], r.get("bound"), r.get("frac"), r.get("extend_mrays_per_s"), r.get("secondary_mrays_per_s"), (d.get("cpu_baseline") or {}).get("value"))
PY
done
tail -n 5 gpurun_out/round/bench_ao.txt; tail -n 6 gpurun_out/round/bench_interactive.txt
