# Round numbers of the shipped build (GPU box): the driver's bench line (headline + "configs" legs), variants, AO and
# interactive benches -> gpurun_out/round/
mkdir -p gpurun_out/round
set -e
T="timeout -k 10 400"
$T python -u bench.py > gpurun_out/round/bench_default.json 2> gpurun_out/round/bench_default.err
$T python -u bench.py --no-cpu-baseline --no-roofline --pipeline 1 > gpurun_out/round/bench_atrium_pipeline.json 2>/dev/null
$T python -u bench.py --no-cpu-baseline --no-roofline --pipeline 1 --sky-mode 1 > gpurun_out/round/bench_atrium_pipeline_skylut.json 2>/dev/null
$T python -u bench.py --workload street_hard --no-cpu-baseline --no-configs > gpurun_out/round/bench_street_hard.json 2>/dev/null
$T python -u bench.py --workload street_hard --no-cpu-baseline --no-configs --shared-triangle-phase 0 > gpurun_out/round/bench_street_hard_per_lane_phase.json 2>/dev/null
$T python -u bench.py --workload street --no-cpu-baseline --no-roofline --material-ext 1 --russian-roulette 1 > gpurun_out/round/bench_street_ext_rr.json 2>/dev/null
$T python -u bench.py --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --no-cpu-baseline --no-configs > gpurun_out/round/bench_courtyard_no_rr.json 2>/dev/null
$T python -u bench.py --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --no-cpu-baseline --no-roofline --material-ext 1 --russian-roulette 1 > gpurun_out/round/bench_courtyard_ext_rr.json 2>/dev/null
$T python -u bench.py --gpus 2 --backend gloo --all-ranks-on-gpu0 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/round/bench_2ranks_gloo_one_gpu_strong.json 2>/dev/null
$T python -u tools/bench_ao.py > gpurun_out/round/bench_ao.txt 2>&1
$T python -u tools/bench_interactive.py > gpurun_out/round/bench_interactive.txt 2>&1
for f in gpurun_out/round/*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], d["value"], d.get("msamples_per_s"), d["ms_per_step"], r.get("bound"), r.get("frac"), r.get("frac_algorithmic"), r.get("pmc_stale"), r.get("extend_mrays_per_s"), r.get("secondary_mrays_per_s"), (d.get("cpu_baseline") or {}).get("value"))
for k, v in (d.get("configs") or {}).items():
    print("   ", k, v.get("mrays_per_s"), v.get("msamples_per_s"), v.get("secondary_mrays_per_s"), v.get("nodes_per_ray"), v.get("tris_per_ray"), (v.get("roofline") or {}).get("bound"), (v.get("roofline") or {}).get("frac"), v.get("error"))
PY
done
tail -n 5 gpurun_out/round/bench_ao.txt; tail -n 6 gpurun_out/round/bench_interactive.txt
