mkdir -p gpurun_out/pipe
for p in 0 1 0 1; do
timeout -k 10 200 python -u bench.py --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --russian-roulette 1 --no-cpu-baseline --no-configs --no-roofline --shared-triangle-phase 1 --pipeline $p > gpurun_out/pipe/c5_$p.json 2>gpurun_out/pipe/c5_$p.err || exit 1
python - gpurun_out/pipe/c5_$p.json $p <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('courtyard pipeline',sys.argv[2],d['value'],d['ms_per_step'],flush=True)
PY
done
for p in 0 1; do
timeout -k 10 200 python -u bench.py --workload street_hard --no-cpu-baseline --no-configs --no-roofline --shared-triangle-phase 1 --pipeline $p > gpurun_out/pipe/sh_$p.json 2>gpurun_out/pipe/sh_$p.err || exit 1
python - gpurun_out/pipe/sh_$p.json $p <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('street_hard pipeline',sys.argv[2],d['value'],d['ms_per_step'],flush=True)
PY
done
