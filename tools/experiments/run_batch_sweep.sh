mkdir -p gpurun_out
for rnd in 1 2; do
for spp in 16 32 64; do
  steps=$((128 / spp))
  timeout -k 10 250 python -u bench.py --no-cpu-baseline --no-roofline --spp-per-step $spp --steps $steps --warmup 1 > gpurun_out/bs_$spp.json 2> gpurun_out/bs_$spp.err || { echo "spp $spp failed"; tail -3 gpurun_out/bs_$spp.err; exit 1; }
  python - $spp <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/bs_{sys.argv[1]}.json"))
print(f"spp/step {sys.argv[1]:>3s}: {d['value']:8.1f} Mrays/s  {d['ms_per_step']:8.3f} ms/step", flush=True)
PY
done
done
