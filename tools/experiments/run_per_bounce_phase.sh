mkdir -p gpurun_out/pb
T="timeout -k 10 170"
for f in 0 1; do
$T python -u bench.py --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --russian-roulette 1 --no-cpu-baseline --no-configs --shared-triangle-phase $f > gpurun_out/pb/courtyard_$f.json 2>gpurun_out/pb/courtyard_$f.err || exit 1
echo courtyard $f done
$T python -u bench.py --workload street_hard --no-cpu-baseline --no-configs --shared-triangle-phase $f > gpurun_out/pb/street_hard_$f.json 2>gpurun_out/pb/street_hard_$f.err || exit 1
echo hard $f done
done
