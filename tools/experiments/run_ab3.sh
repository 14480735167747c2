# same-box A/B on the three bench workloads: tools/run_ab3.sh tagA tagB ...
mkdir -p gpurun_out
set -e
echo "== atrium"; bash tools/ab.sh "--steps 6 --warmup 2" "$@"
echo "== street"; bash tools/ab.sh "--workload street --steps 4 --warmup 1 --spp-per-step 8" "$@"
echo "== courtyard"; bash tools/ab.sh "--workload courtyard --width 3840 --height 2160 --steps 2 --warmup 1 --spp-per-step 4" "$@"
