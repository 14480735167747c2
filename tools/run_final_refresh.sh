# Final refresh of a round's committed measurements on ONE box, in dependency order: kernel-trace stats of the headline
# bench command, PMC passes per workload (with the primary kernel named explicitly, so the one-off timing launches of
# PT_OPT_PRIMARY_PACKET = 2 do not enter the per-launch averages), pmc_traffic.json from those passes (each entry stamped with
# the fingerprint of the kernel sources), and only then the bench lines whose roofline object reads that file.
# usage (GPU box): bash tools/run_final_refresh.sh <round>      e.g. r03 -> gpurun_out/<round>_* ; copy what is kept to profiles/<round>/
set -u
TAG=${1:-r03}
export PT_PROFILE_ROUND=$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/${TAG}_trace profiles/$TAG
# the headline alone (no "configs" legs: their k_extend launches on other scenes would enter the per-kernel averages)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --no-cpu-baseline --no-configs > gpurun_out/${TAG}_trace/bench.json 2> gpurun_out/${TAG}_trace/bench.err || { echo "trace failed"; tail -3 gpurun_out/${TAG}_trace/bench.err; exit 1; }
echo "trace ok"
# which triangle phase does the library pick per workload (PT_OPT_SHARED_TRIANGLE_PHASE = 2 times two batches of each form during the warm-up steps)?
# The PMC passes then name it explicitly, like the primary kernel, so no timing launch enters the per-launch averages.
choice() {
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-configs --steps 1 --warmup 5 "$@" 2> /dev/null | python3 -c "import json,sys; print(max(0, json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']['triangle_phase']['choice']))"
}
pmc_one() {   # <name> <w> <h> <spp> <depth> <extra pmc_traffic key=value or ''> <bench args...>
  local name=$1 w=$2 h=$3 spp=$4 depth=$5 extra=$6; shift 6
  local ch; ch=$(choice "$@") || exit 1
  echo "$name: triangle phase $ch"
  bash tools/prof_pmc.sh ${TAG}_$name "$@" --primary-packet 1 --shared-triangle-phase $ch || exit 1
  python3 tools/summarize_pmc.py ${TAG}_$name > gpurun_out/pmc_${TAG}_${name}_summary.txt
  python3 tools/pmc_traffic.py ${TAG}_$name $name $w $h $spp $depth $extra shared_triangle_phase=$ch > /dev/null || exit 1
}
pmc_one atrium 1920 1080 64 4 "" --workload atrium
pmc_one street 1920 1080 64 8 "" --workload street
pmc_one street_hard 1920 1080 64 8 "" --workload street_hard
pmc_one courtyard 3840 2160 16 16 russian_roulette=1 --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --russian-roulette 1
if [ "${PMC_VARIANTS:-1}" = 1 ]; then   # the variants the bench also reports: C4 with the GGX extension, C5 without Russian roulette
  mv gpurun_out/pmc_${TAG}_street gpurun_out/pmc_${TAG}_street_plain; mv gpurun_out/pmc_${TAG}_courtyard gpurun_out/pmc_${TAG}_courtyard_rr
  cp gpurun_out/pmc_${TAG}_street_summary.txt gpurun_out/pmc_${TAG}_street_plain_summary.txt; cp gpurun_out/pmc_${TAG}_courtyard_summary.txt gpurun_out/pmc_${TAG}_courtyard_rr_summary.txt   # the variants below reuse the names
  pmc_one street 1920 1080 64 8 material_ext=1 --workload street --material-ext 1
  mv gpurun_out/pmc_${TAG}_street_summary.txt gpurun_out/pmc_${TAG}_street_ggx_summary.txt
  pmc_one courtyard 3840 2160 16 16 "" --workload courtyard --width 3840 --height 2160 --spp-per-step 16
  mv gpurun_out/pmc_${TAG}_courtyard_summary.txt gpurun_out/pmc_${TAG}_courtyard_no_rr_summary.txt
  mv gpurun_out/pmc_${TAG}_street_plain_summary.txt gpurun_out/pmc_${TAG}_street_summary.txt; mv gpurun_out/pmc_${TAG}_courtyard_rr_summary.txt gpurun_out/pmc_${TAG}_courtyard_summary.txt
fi
# per-kernel time of the other workloads (the headline's trace is above)
for w in street street_hard courtyard; do
  extra=""; [ $w = courtyard ] && extra="--width 3840 --height 2160 --spp-per-step 16 --russian-roulette 1"
  mkdir -p gpurun_out/${TAG}_trace_$w
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace_$w -- python3 bench.py --workload $w $extra --no-cpu-baseline --no-configs --no-roofline --steps 4 --warmup 1 > gpurun_out/${TAG}_trace_$w/bench.json 2> gpurun_out/${TAG}_trace_$w/bench.err || { echo "trace $w failed"; exit 1; }
  python3 tools/kstats.py gpurun_out/${TAG}_trace_$w > gpurun_out/kernel_stats_${TAG}_$w.txt
done
cp profiles/$TAG/pmc_traffic.json gpurun_out/pmc_traffic_${TAG}.json
echo "pmc ok"
# (SKIP_ROUND=1: the bench lines run in a call of their own, after pmc_traffic.json has been copied into profiles/<round>/)
[ "${SKIP_ROUND:-0}" = 1 ] || bash tools/run_round_numbers.sh
