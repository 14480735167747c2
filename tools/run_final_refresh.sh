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
bash tools/prof_pmc.sh ${TAG}_atrium --primary-packet 1 || exit 1
python3 tools/summarize_pmc.py ${TAG}_atrium > gpurun_out/pmc_${TAG}_atrium_summary.txt
bash tools/prof_pmc.sh ${TAG}_street --workload street --primary-packet 1 || exit 1
python3 tools/summarize_pmc.py ${TAG}_street > gpurun_out/pmc_${TAG}_street_summary.txt
bash tools/prof_pmc.sh ${TAG}_street_hard --workload street_hard --primary-packet 1 || exit 1
python3 tools/summarize_pmc.py ${TAG}_street_hard > gpurun_out/pmc_${TAG}_street_hard_summary.txt
bash tools/prof_pmc.sh ${TAG}_courtyard --workload courtyard --width 3840 --height 2160 --spp-per-step 16 --russian-roulette 1 --primary-packet 1 || exit 1
python3 tools/summarize_pmc.py ${TAG}_courtyard > gpurun_out/pmc_${TAG}_courtyard_summary.txt
python3 tools/pmc_traffic.py ${TAG}_atrium atrium 1920 1080 64 4 > /dev/null || exit 1
python3 tools/pmc_traffic.py ${TAG}_street street 1920 1080 64 8 > /dev/null || exit 1
python3 tools/pmc_traffic.py ${TAG}_street_hard street_hard 1920 1080 64 8 > /dev/null || exit 1
python3 tools/pmc_traffic.py ${TAG}_courtyard courtyard 3840 2160 16 16 russian_roulette=1 > /dev/null || exit 1
cp profiles/$TAG/pmc_traffic.json gpurun_out/pmc_traffic_${TAG}.json
echo "pmc ok"
bash tools/run_round_numbers.sh
