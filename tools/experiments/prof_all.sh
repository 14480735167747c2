#!/bin/bash
# Round profiles of the shipped build: kernel-trace stats of the default bench command + PMC passes per workload.
# usage (on the GPU box): bash tools/prof_all.sh <round-tag>   -> gpurun_out/<tag>_*  (copy summaries into profiles/<round>/)
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/${TAG}_trace
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --no-cpu-baseline > gpurun_out/${TAG}_trace/bench.json 2> gpurun_out/${TAG}_trace/bench.err || { echo "trace failed"; tail -3 gpurun_out/${TAG}_trace/bench.err; exit 1; }
echo "trace ok"
bash tools/prof_pmc.sh ${TAG}_atrium || exit 1
python3 tools/summarize_pmc.py ${TAG}_atrium > gpurun_out/pmc_${TAG}_atrium_summary.txt
bash tools/prof_pmc.sh ${TAG}_street --workload street || exit 1
python3 tools/summarize_pmc.py ${TAG}_street > gpurun_out/pmc_${TAG}_street_summary.txt
bash tools/prof_pmc.sh ${TAG}_courtyard --workload courtyard --width 3840 --height 2160 --spp-per-step 16 || exit 1
python3 tools/summarize_pmc.py ${TAG}_courtyard > gpurun_out/pmc_${TAG}_courtyard_summary.txt
echo "all ok"
