# Final refresh of the round's committed measurements on ONE box: kernel-trace stats of the default bench + round numbers.
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/${TAG}_trace
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --no-cpu-baseline > gpurun_out/${TAG}_trace/bench.json 2> gpurun_out/${TAG}_trace/bench.err || { echo "trace failed"; tail -3 gpurun_out/${TAG}_trace/bench.err; exit 1; }
echo "trace ok"
bash tools/run_round_numbers.sh
