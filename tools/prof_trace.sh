#!/bin/bash
# kernel-trace stats of one bench workload: tools/prof_trace.sh <tag> [bench args...] -> gpurun_out/<tag>_trace
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
mkdir -p gpurun_out/${TAG}_trace
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --no-cpu-baseline --no-roofline "$@" > gpurun_out/${TAG}_trace/bench.json 2> gpurun_out/${TAG}_trace/bench.err || { echo "trace failed"; tail -3 gpurun_out/${TAG}_trace/bench.err; exit 1; }
python3 - "$TAG" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(f"gpurun_out/{sys.argv[1]}_trace/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    print(' ', r['Name'][:60].ljust(60), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:10.1f} us", r['Percentage'])
PY
