#!/bin/bash
# VALU / LDS / wait counters of k_extend on one workload (default street_hard): per-lane triangle phase against the shared one.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/shtri_pmc
ARGS="--workload ${WL:-street_hard} --no-configs --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --primary-packet 1"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P4="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"
for t in 0 1; do
  i=0
  for P in "$P1" "$P4"; do
    i=$((i+1))
    timeout -s KILL 150 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_sh_$t/pass$i -- python3 bench.py $ARGS --shared-triangle-phase $t > gpurun_out/shtri_pmc/$t.pass$i.log 2>&1 || { echo "$t pass $i failed"; tail -5 gpurun_out/shtri_pmc/$t.pass$i.log; exit 1; }
  done
  python3 tools/summarize_pmc.py sh_$t > gpurun_out/shtri_pmc/summary_$t.txt
  timeout -k 10 200 python3 bench.py $ARGS --shared-triangle-phase $t > gpurun_out/shtri_pmc/bench_$t.json 2>gpurun_out/shtri_pmc/bench_$t.err || { tail -3 gpurun_out/shtri_pmc/bench_$t.err; exit 1; }
done
head -20 gpurun_out/shtri_pmc/summary_0.txt; head -20 gpurun_out/shtri_pmc/summary_1.txt
