#!/bin/bash
# Diagnostic PMC passes (pipe utilisation of the extend kernel): one rocprofv3 run per pass, counters only.
# usage: tools/prof_pmc_diag.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/passN/...   (summarise: tools/summarize_pmc.py <tag>)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline $*"
P1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
P2="SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU TA_TOTAL_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TAGRAM0_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum"
P3="SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_CYCLES SQ_WAIT_INST_ANY TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 bench.py $ARGS > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
  echo "pass $i ok"
done
