#!/bin/bash
# PMC passes over a short bench run, one rocprofv3 run per pass (counters only, no trace domains).
# usage: tools/prof_pmc.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/passN/...
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline $*"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="FETCH_SIZE GRBM_GUI_ACTIVE"
P3="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
P4="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"
P5="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
P6="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 bench.py $ARGS > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
  echo "pass $i ok"
done
