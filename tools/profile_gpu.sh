#!/bin/bash
# Runs on the GPU box: kernel trace + PMC passes of a short bench.py run.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj $@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "FETCH_SIZE" "WRITE_SIZE" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
            "GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
            "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
            "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT" \
            "SQ_WAVES_LT_16 SQ_WAVES_LT_32 SQ_WAVES_LT_48 SQ_WAVES_EQ_64 SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 240 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pmc pass failed: $pass"; tail -3 $OUT/pmc_$name.log; }
done

python3 $REPO/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
