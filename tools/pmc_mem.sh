#!/bin/bash
# One PMC pass on the memory side (texture addresser, L1, LDS) of a short one-dispatch-at-a-time bench run.
# usage: tools/pmc_mem.sh <tag> [bench args...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcm_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj "$@" > $OUT/run.log 2>&1 || { echo "pmc pass failed"; tail -3 $OUT/run.log; exit 1; }
python3 $REPO/tools/prof_summary.py $OUT 2>&1 | grep -i "path_"
find $OUT -name "*.db" -delete
