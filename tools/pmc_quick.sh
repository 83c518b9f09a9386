#!/bin/bash
# One PMC pass (instruction counts, lane occupancy, waiting) of a short one-dispatch-at-a-time bench run on the GPU box.
# usage: tools/pmc_quick.sh <tag> [bench args...]   (environment variables such as PTRACE_BACKEND pass through)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD -d $OUT -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj "$@" > $OUT/run.log 2>&1 || { echo "pmc pass failed"; tail -3 $OUT/run.log; exit 1; }
python3 $REPO/tools/prof_summary.py $OUT 2>&1 | grep -i "path_" 
find $OUT -name "*.db" -delete
