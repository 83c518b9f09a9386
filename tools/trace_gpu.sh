#!/bin/bash
# kernel trace only. usage: tools/trace_gpu.sh <tag> [bench args]
set -o pipefail
TAG=${1:-t}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
tail -1 $OUT/trace.log | cut -c1-300
python3 $REPO/tools/prof_summary.py $OUT | head -30
