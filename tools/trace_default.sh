#!/bin/bash
# Kernel trace of bench.py's default pipelined headline run (3 dispatches in flight), secondary measurements off.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_pipelined
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python3 $REPO/bench.py --no-cpu-baseline --no-stereo --no-reproj > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 $REPO/tools/prof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
head -8 $OUT/summary.txt
grep '"metric"' $OUT/bench.log | cut -c1-260
