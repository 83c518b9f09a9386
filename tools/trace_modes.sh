#!/bin/bash
# Kernel trace of one bench.py run that includes the renderMode 0 / 1 / 2 frame times (run on the GPU box).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_modes
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stereo > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 $REPO/tools/prof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
head -12 $OUT/summary.txt
tail -c 700 $OUT/bench.log
