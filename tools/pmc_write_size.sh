#!/bin/bash
# WRITE_SIZE and FETCH_SIZE (separate passes) of a short one-dispatch-at-a-time bench run; LIB selects a library through PTRACE_LIB.
# usage: tools/pmc_write_size.sh <tag> [bench args...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcs_$TAG
mkdir -p $OUT
[ -n "$LIB" ] && export PTRACE_LIB=$LIB
export TMPDIR=/tmp
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c -d $OUT/$c -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj "$@" > $OUT/$c.log 2>&1 || { echo "pass $c failed"; tail -3 $OUT/$c.log; exit 1; }
  python3 $REPO/tools/prof_summary.py $OUT/$c 2>&1 | grep -E "path_" | sed "s/^/[$TAG] /"
  find $OUT/$c -name "*.db" -delete
done
