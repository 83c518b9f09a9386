#!/bin/bash
# Where the dominant kernel's write traffic comes from: L2-side and memory-side write counters of a short
# one-dispatch-at-a-time bench run, one rocprofv3 --pmc pass per counter group (at most four TCC counters per pass).
# usage: tools/pmc_write.sh <tag> [bench args...]   (PTRACE_BLOCK_ORDER etc. pass through the environment)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcw_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pass() {
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" -d $OUT/$name -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj $BENCH_ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }
    python3 $REPO/tools/prof_summary.py $OUT/$name 2>&1 | grep -E "path_|order_" | sed "s/^/[$name] /"
    find $OUT/$name -name "*.db" -delete
}
BENCH_ARGS="$*"
pass ea TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum TCC_EA0_WR_UNCACHED_32B_sum || exit 1
pass l2 TCC_WRITE_sum TCC_ATOMIC_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum || exit 1
pass ws WRITE_SIZE TCC_NORMAL_EVICT_sum || exit 1
