#!/bin/bash
# A/B builds of libptrace.so with the pool kernel on the GPU box: tools/ab_pool.sh lib1.so lib2.so ... (names in csrc/, loaded
# through PTRACE_LIB); every library runs with PTRACE_BACKEND=pool and each vote setting listed in $VOTES (space separated),
# one dispatch at a time.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for lib in "$@"; do
  for votes in ${VOTES:-56,56,32,48}; do
    PTRACE_LIB=$lib PTRACE_BACKEND=pool PTRACE_POOL_VOTES=$votes timeout -k 5 120 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stereo --no-reproj --in-flight ${IN_FLIGHT:-1} $BENCH_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$votes', round(d['value'],1), round(d['ms_per_step'],3), d['image_crc'])" || exit 1
  done
done
