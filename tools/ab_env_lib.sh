#!/bin/bash
# One alternative library under several environments: tools/ab_env_lib.sh lib.so "VAR=a" "VAR=b" ...
# (the library is loaded through PTRACE_LIB; the installed one is not touched)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
export PTRACE_LIB=$1; shift
for envs in "$@"; do
  for rep in 1 2; do
    env $envs timeout -k 5 90 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stereo --no-reproj --in-flight 1 $BENCH_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$envs]', round(d['value'],1), d['image_crc'])" || exit 1
  done
done
