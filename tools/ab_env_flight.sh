#!/bin/bash
# One library, several environments, the bench workload one dispatch at a time and three in flight:
#   tools/ab_env_flight.sh "" "VAR=a" "VAR=b OTHER=c" ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for envs in "$@"; do
  for inflight in 1 3; do
    env $envs timeout -k 5 150 python -u bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-stereo --no-reproj --in-flight $inflight $BENCH_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$envs] in flight $inflight', round(d['value'],1), round(d['value_single_dispatch'],1), d['image_crc'])" || exit 1
  done
done
