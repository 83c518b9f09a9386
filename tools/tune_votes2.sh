#!/bin/bash
# Sweep of the scheduled kernel's phase votes on the bench workload, one dispatch at a time, six steps after two warm-up steps:
#   tools/tune_votes2.sh "<vote_shade list>" "<vote_node list>"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for vs in $1; do for vn in $2; do
  PTRACE_VOTE_SHADE=$vs PTRACE_VOTE_NODE=$vn timeout -k 5 90 python -u bench.py --steps 6 --warmup 2 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('vote_shade $vs vote_node $vn', round(d['value'],1), round(d['value_single_dispatch'],1), d['image_crc'])" || exit 1
done; done
