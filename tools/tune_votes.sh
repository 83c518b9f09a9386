#!/bin/bash
# Sweep the phase thresholds of a backend: tools/tune_votes.sh <backend> "<vote_shade list>" "<vote_node list>" "<swap_min list>"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for vs in $2; do for vn in $3; do for sm in ${4:-1}; do
  PTRACE_BACKEND=$1 PTRACE_VOTE_SHADE=$vs PTRACE_VOTE_NODE=$vn PTRACE_SWAP_MIN=$sm timeout -k 5 60 python -u bench.py --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 vs $vs vn $vn sm $sm', round(d['value'],1), round(d['ms_per_step'],3), d['image_crc'])" || exit 1
done; done; done
