#!/bin/bash
# Pipelined headline rate against the persistent grid's blocks per CU and the number of dispatches in flight.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for cfg in "5 3" "4 3" "3 3" "2 3" "3 4" "2 4" "4 2" "3 2"; do
  set -- $cfg
  for rep in 1 2; do
    PTRACE_BLOCKS_PER_CU=$1 timeout -k 5 120 python -u bench.py --no-cpu-baseline --no-stereo --no-reproj --in-flight $2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocks/CU $1, in flight $2:', round(d['value'],1), 'Mrays/s')" || exit 1
  done
done
