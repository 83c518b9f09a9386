#!/bin/bash
# Headline rate against the number of dispatches kept in flight.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for n in 1 2 3 4 6; do
  for rep in 1 2; do
    timeout -k 5 120 python -u bench.py --no-cpu-baseline --no-stereo --no-reproj --in-flight $n --steps 12 --warmup 3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight $n:', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step')" || exit 1
  done
done
