#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for spp in 1 2 4 8; do for vs in 44 52; do
  PTRACE_VOTE_SHADE=$vs timeout -k 5 90 python -u bench.py --spp $spp --steps $((64/spp)) --warmup 4 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('spp $spp vote_shade $vs', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],4), 'ms/step')" || exit 1
done; done
