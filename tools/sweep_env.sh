#!/bin/bash
# tools/sweep_env.sh VAR "v1 v2 ..." : headline rate (one dispatch at a time) for each value of an environment knob
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for v in $2; do
  env $1=$v timeout -k 5 90 python -u bench.py --steps 4 --warmup 1 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1=$v', round(d['value'],1), round(d['ms_per_step'],3), d['image_crc'])" || exit 1
done
