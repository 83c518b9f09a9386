#!/bin/bash
# One library, several environments, on the one-sample frame times and the 16-spp workload:
#   tools/ab_env_interactive.sh "VAR=a" "VAR=b OTHER=c" ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for envs in "$@"; do
  env $envs timeout -k 5 200 python -u bench.py --steps 4 --warmup 2 --no-cpu-baseline --in-flight 1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['reproj_1080p']; print('[$envs]', '16spp', round(d['value'],1), 'blob70k', round(d['blob70k_1080p']['value'],1), 'stereo', round(d['stereo_1080p']['ms_per_frame'],4), 'plain', round(r['plain_ms_per_frame'],4), 'mode2', round(r['ms_per_frame'],4), 'mode0', round(r['denoise_mode_ms_per_frame'],4), 'stripes', round(r['stripes_ms_per_frame'],4), 'mode10', round(r['other_modes_ms_per_frame']['mode10'],4))" || exit 1
done
