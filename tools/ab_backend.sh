#!/bin/bash
# A/B the path kernels on the GPU box: tools/ab_backend.sh "<env assignments>" ...   (each argument is one configuration,
# e.g. "PTRACE_BACKEND=sched" "PTRACE_BACKEND=pool PTRACE_POOL_VOTES=56,56,32,48"); one dispatch at a time.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg timeout -k 5 120 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stereo --no-reproj --in-flight ${IN_FLIGHT:-1} $BENCH_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],1), round(d['ms_per_step'],3), d['image_mean_rgb'][0], d['image_crc'], round(d['roofline']['nodes_per_ray'],2))" || exit 1
  done
done
