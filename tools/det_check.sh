#!/bin/bash
# determinism / equality check between library builds. usage: tools/det_check.sh libA.so libB.so
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
for cfg in "--scene hall260k --spp 1" "--scene hall260k --spp 4" "--scene blob70k --spp 16" "--scene spheres --spp 16" "--scene hall260k --spp 16 --width 480 --height 270"; do
  for lib in "$@"; do
    PTRACE_LIB=$lib PTRACE_BACKEND=mega timeout -k 5 90 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline $cfg 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', '$lib', round(d['value'],1), d['image_mean_rgb'][0], d['image_crc'])" || exit 1
  done
done
