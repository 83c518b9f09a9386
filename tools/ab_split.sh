#!/bin/bash
# Reference-splitting budgets on the GPU box: tools/ab_split.sh 0 0.1 0.3 ... ; one dispatch at a time, bench workload.
for b in "$@"; do
  PTRACE_SPLIT=$b PTRACE_BUILD_TIMING=1 timeout -k 5 150 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stereo --no-reproj --in-flight 1 $BENCH_ARGS > /tmp/ab_split.log 2>&1 || { tail -5 /tmp/ab_split.log; exit 1; }
  grep "scene build" /tmp/ab_split.log | tail -1
  tail -1 /tmp/ab_split.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('PTRACE_SPLIT=$b', round(d['value'],1), round(d['ms_per_step'],3), d['image_crc'], round(d['roofline']['nodes_per_ray'],2), round(d['roofline']['prims_per_ray'],2))"
done
