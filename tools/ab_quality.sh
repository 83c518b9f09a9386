#!/bin/bash
# A/B builds of libptrace.so by tree quality (deterministic) and rate: tools/ab_quality.sh lib1.so ...  (paths relative to csrc/)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for lib in base "$@"; do
  if [ "$lib" = base ]; then unset PTRACE_LIB; else export PTRACE_LIB=$lib; fi      # loaded through PTRACE_LIB (_ffi.py): the installed library is not touched
  for rep in 1 2; do
    timeout -k 5 90 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stereo --no-reproj --in-flight 1 $BENCH_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', round(d['value'],1), 'nodes/ray', round(r['nodes_per_ray'],3), 'prims/ray', round(r['prims_per_ray'],3), 'build_ms', round(d['bvh']['build_ms'],1), 'nodes', d['bvh']['nodes'], d['image_crc'])" || exit 1
  done
done
