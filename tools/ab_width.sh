#!/bin/bash
# Tree widths side by side on the GPU box: tools/ab_width.sh [scene ...]  (default: hall260k blob70k); per scene and width the
# bench workload one dispatch at a time: Mrays/s, node visits and primitive tests per ray, nodes, build time, image checksum.
# WIDTHS="4 8" and LIB (a library loaded through PTRACE_LIB) select what runs.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
[ -n "$LIB" ] && export PTRACE_LIB=$LIB
for scene in ${@:-hall260k blob70k}; do
  for w in ${WIDTHS:-4 8}; do
    for rep in 1 2; do
      PTRACE_BVH_WIDTH=$w timeout -k 5 150 python -u bench.py --scene $scene --steps 5 --warmup 2 --no-cpu-baseline --no-stereo --no-reproj --in-flight ${IN_FLIGHT:-1} $BENCH_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$scene width $w:', round(d['value'],1), 'Mrays/s; one dispatch', round(d['value_single_dispatch'],1), 'kernel ms', round(r['kernel_ms'],3), 'nodes/ray', round(r['nodes_per_ray'],3), 'prims/ray', round(r['prims_per_ray'],3), 'nodes', d['bvh']['nodes'], 'build ms', round(d['bvh']['build_ms'],1), 'crc', d['image_crc'])" || exit 1
    done
  done
done
