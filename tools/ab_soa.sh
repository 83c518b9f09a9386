#!/bin/bash
# Node layout A/B on the GPU box: 64-byte node records (the build) against the same tree as four component arrays
# (csrc/libptrace_soa.so, built with -DPT_NODES_SOA): rays/s one dispatch at a time, node visits per ray, and the L1's
# requests to L2 per ray.  Output: gpurun_out/r2_soa_ab.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r2_soa_ab.txt
: > $OUT
export TMPDIR=/tmp
for lib in base soa; do
  if [ $lib = base ]; then unset PTRACE_LIB; else export PTRACE_LIB=libptrace_soa.so; fi      # loaded through PTRACE_LIB: the installed library is not touched
  for rep in 1 2; do
    timeout -k 5 120 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stereo --no-reproj --in-flight 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'Mrays/s', round(d['value_single_dispatch'],1), 'nodes/ray', round(d['roofline']['nodes_per_ray'],3), 'rays/launch', int(d['config']['rays_per_step']), 'crc', d['image_crc'])" >> $OUT
  done
  ( cd /tmp && timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum -d $REPO/gpurun_out/soa_$lib -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj > $REPO/gpurun_out/soa_$lib.log 2>&1 )
  python3 tools/prof_summary.py gpurun_out/soa_$lib 2>&1 | grep "path_sched" | sed "s/^/$lib /" >> $OUT
  find gpurun_out/soa_$lib -name "*.db" -delete
done
cat $OUT
