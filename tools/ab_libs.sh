#!/bin/bash
# A/B different builds of libptrace.so on the GPU box: tools/ab_libs.sh lib1.so lib2.so ...  (names in csrc/, or absolute
# paths; loaded through PTRACE_LIB, the installed library stays as it is and runs first as "base")
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for lib in base "$@"; do
  if [ "$lib" = base ]; then unset PTRACE_LIB; else export PTRACE_LIB=$lib; fi
  for rep in 1 2; do
    PTRACE_BACKEND=${PTRACE_BACKEND:-mega} timeout -k 5 90 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --in-flight 1 $BENCH_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), round(d['ms_per_step'],3), d['image_mean_rgb'][0], d['image_crc'])" || exit 1
  done
done
