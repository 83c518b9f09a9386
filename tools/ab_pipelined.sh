#!/bin/bash
# A/B builds of libptrace.so at the default pipelined setting: tools/ab_pipelined.sh lib1.so ...  (paths relative to csrc/)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for lib in base "$@"; do
  if [ "$lib" = base ]; then unset PTRACE_LIB; else export PTRACE_LIB=$lib; fi      # loaded through PTRACE_LIB (_ffi.py): the installed library is not touched
  for rep in 1 2 3; do
    timeout -k 5 120 python -u bench.py --no-cpu-baseline --no-stereo --no-reproj 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step, solo kernel', round(d['roofline']['kernel_ms'],3), 'ms')" || exit 1
  done
done
