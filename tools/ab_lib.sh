#!/bin/bash
# A/B of whole libptrace builds on the GPU box: tools/ab_lib.sh lib1.so lib2.so ... (names in csrc/, or absolute paths); per
# library the bench workload one dispatch at a time and three in flight, and the city10m demo flattened and two-level.
# The candidates are loaded through PTRACE_LIB (pathtracing_amd/_ffi.py): the installed libptrace.so is never touched.
for lib in "$@"; do
  echo "== $lib"
  for inflight in 1 3; do
    PTRACE_LIB=$lib timeout -k 5 150 python -u bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-stereo --no-reproj --in-flight $inflight 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight $inflight', round(d['value'],1), round(d['value_single_dispatch'],1), d['image_crc'])" || exit 1
  done
  [ -n "$NO_DEMO" ] || PTRACE_LIB=$lib timeout -k 10 300 python -u tools/inst_demo.py 2>&1 | grep instancing
done
