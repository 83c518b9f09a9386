#!/bin/bash
# A/B of whole libptrace builds on the GPU box: tools/ab_lib.sh lib1.so lib2.so ... (names in csrc/); per library the bench workload
# one dispatch at a time and three in flight, and the city10m demo flattened and two-level.
cp pathtracing_amd/csrc/libptrace.so /tmp/libptrace_base.so
for lib in "$@"; do
  cp pathtracing_amd/csrc/$lib pathtracing_amd/csrc/libptrace.so
  echo "== $lib"
  for inflight in 1 3; do
    timeout -k 5 150 python -u bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-stereo --no-reproj --in-flight $inflight 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight $inflight', round(d['value'],1), round(d['value_single_dispatch'],1), d['image_crc'])" || { cp /tmp/libptrace_base.so pathtracing_amd/csrc/libptrace.so; exit 1; }
  done
  [ -n "$NO_DEMO" ] || timeout -k 10 300 python -u tools/inst_demo.py 2>&1 | grep instancing
done
cp /tmp/libptrace_base.so pathtracing_amd/csrc/libptrace.so
