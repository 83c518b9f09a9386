#!/bin/bash
# A/B of libptrace builds on the instancing demo (GPU box): tools/ab_inst.sh lib1.so lib2.so ... (names in csrc/)
cp pathtracing_amd/csrc/libptrace.so /tmp/libptrace_base.so
for lib in "$@"; do
  cp pathtracing_amd/csrc/$lib pathtracing_amd/csrc/libptrace.so
  echo "== $lib"
  timeout -k 10 300 python -u tools/inst_demo.py $DEMO_ARGS || { cp /tmp/libptrace_base.so pathtracing_amd/csrc/libptrace.so; exit 1; }
done
cp /tmp/libptrace_base.so pathtracing_amd/csrc/libptrace.so
