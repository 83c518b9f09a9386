#!/bin/bash
# A/B builds of libptrace.so with tools/scene_rate.py: tools/ab_scene.sh "<scene_rate args>" lib1.so ...  (paths relative to csrc/)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
ARGS=$1; shift
cp pathtracing_amd/csrc/libptrace.so /tmp/libptrace_base.so
for lib in base "$@"; do
  if [ "$lib" = base ]; then cp /tmp/libptrace_base.so pathtracing_amd/csrc/libptrace.so; else cp pathtracing_amd/csrc/$lib pathtracing_amd/csrc/libptrace.so; fi
  for rep in 1 2; do echo -n "$lib: "; timeout -k 5 90 python -u tools/scene_rate.py $ARGS || exit 1; done
done
cp /tmp/libptrace_base.so pathtracing_amd/csrc/libptrace.so
