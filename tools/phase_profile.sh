#!/bin/bash
# Per-phase rounds, lane occupancy and wave clocks of path_sched_kernel, from a -DPT_PROFILE_PHASES build
# (pathtracing_amd/csrc/libptrace_prof.so, built by hand with the flags of pathtracing_amd/build.py).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
cp pathtracing_amd/csrc/libptrace.so /tmp/libptrace_base.so
cp pathtracing_amd/csrc/libptrace_prof.so pathtracing_amd/csrc/libptrace.so
timeout -k 5 90 python -u bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj "$@" 2>&1 | grep -E "^phase|value" | cut -c1-400
rc=$?
cp /tmp/libptrace_base.so pathtracing_amd/csrc/libptrace.so
exit $rc
