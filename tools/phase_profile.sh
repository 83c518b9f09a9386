#!/bin/bash
# Per-phase rounds, lane occupancy and wave clocks of path_sched_kernel, from a -DPT_PROFILE_PHASES build
# (pathtracing_amd/csrc/libptrace_prof.so, or $PROF_LIB; built by hand with the flags of pathtracing_amd/build.py and
# loaded through PTRACE_LIB).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
PTRACE_LIB=${PROF_LIB:-libptrace_prof.so} timeout -k 5 90 python -u bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj "$@" 2>&1 | grep -E "^phase|^waves|value" | cut -c1-400
