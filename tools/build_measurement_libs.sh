#!/bin/bash
# The measurement builds of libptrace.so that the profile scripts load through PTRACE_LIB (git-ignored; built here, on the
# CPU, from the same sources and flags as the installed library plus one macro each):
#   libptrace_prof.so   -DPT_PROFILE_PHASES    clock counters per phase of the scheduled kernel
#   libptrace_ovf.so    -DPT_COUNT_OVERFLOW    counts stack entries that leave the LDS part of the stack
#   libptrace_w4.so     -DPT_SCHED_WAVES=4     four waves per SIMD (128 registers, no spill) instead of five
#   libptrace_3load.so  -DPT_NODES_3LOAD       four-wide nodes of 48 bytes without links
cd $(dirname $0)/../pathtracing_amd/csrc
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function"
pids=()
hipcc $F -DPT_PROFILE_PHASES -o libptrace_prof.so ptrace_api.hip bvh.cpp -lpthread & pids+=($!)
hipcc $F -DPT_COUNT_OVERFLOW -o libptrace_ovf.so ptrace_api.hip bvh.cpp -lpthread & pids+=($!)
hipcc $F -DPT_SCHED_WAVES=4 -o libptrace_w4.so ptrace_api.hip bvh.cpp -lpthread & pids+=($!)
hipcc $F -DPT_NODES_3LOAD -o libptrace_3load.so ptrace_api.hip bvh.cpp -lpthread & pids+=($!)
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
ls -la libptrace*.so; exit $rc
