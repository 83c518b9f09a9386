#!/bin/bash
# The part of tools/profile_r02.sh that depends on the scheduled kernel, for a refresh after a kernel change (GPU box;
# outputs under gpurun_out/r02b/).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r02b; mkdir -p $O
tools/profile_gpu.sh r02b > $O/profile_gpu.log 2>&1; cp gpurun_out/prof_r02b/summary.txt $O/rocprofv3_summary_sched.txt
echo "profile_gpu done"
PTRACE_BACKEND=sched tools/phase_profile.sh > $O/phase_sched.txt 2>&1
tools/trace_default.sh > $O/trace_default.log 2>&1; cp gpurun_out/prof_pipelined/summary.txt $O/pipelined_trace.txt
echo "trace_default done"
tools/ab_backend.sh "PTRACE_BACKEND=sched PTRACE_BLOCK_ORDER=0" "PTRACE_BACKEND=sched PTRACE_BLOCK_ORDER=1" "PTRACE_BACKEND=pool PTRACE_BLOCK_ORDER=1" "PTRACE_BACKEND=wavefront" > $O/backends.txt 2>&1
echo "backends done"
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
