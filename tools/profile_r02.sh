#!/bin/bash
# Everything the profiles/r02_* files are made from, in one call on the GPU box (outputs under gpurun_out/r02/).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r02; mkdir -p $O
tools/profile_gpu.sh r02 > $O/profile_gpu.log 2>&1; cp gpurun_out/prof_r02/summary.txt $O/rocprofv3_summary_sched.txt
PTRACE_BACKEND=pool tools/profile_gpu.sh r02pool > $O/profile_gpu_pool.log 2>&1; cp gpurun_out/prof_r02pool/summary.txt $O/rocprofv3_summary_pool.txt
PTRACE_BACKEND=sched tools/phase_profile.sh > $O/phase_sched.txt 2>&1
PTRACE_BACKEND=pool tools/phase_profile.sh > $O/phase_pool.txt 2>&1
tools/trace_default.sh > $O/trace_default.log 2>&1; cp gpurun_out/prof_pipelined/summary.txt $O/pipelined_trace.txt
timeout -k 10 120 tools/ubench/gather_rate > $O/gather_rate_4MiB.txt 2>&1
timeout -k 10 120 tools/ubench/gather_rate 262144 > $O/gather_rate_16MiB.txt 2>&1
tools/ab_backend.sh "PTRACE_BACKEND=sched PTRACE_BLOCK_ORDER=0" "PTRACE_BACKEND=sched PTRACE_BLOCK_ORDER=1" "PTRACE_BACKEND=pool PTRACE_BLOCK_ORDER=1" "PTRACE_BACKEND=wavefront" > $O/backends.txt 2>&1
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
