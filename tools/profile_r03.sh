#!/bin/bash
# Everything the profiles/r03_* files are made from, in one call on the GPU box (outputs under gpurun_out/r03p/).
# Needs the measurement builds csrc/libptrace_prof.so (-DPT_PROFILE_PHASES), csrc/libptrace_w4.so (-DPT_SCHED_WAVES=4) and
# csrc/libptrace_3load.so (-DPT_NODES_3LOAD): tools/build_measurement_libs.sh.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r03p; mkdir -p $O
tools/profile_gpu.sh r03 > $O/profile_gpu.log 2>&1; cp gpurun_out/prof_r03/summary.txt $O/rocprofv3_summary_sched.txt
echo "profile_gpu done"
(tools/ab_width.sh hall260k blob70k; BENCH_ARGS="--spp 4" tools/ab_width.sh city10m) > $O/width_ab.txt 2>&1
(echo "== hall260k"; tools/ab_libs.sh libptrace_3load.so; echo "== blob70k"; BENCH_ARGS="--scene blob70k" tools/ab_libs.sh libptrace_3load.so; echo "== city10m (4 spp)"; BENCH_ARGS="--scene city10m --spp 4" tools/ab_libs.sh libptrace_3load.so) > $O/width_ab_3load.txt 2>&1
echo "width ab done"
for sc in hall260k blob70k; do for w in 4 8; do echo "== $sc width $w"; PTRACE_BVH_WIDTH=$w tools/pmc_quick.sh ${sc}_w$w --scene $sc; done; done > $O/width_pmc.txt 2>&1
for w in 4 8; do echo "== city10m width $w (4 spp)"; PTRACE_BVH_WIDTH=$w tools/pmc_quick.sh city_w$w --scene city10m --spp 4; done >> $O/width_pmc.txt 2>&1
for sc in hall260k blob70k; do echo "== $sc three-load four-wide record"; PTRACE_LIB=libptrace_3load.so tools/pmc_quick.sh ${sc}_3l --scene $sc; done >> $O/width_pmc.txt 2>&1
echo "width pmc done"
(echo "== width 4, exact"; tools/phase_profile.sh; echo "== width 4, fast_intrinsics"; PTRACE_FAST_INTRINSICS=1 tools/phase_profile.sh; echo "== width 8, exact"; PTRACE_BVH_WIDTH=8 tools/phase_profile.sh) 2>&1 | cut -c1-300 > $O/phase_profile.txt
timeout -k 10 300 python -u tools/fast_rmse.py > $O/fast_intrinsics.txt 2>&1
echo "phase + fast done"
(PTRACE_BLOCK_ORDER=1 tools/pmc_write.sh order1; tools/pmc_write_size.sh base; LIB=libptrace_w4.so tools/pmc_write_size.sh w4; LIB=libptrace_w4.so WIDTHS=4 tools/ab_width.sh hall260k) > $O/write_traffic.txt 2>&1
tools/trace_modes.sh > $O/trace_modes.log 2>&1; cp gpurun_out/prof_modes/summary.txt $O/modes_summary.txt
PTRACE_BUILD_TIMING=1 timeout -k 10 500 python -u tools/inst_demo.py > $O/instancing.txt 2>&1
PTRACE_BUILD_TIMING=1 RAYS=20000 timeout -k 10 500 python -u tools/device_build_demo.py > $O/device_build.txt 2>&1
PTRACE_BUILD_TIMING=1 timeout -k 10 200 python -u tools/tree_cost_demo.py blob70k hall260k > $O/tree_cost.txt 2>&1
tools/trace_default.sh > $O/trace_default.log 2>&1; cp gpurun_out/prof_pipelined/summary.txt $O/pipelined_trace.txt
echo "traffic, modes, instancing, pipelined done"
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
