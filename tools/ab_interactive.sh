#!/bin/bash
# A/B of whole libptrace builds on the one-sample frame times: tools/ab_interactive.sh lib1.so lib2.so ... (names in csrc/,
# loaded through PTRACE_LIB); per library the 1080p stereo frame (2 eyes, 1 spp, 4 bounces) and the 1080p frames of
# renderMode 1 / 2 / 0 / 5 / 6 (ms).
for lib in "$@"; do
  PTRACE_LIB=$lib timeout -k 5 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --in-flight 1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['reproj_1080p']; print('$lib', 'stereo', round(d['stereo_1080p']['ms_per_frame'],4), 'plain', round(r['plain_ms_per_frame'],4), 'mode2', round(r['ms_per_frame'],4), 'mode0', round(r['denoise_mode_ms_per_frame'],4), 'stripes', round(r['stripes_ms_per_frame'],4), round(r['stripes_depth_ms_per_frame'],4), 'mode10', round(r['other_modes_ms_per_frame']['mode10'],4), d['image_crc'])" || exit 1
done
