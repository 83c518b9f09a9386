"""Sweeps on the 1080p stereo frame time (2 eyes, 1 spp, 4 bounces; bench.stereo_frame_time), one line per setting:
    python tools/stereo_sweep.py blocks      blocks per CU of each eye's persistent grid
    python tools/stereo_sweep.py votes       phase votes of the scheduled kernel (read at pt_create)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402
import pathtracing_amd as pa  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "blocks"
if what == "blocks":
    for b in (1, 2, 3, 5):
        os.environ["STEREO_BLOCKS"] = str(b)
        r = [bench.stereo_frame_time(pa, torch, frames=60)["ms_per_frame"] for _ in range(2)]
        print("STEREO_BLOCKS", b, [round(x, 4) for x in r], flush=True)
else:
    for vs in (24, 32, 38, 44, 52):
        for vn in (8, 16, 24):
            os.environ["PTRACE_VOTE_SHADE"] = str(vs); os.environ["PTRACE_VOTE_NODE"] = str(vn)
            r = [bench.stereo_frame_time(pa, torch, frames=60)["ms_per_frame"] for _ in range(2)]
            print("vote_shade", vs, "vote_node", vn, [round(x, 4) for x in r], flush=True)
