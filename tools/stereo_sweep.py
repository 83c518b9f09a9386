import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
import pathtracing_amd as pa
for b in (1, 2, 3, 5):
    os.environ["STEREO_BLOCKS"] = str(b)
    r = [bench.stereo_frame_time(pa, torch, frames=60)["ms_per_frame"] for _ in range(2)]
    print("STEREO_BLOCKS", b, [round(x, 4) for x in r], flush=True)
