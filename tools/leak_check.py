"""Device-memory leak check (run on the GPU box): contexts created, used in every renderMode, resized and destroyed in a
loop; free device memory before and after must agree (hipMemGetInfo through torch)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as pa  # noqa: E402

sky = pa.procedural_sky(256, 128)
scenes = [pa.Scene("fixture"), pa.Scene("blob70k")]


def cycle(k):
    s = scenes[k % 2]
    with pa.PathTracer(0) as t:
        t.load_scene(s, sky)
        for W, H in ((64, 36), (200, 120), (33, 17)):
            t.resize(W, H)
            for mode in (1, 2, 0, 3, 10, 12, 6, 8):
                hist = [pa.camera_history_entry(tuple(s.eye), tuple(s.target), s.fov, W / H)]
                t.set_camera_history(hist)
                t.render(pa.make_params(s, W, H, spp=1, max_bounces=3, render_mode=mode, accumulation_frames=4, divisions=3, counter=k % 3))
            t.display_denoise()
        t.update_transforms(s.objects)
        t.set_mesh_materials(pa.materials([((0.5, 0.5, 0.5), 0.5, (0.1, 0.1, 0.1), pa.PT_LOBE_GGX)] * len(s.objects)))
        t.render(pa.make_params(s, 33, 17, spp=1, max_bounces=3))
        t.trace_rays(np.array([[0, 5, 0, 0, -1, 0]], np.float32))
        t.sync()
    if k % 5 == 0:
        with pa.PathTracerGroup([0, 0]) as g:
            for m in g.members:
                m.load_scene(s, sky); m.resize(64, 36)
                m.render(pa.make_params(s, 64, 36, spp=1, max_bounces=2, accumulate=pa.PT_ACCUM_SUM))
            g.reduce(0)
            g.members[0].sync()


torch.cuda.init()
cycle(0); cycle(1)                                   # first use: runtime pools, code objects
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for k in range(N):
    cycle(k)
    if k % 10 == 9:
        print(k + 1, "cycles, free device memory change:", (torch.cuda.mem_get_info()[0] - free0) / 2**20, "MiB", flush=True)
torch.cuda.synchronize()
delta = (torch.cuda.mem_get_info()[0] - free0) / 2**20
print("leak check:", "ok" if delta > -8 else "LEAK", f"({delta:+.1f} MiB after {N} cycles)", flush=True)
sys.exit(0 if delta > -8 else 1)
