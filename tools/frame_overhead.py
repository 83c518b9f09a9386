"""What one 1-spp 1080p frame costs on the host (time inside pt_render) and on the device (events around the frame), back to back frames on one context:
the call returns after 10 us, the frame is the path kernel plus 15 us of copy and fill (rocprofv3 --kernel-trace of this script).
usage: python tools/frame_overhead.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pathtracing_amd as pa
for name, spp, mb in (("spheres", 1, 8), ("xr", 1, 4), ("hall260k", 1, 8)):
    scene = pa.Scene(name); sky = pa.procedural_sky(512, 256)
    W, H = 1920, 1080
    with pa.PathTracer(0) as tr:
        tr.load_scene(scene, sky); tr.resize(W, H)
        ps = [pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=spp, max_bounces=mb) for k in range(8)]
        for k in range(10): tr.render(ps[k % 8])
        tr.sync(); tr.stats()
        N = 200
        t0 = time.perf_counter()
        for k in range(N): tr.render(ps[k % 8])
        t1 = time.perf_counter()
        tr.sync()
        t2 = time.perf_counter()
        st = tr.stats()
        print(f"{name}: wall {1e3*(t2-t0)/N:.3f} ms/frame, host time in pt_render {1e3*(t1-t0)/N:.3f} ms/frame, kernel_ms (events) {st['kernel_ms']/N:.3f} per frame", flush=True)
