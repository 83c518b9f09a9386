"""Experiment: throughput of the headline workload with two contexts (half the resident-block slots each) rendering
alternate frames on their own streams, against one context with the full grid."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pathtracing_amd as pa

scene = pa.Scene("hall260k"); sky = pa.procedural_sky(2048, 1024)
W, H = 1920, 1080
def run(n_ctx, blocks, frames=24):
    trs = []
    for _ in range(n_ctx):
        t = pa.PathTracer(0); t.load_scene(scene, sky); t.resize(W, H); t.set_option("blocks_per_cu", blocks); trs.append(t)
    for k in range(n_ctx):
        trs[k].render(pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=16, max_bounces=8, accumulate=pa.PT_ACCUM_SUM, sample_index=0))
    for t in trs: t.sync(); t.stats()
    t0 = time.perf_counter()
    for k in range(frames):
        trs[k % n_ctx].render(pa.make_params(scene, W, H, seed=pa.frame_seed(10 + k), spp=16, max_bounces=8, accumulate=pa.PT_ACCUM_SUM, sample_index=1 + k // n_ctx))
    for t in trs: t.sync()
    dt = time.perf_counter() - t0
    rays = sum(t.stats()["rays"] for t in trs)
    for t in trs: t.close()
    return rays / dt / 1e6, dt / frames * 1e3
for n, b in ((3, 0), (4, 0), (6, 0), (4, 3)):
    print(n, b, run(n, b), flush=True)
