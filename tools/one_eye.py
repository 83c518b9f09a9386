"""Experiment: one 1080p eye of the xr scene, 1 spp, 4 bounces: frame time and kernel time against blocks_per_cu."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pa
scene = pa.Scene(sys.argv[1] if len(sys.argv) > 1 else "xr"); sky = pa.procedural_sky(2048, 1024)
W, H = 1920, 1080
t = pa.PathTracer(0); t.load_scene(scene, sky); t.resize(W, H)
for b in (0, 3, 2, 1):
    t.set_option("blocks_per_cu", b)
    ps = [pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=1, max_bounces=4, accumulate=pa.PT_ACCUM_MEAN, sample_index=min(k, 10), fov=90.0) for k in range(40)]
    for k in range(5): t.render(ps[k])
    t.sync(); t.stats()
    t0 = time.perf_counter()
    for k in range(30): t.render(ps[5 + k])
    t.sync(); dt = (time.perf_counter() - t0) / 30 * 1e3
    st = t.stats()
    print(b, "frame ms", round(dt, 3), "kernel span ms/frame", round(st["kernel_ms"] / 30, 3), "rays/frame", st["rays"] // 30, "Grays/s", round(st["rays"] / 30 / dt / 1e6, 2), flush=True)
