import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pathtracing_amd as pa
t0 = time.time(); scene = pa.Scene("city10m"); print("scene gen", time.time() - t0, scene.n_triangles, flush=True)
sky = pa.procedural_sky(2048, 1024)
tr = pa.PathTracer(0)
t0 = time.time(); tr.load_scene(scene, sky); tr.resize(3840, 2160)
p = pa.make_params(scene, 3840, 2160, seed=0.5, spp=1, max_bounces=8)
tr.render(p); tr.sync(); print("first render incl. build", time.time() - t0, flush=True)
print(tr.stats(), flush=True)
for k in range(3):
    tr.render(pa.make_params(scene, 3840, 2160, seed=pa.frame_seed(k), spp=1, max_bounces=8))
st = tr.stats(); print("3 frames", st, "Mrays/s", st["rays"] / st["kernel_ms"] / 1e3, flush=True)
tr.render(pa.make_params(scene, 3840, 2160, depth=1)); d = tr.read_result(); ids = tr.read_hit_ids()
print("depth finite frac", np.isfinite(d[..., 0]).mean(), "mesh frac", (ids > len(scene.spheres)).mean(), flush=True)
# BVH vs exhaustive loop on the device
rng = np.random.default_rng(1); n = 1500
o = rng.uniform(-250, 250, (n, 3)).astype(np.float32); o[:, 1] = rng.uniform(1, 60, n)
t = rng.uniform(-200, 200, (n, 3)).astype(np.float32); t[:, 1] = rng.uniform(0, 10, n)
dd = t - o; dd /= np.linalg.norm(dd, axis=1, keepdims=True)
od = np.concatenate([o, dd.astype(np.float32)], 1)
t0 = time.time(); ib, tb = tr.trace_rays(od); t1 = time.time(); ie, te = tr.trace_rays(od, brute_force=True); t2 = time.time()
print("bvh", t1 - t0, "brute", t2 - t1, "id mismatches", int((ib != ie).sum()), "t mismatches", int((tb.view(np.uint32) != te.view(np.uint32)).sum()), "hit frac", (ib > 0).mean())
