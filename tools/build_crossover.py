"""Where the host builder and the device builder cross: build time, rays per second and node visits per ray of both for clouds of 1 000 to 64 000 triangles
(best of four builds each).  usage: python tools/build_crossover.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pathtracing_amd as pa
ref = pa.Scene("fixture"); OBJ = ref.objects.dtype
rng = np.random.default_rng(3)
sky = pa.procedural_sky(64, 32)
with pa.PathTracer(0) as t:
    t.set_sky(sky); t.resize(320, 180); t.set_spheres(ref.spheres)
    for n in (1000, 2000, 4000, 8000, 16000, 32000, 64000):
        c = rng.uniform(-3, 3, (8, 3)); base = c[rng.integers(0, 8, n)] + rng.normal(0, 0.6, (n, 3)); base[:, 1] += 3
        v = (base[:, None, :] + rng.normal(0, 0.05, (n, 3, 3))).reshape(-1, 3).astype(np.float32)
        idx = np.arange(3 * n, dtype=np.int32)
        objs = np.zeros(1, OBJ); objs[0] = (np.eye(4, dtype=np.float32).reshape(16), 0, 3 * n)
        p = pa.make_params(ref, 320, 180, spp=4, max_bounces=4, eye=(0, 4, -9), target=(0, 3, 0))
        line = f"{n:6d} triangles:"
        for build in (0, 1):
            t.set_option("tree_build", build)
            best = 1e9; rate = 0
            for rep in range(4):
                t.set_meshes(objs, v, idx); t.stats(); t.render(p); t.sync(); st = t.stats()
                best = min(best, st["bvh_build_ms"]); rate = max(rate, st["rays"] / st["kernel_ms"] / 1e3)
            line += f"  tree_build {build}: {best:6.2f} ms build, {rate:7.0f} Mrays/s, {st['nodes_visited'] / st['rays']:.2f} visits/ray;"
        print(line, flush=True)
