"""The tree built on the device (option "tree_build" 1: top-down SAH, 3: clustering) against the host's (0) on flattened scenes: build time, tree size,
rays per second, node visits per ray, the image checksum, and an exhaustive-loop ray query on the device-built tree.
usage: python tools/device_build_demo.py [scene ...]      (SPP, RAYS in the environment)"""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pa  # noqa: E402


def main():
    names = sys.argv[1:] or ["blob70k", "hall260k", "city10m"]
    W, H = 1920, 1080
    n_rays = int(os.environ.get("RAYS", "200000"))
    sky = pa.procedural_sky(512, 256)
    for name in names:
        scene = pa.Scene(name)
        spp = int(os.environ.get("SPP", "4" if scene.n_triangles > 2_000_000 else "16"))
        print(f"{name}: {scene.n_triangles} triangles in {len(scene.objects)} mesh objects, {len(scene.spheres)} spheres, {spp} spp", flush=True)
        rng = np.random.default_rng(7)
        lo, hi = scene.vertices.min(0), scene.vertices.max(0)
        crcs = {}
        for build in (int(x) for x in os.environ.get("BUILDS", "0 1 3").split()):
            with pa.PathTracer(0) as tr:
                tr.set_option("instancing", 0); tr.set_option("tree_build", build)
                t0 = time.time()
                tr.load_scene(scene, sky); tr.resize(W, H)
                p = pa.make_params(scene, W, H, seed=0.5, spp=spp, max_bounces=8)
                tr.render(p); tr.sync()
                first = time.time() - t0
                st0 = tr.stats()
                best = None
                for _ in range(3):
                    tr.render(p); tr.sync()
                    st = tr.stats()
                    if best is None or st["kernel_ms"] < best["kernel_ms"]:
                        best = st
                img = tr.read_result(); ids = tr.read_hit_ids()
                crcs[build] = (zlib.crc32(img.tobytes()), zlib.crc32(ids.tobytes()))
                # a second build of the same scene: what a scene change costs once the process is warm
                tr.set_option("tree_build", 1 if build == 0 else 0); tr.set_option("tree_build", build)
                tr.render(p); tr.sync()
                again = tr.stats()["bvh_build_ms"]
                line = (f"  tree_build {build}: build {st0['bvh_build_ms']:8.1f} ms, again {again:8.1f} ms (first frame {first:5.2f} s)  nodes {best['bvh_nodes']:8d} depth {best['bvh_depth']:3d}  "
                        f"{best['rays'] / best['kernel_ms'] / 1e3:8.1f} Mrays/s  kernel {best['kernel_ms']:7.2f} ms  nodes/ray {best['nodes_visited'] / best['rays']:6.2f}  "
                        f"prims/ray {best['prims_tested'] / best['rays']:5.2f}  crc {crcs[build][0]} ids {crcs[build][1]}")
                if build != 0 and n_rays:
                    # rays between random points of the scene's box: the tree against the exhaustive loop, ids and distances bit for bit
                    o = (lo + (hi - lo) * rng.random((n_rays, 3))).astype(np.float32)
                    d = rng.normal(size=(n_rays, 3)).astype(np.float32)
                    d /= np.linalg.norm(d, axis=1, keepdims=True)
                    m = min(n_rays, 2000 if scene.n_triangles > 2_000_000 else 20000)
                    od = np.concatenate([o, d], axis=1)
                    i1, t1 = tr.trace_rays(od)
                    i2, t2 = tr.trace_rays(od[:m], brute_force=True)
                    same = (i1[:m] == i2).all() and (t1[:m].view(np.uint32) == t2.view(np.uint32)).all()
                    line += f"  {m} rays against the exhaustive loop: {'identical' if same else 'DIFFERENT'} ({(i2 > 0).mean():.2f} hit)"
                print(line, flush=True)
        print(f"  images {'identical' if len(set(crcs.values())) == 1 else 'DIFFERENT'}", flush=True)


if __name__ == "__main__":
    main()
