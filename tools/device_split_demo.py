"""A finely tessellated scene inside a room of twelve wall-sized triangles, built by the host (tree_build 0) and on the device
(1) with and without reference splitting: build time, node visits per ray, rays per second, image checksum.
usage: python tools/device_split_demo.py [scene]"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pa  # noqa: E402


def main():
    s = pa.Scene(sys.argv[1] if len(sys.argv) > 1 else "hall260k")
    sky = pa.procedural_sky(256, 128)
    lo = s.vertices.min(0) - np.float32(6.0); hi = s.vertices.max(0) + np.float32(6.0)
    v = np.array([[(hi if (c >> k) & 1 else lo)[k] for k in range(3)] for c in range(8)], np.float32)
    idx = []
    for a, b, c, d in [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]:
        idx += [a, b, c, a, c, d]
    verts = np.concatenate([s.vertices, v]); index = np.concatenate([s.indices, np.array(idx, np.int32) + len(s.vertices)])
    objs = np.zeros(len(s.objects) + 1, s.objects.dtype)
    objs[:len(s.objects)] = s.objects
    objs[-1] = (np.eye(4, dtype=np.float32).reshape(16), len(s.indices), len(idx))
    W, H = 1920, 1080
    p = pa.make_params(s, W, H, seed=0.5, spp=4, max_bounces=8)
    print(f"{s.name} + a room of 12 triangles: {len(index) // 3} triangles", flush=True)
    for build in (0, 1):
        for percent in (0, 30):
            with pa.PathTracer(0) as t:
                t.set_option("tree_build", build); t.set_option("split_percent", percent)
                t.set_spheres(s.spheres); t.set_meshes(objs, verts, index); t.set_sky(sky); t.resize(W, H)
                t.render(p); t.sync(); st0 = t.stats()
                t.render(p); t.sync(); st = t.stats()
                img = t.read_result()
                print(f"  tree_build {build} split_percent {percent:3d}: build {st0['bvh_build_ms']:7.1f} ms  nodes {st['bvh_nodes']:7d}  {st['rays'] / st['kernel_ms'] / 1e3:8.1f} Mrays/s  "
                      f"nodes/ray {st['nodes_visited'] / st['rays']:6.2f}  prims/ray {st['prims_tested'] / st['rays']:5.2f}  crc {zlib.crc32(img.tobytes())}", flush=True)


if __name__ == "__main__":
    main()
