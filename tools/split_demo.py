"""Reference splitting on a scene it is made for: a detailed mesh inside a room of twelve huge triangles, crossed by long thin
diagonal beams.  Prints node visits per ray and Mrays/s per budget (environment PTRACE_SPLIT is read at context creation)."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pa  # noqa: E402

OBJ = np.dtype([("m", np.float32, 16), ("o", np.int32), ("c", np.int32)])


def box(lo, hi):
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    v = np.array([[(hi if (c >> k) & 1 else lo)[k] for k in range(3)] for c in range(8)], np.float32)
    q = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    idx = []
    for a, b, c, d in q:
        idx += [a, b, c, a, c, d]
    return v, np.array(idx, np.int32)


def rot(axis, deg):
    a = np.radians(deg); c, s = np.cos(a), np.sin(a)
    m = np.eye(4, dtype=np.float32)
    i, j = [(1, 2), (0, 2), (0, 1)][axis]
    m[i, i] = c; m[i, j] = -s; m[j, i] = s; m[j, j] = c
    return m


def build():
    base = pa.Scene("blob70k")
    verts = [np.array(base.vertices, np.float32).reshape(-1, 3)]
    index = [np.array(base.indices, np.int32)]
    objs = [(np.array(base.objects["m"][0], np.float32).reshape(16), 0, len(index[0]))]
    nv, ni = len(verts[0]), len(index[0])
    rng = np.random.default_rng(3)
    # the room
    v, i = box((-40, -0.5, -40), (40, 30, 40))
    verts.append(v); index.append(i + nv); objs.append((np.eye(4, dtype=np.float32).T.reshape(16), ni, len(i))); nv += len(v); ni += len(i)
    # beams: unit-square rods 60 long, turned about two axes
    for b in range(60):
        v, i = box((-30, -0.08, -0.08), (30, 0.08, 0.08))
        m = rot(1, rng.uniform(0, 180)) @ rot(2, rng.uniform(-50, 50))
        m[:3, 3] = (rng.uniform(-10, 10), rng.uniform(6, 20), rng.uniform(-10, 10))
        verts.append(v); index.append(i + nv); objs.append((m.T.reshape(16), ni, len(i))); nv += len(v); ni += len(i)
    return base, np.array(objs, OBJ), np.concatenate(verts), np.concatenate(index)


def main():
    base, objs, verts, index = build()
    sky = pa.procedural_sky(256, 128)
    W, H = 960, 540
    for budget in sys.argv[1:] or ["0", "0.3"]:
        os.environ["PTRACE_SPLIT"] = budget
        with pa.PathTracer(0) as tr:
            tr.set_spheres(base.spheres); tr.set_meshes(objs, verts, index); tr.set_sky(sky)
            tr.resize(W, H)
            p = pa.make_params(base, W, H, seed=0.5, spp=8, max_bounces=8)
            tr.render(p); tr.sync()
            tr.render(p); tr.sync()
            st = tr.stats(); img = tr.read_result()
            print(f"PTRACE_SPLIT={budget}: {st['rays'] / st['kernel_ms'] / 1e3:8.1f} Mrays/s  nodes/ray {st['nodes_visited'] / st['rays']:.2f}  "
                  f"prims/ray {st['prims_tested'] / st['rays']:.2f}  build {st['bvh_build_ms']:.1f} ms  crc {zlib.crc32(img.tobytes())}", flush=True)


if __name__ == "__main__":
    main()
