"""What a scene change costs from pt_set_meshes to the first pixel, for the three ways a host can hand over ten million
triangles: objects that share one index range, buffers as RebuildMeshObjectBuffers lays them out (a copy of the mesh per
object, RayTracingMaster.cs:387-408), and the same with every copy different (nothing shared).  Defaults everywhere.
usage: python tools/scene_change_demo.py [scene]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pa  # noqa: E402


def main():
    scene = pa.Scene(sys.argv[1] if len(sys.argv) > 1 else "city10m")
    sky = pa.procedural_sky(64, 32)
    W, H = 640, 360
    objs = scene.objects
    v = np.asarray(scene.vertices, np.float32).reshape(-1, 3); idx = np.asarray(scene.indices, np.int32)

    def copies(nudge):
        out_v, out_i, o2 = [], [], objs.copy()
        n_v = n_i = 0
        for k in range(len(objs)):
            off, cnt = int(objs[k]["offset"]), int(objs[k]["count"])
            ix = idx[off:off + cnt]
            lo, hi = int(ix.min()), int(ix.max()) + 1
            vv = v[lo:hi].copy()
            if nudge:
                vv[k % len(vv), 1] += 1e-3 * (k + 1)
            out_v.append(vv); out_i.append(ix - lo + n_v)
            o2[k]["offset"] = n_i; o2[k]["count"] = cnt
            n_v += len(vv); n_i += cnt
        return o2, np.concatenate(out_v), np.concatenate(out_i).astype(np.int32)

    layouts = [("objects share index ranges", (objs, v, idx)), ("a copy of the mesh per object (the reference's layout)", copies(False)),
               ("a copy per object, every copy different", copies(True))]
    p = pa.make_params(scene, W, H, seed=0.5, spp=1, max_bounces=4)
    with pa.PathTracer(0) as tr:
        tr.set_spheres(scene.spheres); tr.set_sky(sky); tr.resize(W, H)
        for label, (o, vv, ii) in layouts:
            for rep in range(3):
                tr.sync(); t0 = time.perf_counter()
                tr.set_meshes(o, vv, ii)
                t1 = time.perf_counter()
                tr.render(p); tr.sync()
                t2 = time.perf_counter()
                st = tr.stats()
            print(f"{label}: {len(ii) // 3} triangles, {vv.nbytes / 1e6:.0f} MB of vertices + {ii.nbytes / 1e6:.0f} MB of indices; pt_set_meshes {1e3 * (t1 - t0):6.1f} ms, "
                  f"first frame {1e3 * (t2 - t1):6.1f} ms (build {st['bvh_build_ms']:.1f}, kernel {st['kernel_ms']:.2f}), {st['bvh_nodes']} nodes", flush=True)


if __name__ == "__main__":
    main()
