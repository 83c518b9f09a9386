"""Two-level tree (option "instancing" 1) and copies made on the device (2) against the host-built flattened tree (0) on
city10m (38 placements of the 262 k-triangle hall): build time, tree size, rays per second and the image checksum.
usage: python tools/inst_demo.py [width height spp]"""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pa  # noqa: E402


def main():
    W, H, spp = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (1920, 1080, 4)))
    scene = pa.Scene(os.environ.get("SCENE", "city10m"))
    sky = pa.procedural_sky(512, 256)
    print(f"{scene.name}: {scene.n_triangles} triangles in {len(scene.objects)} mesh objects", flush=True)
    for inst in (int(x) for x in os.environ.get("INST", "0 1 2").split()):
        with pa.PathTracer(0) as tr:
            tr.set_option("instancing", inst)
            # the host builder, so that the three modes are compared on what they do with shared meshes (by default the
            # device builds this scene flat, in less time than the host needs for the one shared mesh: TREE_BUILD=2)
            tr.set_option("tree_build", int(os.environ.get("TREE_BUILD", "0")))
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
            img = tr.read_result()
            # one object moves: what the next frame costs on the host side (refit of the flat tree / new top-level tree)
            objs = scene.objects.copy()
            upd = []
            for step in range(3):
                objs["m"][3][12] += 0.25
                tr.sync(); t1 = time.time()
                tr.update_transforms(objs); tr.render(p); tr.sync()
                upd.append((time.time() - t1) * 1e3 - tr.stats()["kernel_ms"])
            print(f"instancing {inst}: build {st0['bvh_build_ms']:8.1f} ms (first frame {first:5.2f} s)  nodes {best['bvh_nodes']:8d}  "
                  f"{best['rays'] / best['kernel_ms'] / 1e3:8.1f} Mrays/s  kernel {best['kernel_ms']:7.2f} ms  nodes/ray {best['nodes_visited'] / best['rays']:6.2f}  "
                  f"prims/ray {best['prims_tested'] / best['rays']:5.2f}  crc {zlib.crc32(img.tobytes())}  a moved object costs {min(upd):6.1f} ms before the frame", flush=True)


if __name__ == "__main__":
    main()
