"""Randomised geometry soak (run on the GPU box): small random scenes full of awkward primitives - coincident and
degenerate triangles, shared vertices, zero / negative / huge radii, NaN and infinite coordinates, empty lists, scaled and
mirrored, turned and sheared object matrices, objects that share a mesh - rendered by libptrace (both tree widths, large
triangles cut into references or not, shared meshes as instances of one tree or flattened, after a build and after an update
that moves some of the objects) and by the
oracle's exhaustive loop; images and hit ids must agree bit for bit.
usage: python tools/soak_geometry.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("PTRACE_INSTANCE_MIN_TRIS", "1")      # the soak's meshes are tiny: let them be instances all the same
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as pa  # noqa: E402
from oracle import pt_oracle as po  # noqa: E402



def run(budget=60.0, seed=1):
    """One soak of `budget` seconds from `seed`; raises AssertionError on the first mismatch, returns the count of checked items."""
    rng = np.random.default_rng(seed)
    skies = [pa.procedural_sky(64, 32), pa.procedural_sky(1, 1), pa.procedural_sky(7, 3), None]
    ref = pa.Scene("fixture")
    OBJ = ref.objects.dtype
    t = pa.PathTracer(0)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 1e-6, 1e6, np.nan, np.inf, -np.inf], np.float32)


    def random_scene():
        ns = int(rng.integers(0, 9)); nv = int(rng.integers(0, 40)); nt = int(rng.integers(0, 70)) if nv else 0
        if nv and rng.random() < 0.15:         # now and then more than one wave of the device builder's last stage takes (pt_build_sah.hpp, kSahSmall)
            nv = int(rng.integers(40, 200)); nt = int(rng.integers(130, 500))
        sph = np.zeros((ns, 14), np.float32)
        sph[:, 0:3] = rng.uniform(-4, 4, (ns, 3)); sph[:, 1] = np.abs(sph[:, 1])
        sph[:, 3] = rng.choice([0.0, -0.7, 0.3, 1.0, 2.5, 40.0], ns)
        sph[:, 4:7] = rng.uniform(0, 1, (ns, 3)); sph[:, 7:10] = rng.uniform(0, 1, (ns, 3)); sph[:, 10] = rng.uniform(0, 1, ns)
        sph[:, 11:14] = rng.choice([0.0, 0.0, 3.0], (ns, 3))
        v = rng.choice(np.array([-2, -1, 0, 1, 2, 3], np.float32), (nv, 3)) + rng.choice([0.0, 0.0, 0.25], (nv, 3)).astype(np.float32)
        v[:, 1] += 1.0
        idx = rng.integers(0, max(nv, 1), (nt, 3)).astype(np.int32)          # repeated indices -> degenerate triangles
        if nt > 4:
            idx[1] = idx[0]                                                  # coincident twins
            idx[3] = idx[2][::-1]                                            # the same triangle seen from behind
        for a in (sph, v):                                                   # a few special values anywhere
            if a.size and rng.random() < 0.5:
                k = int(rng.integers(1, 4))
                a.reshape(-1)[rng.integers(0, a.size, k)] = rng.choice(special, k)
        n_obj = int(rng.integers(1, 4)) if nt else 0
        objs = np.zeros(n_obj, OBJ)
        cuts = np.sort(rng.integers(0, nt + 1, max(n_obj - 1, 0))) if n_obj else []
        bounds = [0] + list(cuts) + [nt]
        for o in range(n_obj):
            m = np.eye(4, dtype=np.float32)
            if rng.random() < 0.5:                                           # turned about a random axis
                ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, 2 * np.pi)
                K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
                m[:3, :3] = (np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K).astype(np.float32)
            m[:3, :3] *= rng.choice([1.0, 1.0, 0.5, 2.0, -1.0])
            if rng.random() < 0.15:
                m[int(rng.integers(0, 3)), int(rng.integers(0, 3))] += rng.choice([0.3, -0.7])      # a shear: never an instance
            m[:3, 3] = rng.uniform(-1, 1, 3)
            objs[o] = (m.T.reshape(16), 3 * bounds[o], 3 * (bounds[o + 1] - bounds[o]))
        if n_obj >= 2 and rng.random() < 0.6:                                # objects that share a mesh: instances when the option is on
            src = int(rng.integers(0, n_obj))
            for o in range(n_obj):
                if o != src and rng.random() < 0.7:
                    objs[o]["offset"] = objs[src]["offset"]; objs[o]["count"] = objs[src]["count"]
        return sph, objs, v, idx.reshape(-1)


    def same(a, b):
        """Bit-identical, except that any NaN equals any NaN (a NaN emission or albedo reaches the image, and x86 and the GPU
        propagate different NaN payloads)."""
        return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


    t0 = time.time(); n = 0
    W, H = 24, 16
    while time.time() - t0 < budget:
        sph, objs, v, idx = random_scene()
        eye = tuple(float(x) for x in rng.uniform(-6, 6, 3)); eye = (eye[0], abs(eye[1]) + 0.2, eye[2])
        depth = int(rng.random() < 0.1)
        p = pa.make_params(ref, W, H, seed=float(rng.random()), spp=int(rng.integers(1, 4)), max_bounces=int(rng.integers(1, 6)), eye=eye,
                           target=(0.0, 1.0, 0.0), depth=depth, fov=float(rng.choice([20.0, 60.0, 120.0])))
        sky = skies[int(rng.integers(0, len(skies)))]
        mats = None
        if len(objs) and rng.random() < 0.4:       # per-object surfaces (extension), now and then with a special value in them
            ent = []
            for _ in range(len(objs)):
                vals = rng.uniform(0, 1, 10).astype(np.float32)
                if rng.random() < 0.2:
                    vals[int(rng.integers(0, 10))] = rng.choice(special)
                ent.append((tuple(vals[0:3]), float(vals[3]), tuple(vals[4:7]), int(rng.choice([pa.PT_LOBE_PHONG, pa.PT_LOBE_GGX])), tuple(3 * vals[7:10])))
            mats = pa.materials(ent)
        width = int(rng.choice([2, 4, 8]))
        t.set_option("bvh_width", width)
        t.set_option("split_percent", int(rng.choice([0, 30, 30, 400])))
        t.set_option("instancing", int(rng.choice([0, 1, 1, 2, 2])))
        t.set_option("tree_build", int(rng.choice([0, 0, 1, 1, 3])))
        t.set_spheres(sph); t.set_meshes(objs, v, idx); t.set_sky(sky); t.resize(W, H)
        t.set_mesh_materials(mats)
        orc = po.Oracle(sph, objs, v, idx, sky=sky, accel=False, materials=mats)
        t.render(p)
        c, cid, _ = orc.render(p, W, H, threads=4)
        g, gid = t.read_result(), t.read_hit_ids()
        ok = same(g, c) and (gid == cid).all()
        stage = "build"
        if ok and len(objs):
            stage = "refit"
            objs2 = objs.copy()
            still = rng.random(len(objs)) < 0.4                             # some objects stay: their split triangles keep their cut boxes
            objs2["m"][:, 12:15] += np.where(still[:, None], 0.0, rng.uniform(-0.5, 0.5, (len(objs), 3))).astype(np.float32)
            t.update_transforms(objs2)
            t.render(p)
            c2, cid2, _ = po.Oracle(sph, objs2, v, idx, sky=sky, accel=False, materials=mats).render(p, W, H, threads=4)
            g, gid, c, cid = t.read_result(), t.read_hit_ids(), c2, cid2
            ok = same(g, c) and (gid == cid).all()
        if not ok:
            bad = np.argwhere(((g.view(np.uint32) != c.view(np.uint32)) & ~(np.isnan(g) & np.isnan(c))).any(-1) | (gid != cid))
            print("stage", stage, "bvh_width", width, "differing pixels", len(bad), "first", bad[:5].tolist(), flush=True)
            for y, x in bad[:5]:
                print("  pixel", (int(x), int(y)), "gpu", g[y, x], int(gid[y, x]), "oracle", c[y, x], int(cid[y, x]), flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "soak_geometry_failure.npz"), sph=sph, objs=objs, v=v, idx=idx, eye=np.array(eye), seed=p.seed)
            print("MISMATCH after", n, "scenes; inputs saved to gpurun_out/soak_geometry_failure.npz", flush=True)
            raise AssertionError("soak mismatch (details printed above)")
        n += 1
        if n % 200 == 0:
            print(n, "scenes identical,", f"{time.time() - t0:.0f} s", flush=True)
    print(f"geometry soak ok: {n} scenes (each after a build and after a refit) bit-identical", flush=True)
    t.close()
    return n


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
