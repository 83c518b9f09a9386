"""Randomised parameter soak (run on the GPU box): odd image sizes down to 1x1, sub-rectangles, zero samples, long bounce
loops, every accumulation mode with arbitrary sample indices, depth dispatches, raw camera matrices (eye on or below the
ground plane, looking straight up or down, tiny and huge fields of view) - Result, the converged target and the hit ids
against the oracle, bit for bit (any NaN equal to any NaN).
usage: python tools/soak_params.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as pa  # noqa: E402
from oracle import pt_oracle as po  # noqa: E402



def run(budget=60.0, seed=1):
    """One soak of `budget` seconds from `seed`; raises AssertionError on the first mismatch, returns the count of checked items."""
    rng = np.random.default_rng(seed)
    sky = pa.procedural_sky(64, 32)
    scenes = {n: pa.Scene(n) for n in ("fixture", "xr")}
    oracles = {n: po.Oracle.from_scene(s, sky, accel=True) for n, s in scenes.items()}
    t = pa.PathTracer(0)


    def same(a, b):
        return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


    t0 = time.time(); n = 0; current = None
    while time.time() - t0 < budget:
        name = str(rng.choice(list(scenes))); s = scenes[name]; o = oracles[name]
        W = int(rng.choice([1, 2, 3, 7, 8, 9, 31, 33, 64, 65, int(rng.integers(1, 90))])); H = int(rng.choice([1, 2, 5, 8, 17, 36, int(rng.integers(1, 70))]))
        width = int(rng.choice([2, 4, 8])); reuse = int(rng.random() < 0.2)
        t.set_option("bvh_width", width)
        t.set_option("reuse_primary_hit", reuse)
        if current != name:
            t.load_scene(s, sky); current = name
        t.resize(W, H)
        conv = rng.uniform(0, 2, (H, W, 4)).astype(np.float32)
        t.write_target(pa.PT_TARGET_CONVERGED, conv)
        marker = np.full((H, W, 4), 0.625, np.float32)
        t.write_target(pa.PT_TARGET_RESULT, marker)
        eye = rng.uniform(-12, 12, 3); eye[1] = rng.choice([abs(eye[1]) + 0.1, 0.0, -1.5, 40.0])
        target = rng.uniform(-3, 3, 3)
        if rng.random() < 0.1:
            target = eye + np.array([0, rng.choice([-1.0, 1.0]), 0])           # straight down / up: the up vector is parallel
        tile = None
        if rng.random() < 0.4:
            x0 = int(rng.integers(0, W)); y0 = int(rng.integers(0, H))
            tile = (x0, y0, int(rng.integers(1, W - x0 + 1)), int(rng.integers(1, H - y0 + 1)))
        acc = int(rng.choice([pa.PT_ACCUM_NONE, pa.PT_ACCUM_MEAN, pa.PT_ACCUM_SUM, pa.PT_ACCUM_ADDMAT]))
        sidx = int(rng.choice([0, 1, 2, 9, 1000, 2**31 - 1, 2**32 - 1]))
        depth = int(rng.random() < 0.15)
        if depth:
            acc = pa.PT_ACCUM_NONE
        p = pa.make_params(s, W, H, seed=float(rng.choice([0.0, 1.0, rng.random(), 123.456])), spp=int(rng.choice([0, 1, 1, 2, 3])),
                           max_bounces=int(rng.choice([1, 2, 4, 8, 50])), depth=depth, accumulate=acc, sample_index=sidx, tile=tile,
                           eye=tuple(eye), target=tuple(target), fov=(fov := float(rng.choice([1.0, 30.0, 60.0, 120.0, 175.0]))))
        t.render(p)
        g, gc, gid = t.read_result(), t.read_converged(), t.read_hit_ids()
        c, cid, _ = o.render(p, W, H, threads=4, out=marker.copy())
        want_conv = conv.copy()
        x0, y0, tw, th = tile if tile else (0, 0, W, H)
        region = (slice(y0, y0 + th), slice(x0, x0 + tw))
        if acc == pa.PT_ACCUM_NONE:
            want_conv[region] = marker[region]      # Result1 = Result before anything else (RayTracingShader.compute:728-731)
        elif p.samples_per_pixel > 0 or depth:
            want_conv[region] = po.accumulate(acc, sidx, np.ascontiguousarray(c[region]), np.ascontiguousarray(conv[region])).reshape(th, tw, 4)
        x0, y0, tw, th = tile if tile else (0, 0, W, H)
        inside = np.zeros((H, W), bool); inside[y0:y0 + th, x0:x0 + tw] = True
        ok = same(g, c) and same(gc, want_conv)
        if p.samples_per_pixel > 0 or depth:
            ok = ok and (gid[inside] == cid[inside]).all()
        if not ok:
            print("MISMATCH after", n, ":", name, W, H, "tile", tile, "acc", acc, "sample_index", sidx, "depth", depth, "spp", p.samples_per_pixel,
                  "bounces", p.max_bounces, "eye", eye.tolist(), "target", list(map(float, target)), "seed", p.seed, "fov", fov, "bvh_width", width,
                  "reuse_primary_hit", reuse, flush=True)
            print("  result same", same(g, c), "converged same", same(gc, want_conv), flush=True)
            bad = np.argwhere(~((g.view(np.uint32) == c.view(np.uint32)) | (np.isnan(g) & np.isnan(c))).all(-1))
            for y, x in bad[:4]:
                print("  result pixel", (int(x), int(y)), "gpu", g[y, x], "oracle", c[y, x], flush=True)
            bad = np.argwhere(~((gc.view(np.uint32) == want_conv.view(np.uint32)) | (np.isnan(gc) & np.isnan(want_conv))).all(-1))
            for y, x in bad[:4]:
                print("  converged pixel", (int(x), int(y)), "gpu", gc[y, x], "oracle", want_conv[y, x], "before", conv[y, x], "fresh", c[y, x], flush=True)
            raise AssertionError("soak mismatch (details printed above)")
        n += 1
        if n % 500 == 0:
            print(n, "dispatches identical,", f"{time.time() - t0:.0f} s", flush=True)
    print(f"parameter soak ok: {n} dispatches bit-identical", flush=True)
    t.close()
    return n


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
