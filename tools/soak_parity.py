"""Randomised parity soak (run on the GPU box): random scene / size / samples / bounces / renderMode / stripe count,
every frame of a short moving-camera sequence compared bit for bit with the oracle.
usage: python tools/soak_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracing_amd as pa  # noqa: E402
from oracle import pt_oracle as po  # noqa: E402
from reproj_util import (gpu_pause_sequence, gpu_sequence, gpu_stripe_sequence, oracle_pause_sequence,  # noqa: E402
                         oracle_sequence, oracle_stripe_sequence)



def run(budget=60.0, seed=1):
    """One soak of `budget` seconds from `seed`; raises AssertionError on the first mismatch, returns the count of checked items."""
    rng = np.random.default_rng(seed)
    sky = pa.procedural_sky(128, 64)
    scenes = {n: pa.Scene(n) for n in os.environ.get("SOAK_SCENES", "fixture,xr,blob70k").split(",")}   # e.g. SOAK_SCENES=hall260k
    oracles = {n: po.Oracle.from_scene(s, sky, accel=True) for n, s in scenes.items()}
    t = pa.PathTracer(0)
    t0 = time.time(); runs = 0; frames = 0
    while time.time() - t0 < budget:
        name = str(rng.choice(list(scenes))); s = scenes[name]; o = oracles[name]
        W = int(rng.integers(9, 97 if name in ("blob70k", "hall260k") else 161)); H = int(rng.integers(9, 90))
        mode = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12]))
        spp = 1 if mode >= 10 else int(rng.integers(1, 4)); maxb = int(rng.integers(1, 7)); n = int(rng.integers(3, 7))
        step = tuple(float(x) for x in rng.uniform(-0.3, 0.3, 3))
        t.set_option("bvh_width", int(rng.choice([2, 4, 8])))
        t.load_scene(s, sky); t.resize(W, H)
        desc = f"{name} {W}x{H} mode {mode} spp {spp} bounces {maxb} frames {n} step {step}"
        if mode in (5, 6, 7, 8):
            d = int(rng.integers(1, 7)); desc += f" divisions {d}"
            want = oracle_stripe_sequence(o, s, W, H, n, d, mode, spp=spp, max_bounces=maxb, step=step, threads=8)
            got = gpu_stripe_sequence(t, s, W, H, n, d, mode, spp=spp, max_bounces=maxb, step=step)
        elif mode in (11, 12):
            want = oracle_pause_sequence(o, s, W, H, 2, n, mode, max_bounces=maxb, step=step, threads=8)
            got = gpu_pause_sequence(t, s, W, H, 2, n, mode, max_bounces=maxb, step=step)
        else:
            kw = dict(left_eye=int(rng.integers(0, 2))) if mode == 4 else {}
            fn = int(rng.integers(2, 9))
            want = oracle_sequence(o, s, W, H, n, spp=spp, max_bounces=maxb, frames_n=fn, mode=mode, step=step, threads=8, **kw)
            got = gpu_sequence(t, s, W, H, n, spp=spp, max_bounces=maxb, frames_n=fn, mode=mode, step=step, **kw)
        for k, (g, w) in enumerate(zip(got, want)):
            targets = [0, 1] + ([4] if len(w) > 4 else [])
            for i in targets:
                if not (g[i].view(np.uint32) == w[i].view(np.uint32)).all():
                    bad = np.argwhere((g[i].view(np.uint32) != w[i].view(np.uint32)).any(-1))
                    print("MISMATCH:", desc, "frame", k, "target", i, "pixels", len(bad), "first", bad[0], flush=True)
                    raise AssertionError("soak mismatch (details printed above)")
            frames += 1
        runs += 1
        if runs % 10 == 0:
            print(f"{runs} sequences, {frames} frames identical, {time.time() - t0:.0f} s", flush=True)
    print(f"soak ok: {runs} sequences, {frames} frames, all bit-identical", flush=True)
    t.close()
    return frames


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
