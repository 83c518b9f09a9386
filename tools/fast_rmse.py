#!/usr/bin/env python3
"""What option "fast_intrinsics" changes: per scene (configurations 1, 2 and 4's geometry at 1920x1080, 8 bounces) the primary
hit ids, the relative RMSE of the 16-spp and the 256-spp image against the bit-exact kernel's (which equals the oracle's), and
the rays per second of both.  usage: fast_rmse.py [scene ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pa


def rel_rmse(a, b):
    a = a[..., :3].astype(np.float64); b = b[..., :3].astype(np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


def frames(t, scene, W, H, n, spp=16):
    """n frames of spp samples summed into the converged target; returns (mean image, ids, rays, seconds)."""
    t.stats(); t.sync()
    t0 = time.perf_counter()
    for k in range(n):
        t.render(pa.make_params(scene, W, H, seed=pa.frame_seed(k), spp=spp, max_bounces=8, accumulate=pa.PT_ACCUM_SUM, sample_index=k))
    t.sync()
    dt = time.perf_counter() - t0
    return t.read_converged() / n, t.read_hit_ids(), t.stats()["rays"], dt


def main(names):
    W, H = 1920, 1080
    sky = pa.procedural_sky(2048, 1024)
    for name in names:
        s = pa.Scene(name)
        out = {}
        for fast in (0, 1):
            with pa.PathTracer(0) as t:
                t.set_option("fast_intrinsics", fast)
                t.load_scene(s, sky); t.resize(W, H)
                frames(t, s, W, H, 2)                                   # build, block order
                out[fast, 16] = frames(t, s, W, H, 1)
                out[fast, 256] = frames(t, s, W, H, 16)
        for n in (16, 256):
            e, f = out[0, n], out[1, n]
            print(f"{name} {n} spp: primary ids differ at {int((e[1] != f[1]).sum())} of {W * H} pixels, relative RMSE {rel_rmse(f[0], e[0]):.5f}, "
                  f"mean {f[0][..., :3].mean():.5f} vs {e[0][..., :3].mean():.5f}, rays {f[2]} vs {e[2]}, "
                  f"Mrays/s {f[2] / f[3] / 1e6:.0f} vs {e[2] / e[3] / 1e6:.0f}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["spheres", "blob70k", "hall260k"])
