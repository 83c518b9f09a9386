"""Option "fast_intrinsics": Shade() on the hardware's approximate reciprocal / square root / exp2 / log2 / sine instead of the
correctly rounded, software-defined intrinsics the oracle shares.  Not bit-exact by construction; what must hold instead:
the primary hit ids are identical (camera rays and the traversal are untouched), the random numbers are the same (rng_next
keeps its own sine), and the N-spp image stays within a stated relative RMSE of the bit-exact one - which other tests prove
equal to the oracle's - at 16 and at 256 samples per pixel, on the scenes of configurations 1, 2 and 4 at 1920x1080."""
import numpy as np
import pytest

import pathtracing_amd as pa
from oracle import pt_oracle as po

pytestmark = pytest.mark.gpu

RMSE_BOUND = 5e-3          # relative RMSE over RGB; measured 0.7e-3 .. 1.6e-3 (tools/fast_rmse.py)


def _rel_rmse(a, b):
    a = a[..., :3].astype(np.float64); b = b[..., :3].astype(np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


def _frames(t, s, W, H, n):
    t.stats()
    for k in range(n):
        t.render(pa.make_params(s, W, H, seed=pa.frame_seed(k), spp=16, max_bounces=8, accumulate=pa.PT_ACCUM_SUM, sample_index=k))
    return t.read_converged() / n, t.read_hit_ids(), t.stats()["rays"]


@pytest.mark.parametrize("name", ["spheres", "blob70k", "hall260k"])
def test_fast_intrinsics_keep_hits_and_stay_within_the_rmse_bound(name):
    W, H = 1920, 1080
    s = pa.Scene(name); sky = pa.procedural_sky(512, 256)
    got = {}
    for fast in (0, 1):
        with pa.PathTracer(0) as t:
            t.set_option("fast_intrinsics", fast)
            t.load_scene(s, sky); t.resize(W, H)
            got[fast, 16] = _frames(t, s, W, H, 1)
            got[fast, 256] = _frames(t, s, W, H, 16)
    for n in (16, 256):
        exact, fast = got[0, n], got[1, n]
        assert (exact[1] == fast[1]).all(), f"{name}, {n} spp: primary hit ids differ"
        r = _rel_rmse(fast[0], exact[0])
        assert 0.0 < r < RMSE_BOUND, (name, n, r)              # > 0: the option did change the arithmetic
        assert abs(fast[2] - exact[2]) < 1e-4 * exact[2]       # a few paths end a bounce earlier or later, no more
        assert np.isfinite(fast[0]).all()
    # and against the oracle itself, on a tile of the 16-spp frame
    tile = (W // 2 - 80, H // 2 - 45, 160, 90)
    p = pa.make_params(s, W, H, seed=pa.frame_seed(0), spp=16, max_bounces=8, tile=tile)
    c, cid, _ = po.Oracle.from_scene(s, sky, accel=(name != "spheres")).render(p, W, H, threads=16)
    ys, xs = slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2])
    assert (got[1, 16][1][ys, xs] == cid[ys, xs]).all()
    assert (got[0, 16][0][ys, xs].view(np.uint32) == c[ys, xs].view(np.uint32)).all()       # the exact kernel: bit for bit
    assert _rel_rmse(got[1, 16][0][ys, xs], c[ys, xs]) < 2 * RMSE_BOUND                     # the fast one: within the bound (a tile is noisier)


def test_fast_intrinsics_leave_the_extended_kernels_alone(sky_small):
    """Materials, instances and the renderModes of the extended kernel keep the exact arithmetic whatever the option says."""
    s = pa.Scene("fixture")
    W, H = 64, 36
    mats = pa.materials([((0.8, 0.3, 0.2), 0.5, (0.1, 0.2, 0.3), pa.PT_LOBE_GGX)] * len(s.objects))
    p = pa.make_params(s, W, H, seed=0.5, spp=4, max_bounces=6)
    c, cid, _ = po.Oracle(s.spheres, s.objects, s.vertices, s.indices, sky=sky_small, materials=mats).render(p, W, H, threads=4)
    with pa.PathTracer(0) as t:
        t.set_option("fast_intrinsics", 1)
        t.load_scene(s, sky_small); t.set_mesh_materials(mats); t.resize(W, H)
        t.render(p)
        assert (t.read_hit_ids() == cid).all()
        assert (t.read_result().view(np.uint32) == c.view(np.uint32)).all()
