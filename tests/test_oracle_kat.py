"""Known-answer tests that pin the CPU oracle (the reference has no tests of its
own, so these analytic cases and the committed golden vectors are the only pins).
SH = Assets/Scripts/RayTracingShader.compute of the reference."""
import os

import numpy as np
import pytest

import pathtracing_amd as pa
from oracle import pt_oracle as po

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "fixture_64x36.npz")
MESH_DT = np.dtype([("m", np.float32, 16), ("offset", np.int32), ("count", np.int32)])
IDENT = np.eye(4, dtype=np.float32).reshape(16)


def one_object(n_idx):
    o = np.zeros(1, MESH_DT); o["m"][0] = IDENT; o["offset"] = 0; o["count"] = n_idx
    return o


def no_mesh():
    return np.zeros(0, MESH_DT), np.zeros((0, 3), np.float32), np.zeros(0, np.int32)


def sphere(pos, r, albedo=(0.8, 0.8, 0.8), spec=(0.04, 0.04, 0.04), smooth=0.5, emission=(0, 0, 0)):
    return np.array([*pos, r, *albedo, *spec, smooth, *emission], np.float32)


# ---------------------------------------------------------------- intrinsics

def test_fmath_accuracy():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-10, 10, 2000), rng.uniform(-1e6, 1e6, 2000), [0.0, 1e-8, 3.1415927, 1.5707964]]).astype(np.float32)
    for x in xs:
        x64 = np.float64(x)
        tol = 2e-7 + 1e-7 * abs(x64) * 6e-8 * 4   # polynomial error + reduction error growing with |x|
        assert abs(po.fmath("sin", x) - np.sin(x64)) < max(tol, 3e-7)
        assert abs(po.fmath("cos", x) - np.cos(x64)) < max(tol, 3e-7)
    for x in rng.uniform(-1, 1, 2000).astype(np.float32):
        assert abs(po.fmath("acos", x) - np.arccos(np.float64(x))) < 5e-7
    for y, x in rng.uniform(-3, 3, (2000, 2)).astype(np.float32):
        assert abs(po.fmath("atan2", y, x) - np.arctan2(np.float64(y), np.float64(x))) < 5e-7
    for x, y in zip(rng.uniform(0.001, 1, 2000).astype(np.float32), rng.uniform(0.001, 1, 2000).astype(np.float32)):
        ref = np.float64(x) ** np.float64(y)
        assert abs(po.fmath("pow", x, y) - ref) < 4e-7 * ref + 1e-12
    for s in np.linspace(0, 1, 50).astype(np.float32):
        ref = 1000.0 ** (np.float64(s) ** 2)
        assert abs(po.fmath("pow", 1000.0, s * s) - ref) < 2e-6 * ref
    # edge definitions
    assert po.fmath("pow", 0.0, 0.5) == 0.0 and po.fmath("pow", 0.0, 0.0) == 1.0
    assert po.fmath("acos", 1.5) == 0.0 and abs(po.fmath("acos", -1.5) - np.pi) < 3e-7
    assert po.fmath("atan2", 0.0, 0.0) == 0.0 and abs(po.fmath("atan2", 0.0, -1.0) - np.pi) < 3e-7
    assert po.fmath("sin", 2e9) == 0.0


def test_rand_kat():
    # pixel (0,0): dot(_Pixel, k) == 0 -> sin(0) == 0 -> every draw is 0 (SH:52, SH:831)
    assert (po.rand_sequence(0.5, 0, 0, 16) == 0).all()
    r = po.rand_sequence(0.5, 959, 539, 64)
    assert ((r >= 0) & (r < 1)).all() and len(np.unique(r)) > 32
    # restated in float64 from the formula: frac(sin(seed/100 * dot) * 43758.5453), seed += 1
    seed = np.float32(0.5); px = np.float32(3 * 2.0); py = np.float32(2 * 2.0)
    d = np.float32(np.float32(px * np.float32(12.9898)) + np.float32(py * np.float32(78.233)))
    got = po.rand_sequence(0.5, 3, 2, 8)
    for k in range(8):
        arg = np.float32(np.float32(seed / np.float32(100.0)) * d)
        v = np.sin(np.float64(arg)) * 43758.5453
        want = v - np.floor(v)
        # |arg| is small here, so fp32 sin error * 43758 stays below 0.02
        diff = abs(got[k] - want); diff = min(diff, 1 - diff)
        assert diff < 0.03, (k, got[k], want)
        seed = np.float32(seed + np.float32(1.0))


def test_rand_seed_is_incremented_in_fp32():
    # _Seed += 1.0f is a float32 addition each draw (SH:53): 0.3 -> 1.3 -> 2.3 ... rounds at each step
    a = po.rand_sequence(0.3, 7, 9, 40)
    b = po.rand_sequence(np.float32(0.3), 7, 9, 40)
    assert (a == b).all()


# -------------------------------------------------------------------- camera

def test_camera_rays_analytic():
    # camera at origin looking down +z world (-z camera), fov 90, aspect 1: corner ray = normalize(+-1, +-1, 1)
    c2w, ip = pa.camera_matrices((0, 0, 0), (0, 0, 1), 90.0, 1.0)
    W = H = 64
    r = po.camera_ray(c2w, ip, 0, 0, W, H)          # uv = (-1,-1): pixel CORNER, no +0.5 (SH:842)
    assert np.allclose(r[:3], 0)
    want = np.array([1, -1, 1]) / np.sqrt(3)          # camera right is world -x when looking down +z with up +y
    assert np.allclose(np.abs(r[3:]), np.abs(want), atol=1e-6)
    assert r[4] < 0 and r[5] > 0
    c = po.camera_ray(c2w, ip, W // 2, H // 2, W, H)  # uv = (0,0): forward
    assert np.allclose(c[3:], [0, 0, 1], atol=1e-6)
    # moving +x in pixels and +u in camera space agree in sign with the right vector of cameraToWorld
    rx = po.camera_ray(c2w, ip, W - 1, H // 2, W, H)
    right = c2w[0:3]
    assert np.dot(rx[3:], right) > 0
    # origin is the translation column of cameraToWorld
    c2w2, ip2 = pa.camera_matrices((1, 2, 3), (4, 2, 3), 60.0, 16 / 9)
    assert np.allclose(po.camera_ray(c2w2, ip2, 5, 5, 32, 18)[:3], [1, 2, 3])


# -------------------------------------------------------------- intersection

def test_sphere_and_plane_kat():
    sp = sphere((0, 1, 0), 1.0)[None]
    o = po.Oracle(sp, *no_mesh())
    ids, ts = o.trace_rays([[0, 1, -5, 0, 0, 1]])
    assert ids[0] == 1 and ts[0] == 4.0                       # SH:170-190
    ids, ts = o.trace_rays([[0, 3, 0, 0, -1, 0]])             # from above: sphere top at y = 2
    assert ids[0] == 1 and ts[0] == 1.0
    ids, ts = o.trace_rays([[5, 2, 0, 0, -0.5, 0]])           # plane: t = -o.y/d.y (SH:157)
    assert ids[0] == 0 and ts[0] == 4.0
    ids, ts = o.trace_rays([[5, 2, 0, 0, 1, 0]])              # up: miss
    assert ids[0] == -1 and np.isinf(ts[0])
    ids, ts = o.trace_rays([[0, 1, 0, 0, 0, 1]])              # from inside: p1-p2 < 0 -> p1+p2 (SH:179)
    assert ids[0] == 1 and ts[0] == 1.0
    ids, ts = o.trace_rays([[5, 0.5, 0, 1, 0, 0]])            # parallel to the plane: t = -inf, rejected
    assert ids[0] == -1


def test_mt97_kat():
    # triangle in the plane z = 2, wound so that its front faces -z (det > 0 for rays travelling +z)
    v = np.array([[0, 1, 2], [0, 2, 2], [1, 1, 2]], np.float32)
    idx = np.array([0, 1, 2], np.int32)
    o = po.Oracle(np.zeros((0, 14), np.float32), one_object(3), v, idx)
    tri = 1
    ids, ts = o.trace_rays([[0.25, 1.25, 0, 0, 0, 1]])
    assert ids[0] == tri and ts[0] == 2.0
    ids, _ = o.trace_rays([[0.25, 1.25, 4, 0, 0, -1]])        # back face: culled (SH:206)
    assert ids[0] != tri
    # edges are inclusive: u == 0, v == 0 and u + v == 1 hit; just outside misses (SH:215, SH:223)
    for p, hit in [((0.0, 1.5), True), ((0.5, 1.0), True), ((0.5, 1.5), True), ((-0.001, 1.5), False),
                   ((0.5, 0.999), False), ((0.51, 1.5), False)]:
        ids, _ = o.trace_rays([[p[0], p[1], 0, 0, 0, 1]])
        assert (ids[0] == tri) == hit, (p, ids[0])
    # first-wins tie: two coincident triangles -> the lower index (SH:245 strict <)
    idx2 = np.array([0, 1, 2, 0, 1, 2], np.int32)
    o2 = po.Oracle(np.zeros((0, 14), np.float32), one_object(6), v, idx2)
    assert o2.trace_rays([[0.25, 1.25, 0, 0, 0, 1]])[0][0] == 1
    o2b = po.Oracle(np.zeros((0, 14), np.float32), one_object(6), v, idx2, accel=True)
    assert o2b.trace_rays([[0.25, 1.25, 0, 0, 0, 1]])[0][0] == 1


def test_mesh_matrix_is_applied_column_major():
    v = np.array([[0, 0, 0], [0, 1, 0], [1, 0, 0]], np.float32)
    obj = one_object(3)
    m = np.eye(4, dtype=np.float32); m[:3, 3] = [10, 1, 5]       # translation in the last COLUMN
    obj["m"][0] = m.T.reshape(16)                                 # Unity memory order is column-major
    o = po.Oracle(np.zeros((0, 14), np.float32), obj, v, np.array([0, 1, 2], np.int32))
    ids, ts = o.trace_rays([[10.25, 1.25, 0, 0, 0, 1]])
    assert ids[0] == 1 and ts[0] == 5.0


# ----------------------------------------------------------------------- sky

def test_sky_mapping_and_bilinear():
    H, W = 8, 16
    sky = np.zeros((H, W, 4), np.float32)
    sky[..., 0] = np.arange(W)[None, :]; sky[..., 1] = np.arange(H)[:, None]; sky[..., 3] = 1
    o = po.Oracle(np.zeros((0, 14), np.float32), *no_mesh(), sky=sky)

    def want(d):
        d = np.asarray(d, np.float64)
        theta = np.arccos(d[1]) / -np.pi; phi = np.arctan2(d[0], -d[2]) / -np.pi * 0.5   # SH:329-330
        x = phi * W - 0.5; y = theta * H - 0.5
        x0 = np.floor(x); y0 = np.floor(y); fx = x - x0; fy = y - y0
        ix0 = int(x0) % W; iy0 = int(y0) % H; ix1 = (ix0 + 1) % W; iy1 = (iy0 + 1) % H        # Repeat wrap
        a = sky[iy0, ix0, :3] * (1 - fx) + sky[iy0, ix1, :3] * fx
        b = sky[iy1, ix0, :3] * (1 - fx) + sky[iy1, ix1, :3] * fx
        return (a * (1 - fy) + b * fy) * 2.0                                                     # SH:331

    for d in [(1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, 0, -1), (0.6, 0.8, 0), (0.36, -0.48, 0.8), (0, 0.999, 0.0447)]:
        d = np.array(d, np.float32); d /= np.linalg.norm(d)
        assert np.allclose(o.sky_sample(d), want(d), atol=2e-4), d
    # straight up and straight down land on the wrap seam of v (theta = 0 and theta = -1)
    assert np.isfinite(o.sky_sample(np.array([0, 1, 0], np.float32))).all()
    assert np.isfinite(o.sky_sample(np.array([0, -1, 0], np.float32))).all()


# -------------------------------------------------------------- kernel quirks

def _render(scene_args, p, W, H, accel=False, sky=None):
    return po.Oracle(*scene_args, sky=sky, accel=accel).render(p, W, H)


def test_sky_pixels_keep_energy_snapshot_and_alpha():
    """SH:866 multiplies Shade() by the energy read BEFORE Shade() zeroes it, so sky pixels are not black;
    their alpha is |camera origin| because hit.position stays (0,0,0) (SH:140, SH:860, SH:935)."""
    sky = np.full((4, 8, 4), 0.25, np.float32)
    s = pa.Scene("fixture")
    p = pa.make_params(s, 16, 8, spp=1, max_bounces=4, eye=(0, 5, -10), target=(0, 50, 0))   # looking up
    img, ids, st = _render((np.zeros((0, 14), np.float32), *no_mesh()), p, 16, 8, sky=sky)
    assert (ids == -1).all()
    assert np.allclose(img[..., :3], 0.5)                        # 0.25 * 2.0
    assert np.allclose(img[..., 3], np.sqrt(125.0), rtol=1e-6)
    assert st["trace_calls"] == 2 * 16 * 8 and st["rays"] == 16 * 8   # the shader re-traces the dead ray once


def test_rng_counter_runs_on_across_samples():
    """_Seed is not reset between the samples of one dispatch (SH:839): a 2-spp render is not the
    mean of two 1-spp renders with the same seed."""
    s = pa.Scene("fixture"); sky = pa.procedural_sky(32, 16)
    o = po.Oracle.from_scene(s, sky)
    W, H = 24, 12
    a, _, _ = o.render(pa.make_params(s, W, H, seed=0.5, spp=1, max_bounces=4), W, H)
    b, _, _ = o.render(pa.make_params(s, W, H, seed=0.5, spp=2, max_bounces=4), W, H)
    assert not np.allclose(a[..., :3], b[..., :3])
    assert (a[..., 3] == b[..., 3]).all()                       # same primary ray every sample: no jitter (SH:842)


def test_depth_mode():
    s = pa.Scene("fixture")
    o = po.Oracle.from_scene(s)
    W, H = 16, 9
    d, ids, _ = o.render(pa.make_params(s, W, H, depth=1), W, H)
    c, ids2, _ = o.render(pa.make_params(s, W, H, spp=1, max_bounces=1), W, H)
    assert (ids == ids2).all()
    assert (d[..., 0] == d[..., 3]).all() and (d[..., 1] == d[..., 2]).all()
    hit = ids >= 0
    assert np.isinf(d[~hit]).all()                             # sky: +inf / 16 (SH:806)
    assert np.allclose(d[hit][:, 0] * 16.0, c[hit][:, 3], rtol=1e-6)


def test_white_furnace():
    """Emission-free scene under a constant sky: no pixel is negative or non-finite and the image
    mean stays at or below the sky radiance (the Phong lobe only loses energy below the horizon)."""
    s = pa.Scene("fixture")
    sph = s.spheres.copy(); sph[:, 11:14] = 0                  # no emitters
    sky = np.full((4, 8, 4), 0.5, np.float32)                  # radiance 1.0 after the shader's * 2
    o = po.Oracle(sph, s.objects, s.vertices, s.indices, sky=sky, accel=True)
    W, H = 48, 27
    img, _, _ = o.render(pa.make_params(s, W, H, seed=0.37, spp=32, max_bounces=8), W, H, threads=4)
    rgb = img[..., :3]
    assert np.isfinite(rgb).all() and (rgb >= 0).all()
    assert 0.3 < rgb.mean() <= 1.02


# ------------------------------------------------------------ accel == brute

def test_oracle_bvh_equals_exhaustive_loop():
    s = pa.Scene("fixture"); sky = pa.procedural_sky(32, 16)
    a = po.Oracle.from_scene(s, sky, accel=False); b = po.Oracle.from_scene(s, sky, accel=True)
    W, H = 64, 36
    p = pa.make_params(s, W, H, seed=0.5, spp=3, max_bounces=8)
    ia, ida, sa = a.render(p, W, H, threads=4); ib, idb, sb = b.render(p, W, H, threads=4)
    assert (ida == idb).all() and (ia.view(np.uint32) == ib.view(np.uint32)).all()
    assert sa["rays"] == sb["rays"] and sb["prims_tested"] < sa["prims_tested"] / 20


def test_oracle_bvh_equals_exhaustive_loop_random_rays_70k():
    s = pa.Scene("blob70k")
    a = po.Oracle.from_scene(s, accel=False); b = po.Oracle.from_scene(s, accel=True)
    rng = np.random.default_rng(3)
    n = 600
    o = rng.uniform(-12, 12, (n, 3)).astype(np.float32); o[:, 1] = rng.uniform(0.2, 12, n)
    t = rng.uniform(-3, 3, (n, 3)).astype(np.float32); t[:, 1] += 4.4
    d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    od = np.concatenate([o, d.astype(np.float32)], 1)
    ia, ta = a.trace_rays(od); ib, tb = b.trace_rays(od)
    assert (ia == ib).all() and (ta.view(np.uint32) == tb.view(np.uint32)).all()
    assert (ia > len(s.spheres)).mean() > 0.3


def test_oracle_bvh_keeps_the_sphere_hits_that_only_rounding_produces():
    # IntersectSphere (SH:170-190) takes the sign of p1^2 - d.d + r^2, a difference of squares of the origin's distance:
    # from 40 000 units out on the ground plane it accepts rays that pass several units beside a sphere.  The exhaustive
    # loop reports those hits, so the oracle's tree must not prune them (a session soak found one such pixel).
    s = pa.Scene("xr")
    a = po.Oracle.from_scene(s, accel=False); b = po.Oracle.from_scene(s, accel=True)
    rng = np.random.default_rng(5)
    sph = np.asarray(s.spheres, np.float64)[:, :4]            # position, radius
    rays = []
    for c in sph:
        for _ in range(200):
            ang = rng.uniform(0, 2 * np.pi)
            o = np.array([c[0] + 39450.0 * np.cos(ang), 1e-3, c[2] + 39450.0 * np.sin(ang)])
            fuzz = np.sqrt(c[3] ** 2 + 16 * 6e-8 * 39450.0 ** 2)
            tgt = c[:3] + rng.normal(size=3) * fuzz * 0.7
            d = tgt - o; d /= np.linalg.norm(d)
            rays.append(np.concatenate([o, d]))
    od = np.array(rays, np.float32)
    ia, ta = a.trace_rays(od); ib, tb = b.trace_rays(od)
    assert (ia == ib).all() and (ta.view(np.uint32) == tb.view(np.uint32)).all()
    # the case is present: sphere hits by rays whose distance from the centre (in float64) exceeds the radius by 1 % or more
    hit = (ia >= 1) & (ia <= len(sph))
    o64 = od[:, :3].astype(np.float64); d64 = od[:, 3:].astype(np.float64); d64 /= np.linalg.norm(d64, axis=1, keepdims=True)
    cc = sph[np.clip(ia - 1, 0, len(sph) - 1)]
    v = cc[:, :3] - o64
    miss = np.linalg.norm(v - (v * d64).sum(1, keepdims=True) * d64, axis=1)
    assert (hit & (miss > 1.01 * cc[:, 3])).sum() >= 20


def _exact_outside(od, tri):
    """float64 Moeller-Trumbore: how far outside [0, 1] the barycentrics of ray `od` lie for triangle `tri` (3x3); <= 0 inside."""
    o = od[:3].astype(np.float64); d = od[3:].astype(np.float64)
    v0, e1, e2 = tri[0].astype(np.float64), (tri[1] - tri[0]).astype(np.float64), (tri[2] - tri[0]).astype(np.float64)
    p = np.cross(d, e2); det = e1 @ p
    u = ((o - v0) @ p) / det
    v = (d @ np.cross(o - v0, e1)) / det
    return max(-u, -v, u + v - 1.0)


@pytest.mark.parametrize("which", ["blob70k", "slivers"])
def test_oracle_bvh_keeps_the_triangle_hits_that_only_rounding_produces(which):
    # Moeller-Trumbore's barycentrics (SH:192-230) come out of cross and dot products with the ray's offset from vertex 0:
    # from 1e3 .. 1e5 units away their rounding error is a visible fraction of a small or thin triangle, and the
    # exhaustive loop reports hits by rays that pass beside it.  The oracle's tree widens its boxes per ray for exactly
    # that; the accelerated and the exhaustive oracle must agree on ids and on the bits of t, and the case must occur.
    rng = np.random.default_rng(11)
    if which == "blob70k":
        s = pa.Scene("blob70k")
        sph, objs = s.spheres, s.objects
        verts = np.array(s.vertices, np.float32).reshape(-1, 3); index = np.array(s.indices, np.int32)
        m = np.array(objs["m"][0], np.float32).reshape(4, 4).T
        world = verts @ m[:3, :3].T + m[:3, 3]
        assert len(objs) == 1
    else:
        # 300 slivers: ten units long, a thousandth wide, anywhere in a 20-unit cube
        a = rng.uniform(-10, 10, (300, 3)); dirs = rng.normal(size=(300, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        side = np.cross(dirs, rng.normal(size=(300, 3))); side /= np.linalg.norm(side, axis=1, keepdims=True)
        verts = np.stack([a, a + 10 * dirs, a + 10 * dirs + 1e-3 * side], 1).reshape(-1, 3).astype(np.float32)
        index = np.arange(900, dtype=np.int32)
        s = pa.Scene("fixture")
        sph = np.zeros((0, 14), np.float32)
        objs = np.zeros(1, s.objects.dtype); objs[0] = (np.eye(4, dtype=np.float32).reshape(16), 0, 900)
        world = verts
    tris = world[index].reshape(-1, 3, 3)
    a = po.Oracle(sph, objs, verts, index, accel=False); b = po.Oracle(sph, objs, verts, index, accel=True)
    rays = []
    for _ in range(1500):
        t = tris[rng.integers(0, len(tris))].astype(np.float64)
        dist = 10.0 ** rng.uniform(3, 5)
        k = rng.integers(0, 3); p0, p1, p2 = t[k], t[(k + 1) % 3], t[(k + 2) % 3]
        on_edge = p0 + rng.uniform(0.1, 0.9) * (p1 - p0)
        n = np.cross(t[1] - t[0], t[2] - t[0]); nn = np.linalg.norm(n)
        if not nn > 0:
            continue
        outward = np.cross(p1 - p0, n); outward /= np.linalg.norm(outward)
        if outward @ (p2 - p0) > 0:
            outward = -outward
        L = max(np.linalg.norm(t[1] - t[0]), np.linalg.norm(t[2] - t[0]), np.linalg.norm(t[2] - t[1]))
        fuzz = 8 * 6e-8 * (L * L / nn) * max(L, dist)
        reach = min(fuzz, nn / L)                      # the rounding's reach, but no farther than the triangle is wide
        target = on_edge + outward * reach * rng.uniform(-0.5, 0.6)
        # from the front side (the test culls back faces), some way off the normal
        dirn = n / nn + 0.7 * rng.normal(size=3); dirn /= np.linalg.norm(dirn)
        if dirn @ n < 0:
            dirn = -dirn
        o = target + dirn * dist
        d = target - o; d /= np.linalg.norm(d)
        rays.append(np.concatenate([o, d]))
    od = np.array(rays, np.float32)
    ia, ta = a.trace_rays(od); ib, tb = b.trace_rays(od)
    assert (ia == ib).all() and (ta.view(np.uint32) == tb.view(np.uint32)).all()
    n_s = len(sph)
    hit = np.flatnonzero(ia > n_s)
    assert len(hit) > 100
    # hits the exact arithmetic would not give: the ray (as the float32 inputs define it) passes outside the triangle
    outside = np.array([_exact_outside(od[r], tris[ia[r] - 1 - n_s]) for r in hit])
    # (on a closed mesh the point beside an edge usually lies inside the neighbouring triangle, which is then the hit: few cases)
    assert (outside > 1e-4).sum() >= (5 if which == "blob70k" else 10), (outside > 1e-4).sum()


# -------------------------------------------------------------- accumulation

def test_accumulate_matches_numpy_restatement():
    rng = np.random.default_rng(5)
    fresh = rng.uniform(0, 1.5, (6, 5, 4)).astype(np.float32); hist = rng.uniform(0, 1.5, (6, 5, 4)).astype(np.float32)
    f32 = np.float32
    for S in (0, 1, 7):
        a = f32(1.0) / (f32(S) + f32(1.0))
        assert (po.accumulate(pa.PT_ACCUM_MEAN, S, fresh, hist) == fresh * a + hist * (f32(1.0) - a)).all()
        assert (po.accumulate(pa.PT_ACCUM_SUM, S, fresh, hist) == (fresh if S == 0 else hist + fresh)).all()
        # AddMat.shader:56-63 + Blend SrcAlpha OneMinusSrcAlpha (:11); history = _MainTex, fresh = destination
        alpha = hist[..., 3].copy()
        fire = (hist[..., :3] > f32(0.8)).any(axis=-1); alpha[fire] = f32(0.1)
        t = f32(1.0) + f32(S) * alpha; A = f32(1.0) - f32(1.0) / t
        want = np.empty_like(hist)
        want[..., :3] = hist[..., :3] * A[..., None] + fresh[..., :3] * (f32(1.0) - A)[..., None]
        want[..., 3] = A * A + fresh[..., 3] * (f32(1.0) - A)
        assert (po.accumulate(pa.PT_ACCUM_ADDMAT, S, fresh, hist) == want).all()
    # with alpha == 1 and no firefly the AddMat blend is the running mean with the roles swapped
    h = np.full((1, 1, 4), 0.5, np.float32); h[..., 3] = 1.0
    f = np.full((1, 1, 4), 0.25, np.float32)
    got = po.accumulate(pa.PT_ACCUM_ADDMAT, 3, f, h)
    assert np.allclose(got[..., :3], 0.5 * 3 / 4 + 0.25 / 4)


# ------------------------------------------------------------------- goldens

def test_oracle_reproduces_committed_golden_vectors():
    g = np.load(GOLDEN)
    for (x, y), want in zip(g["rand_pixels"], g["rand"]):
        assert (po.rand_sequence(0.5, int(x), int(y), 16) == want).all()
    for (x, y), want in zip(g["cam_pixels"], g["cam_rays"]):
        assert (po.camera_ray(g["cam_c2w"], g["cam_ip"], int(x), int(y), 1920, 1080) == want).all()
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    o = po.Oracle.from_scene(s, sky)
    W, H = 64, 36
    for name, kw in {"spp1_b4": dict(spp=1, max_bounces=4), "spp1_b8": dict(spp=1, max_bounces=8),
                     "spp16_b8": dict(spp=16, max_bounces=8), "depth": dict(depth=1)}.items():
        img, ids, st = o.render(pa.make_params(s, W, H, seed=0.5, **kw), W, H, threads=4)
        assert (img.view(np.uint32) == g[name].view(np.uint32)).all(), name
        assert (ids == g["ids"]).all()
        assert [st["rays"], st["trace_calls"], st["paths"]] == g[name + "_rays"].tolist()


def test_threads_do_not_change_the_image():
    s = pa.Scene("fixture"); sky = pa.procedural_sky(32, 16)
    o = po.Oracle.from_scene(s, sky)
    p = pa.make_params(s, 32, 18, spp=2, max_bounces=8)
    a, _, _ = o.render(p, 32, 18, threads=1); b, _, _ = o.render(p, 32, 18, threads=8)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_bunny_obj_from_reference_checkout():
    path = "/root/reference/Assets/Meshes/BunnyLow.obj"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    s = pa.Scene("fixture")
    m = np.eye(4, dtype=np.float32); m[:3, :3] *= 1.5; m[:3, 3] = [-3, 1.0, -2]
    s.add_obj(path, m.T.reshape(16))
    assert s.n_triangles == 320 + 432 and len(s.objects) == 2    # the asset: 220 vertices, 432 triangles
    a = po.Oracle.from_scene(s, accel=False); b = po.Oracle.from_scene(s, accel=True)
    p = pa.make_params(s, 48, 27, spp=1, max_bounces=4)
    ia, ida, _ = a.render(p, 48, 27, threads=4); ib, idb, _ = b.render(p, 48, 27, threads=4)
    assert (ida == idb).all() and (ia.view(np.uint32) == ib.view(np.uint32)).all()
    assert (ida > len(s.spheres) + 320).any()                     # the bunny is hit


# ---------------------------------------------------- renderMode 2 (reprojection)

def test_find_camera_ray_is_the_old_cameras_projection():
    """FindCameraRay (SH:394-411) returns the pixel position (+0.5, and 0.005 less in y) at which the old
    camera saw a world point; checked against projection * worldToCamera done in float64."""
    W, H = 1920, 1080
    eye, target = (1.0, 4.0, -9.0), (0.5, 1.0, 0.0)
    h = pa.camera_history_entry(eye, target, 60.0, W / H)
    w2c = np.array(h.world_to_camera, np.float64).reshape(4, 4).T      # column-major storage
    prj = np.array(h.camera_projection, np.float64).reshape(4, 4).T
    rng = np.random.default_rng(3)
    for _ in range(50):
        x = np.array([rng.uniform(-6, 6), rng.uniform(0, 5), rng.uniform(-2, 8), 1.0])
        clip = prj @ (w2c @ x)
        ndc = clip[:2] / clip[3]
        want = (ndc + 1.0) * np.array([W, H]) / 2.0 + np.array([0.5, 0.5 - 0.005])
        got = po.find_camera_ray(h, W, H, x[:3] - np.array(eye))
        assert np.allclose(got, want, atol=2e-2), (got, want)


def test_find_camera_ray_without_history_is_nan():
    """Before the first upload the shader's history arrays hold zeros: near = -0/0, so no comparison of
    SH:965-968 can hold and the pixel takes the fresh value."""
    z = pa.PtCameraHistory()
    assert np.isnan(po.find_camera_ray(z, 64, 36, (0.0, -1.0, 2.0))).all()


def test_bilinear_texture_sample_kat():
    """BilinearTextureSample (SH:440-463) on a ramp image: away from integer positions it is the usual
    bilinear filter with texel centres at integers + 0.5; at an exactly integer P - 0.5 its index
    arithmetic (floor(...)/size - s/2, truncated) lands one texel low in both axes; loads outside
    the texture read as zero."""
    W, H = 64, 36
    img = np.zeros((H, W, 4), np.float32)
    img[..., 0] = np.arange(W, dtype=np.float32)[None, :]
    img[..., 1] = np.arange(H, dtype=np.float32)[:, None]
    img[..., 3] = 1.0
    got = po.bilinear_sample(img, 10.75 + 0.5, 20.25 + 0.5)
    assert np.allclose(got, [10.75, 20.25, 0.0, 1.0], atol=1e-5)
    got = po.bilinear_sample(img, 10.0 + 0.5, 20.0 + 0.5)              # f = 0: C11 alone, and C11 is texel (9, 19)
    assert (got == np.array([9.0, 19.0, 0.0, 1.0], np.float32)).all()
    got = po.bilinear_sample(img, 63.75 + 0.5, 5.5 + 0.5)              # right neighbours are outside: weight .75 of zero
    assert np.allclose(got, [63.0 * 0.25, 5.5 * 0.25, 0.0, 0.25], atol=1e-5)


def test_reprojection_static_camera_properties():
    from reproj_util import oracle_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    o = po.Oracle.from_scene(s, sky, accel=True)
    W, H, N = 48, 27, 4
    seq = oracle_sequence(o, s, W, H, 3, spp=1, max_bounces=4, frames_n=N, step=(0, 0, 0))
    plain = []
    for k in range(3):
        p = pa.make_params(s, W, H, seed=pa.frame_seed(k), spp=1, max_bounces=4)
        plain.append(o.render(p, W, H)[0])
    r0, h0, ids, _ = seq[0]
    assert (r0 == plain[0]).all()                      # no history yet: the fresh frame (SH:981)
    assert (h0 == 0).all()                             # Result1 = the (empty) previous Result
    r1, h1, _, _ = seq[1]
    assert (h1 == r0).all()                            # SH:730
    sky_px = ids < 0
    assert (r1[sky_px] == plain[1][sky_px]).all()      # !skyboxS fails: fresh value
    hit = ~sky_px
    # same camera: the reprojected distance is the pixel's own distance, so the alpha channel is unchanged ...
    assert np.allclose(r1[hit][:, 3], plain[1][hit][:, 3], rtol=1e-6)
    # ... and wherever the history was taken, colour = (history * (N-1) + fresh) / N with history from frame 0
    taken = hit & (np.abs(r1[..., :3] - plain[1][..., :3]).max(-1) > 0)
    assert taken.sum() > 0.5 * hit.sum()
    lo = np.minimum(plain[1][..., :3] / N, plain[1][..., :3] / N + r0[..., :3].max() * (N - 1) / N)
    assert (r1[..., :3][taken] >= lo[taken] - 1e-6).all()


def test_reprojection_golden_sequence():
    from reproj_util import oracle_sequence
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "reproj_64x36.npz"))
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    o = po.Oracle.from_scene(s, sky)
    seq = oracle_sequence(o, s, 64, 36, 3, spp=2, max_bounces=4, frames_n=5)
    for k, (res, hist, ids, st) in enumerate(seq):
        assert (res.view(np.uint32) == g[f"result{k}"].view(np.uint32)).all(), k
    assert int(g["taken2"]) == int((np.abs(seq[2][0][..., 3] - g["plain_alpha2"]) > 0).sum())


# ------------------------------------------- renderMode 0 (denoiser + TAA)

def test_denoiser_kat():
    """denoiser() (SH:479-615): on a constant image every tap has weight 1, so an interior pixel keeps its
    value; taps outside the image read zero and pull border pixels down; the output is clamped to [0,1] with
    alpha 0; the tap stride is 1 + 3*hash1(641.128752 x + 312.321374 y + 1.92357812 _Seed)."""
    W, H = 40, 30
    img = np.zeros((H, W, 4), np.float32); img[..., :3] = (0.25, 0.5, 0.75); img[..., 3] = 7.0
    out = po.denoiser(img, 20, 15, 3.5)
    assert np.allclose(out, [0.25, 0.5, 0.75, 0.0], atol=1e-6)
    edge = po.denoiser(img, 0, 0, 3.5)
    assert (edge[:3] < np.array([0.25, 0.5, 0.75])).all() and (edge[:3] > 0).all()
    img[..., :3] = 2.0
    assert (po.denoiser(img, 20, 15, 3.5) == np.array([1, 1, 1, 0], np.float32)).all()       # clamp (SH:614)
    # a single bright texel next to a dark pixel is suppressed by the colour weight exp(-|dc|^2)
    img[..., :3] = 0.0; img[15, 21, :3] = 3.0
    strength = 1.0 + 3.0 * po.fmath("frac", po.fmath("sin", np.float32(641.128752) * np.float32(20) + np.float32(312.321374) * np.float32(15)
                                                     + np.float32(1.92357812) * np.float32(3.5)) * np.float32(43758.5453123))
    got = po.denoiser(img, 20, 15, 3.5)
    if strength < 2.0:      # tap (1, 0) * strength lands on texel 21 only while strength < 2
        w = np.exp(-27.0) * (1 / 4 * 3 / 8)
        assert np.allclose(got[:3], 3.0 * w / (1.0 - 3 / 32 + w), rtol=1e-3, atol=1e-12)
    else:
        assert (got[:3] == 0).all()


def test_taa_kat():
    """TAA() (SH:644-714) on constant images: history h, current c -> the clamp to the neighbourhood's range
    returns c, and the mix rate is clamp(1/3 + 4 |yuv(c) - yuv(sqrt((h^2+c^2)/2))|^2, 0.05, 0.5)."""
    W, H = 24, 16
    hist = np.zeros((H, W, 4), np.float32); cur = np.zeros((H, W, 4), np.float32)
    hist[..., :3] = 0.4; cur[..., :3] = 0.4
    out = po.taa(hist, cur, 10, 8)
    assert np.allclose(out[:3], 0.4, atol=2e-6) and np.isclose(out[3], 1.0 / 3.0, rtol=1e-6)
    cur[..., :3] = (0.1, 0.5, 0.9)
    out = po.taa(hist, cur, 10, 8)
    assert np.allclose(out[:3], (0.1, 0.5, 0.9), atol=1e-4)      # the two YUV matrices are 5-digit inverses of each other
    assert 1.0 / 3.0 < out[3] <= 0.5
    # the "+off" neighbours are the pixel itself, the "-off" ones its left / upper neighbours (index truncation):
    # a bright texel to the right or below does not widen the clamp range, one to the left does
    hist[..., :3] = 0.9; cur[..., :3] = 0.2
    base = po.taa(hist, cur, 10, 8)
    cur2 = cur.copy(); cur2[8, 11, :3] = 1.0; cur2[9, 10, :3] = 1.0
    assert (po.taa(hist, cur2, 10, 8) == base).all()
    cur3 = cur.copy(); cur3[8, 9, :3] = 1.0
    assert (po.taa(hist, cur3, 10, 8)[:3] > base[:3]).all()


def test_denoise_mode_sequence_properties_and_golden():
    from reproj_util import oracle_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    o = po.Oracle.from_scene(s, sky)
    W, H = 64, 36
    seq = oracle_sequence(o, s, W, H, 3, spp=2, max_bounces=4, frames_n=5, mode=0)
    ref2 = oracle_sequence(o, s, W, H, 3, spp=2, max_bounces=4, frames_n=5, mode=2)
    r0, h0, _, _, acc0 = seq[0]
    assert (h0 == 0).all()                                      # Result1 = Result2 of the (empty) previous frame, SH:735
    assert (acc0 == ref2[0][0]).all()                           # no history: Result2 = the fresh frame, as renderMode 2 writes Result
    assert (r0[..., :3] == 0).all() and np.allclose(r0[..., 3], 1.0 / 3.0)   # denoiser + TAA of an all-zero history
    r1, h1, _, _, acc1 = seq[1]
    assert (h1 == acc0).all()
    assert r1[..., :3].max() < 1.5 and r1[..., :3].min() >= 0.0 and r1[..., :3].mean() > 0.05   # the YUV round trip is not clamped
    # Result2 follows renderMode 2's rule except below a thread group's last row (SH:908-915)
    same = (acc1 == ref2[1][0]).all(-1)
    assert 0.9 < same.mean() <= 1.0
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "denoise_64x36.npz"))
    for k in range(3):
        assert (seq[k][0].view(np.uint32) == g[f"result{k}"].view(np.uint32)).all(), k
        assert (seq[k][4].view(np.uint32) == g[f"result2_{k}"].view(np.uint32)).all(), k


def test_exp_accuracy():
    xs = np.concatenate([np.linspace(-80, 5, 4001), [-0.0, 0.0, -1e-8]]).astype(np.float32)
    got = np.array([po.fmath("exp", float(x)) for x in xs], np.float64)
    want = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - want) / want) < 2e-5      # |x| * 2^-24 from rounding x*log2(e), plus the exp2 kernel
    assert po.fmath("exp", -0.0) == 1.0


# ------------------------------------------- renderMode 5 / 6 (partial frames)

@pytest.mark.parametrize("mode", [5, 6])
def test_stripe_modes_properties(mode):
    """One stripe of H/divisions rows is path-traced per frame exactly as renderMode 1 renders it and copied into
    Result1; with a static camera every other pixel takes back (a bilinear tap of) what its own stripe rendered."""
    from reproj_util import oracle_stripe_sequence, stripe_frames
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    o = po.Oracle.from_scene(s, sky, accel=True)
    W, H, D = 48, 40, 4
    seq = oracle_stripe_sequence(o, s, W, H, 2 * D, D, mode, step=(0, 0, 0))
    frames = stripe_frames(s, W, H, 2 * D, D, mode, step=(0, 0, 0))
    stripe = H / D
    for k, (res, res1, _, st) in enumerate(seq):
        c = k % D
        rows = [y for y in range(H) if not (y > H - stripe * c or y < H - stripe * (c + 1))]
        assert rows == list(range(int(H - stripe * (c + 1)), min(H - 1, int(H - stripe * c)) + 1))   # both ends inclusive (SH:751)
        p1 = pa.PtFrameParams.from_buffer_copy(frames[k][0]); p1.render_mode = 1
        plain = o.render(p1, W, H)[0]
        assert (res[rows] == plain[rows]).all()                    # SH:933-936
        assert (res1[rows] == res[rows]).all()                     # SH:1108-1112
        others = [y for y in range(H) if y not in rows]
        if k == 0:
            # nothing rendered yet and no cameras uploaded: off-screen value (SH:780) - or the sky in the depth variant
            hit = (res[others][..., 3] == 1.0)
            assert hit.all()
            if mode == 5:
                assert (res[others][..., :3] == 0).all()
        paths = st["paths"]
        assert paths == len(rows) * W + (len(others) * W if mode == 6 else 0)
    # after a full cycle every stripe has been rendered once; with a static camera the reprojected pixels repeat what
    # Result1 holds around their own position, so the displayed frame is close to a plain render of the scene
    last = seq[-1][0]
    p1 = pa.PtFrameParams.from_buffer_copy(frames[-1][0]); p1.render_mode = 1
    plain = o.render(p1, W, H)[0]
    ids = o.render(p1, W, H)[1]
    interior = np.zeros((H, W), bool); interior[2:-2, 2:-2] = True
    close = np.abs(last[..., 3] - plain[..., 3]) < 0.05 * np.maximum(plain[..., 3], 1e-3)
    assert close[interior & (ids >= 0)].mean() > 0.8


# ------------------------------------------- the remaining renderModes: 3, 4, 10, 11, 12, 7, 8

def test_blur_and_new_render_modes_static_camera_properties():
    """Static camera, so a hit pixel's history position is (about) the pixel itself.
    renderMode 3 (SH:986-1019): colour = fresh/2.5 + bilinear/5 + four diagonal history texels/10, alpha = the new
    distance.  renderMode 10 (SH:1048-1069): covered pixels copy the bilinear history sample - alpha too - and trace
    no path; sky pixels and uncovered pixels are rendered as renderMode 1 renders them."""
    from reproj_util import oracle_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    o = po.Oracle.from_scene(s, sky, accel=True)
    W, H = 48, 27
    plain = [o.render(pa.make_params(s, W, H, seed=pa.frame_seed(k), spp=1, max_bounces=4), W, H) for k in range(2)]
    ids = plain[1][1]; hit = ids >= 0
    seq3 = oracle_sequence(o, s, W, H, 2, spp=1, max_bounces=4, mode=3, step=(0, 0, 0))
    assert (seq3[0][0] == plain[0][0]).all() and (seq3[1][1] == seq3[0][0]).all()       # no history yet; Result1 = Result
    r = seq3[1][0]
    assert (r[~hit] == plain[1][0][~hit]).all()
    assert (r[..., 3] == plain[1][0][..., 3]).all()                                     # SH:1005: alpha is always the new distance
    blended = hit & (np.abs(r[..., :3] - plain[1][0][..., :3]).max(-1) > 0)
    assert blended.sum() > 0.5 * hit.sum()
    # the weights sum to 1: the result stays within the range spanned by the fresh value and the history image
    hi = np.maximum(plain[1][0][..., :3], seq3[0][0][..., :3].max()) + 1e-6
    assert (r[..., :3][blended] <= np.broadcast_to(hi, r[..., :3].shape)[blended]).all()

    seq10 = oracle_sequence(o, s, W, H, 2, spp=1, max_bounces=4, mode=10, step=(0, 0, 0))
    assert (seq10[0][0] == plain[0][0]).all()                                           # nothing to copy from: a plain render
    r, r1, _, st = seq10[1]
    assert (r1 == seq10[0][0]).all()
    assert (r[~hit] == plain[1][0][~hit]).all()                                         # !skyboxS fails: traced
    covered = hit & (r != plain[1][0]).any(-1)
    assert covered.sum() > 0.5 * hit.sum()
    assert st["rays"] < plain[1][2]["rays"] and st["rays"] >= W * H                     # one primary ray each, paths only where uncovered
    traced = hit & ~covered
    assert (r[traced] == plain[1][0][traced]).all()


def test_stereo_reproj_mode_offsets():
    """renderMode 4 (SH:1021-1046) finds the history position with FindCameraRayVR (SH:418-437): its per-eye offsets
    (156 / 538 pixels) put every position off screen in a small image, which leaves the plain render; in a tall image
    the two eyes blend different pixels."""
    from reproj_util import oracle_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    o = po.Oracle.from_scene(s, sky, accel=True)
    W, H = 48, 27
    seq = oracle_sequence(o, s, W, H, 2, spp=1, max_bounces=4, mode=4, step=(0, 0, 0), left_eye=1)
    plain = o.render(pa.make_params(s, W, H, seed=pa.frame_seed(1), spp=1, max_bounces=4), W, H)[0]
    assert (seq[1][0] == plain).all()
    W, H = 160, 400
    out = {}
    for eye in (0, 1):
        seq = oracle_sequence(o, s, W, H, 2, spp=1, max_bounces=4, mode=4, step=(0, 0, 0), left_eye=eye, threads=8)
        plain = o.render(pa.make_params(s, W, H, seed=pa.frame_seed(1), spp=1, max_bounces=4), W, H, threads=8)[0]
        out[eye] = (seq[1][0] != plain).any(-1)
    assert out[1].sum() > 100 and not (out[0] == out[1]).all()


@pytest.mark.parametrize("mode", [11, 12])
def test_pause_modes_properties(mode):
    """renderMode 11 / 12 (SH:1071-1104): nothing is path-traced, Result1 is not written, and with the camera where the
    history camera was a pixel shows (a bilinear tap of) Result1 around its own position - or (0,0,0,1) where the depth
    variant's primary ray misses."""
    from reproj_util import oracle_pause_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    o = po.Oracle.from_scene(s, sky, accel=True)
    W, H = 48, 27
    seq = oracle_pause_sequence(o, s, W, H, 2, 2, mode, step=(0, 0, 0))
    frozen = seq[1][1]                                   # Result1 after the last rendered frame
    for k in (2, 3):
        res, res1, ids, st = seq[k]
        assert (res1 == frozen).all()
        assert st["rays"] == (0 if mode == 11 else W * H)
        assert (res == seq[2][0]).all()                  # no randomness, no state: every paused frame is the same
    res, ids = seq[2][0], seq[2][2]
    if mode == 12:
        assert (res[ids < 0] == np.array([0, 0, 0, 1], np.float32)).all()
        on = ids >= 0
    else:
        on = np.ones((H, W), bool)
    interior = np.zeros((H, W), bool); interior[2:-2, 2:-2] = True
    lo = np.minimum.reduce([np.roll(frozen, (dy, dx), (0, 1)) for dy in (0, 1, 2) for dx in (0, 1, 2)])
    hi = np.maximum.reduce([np.roll(frozen, (dy, dx), (0, 1)) for dy in (0, 1, 2) for dx in (0, 1, 2)])
    sel = interior & on
    inside = ((res >= lo - 1e-5) & (res <= hi + 1e-5)).all(-1)
    assert inside[sel].mean() > 0.95                     # a bilinear tap lies between the texels it reads


@pytest.mark.parametrize("mode", [7, 8])
def test_full_frame_reproj_modes_properties(mode):
    """renderMode 7 / 8 (SH:751-789, 850-853, 938-948) with a static camera and 4 stripes: the dispatch with counter 0
    renders the whole frame (from "history camera -1", the current one) into Result2 and copies it to Result1 and
    Result; the following dispatches re-render one stripe each into Result2, show Result1 reprojected in Result and
    leave Result1 alone."""
    from reproj_util import oracle_stripe_sequence, stripe_frames
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    o = po.Oracle.from_scene(s, sky, accel=True)
    W, H, D = 48, 40, 4
    seq = oracle_stripe_sequence(o, s, W, H, D + 1, D, mode, step=(0, 0, 0))
    frames = stripe_frames(s, W, H, D + 1, D, mode, step=(0, 0, 0))
    stripe = H / D
    p1 = pa.PtFrameParams.from_buffer_copy(frames[0][0]); p1.render_mode = 1
    plain0, ids0, st0 = o.render(p1, W, H)
    sky_px = ids0 < 0
    res, res1, _, st, res2 = seq[0]
    keep = ~sky_px if mode == 8 else np.ones((H, W), bool)           # mode 8: a primary miss returns at SH:760-763
    assert (res2[keep] == plain0[keep]).all() and (res1[keep] == plain0[keep]).all() and (res[keep] == plain0[keep]).all()
    if mode == 8:
        assert (res2[sky_px] == 0).all() and (res[sky_px][:, 3] == 1.0).all() and (res[sky_px][:, :3] > 0).any()
    for k in range(1, D):
        res, r1k, _, st, r2k = seq[k]
        assert (r1k == res1).all()                                   # Result1 only changes when counter is 0
        rows = [y for y in range(H) if not (y > H - stripe * k or y < H - stripe * (k + 1))]
        others = [y for y in range(H) if y not in rows]
        assert (r2k[others] == seq[k - 1][4][others]).all()          # Result2: only the stripe is re-rendered
        pk = pa.PtFrameParams.from_buffer_copy(frames[k][0]); pk.render_mode = 1
        plain_k = o.render(pk, W, H)[0]
        sel = np.zeros((H, W), bool); sel[rows] = True; sel &= keep
        assert (r2k[sel] == plain_k[sel]).all()                      # same camera: the stripe is what renderMode 1 renders
        n_stripe = len(rows) * W - (int((sky_px[rows]).sum()) if mode == 8 else 0)
        assert st["paths"] == n_stripe + (W * H if mode == 8 else 0)
    # the first dispatch of the next cycle publishes the assembled frame
    res, r1n, _, _, r2n = seq[D]
    assert (r1n[keep] == r2n[keep]).all() and (res[keep] == r2n[keep]).all()


def test_camera_ray_old_matches_camera_ray_for_a_perspective_camera():
    """CreateCameraRayOld (SH:374-387) feeds clip z = 1 where CreateCameraRay (SH:108-120) feeds 0; the inverse of a
    perspective projection has an empty third column in its xyz rows, so renderMode 7's first dispatch - history
    element -1, the current camera - renders exactly what renderMode 1 renders (checked above); here the matrix itself."""
    s = pa.Scene("fixture")
    p = pa.make_params(s, 64, 36)
    ip = np.array(p.camera_inverse_projection[:], np.float32).reshape(4, 4).T      # [row, col]
    assert (ip[:3, 2] == 0).all() and ip[3, 2] != 0


def test_other_modes_golden():
    from reproj_util import oracle_pause_sequence, oracle_sequence, oracle_stripe_sequence
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "othermodes_64x36.npz"))
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    o = po.Oracle.from_scene(s, sky)
    W, H = 64, 36
    for mode, spp in ((3, 2), (10, 1)):
        got = oracle_sequence(o, s, W, H, 3, spp=spp, max_bounces=4, frames_n=5, mode=mode)[2][0]
        assert (got.view(np.uint32) == g[f"mode{mode}_result2"].view(np.uint32)).all(), mode
    for mode in (11, 12):
        got = oracle_pause_sequence(o, s, W, H, 2, 3, mode)[4][0]
        assert (got.view(np.uint32) == g[f"mode{mode}_result4"].view(np.uint32)).all(), mode
    for mode in (7, 8):
        seq = oracle_stripe_sequence(o, s, W, H, 10, 4, mode, spp=1, max_bounces=4)
        assert (seq[9][0].view(np.uint32) == g[f"mode{mode}_result9"].view(np.uint32)).all(), mode
        assert (seq[9][4].view(np.uint32) == g[f"mode{mode}_result2_9"].view(np.uint32)).all(), mode


# ------------------------------------------- per-mesh materials and the GGX lobe (extensions)

def _one_quad_scene(material):
    """A large upward-facing quad at y = 1 (two triangles, one object) under a constant sky."""
    v = np.array([[-50, 1, -50], [50, 1, -50], [50, 1, 50], [-50, 1, 50]], np.float32)
    idx = np.array([0, 2, 1, 0, 3, 2], np.int32)            # counter-clockwise seen from above
    objs = np.zeros(1, dtype=np.dtype([("m", np.float32, 16), ("offset", np.int32), ("count", np.int32)]))
    objs[0] = (np.eye(4, dtype=np.float32).reshape(16), 0, 6)
    return (np.zeros((0, 14), np.float32), objs, v, idx), pa.materials([material])


def test_hard_coded_material_equals_its_explicit_form():
    """Supplying the shader's own constants (SH:250-253) as a PtMaterial changes nothing."""
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    W, H = 32, 18
    p = pa.make_params(s, W, H, spp=2, max_bounces=4)
    a = po.Oracle.from_scene(s, sky).render(p, W, H)[0]
    m = pa.materials([((0, 0, 0), 0.99, (0.65, 0.65, 0.65), pa.PT_LOBE_PHONG)] * len(s.objects))
    b = po.Oracle.from_scene(s, sky, materials=m).render(p, W, H)[0]
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_material_emission_and_albedo_reach_the_image():
    sky = np.zeros((4, 8, 4), np.float32)
    scene, m = _one_quad_scene(((0, 0, 0), 0.5, (0, 0, 0), pa.PT_LOBE_PHONG, (0.5, 1.0, 2.0)))
    s = pa.Scene("fixture")
    p = pa.make_params(s, 16, 8, spp=1, max_bounces=2, eye=(0, 5, 0), target=(0, 0, 0.01))
    img, ids, _ = po.Oracle(*scene, sky=sky, materials=m).render(p, 16, 8)
    assert (ids == 1).all() or (ids >= 1).all()
    assert np.allclose(img[..., :3], (0.5, 1.0, 2.0))             # black, non-reflecting emitter: the emission alone


def test_ggx_lobe_properties():
    """White furnace: under a constant sky of radiance 1 a GGX surface never returns more than it receives
    (the estimator F*G*(v.h)/((n.v)(n.h)) <= 1/specChance per sample has mean <= 1), a smooth one
    returns nearly everything near normal incidence, and a rough one visibly less (no multiple scattering)."""
    sky = np.full((4, 8, 4), 0.5, np.float32)                     # SkyboxSample doubles it (SH:331)
    s = pa.Scene("fixture")
    W, H = 24, 12
    p = pa.make_params(s, W, H, spp=64, max_bounces=2, eye=(0, 6, 0), target=(0, 0, 0.01), fov=40)
    means = {}
    for smooth in (0.999, 0.7, 0.2):
        scene, m = _one_quad_scene(((0, 0, 0), smooth, (1, 1, 1), pa.PT_LOBE_GGX))
        img, ids, _ = po.Oracle(*scene, sky=sky, materials=m).render(p, W, H, threads=4)
        assert (ids >= 1).all()
        assert np.isfinite(img).all() and (img[..., :3] >= 0).all()
        means[smooth] = float(img[..., :3].mean())
    assert 0.97 < means[0.999] <= 1.0 + 1e-3
    assert means[0.2] < means[0.7] <= means[0.999] + 1e-3
    assert means[0.2] > 0.3
    # the Phong lobe of the shader on the same surface, for scale
    scene, m = _one_quad_scene(((0, 0, 0), 0.999, (1, 1, 1), pa.PT_LOBE_PHONG))
    ph = po.Oracle(*scene, sky=sky, materials=m).render(p, W, H, threads=4)[0]
    assert abs(float(ph[..., :3].mean()) - means[0.999]) < 0.15   # the Phong lobe loses what it samples below the surface


# ------------------------------------------- the display blit (Denoiser.shader)

def test_display_denoiser_kat():
    """Denoiser.shader: a constant image stays constant everywhere (clamp addressing: no dark border, unlike the
    compute kernel's filter), values above 1 are clamped, alpha is 0, and an isolated outlier is suppressed by
    the two colour weights exp(-d2) * exp(-d2 / 0.5)."""
    W, H = 40, 30
    img = np.zeros((H, W, 4), np.float32); img[..., :3] = (0.25, 0.5, 0.75); img[..., 3] = 3.0
    out = po.display_denoise(img)
    assert np.allclose(out[..., :3], (0.25, 0.5, 0.75), atol=1e-6) and (out[..., 3] == 0).all()
    img[..., :3] = 1.5
    assert (po.display_denoise(img)[..., :3] == 1.0).all()
    img[..., :3] = 0.1; img[15, 20, :3] = 2.0
    out = po.display_denoise(img)
    assert out[15, 21, 0] < 0.1 + 1e-3 and out[15, 19, 0] < 0.1 + 1e-3          # neighbours ignore the outlier
    assert out[15, 20, 0] > 0.9                                                 # the outlier's own centre tap dominates (and is clamped)
