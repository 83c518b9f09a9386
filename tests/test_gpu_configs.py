"""The benchmark configurations at their full sizes, each against the oracle: the sphere field (whole frame, exhaustive
loop), the 70 k-triangle mesh (a tile of the 1080p frame, 16 samples), the 260 k-triangle scene with GGX surfaces (a
tile, 16 samples), the 10 M-triangle scene as a 4K stereo pair with its depth output fed to a reprojection frame."""
import numpy as np
import pytest

import pathtracing_amd as pa
from oracle import pt_oracle as po

pytestmark = pytest.mark.gpu


def _same(a, b):
    return (np.ascontiguousarray(a).view(np.uint32) == np.ascontiguousarray(b).view(np.uint32)).all()


def test_sphere_field_full_frame_against_the_exhaustive_loop(tracer):
    """Configuration 1: the sphere field of RayTracingMaster.cs:32-34 (SpheresMax 100, radii 3..8, placement radius 100)
    on the ground plane under the sky, 1920x1080, 8 bounces, 1 sample: every pixel and every primary hit id against the
    oracle's loop over all spheres (RayTracingShader.compute:262-285), and the ray counts."""
    s = pa.Scene("spheres"); sky = pa.procedural_sky(512, 256)
    W, H = 1920, 1080
    tracer.load_scene(s, sky); tracer.set_mesh_materials(None); tracer.resize(W, H)
    p = pa.make_params(s, W, H, seed=0.5, spp=1, max_bounces=8)
    tracer.stats(); tracer.render(p)
    g = tracer.read_result(); gid = tracer.read_hit_ids(); st = tracer.stats()
    c, cid, cst = po.Oracle.from_scene(s, sky, accel=False).render(p, W, H, threads=16)
    assert (gid == cid).all()
    assert _same(g, c)
    assert st["rays"] == cst["rays"] and st["paths"] == W * H
    assert (gid > 0).mean() > 0.2 and (gid == 0).mean() > 0.2 and (gid < 0).mean() > 0.2    # spheres, plane and sky all in view


def test_mesh_70k_tile_at_full_size(tracer):
    """Configuration 2 as stated: the ~70 k-triangle mesh through the tree, 1920x1080, 8 bounces, 16 samples in one
    dispatch - the settings bench.py quotes a rate for.  A 160x90 tile of that frame with the mesh in it against the
    oracle (its own tree), hit ids and bits; the whole frame rendered at once must show the same tile."""
    s = pa.Scene("blob70k"); sky = pa.procedural_sky(512, 256)
    W, H = 1920, 1080
    tracer.load_scene(s, sky); tracer.set_mesh_materials(None); tracer.resize(W, H)
    whole = pa.make_params(s, W, H, seed=pa.frame_seed(2), spp=16, max_bounces=8)
    tracer.render(whole)
    full = tracer.read_result(); full_ids = tracer.read_hit_ids()
    # the tile with the most mesh pixels among a coarse grid of candidates
    best, tile = -1, None
    for y0 in range(0, H - 90 + 1, 90):
        for x0 in range(0, W - 160 + 1, 160):
            n = int((full_ids[y0:y0 + 90, x0:x0 + 160] > len(s.spheres)).sum())
            if n > best:
                best, tile = n, (x0, y0, 160, 90)
    ys, xs = slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2])
    assert best > 0.3 * 160 * 90
    p = pa.make_params(s, W, H, seed=pa.frame_seed(2), spp=16, max_bounces=8, tile=tile)
    c, cid, cst = po.Oracle.from_scene(s, sky, accel=True).render(p, W, H, threads=16)
    assert (full_ids[ys, xs] == cid[ys, xs]).all()
    assert _same(full[ys, xs], c[ys, xs])
    tracer.resize(W, H); tracer.stats(); tracer.render(p)            # and dispatched as a tile: same pixels, the oracle's ray count
    assert _same(tracer.read_result()[ys, xs], c[ys, xs]) and tracer.stats()["rays"] == cst["rays"]


def test_ggx_hall_tile_at_full_size(tracer):
    """Configuration 4's scene and settings on one GPU: 262 k triangles with a GGX surface on every mesh, 1920x1080,
    8 bounces, 16 samples in one dispatch - a 160x90 tile against the oracle (its own tree), bit for bit."""
    s = pa.Scene("hall260k"); sky = pa.procedural_sky(512, 256)
    W, H = 1920, 1080
    mats = pa.materials([((0.05, 0.03, 0.02), 0.85, (0.95, 0.64, 0.54), pa.PT_LOBE_GGX)] * len(s.objects))
    tracer.load_scene(s, sky); tracer.set_mesh_materials(mats); tracer.resize(W, H)
    tile = (W // 2 - 80, H // 2 - 45, 160, 90)
    p = pa.make_params(s, W, H, seed=pa.frame_seed(3), spp=16, max_bounces=8, tile=tile)
    tracer.render(p)
    g = tracer.read_result(); gid = tracer.read_hit_ids()
    orc = po.Oracle(s.spheres, s.objects, s.vertices, s.indices, sky=sky, accel=True, materials=mats)
    c, cid, _ = orc.render(p, W, H, threads=16)
    ys, xs = slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2])
    assert (gid[ys, xs] == cid[ys, xs]).all()
    assert _same(g[ys, xs], c[ys, xs])
    assert (gid[ys, xs] > len(s.spheres)).mean() > 0.3
    tracer.set_mesh_materials(None)


def test_ten_million_triangles_4k_stereo_with_depth_for_reprojection():
    """Configuration 5: ~10 M triangles, two eyes of 3840x2160, 8 bounces.  Per eye: a 128x72 tile of the path-traced
    frame and of the _Depth dispatch against the oracle, bit for bit; then the frame's alpha channel - the depth buffer
    the shader keeps for reprojection (RayTracingShader.compute:935) - drives one renderMode-2 frame from a moved
    camera, whose tile (history lookups included) equals the oracle's again."""
    s = pa.Scene("city10m"); sky = pa.procedural_sky(256, 128)
    W, H = 3840, 2160
    orc = po.Oracle.from_scene(s, sky, accel=True)
    tile = (1640, 500, 128, 72)                                  # buildings, spheres and ground in it
    ys, xs = slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2])
    sep = 0.064
    with pa.PathTracer(0) as t:
        t.load_scene(s, sky); t.resize(W, H)
        for e in range(2):
            off = np.array([(-0.5 + e) * sep, 0, 0], np.float32)
            eye, target = s.eye + off, s.target + off
            kw = dict(spp=1, max_bounces=8, eye=eye, target=target)
            # frame 0: plain path tracing (renderMode 1), whole eye
            p0 = pa.make_params(s, W, H, seed=pa.frame_seed(0), **kw)
            t.stats(); t.render(p0)
            f0 = t.read_result(); ids = t.read_hit_ids(); st = t.stats()
            assert st["paths"] == W * H and np.isfinite(f0).all()
            res = np.zeros((H, W, 4), np.float32); res1 = np.zeros_like(res)
            cid, _ = orc.render_history(pa.make_params(s, W, H, seed=pa.frame_seed(0), tile=tile, **kw), [], W, H, res, res1, threads=16)
            assert (ids[ys, xs] == cid[ys, xs]).all() and _same(f0[ys, xs], res[ys, xs])
            assert (ids[ys, xs] > len(s.spheres)).mean() > 0.3
            # the _Depth dispatch of the same camera agrees with the alpha channel and with the oracle
            t.render(pa.make_params(s, W, H, depth=1, eye=eye, target=target))
            dep = t.read_result()
            hit = ids >= 0
            assert np.allclose(dep[hit][:, 0] * 16.0, f0[hit][:, 3], rtol=2e-6) and np.isinf(dep[~hit][:, 0]).all()
            dres = np.zeros((H, W, 4), np.float32)
            orc.render_history(pa.make_params(s, W, H, depth=1, eye=eye, target=target, tile=tile), [], W, H, dres, np.zeros_like(dres), threads=16)
            assert _same(dep[ys, xs], dres[ys, xs])
            # frame 1: renderMode 2 from a camera that has moved; the history is frame 0 with its depth in alpha.  The
            # oracle gets the GPU's frame 0 as Result and Result1 (it rendered only the tile itself): what it reads
            # outside the tile - the reprojected positions - is then the same data the kernel reads.
            t.write_target(pa.PT_TARGET_RESULT, f0)
            eye1 = eye + np.array([0.3, 0.1, 0.5], np.float32)
            hist = [pa.camera_history_entry(tuple(eye), tuple(target), s.fov, W / H)]
            t.set_camera_history(hist)
            p1 = pa.make_params(s, W, H, seed=pa.frame_seed(1), render_mode=2, accumulation_frames=4, eye=eye1, target=target, spp=1, max_bounces=8)
            t.render(p1)
            f1 = t.read_result(); h1 = t.read_converged()
            assert _same(h1, f0)                                            # Result1 = Result (SH:728-731)
            res = f0.copy(); res1 = f0.copy()
            orc.render_history(pa.make_params(s, W, H, seed=pa.frame_seed(1), render_mode=2, accumulation_frames=4, eye=eye1, target=target,
                                              spp=1, max_bounces=8, tile=tile), hist, W, H, res, res1, threads=16)
            assert _same(f1[ys, xs], res[ys, xs])
            blended = ~_bitsame(f1[ys, xs], _plain_tile(orc, s, W, H, tile, eye1, target))
            assert blended.mean() > 0.3                                     # the history was actually used
            if e == 0:
                # The frame above came from the default build: all ten million world-space triangles built flat on the device
                # (quicker than the host builds the one shared mesh of the 38 placements).  The same eye through the copies of
                # that mesh made and refitted on the device (instancing 2 with the host builder), through the two-level tree,
                # through the tree the host builds over all triangles and through the device's clustering builder: the whole
                # 4K frame, its hit ids and the _Depth frame are the same bit for bit.  The two-level tree is a fraction of the
                # size; the flat trees are of one size, the host's more than ten times as long in the making, the device's
                # within half a second - and the top-down one within a tenth of the host's node visits per ray.
                nodes_copies, build_copies = st["bvh_nodes"], st["bvh_build_ms"]
                visits = {}
                for inst, build in ((2, 0), (1, 0), (0, 0), (0, 1), (0, 3)):
                    t.set_option("instancing", inst); t.set_option("tree_build", build)
                    t.stats(); t.render(p0)
                    g0 = t.read_result(); gids = t.read_hit_ids(); st1 = t.stats()
                    assert (gids == ids).all() and _same(g0, f0) and st1["rays"] == st["rays"], (inst, build)
                    assert st1["bvh_prims"] == st["bvh_prims"]
                    if inst == 0:
                        visits[build] = st1["nodes_visited"] / st1["rays"]
                    if inst == 2:
                        assert 0.9 * nodes_copies < st1["bvh_nodes"] < 1.1 * nodes_copies and st1["bvh_build_ms"] < 1000.0
                    elif inst == 1:
                        assert st1["bvh_nodes"] < nodes_copies // 20
                    else:
                        assert 0.9 * nodes_copies < st1["bvh_nodes"] < 1.1 * nodes_copies
                        if build == 0:
                            assert st1["bvh_build_ms"] > 10 * build_copies
                        else:
                            assert st1["bvh_build_ms"] < 500.0, st1["bvh_build_ms"]
                    t.render(pa.make_params(s, W, H, depth=1, eye=eye, target=target))
                    assert _same(t.read_result(), dep), (inst, build)
                assert visits[1] < 1.1 * visits[0] and visits[3] < 1.5 * visits[0], visits
                t.set_option("tree_build", 2); t.set_option("instancing", 2)

def _bitsame(a, b):
    return (np.ascontiguousarray(a).view(np.uint32) == np.ascontiguousarray(b).view(np.uint32)).all(-1)


def _plain_tile(orc, s, W, H, tile, eye, target):
    res = np.zeros((H, W, 4), np.float32)
    orc.render_history(pa.make_params(s, W, H, seed=pa.frame_seed(1), eye=eye, target=target, spp=1, max_bounces=8, tile=tile), [], W, H, res,
                       np.zeros_like(res), threads=16)
    return res[tile[1]:tile[1] + tile[3], tile[0]:tile[0] + tile[2]]
