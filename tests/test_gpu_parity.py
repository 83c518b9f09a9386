"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
same inputs.  fp32 images are compared BIT-EXACT (tolerance 0): both sides are
written on the intrinsics of include/ptrace_fmath.h with contraction off."""
import os

import numpy as np
import pytest

import pathtracing_amd as pa
from oracle import pt_oracle as po

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _assert_same_image(gpu, cpu, what):
    neq = _bits(gpu) != _bits(cpu)
    # NaN == NaN bitwise here, so any difference is a real one
    assert not neq.any(), f"{what}: {int(neq.sum())} of {neq.size} floats differ; first at {np.argwhere(neq)[0]}: " \
                          f"gpu={gpu[tuple(np.argwhere(neq)[0])]} cpu={cpu[tuple(np.argwhere(neq)[0])]}"


_FIXTURE = {}


@pytest.fixture
def fixture_scene(tracer, sky_small):
    """The 20-sphere + 320-triangle scene, loaded into the shared context for every test that asks for it (other tests
    leave their own scenes, materials and options behind), with its exhaustive-loop oracle built once."""
    if not _FIXTURE:
        s = pa.Scene("fixture")
        _FIXTURE["scene"] = s; _FIXTURE["oracle"] = po.Oracle.from_scene(s, sky_small, accel=False)
    tracer.load_scene(_FIXTURE["scene"], sky_small)
    tracer.set_mesh_materials(None)
    tracer.set_option("reuse_primary_hit", 0)
    return _FIXTURE["scene"], _FIXTURE["oracle"]


@pytest.mark.parametrize("spp,maxb", [(1, 4), (1, 8), (4, 8), (16, 4)])
def test_fixture_image_bit_exact(tracer, fixture_scene, spp, maxb):
    s, orc = fixture_scene
    W, H = 64, 36
    tracer.resize(W, H)
    p = pa.make_params(s, W, H, seed=0.5, spp=spp, max_bounces=maxb)
    tracer.stats()   # counters run from the previous pt_get_stats: start a fresh epoch
    tracer.render(p)
    g = tracer.read_result(); gid = tracer.read_hit_ids(); st = tracer.stats()
    c, cid, cst = orc.render(p, W, H, threads=4)
    assert (gid == cid).all(), f"{int((gid != cid).sum())} primary hit ids differ"
    _assert_same_image(g, c, f"spp={spp} maxb={maxb}")
    assert st["rays"] == cst["rays"] and st["paths"] == cst["paths"]
    assert np.isfinite(g).all() and g[..., :3].max() > 0


def test_depth_mode(tracer, fixture_scene):
    s, orc = fixture_scene
    W, H = 64, 36
    tracer.resize(W, H)
    p = pa.make_params(s, W, H, depth=1)
    tracer.render(p)
    g = tracer.read_result()
    c, cid, _ = orc.render(p, W, H)
    _assert_same_image(g, c, "depth")
    assert np.isinf(g).any() and (tracer.read_hit_ids() == cid).all()


def test_odd_size_and_tile(tracer, fixture_scene):
    s, orc = fixture_scene
    W, H = 37, 23          # not multiples of 8 (the reference's 1080 is not a multiple of 32 either)
    tracer.resize(W, H)
    p = pa.make_params(s, W, H, spp=2, max_bounces=8, seed=0.25)
    tracer.render(p)
    g = tracer.read_result()
    c, _, _ = orc.render(p, W, H)
    _assert_same_image(g, c, "odd size")
    # a tile leaves every other pixel untouched
    tracer.resize(W, H)
    pt = pa.make_params(s, W, H, spp=2, max_bounces=8, seed=0.25, tile=(5, 3, 13, 9))
    tracer.render(pt)
    gt = tracer.read_result()
    _assert_same_image(gt[3:12, 5:18], c[3:12, 5:18], "tile")
    mask = np.ones((H, W), bool); mask[3:12, 5:18] = False
    assert (gt[mask] == 0).all()


@pytest.mark.parametrize("mode", [pa.PT_ACCUM_MEAN, pa.PT_ACCUM_SUM, pa.PT_ACCUM_ADDMAT])
def test_accumulation_modes(tracer, fixture_scene, mode):
    s, orc = fixture_scene
    W, H = 48, 27
    tracer.resize(W, H)
    conv = np.zeros((H, W, 4), np.float32)
    for k in range(4):
        p = pa.make_params(s, W, H, seed=pa.frame_seed(k), spp=1, max_bounces=4, accumulate=mode, sample_index=k)
        tracer.render(p)
        fresh, _, _ = orc.render(p, W, H)
        conv = po.accumulate(mode, k, fresh, conv)
    _assert_same_image(tracer.read_converged(), conv, f"accumulate mode {mode}")


def test_empty_scene_and_no_sky(tracer):
    W, H = 32, 16
    e = np.zeros((0, 14), np.float32)
    tracer.set_spheres(e); tracer.set_meshes(np.zeros(0, np.dtype([("m", np.float32, 16), ("o", np.int32), ("c", np.int32)])),
                                             np.zeros((0, 3), np.float32), np.zeros(0, np.int32))
    tracer.set_sky(None)
    tracer.resize(W, H)
    s = pa.Scene("fixture")
    p = pa.make_params(s, W, H, spp=1, max_bounces=4)
    tracer.render(p)
    g = tracer.read_result(); ids = tracer.read_hit_ids()
    orc = po.Oracle(e, np.zeros(0, np.dtype([("m", np.float32, 16), ("o", np.int32), ("c", np.int32)])),
                    np.zeros((0, 3), np.float32), np.zeros(0, np.int32))
    c, cid, _ = orc.render(p, W, H)
    _assert_same_image(g, c, "empty scene")
    assert (ids == cid).all() and set(np.unique(ids)) <= {-1, 0}


def test_bvh_equals_exhaustive_loop_on_device(tracer, sky_small):
    """The acceleration structure returns the hit of the reference's loop over every primitive."""
    s = pa.Scene("blob70k")
    tracer.load_scene(s, sky_small)
    rng = np.random.default_rng(7)
    n = 20000
    o = rng.uniform(-25, 25, (n, 3)).astype(np.float32); o[:, 1] = rng.uniform(0.1, 20, n)
    tgt = rng.uniform(-5, 5, (n, 3)).astype(np.float32); tgt[:, 1] += 4
    d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    od = np.concatenate([o, d.astype(np.float32)], axis=1)
    ib, tb = tracer.trace_rays(od, brute_force=False)
    ie, te = tracer.trace_rays(od, brute_force=True)
    assert (ib == ie).all(), f"{int((ib != ie).sum())} of {n} hit ids differ"
    assert (_bits(tb) == _bits(te)).all()
    assert (ib > 0).mean() > 0.2
    # and the oracle's exhaustive loop agrees on a subset
    orc = po.Oracle.from_scene(s, accel=False)
    io, to = orc.trace_rays(od[:300])
    assert (io == ib[:300]).all() and (_bits(to) == _bits(tb[:300])).all()


def test_mesh_scene_against_oracle_bvh(tracer, sky_small):
    """~70 k triangles at 256x144: too big for the exhaustive oracle, so the oracle's own BVH
    (proved equal to the exhaustive loop in the CPU tests) is the checker."""
    s = pa.Scene("blob70k")
    tracer.load_scene(s, sky_small)
    W, H = 256, 144
    tracer.resize(W, H)
    p = pa.make_params(s, W, H, seed=0.5, spp=2, max_bounces=8)
    tracer.stats()   # counters run from the previous pt_get_stats: start a fresh epoch
    tracer.render(p)
    g = tracer.read_result(); gid = tracer.read_hit_ids(); st = tracer.stats()
    orc = po.Oracle.from_scene(s, sky_small, accel=True)
    c, cid, cst = orc.render(p, W, H, threads=8)
    mism = (gid != cid).mean()
    assert mism == 0, f"primary hit-id mismatch rate {mism}"
    _assert_same_image(g, c, "blob70k")
    assert st["rays"] == cst["rays"]
    assert (gid > len(s.spheres)).mean() > 0.05   # the mesh is actually in view


def test_committed_golden_vectors(tracer):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fixture_64x36.npz"))
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    tracer.load_scene(s, sky)
    W, H = 64, 36
    tracer.resize(W, H)
    for name, kw in {"spp1_b4": dict(spp=1, max_bounces=4), "spp1_b8": dict(spp=1, max_bounces=8),
                     "spp16_b8": dict(spp=16, max_bounces=8), "depth": dict(depth=1)}.items():
        tracer.stats()
        tracer.render(pa.make_params(s, W, H, seed=0.5, **kw))
        _assert_same_image(tracer.read_result(), g[name], name)
        assert (tracer.read_hit_ids() == g["ids"]).all()
        st = tracer.stats()
        assert [st["rays"], st["paths"]] == [g[name + "_rays"][0], g[name + "_rays"][2]]


def test_master_mirror_renders_the_reference_default_scene(sky_small):
    """The C++ RayTracingMaster mirror drives libptrace; the oracle renders the same frames from the
    parameters the mirror uploads.  Three frames with the AddMat blend and its sample counter."""
    import ctypes as C
    from pathtracing_amd import _ffi
    lib = _ffi.ptmaster()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    try:
        s = pa.Scene("xr")
        W, H = 96, 54
        assert lib.ptm_master_load_scene(h, s.handle, 128, 64) == 0, lib.ptm_master_error(h)
        lib.ptm_master_configure(h, W, H, 1, 4, 0, pa.PT_ACCUM_ADDMAT, 2)
        eye = (C.c_float * 3)(*s.eye); tgt = (C.c_float * 3)(*s.target)
        lib.ptm_master_set_camera(h, eye, tgt, 90.0, 0.011)
        orc = po.Oracle.from_scene(s, pa.procedural_sky(128, 64), accel=False)
        conv = np.zeros((H, W, 4), np.float32)
        ctx = lib.ptm_master_context(h)
        samples = []
        for _ in range(4):
            p = pa.PtFrameParams()
            lib.ptm_master_peek_params(h, C.byref(p))
            samples.append(p.sample_index)
            assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
            fresh, _, _ = orc.render(p, W, H, threads=4)
            conv = po.accumulate(p.accumulate, p.sample_index, fresh, conv)
            got = np.empty((H, W, 4), np.float32)
            assert _ffi.ptrace().pt_read_converged(ctx, got.ctypes.data) == 0
            _assert_same_image(got, conv, "master frame")
        assert samples == [0, 1, 2, 2]          # _currentSample saturates at sampleFrames (RayTracingMaster.cs:885-888)
    finally:
        lib.ptm_master_destroy(h)


def test_update_transforms(tracer, sky_small):
    s = pa.Scene("xr")
    tracer.load_scene(s, sky_small)
    W, H = 64, 36
    tracer.resize(W, H)
    objs = s.objects.copy()
    m = objs["m"][0].reshape(4, 4).T.copy()
    a = np.radians(25.0)
    rot = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]], np.float32)
    objs["m"][0] = (m @ rot).T.reshape(16)
    tracer.update_transforms(objs)
    p = pa.make_params(s, W, H, spp=1, max_bounces=4, eye=(0, 4.6, -2), target=(0, 3, 6), fov=60)
    tracer.render(p)
    g = tracer.read_result(); gid = tracer.read_hit_ids()
    c, cid, _ = po.Oracle(s.spheres, objs, s.vertices, s.indices, sky=sky_small).render(p, W, H, threads=4)
    assert (gid == cid).all() and (gid > len(s.spheres)).any()
    _assert_same_image(g, c, "moved cube")
    with pytest.raises(pa.PtraceError):
        tracer.update_transforms(objs[:0])


def _rot_y(deg, t=(0, 0, 0), scale=1.0):
    a = np.radians(deg)
    m = np.array([[np.cos(a) * scale, 0, np.sin(a) * scale, t[0]], [0, scale, 0, t[1]],
                  [-np.sin(a) * scale, 0, np.cos(a) * scale, t[2]], [0, 0, 0, 1]], np.float32)
    return m.T.reshape(16)      # Matrix4x4 memory order


def test_refit_follows_moving_objects(tracer, sky_small):
    """pt_update_transforms on a built scene refits the tree on the device (no host build).  Three objects that
    share one mesh move for several frames - rotation, a large translation that inflates the root, a scale
    change; every frame equals the oracle's render of the moved scene bit for bit, the tree still returns the
    hit of the exhaustive loop, and a forced full build gives the same image."""
    f = pa.Scene("fixture")
    n_idx = len(f.indices)
    objs = np.zeros(3, dtype=f.objects.dtype)
    for k in range(3):
        objs[k] = (_rot_y(0, (-3.0 + 3.0 * k, 0.0, 2.0 * k)), 0, n_idx)
    tracer.set_spheres(f.spheres); tracer.set_meshes(objs, f.vertices, f.indices); tracer.set_sky(sky_small)
    W, H = 96, 54
    tracer.resize(W, H)
    cam = dict(eye=(0, 5, -12), target=(0, 1, 2), fov=60)
    p = pa.make_params(f, W, H, spp=1, max_bounces=4, seed=0.3, **cam)
    tracer.render(p); tracer.sync()                                  # full build happens here
    rng = np.random.default_rng(5)
    for frame in range(4):
        moved = objs.copy()
        moved["m"][0] = _rot_y(20.0 * (frame + 1), (-3.0, 0.2 * frame, 0.0))
        moved["m"][1] = _rot_y(-35.0 * frame, (0.0 + 1.5 * frame, 0.0, 2.0 + 10.0 * (frame == 2)))   # frame 2: far away
        moved["m"][2] = _rot_y(10.0, (3.0, 0.0, 4.0), scale=1.0 + 0.25 * frame)
        tracer.update_transforms(moved)
        tracer.render(p)
        g = tracer.read_result(); gid = tracer.read_hit_ids()
        c, cid, _ = po.Oracle(f.spheres, moved, f.vertices, f.indices, sky=sky_small).render(p, W, H, threads=4)
        assert (gid == cid).all(), f"frame {frame}"
        _assert_same_image(g, c, f"refit frame {frame}")
        o = rng.uniform(-8, 8, (200, 3)).astype(np.float32); o[:, 1] = rng.uniform(0.5, 8, 200)
        d = rng.normal(size=(200, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
        od = np.concatenate([o, d], 1)
        ib, tb = tracer.trace_rays(od); ie, te = tracer.trace_rays(od, brute_force=True)
        assert (ib == ie).all() and (_bits(tb) == _bits(te)).all()
    assert (gid > len(f.spheres)).any()
    tracer.set_option("refit", 0)
    tracer.update_transforms(moved)                                  # now a full host build at the next render
    tracer.render(p)
    _assert_same_image(tracer.read_result(), g, "rebuild after refits")
    tracer.set_option("refit", 1)


def test_caller_owned_targets_and_stream(tracer, fixture_scene):
    import torch
    s, orc = fixture_scene
    s2 = pa.Scene("fixture")
    tracer.load_scene(s2, pa.procedural_sky(128, 64))
    W, H = 64, 36
    tracer.resize(W, H)
    res = torch.full((H, W, 4), -1.0, device="cuda"); ids = torch.full((H, W), -7, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    tracer.bind_target(pa.PT_TARGET_RESULT, res.data_ptr())
    tracer.bind_target(pa.PT_TARGET_HIT_IDS, ids.data_ptr())
    tracer.set_stream(st.cuda_stream)
    p = pa.make_params(s2, W, H, seed=0.5, spp=1, max_bounces=4)
    with torch.cuda.stream(st):
        tracer.render(p)
        doubled = res * 2          # ordered after the render on the same stream
    st.synchronize()
    c, cid, _ = po.Oracle.from_scene(s2, pa.procedural_sky(128, 64)).render(p, W, H, threads=4)
    _assert_same_image(res.cpu().numpy(), c, "caller-owned Result")
    assert (ids.cpu().numpy() == cid).all()
    assert torch.equal(doubled, res * 2)
    tracer.set_stream(None); tracer.bind_target(pa.PT_TARGET_RESULT, None); tracer.bind_target(pa.PT_TARGET_HIT_IDS, None)


def test_error_paths(tracer):
    tracer.resize(16, 16)
    s = pa.Scene("fixture")
    for bad in (dict(max_bounces=0), dict(tile=(8, 8, 16, 16)), dict(accumulate=9)):
        with pytest.raises(pa.PtraceError):
            tracer.render(pa.make_params(s, 16, 16, **bad))
    p = pa.make_params(s, 16, 16); p.render_mode = 9
    with pytest.raises(pa.PtraceError, match="renderMode"):
        tracer.render(p)
    with pytest.raises(pa.PtraceError, match="outside"):
        tracer.set_meshes(np.array([(np.eye(4, dtype=np.float32).reshape(16), 0, 3)],
                                   np.dtype([("m", np.float32, 16), ("o", np.int32), ("c", np.int32)])),
                          np.zeros((2, 3), np.float32), np.array([0, 1, 5], np.int32))
    # spp <= 0: the shader's sample loop does not run, nothing is written
    tracer.resize(16, 16)
    tracer.render(pa.make_params(s, 16, 16, spp=0))
    assert (tracer.read_result() == 0).all()


def test_full_size_frame_properties(tracer):
    """1920x1080, ~260 k triangles, 8 bounces: size-independent properties plus a bit-exact central tile."""
    s = pa.Scene("hall260k"); sky = pa.procedural_sky(512, 256)
    tracer.load_scene(s, sky)
    W, H = 1920, 1080
    tracer.resize(W, H)
    p = pa.make_params(s, W, H, seed=0.5, spp=1, max_bounces=8)
    tracer.stats()
    tracer.render(p)
    full = tracer.read_result(); ids = tracer.read_hit_ids(); st = tracer.stats()
    assert np.isfinite(full).all() and (full[..., :3] >= 0).all()
    assert st["paths"] == W * H and W * H < st["rays"] <= 8 * W * H
    assert (ids > len(s.spheres)).mean() > 0.3                            # mostly mesh
    # determinism: the same frame again is identical although lanes pull pixels dynamically
    tracer.render(p)
    _assert_same_image(tracer.read_result(), full, "re-render")
    # tile decomposition: four quadrant dispatches reproduce the full dispatch bit for bit, rays add up
    tracer.resize(W, H)
    tracer.stats()
    for tx, ty in ((0, 0), (W // 2, 0), (0, H // 2), (W // 2, H // 2)):
        tracer.render(pa.make_params(s, W, H, seed=0.5, spp=1, max_bounces=8, tile=(tx, ty, W // 2, H // 2)))
    _assert_same_image(tracer.read_result(), full, "quadrants")
    assert tracer.stats()["rays"] == st["rays"]
    # depth dispatch agrees with the alpha channel wherever the primary ray hit something
    tracer.render(pa.make_params(s, W, H, depth=1))
    d = tracer.read_result()
    hit = ids >= 0
    assert np.allclose(d[hit][:, 0] * 16.0, full[hit][:, 3], rtol=2e-6)
    # central tile against the oracle (its own BVH), bit for bit
    tile = (W // 2 - 96, H // 2 - 54, 192, 108)
    orc = po.Oracle.from_scene(s, sky, accel=True)
    c, cid, _ = orc.render(pa.make_params(s, W, H, seed=0.5, spp=1, max_bounces=8, tile=tile), W, H, threads=8)
    ys, xs = slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2])
    assert (ids[ys, xs] == cid[ys, xs]).all()
    _assert_same_image(full[ys, xs], c[ys, xs], "central tile vs oracle")


def test_ten_million_triangles(tracer):
    """~10 M triangles (38 instanced halls): the acceleration structure still returns the hit of the exhaustive
    loop, and a 4K depth dispatch agrees with the alpha channel of a path-traced one."""
    s = pa.Scene("city10m")
    tracer.load_scene(s, pa.procedural_sky(256, 128))
    rng = np.random.default_rng(11)
    n = 300
    o = rng.uniform(-250, 250, (n, 3)).astype(np.float32); o[:, 1] = rng.uniform(1, 60, n)
    t = rng.uniform(-200, 200, (n, 3)).astype(np.float32); t[:, 1] = rng.uniform(0, 10, n)
    d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    od = np.concatenate([o, d.astype(np.float32)], 1)
    ib, tb = tracer.trace_rays(od); ie, te = tracer.trace_rays(od, brute_force=True)
    assert (ib == ie).all() and (_bits(tb) == _bits(te)).all() and (ib > len(s.spheres)).mean() > 0.3
    W, H = 3840, 2160
    tracer.resize(W, H)
    tracer.stats()
    tracer.render(pa.make_params(s, W, H, seed=0.5, spp=1, max_bounces=8))
    img = tracer.read_result(); ids = tracer.read_hit_ids(); st = tracer.stats()
    assert st["paths"] == W * H and st["bvh_prims"] == s.n_triangles + len(s.spheres)
    tracer.render(pa.make_params(s, W, H, depth=1))
    dep = tracer.read_result()
    hit = ids >= 0
    assert np.isfinite(img).all() and np.allclose(dep[hit][:, 0] * 16.0, img[hit][:, 3], rtol=2e-6) and np.isinf(dep[~hit]).all()


def test_xr_stereo_frame_full_size():
    """The stereo XR case at its full size: two eyes of 2064x2208, 4 bounces, 1 spp + running-mean accumulation,
    one context per eye (both on this GPU; on two GPUs each rank owns one).  A tile of each eye is compared with
    the oracle bit for bit over two accumulated frames; whole-frame properties cover the rest."""
    s = pa.Scene("xr"); sky = pa.procedural_sky(512, 256)
    W, H = 2064, 2208
    sep = 0.022                                                # TestScene.unity:870
    tile = (W // 2 - 64, H // 2 - 40, 128, 80)
    ys, xs = slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2])
    orc = po.Oracle.from_scene(s, sky, accel=True)
    images = []
    for e in range(2):
        off = np.array([(-0.5 + e) * sep, 0, 0], np.float32)
        eye, target = s.eye + off, s.target + off
        with pa.PathTracer(0) as tr:
            tr.load_scene(s, sky); tr.resize(W, H)
            conv = np.zeros((tile[3], tile[2], 4), np.float32)
            for k in range(2):
                kw = dict(seed=pa.frame_seed(k), spp=1, max_bounces=4, eye=eye, target=target, fov=90.0)
                tr.stats()
                tr.render(pa.make_params(s, W, H, accumulate=pa.PT_ACCUM_MEAN, sample_index=k, **kw))
                st = tr.stats()
                assert st["paths"] == W * H and W * H < st["rays"] <= 4 * W * H
                fresh, _, _ = orc.render(pa.make_params(s, W, H, tile=tile, **kw), W, H, threads=8)
                conv = po.accumulate(pa.PT_ACCUM_MEAN, k, fresh[ys, xs], conv)
                _assert_same_image(tr.read_result()[ys, xs], fresh[ys, xs], f"eye {e} frame {k}")
            full = tr.read_converged()
            _assert_same_image(full[ys, xs], conv, f"eye {e} converged")
            assert np.isfinite(full).all() and (full[..., :3] >= 0).all()
            images.append(full)
    assert not (images[0] == images[1]).all()                  # the eyes see different images ...
    assert abs(float(images[0][..., :3].mean()) - float(images[1][..., :3].mean())) < 0.02   # ... of the same scene


# ---------------------------------------------------- renderMode 2 (reprojection)

def test_reprojection_sequence_matches_oracle_bit_exact(tracer):
    """Three frames with a moving camera: Result and Result1 of every frame equal the oracle's, bit for bit."""
    from reproj_util import gpu_sequence, oracle_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    W, H = 64, 36
    o = po.Oracle.from_scene(s, sky)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_sequence(o, s, W, H, 3, spp=2, max_bounces=4, frames_n=5)
    got = gpu_sequence(tracer, s, W, H, 3, spp=2, max_bounces=4, frames_n=5)
    for k in range(3):
        assert (got[k][0].view(np.uint32) == want[k][0].view(np.uint32)).all(), f"Result, frame {k}"
        assert (got[k][1].view(np.uint32) == want[k][1].view(np.uint32)).all(), f"Result1, frame {k}"
        assert (got[k][2] == want[k][2]).all()
        assert got[k][3]["rays"] == want[k][3]["rays"]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reproj_64x36.npz"))
    assert (got[2][0].view(np.uint32) == g["result2"].view(np.uint32)).all()
    taken = (np.abs(got[2][0][..., 3] - g["plain_alpha2"]) > 0).sum()
    assert taken == int(g["taken2"]) and taken > 1000       # the history really is used


def test_reprojection_mesh_scene_odd_size_and_tile(tracer):
    from reproj_util import frame_params, gpu_sequence, oracle_sequence
    s = pa.Scene("blob70k"); sky = pa.procedural_sky(256, 128)
    W, H = 131, 77
    o = po.Oracle.from_scene(s, sky, accel=True)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_sequence(o, s, W, H, 2, spp=1, max_bounces=8, frames_n=10, threads=8)
    got = gpu_sequence(tracer, s, W, H, 2, spp=1, max_bounces=8, frames_n=10)
    for k in range(2):
        assert (got[k][0].view(np.uint32) == want[k][0].view(np.uint32)).all(), f"frame {k}"
    # a third frame over a sub-rectangle only: Result1 and Result change inside it and nowhere else
    tile = (16, 8, 70, 40)
    p, h = frame_params(s, W, H, 2, 1, 8, 10, tile=tile)
    hist = [frame_params(s, W, H, 1, 1, 8, 10)[1]]
    res, res1 = want[1][0].copy(), want[1][1].copy()
    o.render_history(p, hist, W, H, res, res1, threads=8)
    tracer.set_camera_history(hist); tracer.render(p)
    assert (tracer.read_result().view(np.uint32) == res.view(np.uint32)).all()
    assert (tracer.read_converged().view(np.uint32) == res1.view(np.uint32)).all()
    x0, y0, tw, th = tile
    outside = np.ones((H, W), bool); outside[y0:y0 + th, x0:x0 + tw] = False
    assert (res[outside] == want[1][0][outside]).all() and (res1[outside] == want[1][1][outside]).all()


def test_reprojection_without_history_and_errors(tracer):
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    W, H = 40, 24
    tracer.load_scene(s, sky); tracer.resize(W, H)
    tracer.clear_camera_history()
    plain = pa.make_params(s, W, H, seed=0.25, spp=1, max_bounces=4)
    tracer.render(plain); a = tracer.read_result()
    p2 = pa.make_params(s, W, H, seed=0.25, spp=1, max_bounces=4, render_mode=2, accumulation_frames=8)
    tracer.render(p2)
    assert (tracer.read_result().view(np.uint32) == a.view(np.uint32)).all()     # zero matrices: fresh frame (SH:981)
    assert (tracer.read_converged().view(np.uint32) == a.view(np.uint32)).all()  # Result1 = previous Result (SH:730)
    bad = pa.make_params(s, W, H, render_mode=2, accumulate=pa.PT_ACCUM_MEAN)
    with pytest.raises(pa.PtraceError, match="accumulate"):
        tracer.render(bad)
    with pytest.raises(pa.PtraceError, match="renderMode"):
        tracer.render(pa.make_params(s, W, H, render_mode=9))
    # depth dispatch in renderMode 2: SH:799-809 returns before any reprojection
    d2 = pa.make_params(s, W, H, depth=1, render_mode=2)
    d1 = pa.make_params(s, W, H, depth=1)
    tracer.render(d1); want = tracer.read_result()
    tracer.render(d2)
    assert (tracer.read_result().view(np.uint32) == want.view(np.uint32)).all()


def test_master_mirror_reproj_mode_keeps_the_camera_history():
    """The C++ RayTracingMaster mirror in renderMode 2 (RayTracingMaster.cs:492-526): frame k reprojects from
    the camera of frame k-1.  The oracle renders the same frames from the parameters the mirror uploads and a
    history list kept the way the C# keeps it."""
    import ctypes as C
    from pathtracing_amd import _ffi
    lib = _ffi.ptmaster()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    try:
        s = pa.Scene("fixture")
        W, H = 64, 36
        assert lib.ptm_master_load_scene(h, s.handle, 128, 64) == 0, lib.ptm_master_error(h)
        lib.ptm_master_configure(h, W, H, 1, 4, 0, pa.PT_ACCUM_NONE, 6)
        lib.ptm_master_set_render_mode(h, 2)
        orc = po.Oracle.from_scene(s, pa.procedural_sky(128, 64), accel=False)
        ctx = lib.ptm_master_context(h)
        res = np.zeros((H, W, 4), np.float32); res1 = np.zeros((H, W, 4), np.float32)
        hist = []
        taken = 0
        for k in range(3):
            eye = (s.eye[0] + 0.1 * k, s.eye[1], s.eye[2] + 0.15 * k)
            lib.ptm_master_set_camera(h, (C.c_float * 3)(*eye), (C.c_float * 3)(*s.target), C.c_float(s.fov), C.c_float(0.0))
            p = pa.PtFrameParams()
            lib.ptm_master_peek_params(h, C.byref(p))
            assert p.render_mode == 2 and p.accumulation_frames == 6 and p.accumulate == pa.PT_ACCUM_NONE
            assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
            orc.render_history(p, hist, W, H, res, res1, threads=4)
            hist = [pa.camera_history_entry(eye, s.target, s.fov, W / H)]
            got = np.empty((H, W, 4), np.float32)
            assert _ffi.ptrace().pt_read_result(ctx, got.ctypes.data) == 0
            _assert_same_image(got, res, f"master reproj frame {k}")
            plain = pa.PtFrameParams.from_buffer_copy(p); plain.render_mode = 1
            taken += int((orc.render(plain, W, H, threads=4)[0][..., 3] != res[..., 3]).sum())
        assert taken > 500
    finally:
        lib.ptm_master_destroy(h)


# ------------------------------------------- renderMode 0 (reprojection + denoiser + TAA)

def test_denoise_mode_sequence_matches_oracle_bit_exact(tracer):
    """Three frames with a moving camera in the reference's start-up mode: Result (what the app displays), Result1
    (history) and Result2 (accumulated radiance) of every frame equal the oracle's, bit for bit."""
    from reproj_util import gpu_sequence, oracle_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    W, H = 64, 36
    o = po.Oracle.from_scene(s, sky)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_sequence(o, s, W, H, 3, spp=2, max_bounces=4, frames_n=5, mode=0)
    got = gpu_sequence(tracer, s, W, H, 3, spp=2, max_bounces=4, frames_n=5, mode=0)
    for k in range(3):
        for name, i in (("Result2", 4), ("Result1", 1), ("Result", 0)):
            assert (got[k][i].view(np.uint32) == want[k][i].view(np.uint32)).all(), f"{name}, frame {k}"
        assert got[k][3]["rays"] == want[k][3]["rays"]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "denoise_64x36.npz"))
    assert (got[2][0].view(np.uint32) == g["result2"].view(np.uint32)).all()
    assert (got[2][4].view(np.uint32) == g["result2_2"].view(np.uint32)).all()


def test_denoise_mode_mesh_scene_odd_size(tracer):
    from reproj_util import gpu_sequence, oracle_sequence
    s = pa.Scene("blob70k"); sky = pa.procedural_sky(256, 128)
    W, H = 131, 77                                        # neither a multiple of the 64x4 launch tile nor of 32 rows
    o = po.Oracle.from_scene(s, sky, accel=True)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_sequence(o, s, W, H, 3, spp=1, max_bounces=8, frames_n=10, threads=8, mode=0)
    got = gpu_sequence(tracer, s, W, H, 3, spp=1, max_bounces=8, frames_n=10, mode=0)
    for k in range(3):
        for name, i in (("Result2", 4), ("Result1", 1), ("Result", 0)):
            assert (got[k][i].view(np.uint32) == want[k][i].view(np.uint32)).all(), f"{name}, frame {k}"
    assert got[2][0][..., :3].mean() > 0.02


def test_denoise_mode_errors_and_depth(tracer):
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    W, H = 40, 24
    tracer.load_scene(s, sky); tracer.resize(W, H)
    with pytest.raises(pa.PtraceError, match="whole frames"):
        tracer.render(pa.make_params(s, W, H, render_mode=0, tile=(0, 0, 20, 10)))
    with pytest.raises(pa.PtraceError, match="accumulate"):
        tracer.render(pa.make_params(s, W, H, render_mode=0, accumulate=pa.PT_ACCUM_SUM))
    tracer.render(pa.make_params(s, W, H, depth=1)); want = tracer.read_result()
    marker = np.full((H, W, 4), 0.125, np.float32)
    tracer.write_target(pa.PT_TARGET_DENOISED, marker)
    tracer.render(pa.make_params(s, W, H, depth=1, render_mode=0))
    assert (tracer.read_result().view(np.uint32) == want.view(np.uint32)).all()   # SH:799-809 returns before the mode-0 block
    assert (tracer.read_converged() == marker).all()                              # but after Result1 = Result2 (SH:733-736)
    assert (tracer.read_target(pa.PT_TARGET_DENOISED) == marker).all()


# ------------------------------------------- renderMode 5 / 6 (partial frames)

@pytest.mark.parametrize("mode", [5, 6])
def test_stripe_modes_match_oracle_bit_exact(tracer, mode):
    """Six frames, four stripes per cycle, moving camera: the displayed frame and the history mosaic of every frame
    equal the oracle's bit for bit; the camera arrays keep the previous cycle's tail."""
    from reproj_util import gpu_stripe_sequence, oracle_stripe_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    W, H = 64, 36
    o = po.Oracle.from_scene(s, sky)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_stripe_sequence(o, s, W, H, 6, 4, mode, spp=1, max_bounces=4)
    got = gpu_stripe_sequence(tracer, s, W, H, 6, 4, mode, spp=1, max_bounces=4)
    for k in range(6):
        assert (got[k][0].view(np.uint32) == want[k][0].view(np.uint32)).all(), f"Result, frame {k}"
        assert (got[k][1].view(np.uint32) == want[k][1].view(np.uint32)).all(), f"Result1, frame {k}"
        assert got[k][3]["rays"] == want[k][3]["rays"] and got[k][3]["paths"] == want[k][3]["paths"]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "stripes_64x36.npz"))
    assert (got[5][0].view(np.uint32) == g[f"mode{mode}_result5"].view(np.uint32)).all()
    assert (got[5][1].view(np.uint32) == g[f"mode{mode}_result1_5"].view(np.uint32)).all()
    assert (got[5][0][..., :3] > 0).mean() > 0.5                       # reprojection really fills the other stripes


def test_stripe_modes_mesh_scene_uneven_rows(tracer):
    """70 k triangles, 131x77 with 10 stripes: rows per stripe is not an integer, so which rows belong to a stripe
    is decided by the shader's float comparisons."""
    from reproj_util import gpu_stripe_sequence, oracle_stripe_sequence
    s = pa.Scene("blob70k"); sky = pa.procedural_sky(256, 128)
    W, H = 131, 77
    o = po.Oracle.from_scene(s, sky, accel=True)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    for mode in (5, 6):
        want = oracle_stripe_sequence(o, s, W, H, 12, 10, mode, spp=1, max_bounces=4, threads=8)
        got = gpu_stripe_sequence(tracer, s, W, H, 12, 10, mode, spp=1, max_bounces=4)
        for k in range(12):
            assert (got[k][0].view(np.uint32) == want[k][0].view(np.uint32)).all(), f"mode {mode} Result, frame {k}"
            assert (got[k][1].view(np.uint32) == want[k][1].view(np.uint32)).all(), f"mode {mode} Result1, frame {k}"


def test_stripe_mode_errors(tracer):
    s = pa.Scene("fixture")
    W, H = 40, 24
    tracer.load_scene(s, None); tracer.resize(W, H)
    for kw in (dict(divisions=0, counter=0), dict(divisions=4, counter=4), dict(divisions=101, counter=0),
               dict(divisions=4, counter=1, tile=(0, 0, 20, 10)), dict(divisions=4, counter=1, accumulate=pa.PT_ACCUM_MEAN)):
        with pytest.raises(pa.PtraceError):
            tracer.render(pa.make_params(s, W, H, render_mode=5, **kw))


# ------------------------------------------- renderMode 3, 4, 10 (reprojection variants), 11 / 12 (pause), 7 / 8 (full-frame)

@pytest.mark.parametrize("mode,kw", [(3, {}), (4, dict(left_eye=1)), (4, dict(left_eye=0)), (10, {})])
def test_reprojection_variants_match_oracle_bit_exact(tracer, mode, kw):
    """BlurAndReproj, StereoReproj (both eyes' offsets) and NewRender over three frames with a moving camera."""
    from reproj_util import gpu_sequence, oracle_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    # renderMode 4's hard-coded pixel offsets (156 / 538, made for a 2016-row eye texture) need a tall image to land on screen
    W, H = ((320, 400) if kw["left_eye"] else (240, 1300)) if mode == 4 else (64, 36)
    spp = 1 if mode == 10 else 2                                # the host sets 1 sample from mode 10 on (RayTracingMaster.cs:531)
    o = po.Oracle.from_scene(s, sky, accel=True)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_sequence(o, s, W, H, 3, spp=spp, max_bounces=4, frames_n=5, mode=mode, threads=8, **kw)
    got = gpu_sequence(tracer, s, W, H, 3, spp=spp, max_bounces=4, frames_n=5, mode=mode, **kw)
    plain_alpha = None
    for k in range(3):
        assert (got[k][0].view(np.uint32) == want[k][0].view(np.uint32)).all(), f"Result, frame {k}"
        assert (got[k][1].view(np.uint32) == want[k][1].view(np.uint32)).all(), f"Result1, frame {k}"
        assert got[k][3]["rays"] == want[k][3]["rays"], k
    # the history really is used: frame 2 differs from a plain render of the same frame in many pixels
    from reproj_util import frame_params
    p1, _ = frame_params(s, W, H, 2, spp, 4, 5, mode=1)
    plain = o.render(p1, W, H, threads=8)[0]
    changed = (np.abs(got[2][0][..., :3] - plain[..., :3]).max(-1) > 0).sum()
    assert changed > (20 if mode == 4 else 500), changed
    if mode == 10:
        assert got[2][3]["rays"] < 0.8 * o.render(p1, W, H, threads=8)[2]["rays"]      # covered pixels trace no path


def test_new_render_mode_several_samples(tracer):
    """renderMode 10 with more than one sample (the host never asks for it, the shader allows it): the decision to
    show the history is the same for every sample of a pixel, so the images agree; only the count of repeated
    primary rays differs (the shader re-traces the covered pixels' camera rays)."""
    from reproj_util import gpu_sequence, oracle_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    W, H = 48, 27
    o = po.Oracle.from_scene(s, sky)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_sequence(o, s, W, H, 3, spp=3, max_bounces=4, frames_n=5, mode=10)
    got = gpu_sequence(tracer, s, W, H, 3, spp=3, max_bounces=4, frames_n=5, mode=10)
    for k in range(3):
        assert (got[k][0].view(np.uint32) == want[k][0].view(np.uint32)).all(), f"Result, frame {k}"
        assert got[k][3]["rays"] <= want[k][3]["rays"]


@pytest.mark.parametrize("mode", [11, 12])
def test_pause_modes_match_oracle_bit_exact(tracer, mode):
    """Two frames of renderMode 2, then three of PlanerPause / DepthPause with the camera moving on: the frozen frame
    re-projected, bit for bit; Result1 and the history cameras stay as the last rendered frame left them."""
    from reproj_util import gpu_pause_sequence, oracle_pause_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    W, H = 64, 36
    o = po.Oracle.from_scene(s, sky)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_pause_sequence(o, s, W, H, 2, 3, mode)
    got = gpu_pause_sequence(tracer, s, W, H, 2, 3, mode)
    for k in range(5):
        assert (got[k][0].view(np.uint32) == want[k][0].view(np.uint32)).all(), f"Result, frame {k}"
        assert (got[k][1].view(np.uint32) == want[k][1].view(np.uint32)).all(), f"Result1, frame {k}"
        assert got[k][3]["rays"] == want[k][3]["rays"], k
    assert (got[4][1] == got[2][1]).all()                                   # Result1 is not touched while paused
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "othermodes_64x36.npz"))
    assert (got[4][0].view(np.uint32) == g[f"mode{mode}_result4"].view(np.uint32)).all()
    assert (got[4][0][..., :3] > 0).mean() > 0.4 and not (got[4][0] == got[3][0]).all()
    assert got[3][3]["rays"] == (0 if mode == 11 else W * H)


@pytest.mark.parametrize("mode", [11, 12])
def test_plain_mode_keeps_the_previous_frame_for_the_pause_modes(tracer, mode):
    """RayTracingShader.compute:728-731 copies Result to Result1 at the start of a renderMode 1 dispatch too (the
    condition covers modes 1, 2, 3, 4 and 10), so a switch from the plain mode to PlanerPause / DepthPause re-projects
    the second-last plain frame.  Two plain frames, then three paused ones: after the second plain frame Result1 IS
    the first plain frame - a fact about the reference, not a choice shared with the oracle - and every frame of the
    sequence equals the oracle's in both targets."""
    from reproj_util import gpu_pause_sequence, oracle_pause_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    W, H = 64, 36
    o = po.Oracle.from_scene(s, sky)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_pause_sequence(o, s, W, H, 2, 3, mode, warm_mode=1)
    got = gpu_pause_sequence(tracer, s, W, H, 2, 3, mode, warm_mode=1)
    assert (got[1][1].view(np.uint32) == got[0][0].view(np.uint32)).all() and (got[0][0][..., :3] > 0).mean() > 0.4
    assert not (got[1][0] == got[0][0]).all()
    for k in range(5):
        assert (got[k][0].view(np.uint32) == want[k][0].view(np.uint32)).all(), f"Result, frame {k}"
        assert (got[k][1].view(np.uint32) == want[k][1].view(np.uint32)).all(), f"Result1, frame {k}"
    assert (got[4][1] == got[1][1]).all()                                   # untouched while paused
    assert (got[4][0][..., :3] > 0).mean() > 0.3


def test_pause_mode_depth_and_errors(tracer):
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    W, H = 40, 24
    tracer.load_scene(s, sky); tracer.resize(W, H)
    tracer.render(pa.make_params(s, W, H, depth=1)); want = tracer.read_result()
    tracer.render(pa.make_params(s, W, H, depth=1, render_mode=12))
    assert (tracer.read_result().view(np.uint32) == want.view(np.uint32)).all()   # SH:799-809 comes before the pause block
    with pytest.raises(pa.PtraceError, match="whole frames"):
        tracer.render(pa.make_params(s, W, H, render_mode=11, tile=(0, 0, 20, 10)))
    marker = np.full((H, W, 4), 0.375, np.float32)
    tracer.write_target(pa.PT_TARGET_RESULT, marker)
    tracer.render(pa.make_params(s, W, H, render_mode=11, spp=0))                  # the sample loop does not run: nothing written
    assert (tracer.read_result() == marker).all()


@pytest.mark.parametrize("mode", [7, 8])
def test_full_frame_reproj_modes_match_oracle_bit_exact(tracer, mode):
    """Ten frames, four stripes per cycle, moving camera: Result (the reprojected display), Result1 (the cycle's
    reference frame) and Result2 (the frame being assembled) of every dispatch equal the oracle's."""
    from reproj_util import gpu_stripe_sequence, oracle_stripe_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(128, 64)
    W, H = 64, 36
    o = po.Oracle.from_scene(s, sky)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_stripe_sequence(o, s, W, H, 10, 4, mode, spp=1, max_bounces=4)
    got = gpu_stripe_sequence(tracer, s, W, H, 10, 4, mode, spp=1, max_bounces=4)
    for k in range(10):
        assert (got[k][0].view(np.uint32) == want[k][0].view(np.uint32)).all(), f"Result, frame {k}"
        assert (got[k][1].view(np.uint32) == want[k][1].view(np.uint32)).all(), f"Result1, frame {k}"
        assert (got[k][4].view(np.uint32) == want[k][4].view(np.uint32)).all(), f"Result2, frame {k}"
        assert got[k][3]["rays"] == want[k][3]["rays"] and got[k][3]["paths"] == want[k][3]["paths"], k
    assert (got[9][0][..., :3] > 0).mean() > 0.5
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "othermodes_64x36.npz"))
    assert (got[9][0].view(np.uint32) == g[f"mode{mode}_result9"].view(np.uint32)).all()
    assert (got[9][4].view(np.uint32) == g[f"mode{mode}_result2_9"].view(np.uint32)).all()


@pytest.mark.parametrize("mode,divisions", [(7, 1), (8, 1), (7, 3), (5, 1), (6, 1)])
def test_stripe_modes_degenerate_divisions(tracer, mode, divisions):
    """One stripe per cycle: every dispatch has counter 0, so renderMode 7 / 8 render the whole frame from "history
    camera -1" and publish it at once, and renderMode 5 / 6 reproject nothing.  Three stripes: 36 rows / 3."""
    from reproj_util import gpu_stripe_sequence, oracle_stripe_sequence
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    W, H = 48, 36
    o = po.Oracle.from_scene(s, sky)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    want = oracle_stripe_sequence(o, s, W, H, 4, divisions, mode, spp=1, max_bounces=3)
    got = gpu_stripe_sequence(tracer, s, W, H, 4, divisions, mode, spp=1, max_bounces=3)
    for k in range(4):
        for t in (0, 1) + ((4,) if mode in (7, 8) else ()):
            assert (got[k][t].view(np.uint32) == want[k][t].view(np.uint32)).all(), f"target {t}, frame {k}"
        assert got[k][3]["rays"] == want[k][3]["rays"], k


def test_full_frame_reproj_mesh_scene_uneven_rows_and_samples(tracer):
    from reproj_util import gpu_stripe_sequence, oracle_stripe_sequence
    s = pa.Scene("blob70k"); sky = pa.procedural_sky(256, 128)
    W, H = 131, 77
    o = po.Oracle.from_scene(s, sky, accel=True)
    tracer.load_scene(s, sky); tracer.resize(W, H)
    for mode in (7, 8):
        want = oracle_stripe_sequence(o, s, W, H, 12, 10, mode, spp=2, max_bounces=4, threads=8)
        got = gpu_stripe_sequence(tracer, s, W, H, 12, 10, mode, spp=2, max_bounces=4)
        for k in range(12):
            for t, name in ((0, "Result"), (1, "Result1"), (4, "Result2")):
                assert (got[k][t].view(np.uint32) == want[k][t].view(np.uint32)).all(), f"mode {mode} {name}, frame {k}"


# ------------------------------------------- per-mesh materials and the GGX lobe (extensions)

def test_mesh_materials_and_ggx_match_oracle(tracer, sky_small):
    """Three instances of one mesh with different surfaces - a GGX metal, a diffuse emitter, the shader's own
    constants given explicitly - render bit-identically on both sides; dropping the materials restores the
    hard-coded surface; a refit keeps every triangle's material."""
    f = pa.Scene("fixture")
    n_idx = len(f.indices)
    objs = np.zeros(3, dtype=f.objects.dtype)
    for k in range(3):
        objs[k] = (_rot_y(15.0 * k, (-3.0 + 3.0 * k, 0.0, 1.5 * k)), 0, n_idx)
    mats = pa.materials([((0.0, 0.0, 0.0), 0.8, (0.95, 0.64, 0.54), pa.PT_LOBE_GGX),
                         ((0.7, 0.2, 0.1), 0.3, (0.04, 0.04, 0.04), pa.PT_LOBE_PHONG, (0.2, 0.2, 0.6)),
                         ((0.0, 0.0, 0.0), 0.99, (0.65, 0.65, 0.65), pa.PT_LOBE_PHONG)])
    tracer.set_spheres(f.spheres); tracer.set_meshes(objs, f.vertices, f.indices); tracer.set_sky(sky_small)
    tracer.set_mesh_materials(mats)
    W, H = 96, 54
    tracer.resize(W, H)
    p = pa.make_params(f, W, H, spp=4, max_bounces=6, seed=0.7, eye=(0, 5, -12), target=(0, 1, 2), fov=60)
    tracer.stats(); tracer.render(p)
    g = tracer.read_result(); gid = tracer.read_hit_ids(); st = tracer.stats()
    c, cid, cst = po.Oracle(f.spheres, objs, f.vertices, f.indices, sky=sky_small, materials=mats).render(p, W, H, threads=4)
    assert (gid == cid).all() and st["rays"] == cst["rays"]
    _assert_same_image(g, c, "materials")
    plain = po.Oracle(f.spheres, objs, f.vertices, f.indices, sky=sky_small).render(p, W, H, threads=4)[0]
    assert not (plain == c).all()
    # refit: the records are rewritten on the device, each with its object's index
    moved = objs.copy(); moved["m"][0] = _rot_y(40.0, (-3.5, 0.3, 0.5))
    tracer.update_transforms(moved); tracer.render(p)
    c2 = po.Oracle(f.spheres, moved, f.vertices, f.indices, sky=sky_small, materials=mats).render(p, W, H, threads=4)[0]
    _assert_same_image(tracer.read_result(), c2, "materials after refit")
    # back to the shader's surface
    tracer.set_mesh_materials(None); tracer.render(p)
    c3 = po.Oracle(f.spheres, moved, f.vertices, f.indices, sky=sky_small).render(p, W, H, threads=4)[0]
    _assert_same_image(tracer.read_result(), c3, "materials removed")
    with pytest.raises(pa.PtraceError):
        tracer.set_mesh_materials(mats[:2])
    bad = mats.copy(); bad["lobe"][0] = 7
    with pytest.raises(pa.PtraceError, match="lobe"):
        tracer.set_mesh_materials(bad)


# ------------------------------------------- the display blit (Denoiser.shader)

def test_display_denoise_matches_oracle(tracer, sky_small):
    s = pa.Scene("fixture")
    tracer.load_scene(s, sky_small)
    for W, H in ((64, 36), (131, 77)):
        tracer.resize(W, H)
        tracer.render(pa.make_params(s, W, H, spp=2, max_bounces=4, seed=0.4))
        tracer.display_denoise()
        res = tracer.read_result()
        got = tracer.read_target(pa.PT_TARGET_DISPLAY)
        want = po.display_denoise(res)
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), (W, H)
        assert (tracer.read_result() == res).all()                  # Result itself is untouched
        assert 0.5 * res[..., :3].clip(0, 1).mean() < got[..., :3].mean() < 1.5 * res[..., :3].clip(0, 1).mean()


def test_reuse_primary_hit_option(tracer, sky_small):
    """The shader applies no jitter, so every sample of a pixel traces the same camera ray; with the option on that ray
    is traced once per pixel.  Same image bit for bit, fewer rays: one primary per pixel instead of one per sample."""
    s = pa.Scene("fixture")
    tracer.load_scene(s, sky_small)
    W, H = 64, 36
    tracer.resize(W, H)
    o = po.Oracle.from_scene(s, sky_small)
    try:
        for spp, maxb in ((1, 4), (5, 4), (16, 8)):
            p = pa.make_params(s, W, H, spp=spp, max_bounces=maxb, seed=0.37)
            want, ids, st = o.render(p, W, H, threads=4)
            tracer.set_option("reuse_primary_hit", 1)
            tracer.stats(); tracer.render(p)
            got = tracer.read_result(); gst = tracer.stats()
            _assert_same_image(got, want, f"reuse, {spp} spp")
            assert (tracer.read_hit_ids() == ids).all()
            assert gst["paths"] == st["paths"] == W * H * spp
            assert gst["rays"] == st["rays"] - (st["paths"] - W * H)
            tracer.set_option("reuse_primary_hit", 0)
            tracer.stats(); tracer.render(p)
            assert tracer.stats()["rays"] == st["rays"]
    finally:
        tracer.set_option("reuse_primary_hit", 0)


def test_master_mirror_partial_frame_mode_cycles_the_counter():
    """The C++ RayTracingMaster mirror in renderMode 5 (RayTracingMaster.cs:513-526, 736-737): _Counter = counter++ %
    divisions, the camera lists grow through a cycle and restart at its beginning, a shorter list leaves the tail of
    the shader's arrays alone.  The oracle follows from the parameters the mirror uploads."""
    import ctypes as C
    from pathtracing_amd import _ffi
    lib = _ffi.ptmaster()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    try:
        s = pa.Scene("fixture")
        W, H, D = 64, 36, 4
        assert lib.ptm_master_load_scene(h, s.handle, 128, 64) == 0, lib.ptm_master_error(h)
        lib.ptm_master_configure(h, W, H, 1, 4, 0, pa.PT_ACCUM_NONE, 6)
        lib.ptm_master_set_render_mode(h, 5)
        lib.ptm_master_set_divisions(h, D)
        orc = po.Oracle.from_scene(s, pa.procedural_sky(128, 64), accel=False)
        ctx = lib.ptm_master_context(h)
        res = np.zeros((H, W, 4), np.float32); res1 = np.zeros((H, W, 4), np.float32)
        state = [pa.PtCameraHistory() for _ in range(100)]
        old = []
        counters = []
        for k in range(2 * D + 1):
            eye = (s.eye[0] + 0.1 * k, s.eye[1], s.eye[2] + 0.15 * k)
            lib.ptm_master_set_camera(h, (C.c_float * 3)(*eye), (C.c_float * 3)(*s.target), C.c_float(s.fov), C.c_float(0.0))
            p = pa.PtFrameParams()
            lib.ptm_master_peek_params(h, C.byref(p))
            counters.append(p.counter)
            assert p.render_mode == 5 and p.divisions == D
            assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
            state[:len(old)] = old
            orc.render_history(p, state, W, H, res, res1, threads=4)
            if p.counter == 0:
                old = []
            old.append(pa.camera_history_entry(eye, s.target, s.fov, W / H))
            got = np.empty((H, W, 4), np.float32)
            assert _ffi.ptrace().pt_read_result(ctx, got.ctypes.data) == 0
            _assert_same_image(got, res, f"master stripes frame {k}")
            assert _ffi.ptrace().pt_read_converged(ctx, got.ctypes.data) == 0
            _assert_same_image(got, res1, f"master stripes history {k}")
        assert counters == [0, 1, 2, 3, 0, 1, 2, 3, 0]
    finally:
        lib.ptm_master_destroy(h)


def test_master_mirror_switches_through_every_render_mode():
    """The C++ RayTracingMaster mirror driven through the host's mode keys (RayTracingMaster.cs:177-212) while the
    camera moves: Reproj, BlurAndReproj, NewRender (one sample, :531), DepthPause (empty camera lists, :513-526),
    FullFrameReprojDepth across a cycle boundary, Denoise.  The counter runs on in every mode (:737).  The oracle is
    fed the parameters the mirror uploads and a Python restatement of the list handling."""
    import ctypes as C
    from pathtracing_amd import _ffi
    lib = _ffi.ptmaster()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    try:
        s = pa.Scene("fixture")
        W, H, D = 64, 36, 4
        assert lib.ptm_master_load_scene(h, s.handle, 128, 64) == 0, lib.ptm_master_error(h)
        lib.ptm_master_configure(h, W, H, 2, 4, 0, pa.PT_ACCUM_NONE, 6)
        lib.ptm_master_set_divisions(h, D)
        orc = po.Oracle.from_scene(s, pa.procedural_sky(128, 64), accel=False)
        ctx = lib.ptm_master_context(h)
        res = np.zeros((H, W, 4), np.float32); res1 = np.zeros((H, W, 4), np.float32); res2 = np.zeros((H, W, 4), np.float32)
        state = [pa.PtCameraHistory() for _ in range(100)]
        old = []
        modes = [2, 2, 3, 10, 12, 12, 8, 8, 8, 8, 8, 8, 0, 0]
        for k, mode in enumerate(modes):
            lib.ptm_master_set_render_mode(h, mode)
            eye = (s.eye[0] + 0.1 * k, s.eye[1], s.eye[2] + 0.15 * k)
            lib.ptm_master_set_camera(h, (C.c_float * 3)(*eye), (C.c_float * 3)(*s.target), C.c_float(s.fov), C.c_float(0.0))
            p = pa.PtFrameParams()
            lib.ptm_master_peek_params(h, C.byref(p))
            assert p.render_mode == mode and p.counter == k % D
            assert p.samples_per_pixel == (2 if mode < 10 else 1)
            assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
            state[:len(old)] = old
            orc.render_history(p, state, W, H, res, res1, threads=4, result2=res2)
            if mode < 5 or mode > 8 or p.counter == 0:
                old = []
            if mode < 11 and mode != 1:
                old.append(pa.camera_history_entry(eye, s.target, s.fov, W / H))
            got = np.empty((H, W, 4), np.float32)
            assert _ffi.ptrace().pt_read_result(ctx, got.ctypes.data) == 0
            _assert_same_image(got, res, f"frame {k} (mode {mode}) Result")
            assert _ffi.ptrace().pt_read_converged(ctx, got.ctypes.data) == 0
            _assert_same_image(got, res1, f"frame {k} (mode {mode}) Result1")
            assert _ffi.ptrace().pt_read_target(ctx, pa.PT_TARGET_DENOISED, got.ctypes.data) == 0
            _assert_same_image(got, res2, f"frame {k} (mode {mode}) Result2")
    finally:
        lib.ptm_master_destroy(h)


def test_group_reduce_adds_the_members_images_in_index_order(sky_small):
    """pt_group_*: three contexts (all on device 0 here; one per GPU in use) each add their share of six sample
    indices into their own converged image (PT_ACCUM_SUM); the reduce leaves (member0 + member1) + member2 on the
    root, bit for bit, and the members' next renders are ordered after it."""
    W, H = 64, 36
    fixture_scene = pa.Scene("fixture")
    with pa.PathTracerGroup([0, 0, 0]) as grp:
        assert len(grp.members) == 3
        for m in grp.members:
            m.load_scene(fixture_scene, sky_small); m.resize(W, H)
        for k in range(6):
            m = grp.members[k % 3]
            m.render(pa.make_params(fixture_scene, W, H, seed=pa.frame_seed(k), spp=2, max_bounces=4,
                                    accumulate=pa.PT_ACCUM_SUM, sample_index=k // 3))
        parts = [m.read_converged() for m in grp.members]
        grp.reduce(root=0)
        # the members go on rendering straight away; member 1 overwrites its image (sample_index 0 restarts the sum)
        grp.members[1].render(pa.make_params(fixture_scene, W, H, seed=0.125, spp=1, max_bounces=4,
                                             accumulate=pa.PT_ACCUM_SUM, sample_index=0))
        total = grp.members[0].read_converged()
        want = (parts[0] + parts[1]) + parts[2]
        assert (total.view(np.uint32) == want.view(np.uint32)).all()
        assert (grp.members[2].read_converged() == parts[2]).all()            # the other members keep theirs
        # against one context taking all six samples: the same image up to float summation order
        with pa.PathTracer(0) as one:
            one.load_scene(fixture_scene, sky_small); one.resize(W, H)
            for k in range(6):
                one.render(pa.make_params(fixture_scene, W, H, seed=pa.frame_seed(k), spp=2, max_bounces=4,
                                          accumulate=pa.PT_ACCUM_SUM, sample_index=k))
            ref = one.read_converged()
        assert np.allclose(total, ref, rtol=1e-5, atol=1e-5)
        with pytest.raises(pa.PtraceError, match="root"):
            grp.reduce(root=3)
        grp.members[2].resize(32, 18)
        with pytest.raises(pa.PtraceError, match="sizes differ"):
            grp.reduce(root=0)
    with pytest.raises(pa.PtraceError, match="ordinal"):
        pa.PathTracerGroup([0, 4096])


def test_deep_traversal_stack_spills_to_the_overflow_buffer(tracer):
    """16 384 large, nearly coincident triangles stacked 0.01 apart in depth: a ray through the stack enters every
    child box of every node on its way down, so the traversal stack grows by three entries per level of the four-wide
    tree (one per level of the binary tree) - far past the entries kept in LDS - before the first hit prunes anything.
    Image, hit ids and ray queries must not notice where an entry is stored."""
    n = 16384
    rng = np.random.default_rng(11)
    z = (np.arange(n) * 0.01).astype(np.float32)
    jit = rng.uniform(-0.5, 0.5, (n, 2)).astype(np.float32)
    v = np.zeros((n, 3, 3), np.float32)
    base = np.array([[-40.0, -30.0], [0.0, 45.0], [40.0, -30.0]], np.float32)
    for c in range(3):
        v[:, c, 0] = base[c, 0] + jit[:, 0]; v[:, c, 1] = base[c, 1] + jit[:, 1]; v[:, c, 2] = z
    idx = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    idx[1::2] = idx[1::2, ::-1]                                   # every other one faces away (culled, SH:206)
    objs = np.zeros(1, dtype=np.dtype([("m", np.float32, 16), ("offset", np.int32), ("count", np.int32)]))
    objs[0] = (np.eye(4, dtype=np.float32).reshape(16), 0, 3 * n)
    spheres = np.zeros((0, 14), np.float32)
    sky = pa.procedural_sky(64, 32)
    verts = v.reshape(-1, 3); index = idx.reshape(-1)
    tracer.set_spheres(spheres); tracer.set_meshes(objs, verts, index); tracer.set_sky(sky)
    orc = po.Oracle(spheres, objs, verts, index, sky=sky, accel=True)
    W, H = 64, 36
    tracer.resize(W, H)
    f = pa.Scene("fixture")
    for eye, target in (((0.0, 2.0, -6.0), (0.0, 2.0, 0.0)), ((3.0, -1.0, n * 0.01 + 6.0), (0.0, 0.0, 0.0))):
        p = pa.make_params(f, W, H, seed=0.375, spp=2, max_bounces=4, eye=eye, target=target)
        tracer.stats(); tracer.render(p)
        g = tracer.read_result(); gid = tracer.read_hit_ids(); st = tracer.stats()
        c, cid, cst = orc.render(p, W, H, threads=8)
        assert (gid == cid).all()
        _assert_same_image(g, c, f"stack of triangles seen from {eye}")
        assert st["rays"] == cst["rays"]
        assert (gid > 0).mean() > 0.5                             # the stack fills most of the view
        # a camera ray descends ~7 levels pushing 3 siblings at each (21 entries) and pops every one of them again;
        # the reflected rays leave the stack at once, which pulls the mean down
        assert st["nodes_visited"] / st["rays"] > 8
    # the one-pass primary kernel (_Depth dispatch) and the extended path kernel (a material on the mesh) share the stack code
    pd = pa.make_params(f, W, H, depth=1, eye=(0.0, 2.0, -6.0), target=(0.0, 2.0, 0.0))
    tracer.render(pd)
    _assert_same_image(tracer.read_result(), orc.render(pd, W, H, threads=8)[0], "depth dispatch through the stack of triangles")
    mats = pa.materials([((0.2, 0.3, 0.4), 0.6, (0.5, 0.5, 0.5), pa.PT_LOBE_GGX)])
    tracer.set_mesh_materials(mats)
    orc_m = po.Oracle(spheres, objs, verts, index, sky=sky, accel=True, materials=mats)
    pm = pa.make_params(f, W, H, seed=0.625, spp=2, max_bounces=4, eye=(0.0, 2.0, -6.0), target=(0.0, 2.0, 0.0))
    tracer.render(pm)
    _assert_same_image(tracer.read_result(), orc_m.render(pm, W, H, threads=8)[0], "extended kernel through the stack of triangles")
    tracer.set_mesh_materials(None)
    # ray queries: the tree against the exhaustive loop on the device
    od = np.zeros((2000, 6), np.float32)
    od[:, :2] = rng.uniform(-20, 20, (2000, 2)); od[:, 2] = -5.0
    d = rng.normal(0, 0.15, (2000, 3)); d[:, 2] = 1.0
    od[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    ib, tb = tracer.trace_rays(od); ie, te = tracer.trace_rays(od, brute_force=True)
    assert (ib == ie).all() and (tb.view(np.uint32) == te.view(np.uint32)).all()


def test_compiled_host_renders_what_the_ctypes_harness_renders():
    """host/ptrace_demo (C++ RayTracingMaster mirror + libptrace, no Python in the process) against the same mirror
    driven through ctypes: same frames, same checksum."""
    import ctypes as C
    import json
    import subprocess
    from pathtracing_amd import _ffi
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pathtracing_amd", "host", "ptrace_demo")
    assert os.path.exists(exe), "host/ptrace_demo is not built (python -m pathtracing_amd.build)"
    W, H, frames = 96, 54, 3
    out = subprocess.run([exe, "--scene", "fixture", "--size", f"{W}x{H}", "--spp", "2", "--bounces", "4", "--frames", str(frames)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["rays"] > frames * W * H * 2 and rep["mrays_per_s"] > 0
    lib = _ffi.ptmaster()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    try:
        s = pa.Scene("fixture")
        assert lib.ptm_master_load_scene(h, s.handle, 2048, 1024) == 0, lib.ptm_master_error(h)
        lib.ptm_master_configure(h, W, H, 2, 4, 0, pa.PT_ACCUM_NONE, 10)
        lib.ptm_master_set_camera(h, (C.c_float * 3)(*s.eye), (C.c_float * 3)(*s.target), C.c_float(s.fov), C.c_float(0.0))
        for _ in range(frames + 1):
            assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
        img = np.empty((H, W, 4), np.float32)
        assert _ffi.ptrace().pt_read_result(lib.ptm_master_context(h), img.ctypes.data) == 0
    finally:
        lib.ptm_master_destroy(h)
    assert int(np.bitwise_xor.reduce(img.view(np.uint32).reshape(-1))) == rep["image_crc"]
    assert np.allclose(img[..., :3].reshape(-1, 3).mean(0), rep["image_mean_rgb"], rtol=1e-4)
    bad = subprocess.run([exe, "--scene", "nowhere"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "unknown scene" in bad.stderr
    # options before the first frame: another kernel, the same frames
    for opt in ("backend=2", "split_percent=0", "instancing=1"):
        alt = subprocess.run([exe, "--scene", "fixture", "--size", f"{W}x{H}", "--spp", "2", "--bounces", "4", "--frames", str(frames), "--option", opt],
                             capture_output=True, text=True, timeout=120)
        assert alt.returncode == 0, alt.stderr
        assert json.loads(alt.stdout.strip().splitlines()[-1])["image_crc"] == rep["image_crc"], opt
    bad = subprocess.run([exe, "--scene", "fixture", "--option", "backend=9"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "backend" in bad.stderr


def test_non_finite_geometry_does_not_hide_the_rest_of_the_scene(tracer, sky_small):
    """Vertices and spheres holding NaN or infinity: the shader's tests on them come out false (every comparison with
    NaN does), so they are never hit - and everything else must render as if they were not there, after a build and
    after a refit alike."""
    s = pa.Scene("fixture")
    verts = np.array(s.vertices, np.float32).reshape(-1, 3).copy()
    verts[5, 0] = np.nan; verts[40, 1] = np.inf; verts[77] = (-np.inf, np.nan, 3.0); verts[120, 2] = -np.inf
    spheres = np.array(s.spheres, np.float32).reshape(-1, 14).copy()
    spheres[2, 0] = np.nan                        # position.x
    spheres[6, 3] = np.inf                        # radius
    spheres[9, 1] = -np.inf
    tracer.set_spheres(spheres); tracer.set_meshes(s.objects, verts, s.indices); tracer.set_sky(sky_small)
    orc = po.Oracle(spheres, s.objects, verts, s.indices, sky=sky_small, accel=False)
    W, H = 64, 36
    tracer.resize(W, H)
    p = pa.make_params(s, W, H, seed=0.5, spp=2, max_bounces=4)
    tracer.render(p)
    c, cid, _ = orc.render(p, W, H, threads=8)
    assert (cid > 0).sum() > 300                                  # plenty of ordinary geometry left to see
    assert (tracer.read_hit_ids() == cid).all()
    _assert_same_image(tracer.read_result(), c, "scene with non-finite vertices and spheres")
    objs = np.array(s.objects).copy()
    objs["m"][0][12] += 0.75                                      # translate in x: a refit on the device
    tracer.update_transforms(objs)
    orc2 = po.Oracle(spheres, objs, verts, s.indices, sky=sky_small, accel=False)
    tracer.render(p)
    c2, cid2, _ = orc2.render(p, W, H, threads=8)
    assert (tracer.read_hit_ids() == cid2).all()
    _assert_same_image(tracer.read_result(), c2, "the same after a refit")
    assert not (cid2 == cid).all()


def test_tree_keeps_the_hits_that_only_rounding_produces(tracer, sky_small):
    """The shader's sphere test accepts rays that pass within |d| * ~7e-4 of a zero-radius sphere's centre (a difference
    of squares of the distance), and its triangle test reports hits well outside a sliver triangle that runs out to 1e6.
    Such hits are artefacts, but the exhaustive loop finds them, so the tree must too: primitive boxes are padded by the
    reach of each test's rounding (found by tools/soak_geometry.py)."""
    rng = np.random.default_rng(21)
    spheres = np.zeros((6, 14), np.float32)
    spheres[:, 0:3] = rng.uniform(-3, 3, (6, 3)); spheres[:, 1] += 4.0
    spheres[:, 3] = [0.0, 0.0, 0.0, 1e-3, 1e-2, 0.5]
    spheres[:, 11:14] = 3.0
    v = np.array([[-0.75, 1, 2.25], [2, 1, 1], [2, 2, 0], [-2, 3, -2], [1e6, 4, 1e6], [0, 1e-6, -1], [3, -1, 3], [-1.75, -1, 3],
                  [2.25, -1, -2], [0, 4, 3]], np.float32)
    idx = np.array([[4, 6, 2], [2, 6, 4], [3, 4, 2], [4, 8, 1], [6, 4, 9], [4, 2, 0], [1, 3, 2], [5, 0, 2]], np.int32).reshape(-1)
    objs = np.zeros(1, dtype=np.dtype([("m", np.float32, 16), ("offset", np.int32), ("count", np.int32)]))
    objs[0] = (np.eye(4, dtype=np.float32).reshape(16), 0, len(idx))
    tracer.set_spheres(spheres); tracer.set_meshes(objs, v, idx); tracer.set_sky(sky_small)
    n = 40000
    od = np.zeros((n, 6), np.float32)
    od[:, :3] = rng.uniform(-25, 25, (n, 3)); od[:, 1] = np.abs(od[:, 1]) + 0.01
    aim = spheres[rng.integers(0, 6, n), 0:3] + (rng.random((n, 1)) < 0.3) * rng.normal(0, 2e-3, (n, 3))   # mostly dead centre
    d = aim - od[:, :3]
    d[n // 2:] = rng.normal(0, 1, (n - n // 2, 3))                # the other half: anywhere, for the sliver triangles
    od[:, 3:] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ib, tb = tracer.trace_rays(od); ie, te = tracer.trace_rays(od, brute_force=True)
    assert (ib == ie).all(), f"{int((ib != ie).sum())} of {n} rays differ"
    assert (tb.view(np.uint32) == te.view(np.uint32)).all()
    zero_radius_hits = int(np.isin(ie[: n // 2], [1, 2, 3]).sum())
    assert zero_radius_hits > 50, zero_radius_hits               # the artefact exists: a point gets hit
    # a camera very far away asks for wider sphere boxes (a rebuild) instead of losing such hits
    far = od[: n // 2].copy()
    far[:, :3] = aim[: n // 2] - od[: n // 2, 3:] * 5000.0
    ib, tb = tracer.trace_rays(far); ie, te = tracer.trace_rays(far, brute_force=True)
    assert (ib == ie).all() and (tb.view(np.uint32) == te.view(np.uint32)).all()


def test_contexts_driven_from_several_threads_at_once(sky_small):
    """A context is single-threaded, but different contexts may be driven from different threads (ctypes drops the GIL
    during every call): four threads, each with its own context, scene, size and renderMode sequence, all running at the
    same time, each frame checked against the oracle."""
    import threading
    from reproj_util import gpu_sequence, gpu_stripe_sequence, oracle_sequence, oracle_stripe_sequence
    jobs = [("fixture", 64, 36, 2, None), ("xr", 48, 40, 0, None), ("fixture", 57, 33, 7, 3), ("blob70k", 40, 30, 10, None)]
    wanted = []
    for name, W, H, mode, div in jobs:
        s = pa.Scene(name)
        o = po.Oracle.from_scene(s, sky_small, accel=True)
        if div:
            wanted.append(oracle_stripe_sequence(o, s, W, H, 6, div, mode, spp=1, max_bounces=4, threads=4))
        else:
            wanted.append(oracle_sequence(o, s, W, H, 4, spp=1 if mode >= 10 else 2, max_bounces=4, frames_n=5, mode=mode, threads=4))
    errors = []
    start = threading.Barrier(len(jobs))

    def work(k):
        try:
            name, W, H, mode, div = jobs[k]
            s = pa.Scene(name)
            with pa.PathTracer(0) as t:
                t.load_scene(s, sky_small); t.resize(W, H)
                start.wait(timeout=60)
                for _ in range(3):                                   # a few rounds, to keep the threads overlapping
                    if div:
                        got = gpu_stripe_sequence(t, s, W, H, 6, div, mode, spp=1, max_bounces=4)
                    else:
                        got = gpu_sequence(t, s, W, H, 4, spp=1 if mode >= 10 else 2, max_bounces=4, frames_n=5, mode=mode)
                    for f, (g, w) in enumerate(zip(got, wanted[k])):
                        for i in (0, 1):
                            if not (g[i].view(np.uint32) == w[i].view(np.uint32)).all():
                                errors.append(f"thread {k} ({name}, mode {mode}) frame {f} target {i} differs")
        except Exception as e:       # noqa: BLE001 - reported by the main thread
            errors.append(f"thread {k}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in threads), "a worker thread did not finish"
    assert not errors, errors[:5]


def test_many_samples_and_long_paths_in_one_dispatch(tracer, fixture_scene, sky_small):
    """The lane state packs sample and bounce counters into one word (sample < 2^19, bounce < 4096): 5 000 samples per
    pixel, and separately 300-segment paths, still match the oracle bit for bit; the limits themselves are rejected."""
    s, orc = fixture_scene
    tracer.load_scene(s, sky_small); tracer.set_mesh_materials(None)      # the shared context may hold another test's scene
    W, H = 5, 3
    tracer.resize(W, H)
    for spp, maxb in ((5000, 3), (2, 300)):
        p = pa.make_params(s, W, H, seed=0.25, spp=spp, max_bounces=maxb)
        tracer.stats(); tracer.render(p)
        g = tracer.read_result(); st = tracer.stats()
        c, _, cst = orc.render(p, W, H, threads=4)
        _assert_same_image(g, c, f"spp={spp} max_bounces={maxb}")
        assert st["rays"] == cst["rays"] and st["paths"] == W * H * spp
    for bad in (dict(spp=1 << 20), dict(max_bounces=4096)):
        with pytest.raises(pa.PtraceError):
            tracer.render(pa.make_params(s, W, H, **bad))


def test_expensive_blocks_first_renders_the_same_image(tracer, sky_small):
    """From the second dispatch over the same rectangle on, the path kernels hand out the 8x8 pixel blocks by falling cost
    of the previous dispatch (the tail of a persistent dispatch then consists of cheap pixels).  Any order must give the
    same image, hit ids and ray count: an odd size (the last work counter gets fewer blocks), a sub-rectangle, and the
    order switched off."""
    s = pa.Scene("fixture")
    tracer.load_scene(s, sky_small)
    tracer.set_mesh_materials(None)
    orc = po.Oracle.from_scene(s, sky_small, accel=True)
    W, H = 397, 283
    tracer.resize(W, H)
    for tile in (None, (13, 9, 371, 270)):
        p = pa.make_params(s, W, H, seed=0.25, spp=4, max_bounces=8, tile=tile)
        want, want_ids, wst = orc.render(p, W, H, threads=8)
        for k in range(3):                       # the first dispatch measures, the later ones use the order
            tracer.set_option("block_order", 0 if k == 2 else 1)
            tracer.write_target(pa.PT_TARGET_RESULT, np.zeros((H, W, 4), np.float32))
            tracer.stats(); tracer.render(p)
            g = tracer.read_result(); gid = tracer.read_hit_ids(); st = tracer.stats()
            x0, y0, tw, th = tile if tile else (0, 0, W, H)
            _assert_same_image(g[y0:y0 + th, x0:x0 + tw], want[y0:y0 + th, x0:x0 + tw], f"dispatch {k}, tile {tile}")
            assert (gid[y0:y0 + th, x0:x0 + tw] == want_ids[y0:y0 + th, x0:x0 + tw]).all()
            assert st["rays"] == wst["rays"] and st["paths"] == wst["paths"]
            if tile:
                outside = np.ones((H, W), bool); outside[y0:y0 + th, x0:x0 + tw] = False
                assert (g[outside] == 0).all()
    tracer.set_option("block_order", 1)


def _room_with_beams():
    """The fixture's mesh inside a room of twelve wall-sized triangles, crossed by long thin beams at odd angles: what
    reference splitting is for."""
    f = pa.Scene("fixture")
    verts = [np.array(f.vertices, np.float32).reshape(-1, 3)]
    index = [np.array(f.indices, np.int32)]
    objs = [(np.array(f.objects["m"][0], np.float32), 0, len(index[0]))]
    nv, ni = len(verts[0]), len(index[0])

    def box(lo, hi):
        lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
        v = np.array([[(hi if (c >> k) & 1 else lo)[k] for k in range(3)] for c in range(8)], np.float32)
        idx = []
        for a, b, c, d in [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]:
            idx += [a, b, c, a, c, d]
        return v, np.array(idx, np.int32)

    v, i = box((-16, -0.5, -16), (16, 14, 16))
    verts.append(v); index.append(i + nv); objs.append((np.eye(4, dtype=np.float32).reshape(16), ni, len(i))); nv += len(v); ni += len(i)
    rng = np.random.default_rng(8)
    v, i = box((-12, -0.06, -0.06), (12, 0.06, 0.06))
    verts.append(v); index.append(i + nv)                            # one beam mesh, many objects
    for _ in range(14):
        ya, za = np.radians(rng.uniform(0, 180)), np.radians(rng.uniform(-50, 50))
        ry = np.array([[np.cos(ya), 0, np.sin(ya)], [0, 1, 0], [-np.sin(ya), 0, np.cos(ya)]])
        rz = np.array([[np.cos(za), -np.sin(za), 0], [np.sin(za), np.cos(za), 0], [0, 0, 1]])
        m = np.eye(4, dtype=np.float32); m[:3, :3] = ry @ rz; m[:3, 3] = (rng.uniform(-4, 4), rng.uniform(2, 9), rng.uniform(-4, 4))
        objs.append((m.T.reshape(16), ni, len(i)))
    return f, np.array(objs, f.objects.dtype), np.concatenate(verts), np.concatenate(index)


def test_large_triangles_are_split_into_references(tracer, sky_small):
    """Option "split_percent": wall-sized triangles and long beams are cut into several boxes before the tree is built.
    The image, the hit ids and the ray count equal the oracle's and those of an unsplit build bit for bit; the rays
    visit fewer nodes and test fewer primitives; a refit that moves one beam keeps the other objects' cuts, still equals
    the oracle and equals a full build of the moved scene; the exhaustive loop sees every primitive once."""
    f, objs, verts, index = _room_with_beams()
    W, H = 128, 72
    cam = dict(eye=(2.0, 6.0, -13.0), target=(0.0, 4.0, 0.0), fov=60)
    p = pa.make_params(f, W, H, spp=2, max_bounces=5, seed=0.41, **cam)
    orc = po.Oracle(f.spheres, objs, verts, index, sky=sky_small)
    c, cid, cst = orc.render(p, W, H, threads=4)
    tracer.set_spheres(f.spheres); tracer.set_meshes(objs, verts, index); tracer.set_sky(sky_small)
    tracer.resize(W, H)
    seen = {}
    try:
        for percent in (0, 30, 400):
            tracer.set_option("split_percent", percent)
            tracer.stats()                                                   # a fresh epoch for the counters
            tracer.render(p)
            g = tracer.read_result(); gid = tracer.read_hit_ids(); st = tracer.stats()
            assert (gid == cid).all(), percent
            _assert_same_image(g, c, f"split_percent {percent}")
            assert st["rays"] == cst["rays"]
            seen[percent] = (st["nodes_visited"], st["prims_tested"], st["bvh_nodes"], st["bvh_prims"])
        assert seen[0][3] == seen[30][3] == seen[400][3]                     # primitives, not references
        assert seen[30][2] > seen[0][2] and seen[400][2] > seen[30][2]       # more references, more nodes
        # (the eight-wide tree tests every primitive's own box in its parent: fewer visits to begin with, a smaller share to win)
        wide8 = tracer.variant.endswith("bvh8")
        assert seen[30][0] < (0.95 if wide8 else 0.85) * seen[0][0] and seen[30][1] < 0.6 * seen[0][1], seen
        assert seen[400][0] < (0.8 if wide8 else 0.7) * seen[0][0] and seen[400][1] < 0.2 * seen[0][1], seen
        tracer.set_option("split_percent", 30)
        tracer.stats()
        tracer.render(p); before = tracer.stats()["nodes_visited"]
        # the same matrices again: a refit, and every reference keeps its cut box
        tracer.update_transforms(objs)
        tracer.render(p)
        _assert_same_image(tracer.read_result(), c, "refit with unchanged matrices")
        assert tracer.stats()["nodes_visited"] == before
        # one beam moves
        moved = objs.copy()
        m = moved["m"][5].reshape(4, 4).T.copy(); m[:3, 3] += (1.5, -0.5, 2.0); moved["m"][5] = m.T.reshape(16)
        tracer.update_transforms(moved)
        tracer.render(p)
        g = tracer.read_result(); gid = tracer.read_hit_ids(); after = tracer.stats()["nodes_visited"]
        c2, cid2, _ = po.Oracle(f.spheres, moved, verts, index, sky=sky_small).render(p, W, H, threads=4)
        assert (gid == cid2).all()
        _assert_same_image(g, c2, "refit after a beam moved")
        assert after < 1.5 * before, (after, before)               # the other objects kept their cuts
        rng = np.random.default_rng(2)
        o = rng.uniform(-10, 10, (300, 3)).astype(np.float32); o[:, 1] = rng.uniform(0.5, 12, 300)
        d = rng.normal(size=(300, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
        od = np.concatenate([o, d], 1)
        ib, tb = tracer.trace_rays(od); ie, te = tracer.trace_rays(od, brute_force=True)
        assert (ib == ie).all() and (_bits(tb) == _bits(te)).all()
        tracer.set_option("refit", 0)
        tracer.update_transforms(moved)
        tracer.render(p)
        _assert_same_image(tracer.read_result(), g, "full build of the moved scene")
    finally:
        tracer.set_option("refit", 1); tracer.set_option("split_percent", 30)


def _trs(angle_deg, t, scale=1.0, axis=1, mirror=False):
    a = np.radians(angle_deg); c, s = np.cos(a), np.sin(a)
    r = np.eye(3)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    r[i, i] = c; r[i, j] = -s; r[j, i] = s; r[j, j] = c
    if mirror:
        r = r @ np.diag([-1.0, 1.0, 1.0])
    m = np.eye(4, dtype=np.float32); m[:3, :3] = (r * scale).astype(np.float32); m[:3, 3] = t
    return m.T.reshape(16).copy()                                    # UnityEngine.Matrix4x4 memory order


def test_instances_share_one_tree(tracer, sky_small):
    """Option "instancing": mesh objects with the same index range, placed by rotation / mirror / uniform scale /
    translation, keep ONE tree over the mesh's object-space triangles under a top-level tree of instance boxes; a sheared
    copy and a mesh used once stay flattened.  Image, hit ids and ray count equal the oracle's (which flattens, as the
    reference's shader does) and the flattened build's bit for bit - with per-object materials too; the tree is a
    fraction of the size; the exhaustive loop agrees on random rays; after objects move, the next frame still equals
    the oracle."""
    f = pa.Scene("fixture")
    mesh_v = np.array(f.vertices, np.float32).reshape(-1, 3); mesh_i = np.array(f.indices, np.int32)
    lo, hi = np.float32([-0.7, 0, -0.7]), np.float32([0.7, 1.4, 0.7])
    box_v = np.array([[(hi if (c >> k) & 1 else lo)[k] for k in range(3)] for c in range(8)], np.float32)
    box_i = np.array([q for a, b, c, d in [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)] for q in (a, b, c, a, c, d)], np.int32)
    verts = np.concatenate([mesh_v, box_v]); index = np.concatenate([mesh_i, box_i + len(mesh_v)])
    n_mesh, n_box = len(mesh_i), len(box_i)
    shear = _trs(0, (5.0, 1.6, 3.0)).reshape(4, 4).copy(); shear[1, 0] = 0.4           # x picks up y: not a similarity
    placements = [_trs(0, (-4.0, 1.5, 2.0)), _trs(70, (-1.0, 1.2, 4.0), 0.7), _trs(-130, (2.5, 2.0, 1.0), 1.3, axis=0),
                  _trs(25, (0.5, 1.4, -2.0), 1.0, axis=2, mirror=True), _trs(200, (4.0, 3.0, 5.0), 2.0), shear.reshape(16)]
    objs = np.zeros(len(placements) + 2, f.objects.dtype)
    for k, m in enumerate(placements):
        objs[k] = (m, 0, n_mesh)
    objs[-2] = (_trs(15, (-2.5, 0.0, -1.0)), n_mesh, n_box)                              # two small boxes: below the size for instancing
    objs[-1] = (_trs(-40, (3.0, 0.0, -1.5), 1.5), n_mesh, n_box)
    W, H = 128, 72
    cam = dict(eye=(0.5, 5.0, -11.0), target=(0.5, 1.5, 2.0), fov=60)
    p = pa.make_params(f, W, H, spp=2, max_bounces=5, seed=0.77, **cam)
    c, cid, cst = po.Oracle(f.spheres, objs, verts, index, sky=sky_small).render(p, W, H, threads=4)
    tracer.set_spheres(f.spheres); tracer.set_meshes(objs, verts, index); tracer.set_sky(sky_small)
    tracer.resize(W, H)
    rng = np.random.default_rng(6)
    o = rng.uniform(-8, 8, (400, 3)).astype(np.float32); o[:, 1] = rng.uniform(0.3, 8, 400)
    d = rng.normal(size=(400, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    od = np.concatenate([o, d], 1)
    try:
        nodes = {}
        for inst in (0, 1):
            tracer.set_option("instancing", inst)
            tracer.stats()
            tracer.render(p)
            g = tracer.read_result(); gid = tracer.read_hit_ids(); st = tracer.stats()
            assert (gid == cid).all(), inst
            _assert_same_image(g, c, f"instancing {inst}")
            assert st["rays"] == cst["rays"] and st["bvh_prims"] == len(f.spheres) + int(objs["count"].sum()) // 3
            nodes[inst] = st["bvh_nodes"]
            ib, tb = tracer.trace_rays(od); ie, te = tracer.trace_rays(od, brute_force=True)
            assert (ib == ie).all() and (_bits(tb) == _bits(te)).all(), inst
            if inst == 0:
                ids_flat, t_flat = ib, tb
            else:
                assert (ib == ids_flat).all() and (_bits(tb) == _bits(t_flat)).all()
        assert (cid > len(f.spheres)).mean() > 0.05
        if tracer.variant.endswith("bvh4"):
            assert nodes[1] < 0.6 * nodes[0], nodes                  # five of the six copies share one tree
        else:
            assert nodes[1] == nodes[0], nodes                       # the two-child tree is always flat
        # per-object materials reach instanced hits through the object index
        mats = pa.materials([((0.8, 0.3 + 0.1 * k, 0.2), 0.3 + 0.1 * k, (0.1, 0.2, 0.3), pa.PT_LOBE_GGX if k % 2 else pa.PT_LOBE_PHONG, (0.0, 0.0, 0.5 * (k == 2)))
                             for k in range(len(objs))])
        tracer.set_mesh_materials(mats)
        tracer.render(p)
        cm, cidm, _ = po.Oracle(f.spheres, objs, verts, index, sky=sky_small, materials=mats).render(p, W, H, threads=4)
        assert (tracer.read_hit_ids() == cidm).all()
        _assert_same_image(tracer.read_result(), cm, "instancing with materials")
        tracer.set_mesh_materials(None)
        # objects move (one of them into a shear: it leaves the instances)
        moved = objs.copy()
        moved["m"][1] = _trs(110, (-1.5, 1.6, 3.0), 0.9)
        moved["m"][4] = shear.reshape(16) + np.float32(0.0); moved["m"][4][12:15] = (4.0, 3.0, 5.0)
        moved["m"][0] = _trs(5, (-4.2, 1.5, 2.2))
        tracer.update_transforms(moved)
        tracer.render(p)
        c2, cid2, _ = po.Oracle(f.spheres, moved, verts, index, sky=sky_small).render(p, W, H, threads=4)
        assert (tracer.read_hit_ids() == cid2).all()
        _assert_same_image(tracer.read_result(), c2, "instances moved")
        # _Depth output goes through the same two-level walk
        pd = pa.make_params(f, W, H, spp=1, max_bounces=1, seed=0.77, depth=1, **cam)
        tracer.render(pd)
        cd, cidd, _ = po.Oracle(f.spheres, moved, verts, index, sky=sky_small).render(pd, W, H, threads=4)
        assert (tracer.read_hit_ids() == cidd).all()
        _assert_same_image(tracer.read_result(), cd, "instances, _Depth")
    finally:
        tracer.set_option("instancing", 0); tracer.set_mesh_materials(None)


def test_wavefront_stages_match_golden_vectors_and_oracle(sky_small):
    """The separate trace / shade stages with compaction and regeneration (pt_set_option("backend", 1)): the committed golden
    vectors, and the 70 k-triangle mesh against the oracle."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fixture_64x36.npz"))
    with pa.PathTracer(0) as t:
        t.set_option("backend", 1)
        s = pa.Scene("fixture")
        t.load_scene(s, pa.procedural_sky(128, 64))
        W, H = 64, 36
        t.resize(W, H)
        for name, kw in {"spp1_b4": dict(spp=1, max_bounces=4), "spp16_b8": dict(spp=16, max_bounces=8)}.items():
            t.stats()
            t.render(pa.make_params(s, W, H, seed=0.5, **kw))
            _assert_same_image(t.read_result(), g[name], "wavefront " + name)
            assert (t.read_hit_ids() == g["ids"]).all()
            st = t.stats()
            assert [st["rays"], st["paths"]] == [g[name + "_rays"][0], g[name + "_rays"][2]]
        for width in (4, 2):
            t.set_option("bvh_width", width)
            m = pa.Scene("blob70k")
            t.load_scene(m, sky_small)
            W, H = 128, 72
            t.resize(W, H)
            p = pa.make_params(m, W, H, seed=0.5, spp=3, max_bounces=8)
            t.stats(); t.render(p)
            c, cid, cst = po.Oracle.from_scene(m, sky_small, accel=True).render(p, W, H, threads=8)
            assert (t.read_hit_ids() == cid).all()
            _assert_same_image(t.read_result(), c, f"wavefront blob70k width {width}")
            assert t.stats()["rays"] == cst["rays"]


def test_instances_are_recognised_in_buffers_laid_out_as_the_reference_lays_them_out(sky_small):
    """RebuildMeshObjectBuffers (RayTracingMaster.cs:387-408) appends, per object, a copy of its mesh's vertices and the
    mesh's indices offset by that copy's first vertex: objects that share a mesh arrive with DIFFERENT index ranges of
    equal content.  With option "instancing" they must still share one tree (recognised by content), the image must not
    change, and a copy that differs in one position or one index must stay out of the group."""
    f = pa.Scene("fixture")
    mesh_v = np.array(f.vertices, np.float32).reshape(-1, 3); mesh_i = np.array(f.indices, np.int32)
    placements = [_trs(0, (-4.0, 1.5, 2.0)), _trs(70, (-1.0, 1.2, 4.0), 0.7), _trs(-130, (2.5, 2.0, 1.0), 1.3, axis=0),
                  _trs(25, (0.5, 1.4, -2.0), 1.0, axis=2, mirror=True), _trs(200, (4.0, 3.0, 5.0), 2.0), _trs(40, (-3.0, 3.5, 6.0), 1.1)]

    def reference_layout(nudged=None, reindexed=None):
        verts, index, objs = [], [], np.zeros(len(placements), f.objects.dtype)
        n_v = n_i = 0
        for k, m in enumerate(placements):
            v = mesh_v.copy(); i = mesh_i.copy()
            if k == nudged:
                v[17, 1] += 0.25                                    # one position differs: another mesh
            if k == reindexed:
                i[0:3] = i[[0, 2, 1]]                               # one triangle wound the other way: another mesh
            verts.append(v); index.append(i + n_v)                  # indices.Select(index => index + firstVertex)
            objs[k] = (m, n_i, len(i))
            n_v += len(v); n_i += len(i)
        return objs, np.concatenate(verts), np.concatenate(index)

    W, H = 128, 72
    cam = dict(eye=(0.5, 5.0, -11.0), target=(0.5, 1.5, 2.0), fov=60)
    p = pa.make_params(f, W, H, spp=2, max_bounces=5, seed=0.77, **cam)
    nodes = {}
    with pa.PathTracer(0) as t:
        t.set_sky(sky_small); t.resize(W, H); t.set_spheres(f.spheres)
        for label, kw in (("copies", {}), ("one nudged", dict(nudged=2)), ("one reindexed", dict(reindexed=4))):
            objs, verts, index = reference_layout(**kw)
            assert len({(int(o["offset"]), int(o["count"])) for o in objs}) == len(objs)            # no two objects share a range
            c, cid, cst = po.Oracle(f.spheres, objs, verts, index, sky=sky_small).render(p, W, H, threads=4)
            for inst in (0, 1, 2):
                t.set_option("instancing", inst)
                t.set_meshes(objs, verts, index)
                t.stats(); t.render(p)
                st = t.stats()
                assert (t.read_hit_ids() == cid).all(), (label, inst)
                _assert_same_image(t.read_result(), c, f"{label}, instancing {inst}")
                assert st["rays"] == cst["rays"]
                nodes[label, inst] = st["bvh_nodes"]
        assert (cid > len(f.spheres)).mean() > 0.05
        # instancing 2 - the shared tree copied and refitted per object on the device - is an ordinary flat tree: the exhaustive
        # loop agrees with it on random rays, the other kernels walk it, and it follows the objects when they move
        rng = np.random.default_rng(6)
        o = rng.uniform(-8, 8, (2000, 3)).astype(np.float32); o[:, 1] = rng.uniform(0.3, 8, 2000)
        d = rng.normal(size=(2000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
        od = np.concatenate([o, d], 1)
        objs, verts, index = reference_layout()
        t.set_option("instancing", 2); t.set_meshes(objs, verts, index)
        ib, tb = t.trace_rays(od); ie, te = t.trace_rays(od, brute_force=True)
        assert (ib == ie).all() and (_bits(tb) == _bits(te)).all() and (ib > len(f.spheres)).mean() > 0.02
        c, cid, cst = po.Oracle(f.spheres, objs, verts, index, sky=sky_small).render(p, W, H, threads=4)
        for backend in (2, 1, 0):
            t.set_option("backend", backend); t.render(p)
            assert (t.read_hit_ids() == cid).all(), backend
            _assert_same_image(t.read_result(), c, f"copies on the device, backend {backend}")
        moved = objs.copy()
        moved["m"][1] = _trs(110, (-1.5, 1.6, 3.0), 0.9); moved["m"][3] = _trs(-60, (3.0, 2.2, -1.0), 1.4, axis=0)
        sheared = _trs(0, (5.0, 1.6, 3.0)).reshape(4, 4).copy(); sheared[1, 0] = 0.4          # not a similarity: fine for copies in world space
        moved["m"][5] = sheared.reshape(16)
        t.update_transforms(moved); t.render(p)
        c2, cid2, _ = po.Oracle(f.spheres, moved, verts, index, sky=sky_small).render(p, W, H, threads=4)
        assert (t.read_hit_ids() == cid2).all()
        _assert_same_image(t.read_result(), c2, "copies on the device, objects moved")
    assert nodes["copies", 1] < 0.35 * nodes["copies", 0], nodes                  # six copies, one tree
    assert nodes["copies", 1] < nodes["one nudged", 1] < nodes["copies", 0], nodes            # five share a tree, the nudged copy is flattened
    assert nodes["copies", 1] < nodes["one reindexed", 1] < nodes["copies", 0], nodes
    assert 5 * nodes["copies", 1] < nodes["copies", 2] < 1.5 * nodes["copies", 0], nodes       # a flat tree again, made of six copies of the shared one


def test_shared_meshes_are_recognised_across_the_threads_of_the_search(sky_small):
    """The search for shared meshes hashes and compares large ranges in pieces on several host threads.  Six copies of the
    70 k-triangle mesh laid out as the reference lays them out: all six share one tree; a copy that differs in its LAST
    vertex, or in its last triangle's winding - differences only the last piece of a range sees - stays out of the group;
    the image is that of the flattened scene every time."""
    b = pa.Scene("blob70k")
    o0 = b.objects[0]
    idx = np.array(b.indices, np.int32)[int(o0["offset"]):int(o0["offset"]) + int(o0["count"])]
    lo, hi = int(idx.min()), int(idx.max()) + 1
    mesh_v = np.array(b.vertices, np.float32).reshape(-1, 3)[lo:hi]; mesh_i = idx - lo
    assert len(mesh_i) > 3 * 65536                                  # more than one piece per range
    placements = [_trs(20 * k, (3.0 * (k % 3) - 3.0, 1.5, 4.0 * (k // 3)), 0.5 + 0.1 * k) for k in range(6)]

    def layout(nudged=None, reindexed=None):
        verts, index, objs = [], [], np.zeros(len(placements), b.objects.dtype)
        n_v = n_i = 0
        for k, m in enumerate(placements):
            v = mesh_v.copy(); i = mesh_i.copy()
            if k == nudged:
                v[int(i[-1]), 1] += 0.25
            if k == reindexed:
                i[-3:] = i[[-3, -1, -2]]
            verts.append(v); index.append(i + n_v)
            objs[k] = (m, n_i, len(i))
            n_v += len(v); n_i += len(i)
        return objs, np.concatenate(verts), np.concatenate(index).astype(np.int32)

    W, H = 160, 90
    p = pa.make_params(b, W, H, spp=1, max_bounces=3, seed=0.31, eye=(0.0, 4.0, -9.0), target=(0.0, 1.5, 2.0), fov=60)
    nodes = {}
    with pa.PathTracer(0) as t:
        t.set_sky(sky_small); t.resize(W, H); t.set_spheres(b.spheres)
        t.set_option("tree_build", 0)                               # the host builder: instancing decides what it builds
        for label, kw in (("copies", {}), ("last vertex nudged", dict(nudged=4)), ("last triangle turned", dict(reindexed=2))):
            objs, verts, index = layout(**kw)
            images = {}
            for inst in (0, 1, 2):
                t.set_option("instancing", inst)
                t.set_meshes(objs, verts, index)
                t.stats(); t.render(p)
                nodes[label, inst] = t.stats()["bvh_nodes"]
                images[inst] = (t.read_result(), t.read_hit_ids())
            for inst in (1, 2):
                assert (images[inst][1] == images[0][1]).all(), (label, inst)
                _assert_same_image(images[inst][0], images[0][0], f"{label}, instancing {inst}")
            assert (images[0][1] > len(b.spheres)).mean() > 0.05
    assert nodes["copies", 1] < 0.25 * nodes["copies", 0], nodes                           # six copies, one tree
    for label in ("last vertex nudged", "last triangle turned"):                          # five share a tree, the sixth is flattened
        assert 1.5 * nodes["copies", 1] < nodes[label, 1] < 0.5 * nodes["copies", 0], (label, nodes)
