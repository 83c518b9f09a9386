"""Regression cases found by the randomised soaks (tools/soak_*.py), replayed as fixed sequences."""
import ctypes as C

import numpy as np
import pytest

import pathtracing_amd as pa
from oracle import pt_oracle as po
from pathtracing_amd import _ffi

pytestmark = pytest.mark.gpu


def _same(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def test_master_mirror_keeps_its_targets_when_the_size_does_not_change():
    """Three frames a run of tools/soak_master.py recorded as a mismatch (camera below the ground plane, renderMode 4
    with _Depth, renderMode 4, renderMode 3; 57x22, then 35x39 twice).  The third frame followed a "resize" to the size
    the targets already had: InitRenderTexture (RayTracingMaster.cs:545) then keeps the targets, so renderMode 3's
    prologue copy (RayTracingShader.compute:728-731) puts the second frame into Result1.  The soak's model had made fresh,
    zeroed targets instead - the disagreement was the model's, not the library's or the mirror's.  Here the model keeps
    its targets and all three targets of all three frames must be identical."""
    lib = _ffi.ptmaster(); pt = _ffi.ptrace()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    try:
        s = pa.Scene("fixture")
        assert lib.ptm_master_load_scene(h, s.handle, 64, 32) == 0, lib.ptm_master_error(h)
        orc = po.Oracle.from_scene(s, pa.procedural_sky(64, 32), accel=True)
        ctx = lib.ptm_master_context(h)
        lib.ptm_master_set_divisions(h, 3)
        # (W, H, renderMode, _Depth, spp, bounces, _AccumulationFrames, eye); two frames first so that _Counter is 2, 0, 1
        frames = [(57, 22, 2, 0, 1, 2, 4, (5.9, -14.2, 8.1)), (57, 22, 4, 0, 1, 2, 4, (5.9, -14.25, 8.15)),
                  (57, 22, 4, 1, 1, 2, 2, (5.838001251220703, -14.280645370483398, 8.194798469543457)),
                  (35, 39, 4, 0, 1, 2, 8, (5.768211364746094, -14.37048053741455, 8.319742202758789)),
                  (35, 39, 3, 0, 2, 1, 6, (5.678927898406982, -14.240138053894043, 8.398101806640625))]
        size = None; old = []; cams = [pa.PtCameraHistory() for _ in range(100)]
        counters = []
        for k, (W, H, mode, depth, spp, maxb, accf, eye) in enumerate(frames):
            if (W, H) != size:
                size = (W, H)
                res = np.zeros((H, W, 4), np.float32); res1 = np.zeros_like(res); res2 = np.zeros_like(res)
            lib.ptm_master_configure(h, W, H, spp, maxb, depth, pa.PT_ACCUM_NONE, accf)      # every frame, as the soak does
            lib.ptm_master_set_render_mode(h, mode)
            lib.ptm_master_set_camera(h, (C.c_float * 3)(*eye), (C.c_float * 3)(*s.target), C.c_float(s.fov), C.c_float(0.0))
            p = pa.PtFrameParams()
            lib.ptm_master_peek_params(h, C.byref(p))
            counters.append(p.counter)
            assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
            cams[:len(old)] = old
            orc.render_history(p, cams, W, H, res, res1, threads=4, result2=res2)
            old = [pa.camera_history_entry(eye, tuple(s.target), s.fov, W / H)]              # modes 2 .. 4: the list restarts every frame
            got = [np.empty((H, W, 4), np.float32) for _ in range(3)]
            assert pt.pt_read_result(ctx, got[0].ctypes.data) == 0 and pt.pt_read_converged(ctx, got[1].ctypes.data) == 0
            assert pt.pt_read_target(ctx, pa.PT_TARGET_DENOISED, got[2].ctypes.data) == 0
            for name, a, b in (("Result", got[0], res), ("Result1", got[1], res1), ("Result2", got[2], res2)):
                assert _same(a, b), f"frame {k}: {name} differs"
        assert counters == [0, 1, 2, 0, 1]
        assert res1[..., 3].min() > 10.0        # the history of the last frame is the frame before it, not zeros
    finally:
        lib.ptm_master_destroy(h)


def test_tree_that_could_overflow_the_stack_is_walked_as_a_two_child_tree(monkeypatch):
    """A four-wide tree stacks up to three siblings per level; when its worst case (BvhResult::stack_need4) exceeds what a
    lane's stack holds, the library walks the two-child tree of the same build instead.  PTRACE_STACK_LIMIT = 20 makes the
    70 k-triangle scene (worst case 33) such a tree: same image and hits as the oracle, statistics of the binary tree."""
    s = pa.Scene("blob70k"); sky = pa.procedural_sky(64, 32)
    W, H = 96, 54
    p = pa.make_params(s, W, H, seed=0.5, spp=2, max_bounces=6)
    want, want_ids, _ = po.Oracle.from_scene(s, sky, accel=True).render(p, W, H, threads=8)
    nodes = {}
    for limit in (None, "20"):
        if limit:
            monkeypatch.setenv("PTRACE_STACK_LIMIT", limit)
        t = pa.PathTracer(0)
        try:
            t.load_scene(s, sky); t.resize(W, H); t.render(p)
            assert _same(t.read_result(), want) and (t.read_hit_ids() == want_ids).all()
            rng = np.random.default_rng(5)
            od = np.zeros((3000, 6), np.float32); od[:, :3] = rng.uniform(-4, 4, (3000, 3)); od[:, 1] = np.abs(od[:, 1]) + 3
            d = rng.normal(0, 1, (3000, 3)); d[:, 1] = -np.abs(d[:, 1]); od[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
            ib, tb = t.trace_rays(od); ie, te = t.trace_rays(od, brute_force=True)
            assert (ib == ie).all() and (tb.view(np.uint32) == te.view(np.uint32)).all()
            nodes[limit] = t.stats()["bvh_nodes"]
        finally:
            t.close()
    assert nodes["20"] > 1.5 * nodes[None]          # the two-child tree has about twice the nodes


def test_eight_wide_tree_too_deep_for_the_group_stack_is_walked_as_a_two_child_tree(monkeypatch):
    """Under option bvh_width 8 a ray holds one group per level plus one; a tree deeper than a lane's stack of groups (a
    degenerate chain; here PTRACE_STACK_LIMIT = 8 against the 70 k-triangle scene's 11) is walked as the two-child tree of
    the same build, whose records follow the other order.  Same image and hits either way."""
    s = pa.Scene("blob70k"); sky = pa.procedural_sky(64, 32)
    W, H = 96, 54
    p = pa.make_params(s, W, H, seed=0.5, spp=2, max_bounces=6)
    want, want_ids, _ = po.Oracle.from_scene(s, sky, accel=True).render(p, W, H, threads=8)
    nodes = {}
    for limit in (None, "8"):
        if limit:
            monkeypatch.setenv("PTRACE_STACK_LIMIT", limit)
        t = pa.PathTracer(0)
        try:
            t.set_option("bvh_width", 8)
            t.load_scene(s, sky); t.resize(W, H); t.render(p)
            assert _same(t.read_result(), want) and (t.read_hit_ids() == want_ids).all()
            rng = np.random.default_rng(5)
            od = np.zeros((3000, 6), np.float32); od[:, :3] = rng.uniform(-4, 4, (3000, 3)); od[:, 1] = np.abs(od[:, 1]) + 3
            d = rng.normal(0, 1, (3000, 3)); d[:, 1] = -np.abs(d[:, 1]); od[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
            ib, tb = t.trace_rays(od); ie, te = t.trace_rays(od, brute_force=True)
            assert (ib == ie).all() and (tb.view(np.uint32) == te.view(np.uint32)).all()
            nodes[limit] = t.stats()["bvh_nodes"]
            # and the other widths from the same context: the records change their order with the option
            for width in (4, 8):
                t.set_option("bvh_width", width); t.render(p)
                assert _same(t.read_result(), want) and (t.read_hit_ids() == want_ids).all(), width
        finally:
            t.close()
    assert nodes["8"] > 2.5 * nodes[None]           # one primitive per leaf: a two-child tree of 70 k nodes against 23 k eight-wide ones


@pytest.mark.parametrize("left", [1, 0])
def test_master_mirror_stereo_branch(left):
    """The stereo branch of SetShaderParameters (RayTracingMaster.cs:459-485) through the C++ mirror, one master per eye
    as the reference has one RayTracingMaster per eye camera: the eye's own view and (asymmetric) projection matrices,
    inverted, become _CameraToWorld / _CameraInverseProjection; _leftEye is uploaded; and the camera lists follow the
    stereo rule - kept only in renderMode 5 / 6 and only across the frame that starts a cycle (counter % divisions != 0
    clears them, :473), where the mono branch clears at == 0.  renderMode 2, 4 (whose reprojection reads _leftEye) and 5
    with a moving head; all targets against the oracle fed the uploaded parameters and a restatement of the list rule."""
    lib = _ffi.ptmaster(); pt = _ffi.ptrace()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    try:
        s = pa.Scene("xr")
        W, H, D = 86, 92, 3                                  # the headset's 2064 x 2208 eye, scaled down
        assert lib.ptm_master_load_scene(h, s.handle, 64, 32) == 0, lib.ptm_master_error(h)
        orc = po.Oracle.from_scene(s, pa.procedural_sky(64, 32), accel=True)
        ctx = lib.ptm_master_context(h)
        lib.ptm_master_set_divisions(h, D)
        res = np.zeros((H, W, 4), np.float32); res1 = np.zeros_like(res); res2 = np.zeros_like(res)
        cams = [pa.PtCameraHistory() for _ in range(100)]; old = []
        modes = [2, 2, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 2]
        kept = []
        for k, mode in enumerate(modes):
            head = s.eye + np.array([0.05 * k, 0.01 * k, 0.08 * k], np.float32)
            view, proj = pa.stereo_eye_matrices(head, s.target, 90.0, W / H, -0.011 if left else 0.011, 0.07 if left else -0.07)
            lib.ptm_master_configure(h, W, H, 1, 4, 0, pa.PT_ACCUM_NONE, 4)
            lib.ptm_master_set_render_mode(h, mode)
            assert lib.ptm_master_set_stereo(h, 1, left, view.ctypes.data, proj.ctypes.data) == 0
            p = pa.PtFrameParams()
            lib.ptm_master_peek_params(h, C.byref(p))
            assert p.left_eye == left and p.counter == k % D and p.render_mode == mode
            assert np.allclose(np.array(p.camera_to_world), pa.matrix_inverse(view), atol=0) and \
                np.allclose(np.array(p.camera_inverse_projection), pa.matrix_inverse(proj), atol=0)
            assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
            cams[:len(old)] = old
            orc.render_history(p, cams, W, H, res, res1, threads=8, result2=res2)
            kept.append(len(old))
            if mode not in (5, 6) or p.counter != 0:          # :473-478
                old = []
            if mode < 11 and mode != 1:                       # :480-485
                old = old + [pa.stereo_history_entry(view, proj)]
            got = [np.empty((H, W, 4), np.float32) for _ in range(3)]
            assert pt.pt_read_result(ctx, got[0].ctypes.data) == 0 and pt.pt_read_converged(ctx, got[1].ctypes.data) == 0
            assert pt.pt_read_target(ctx, pa.PT_TARGET_DENOISED, got[2].ctypes.data) == 0
            for name, a, b in (("Result", got[0], res), ("Result1", got[1], res1), ("Result2", got[2], res2)):
                assert _same(a, b), f"frame {k} (renderMode {mode}, {'left' if left else 'right'} eye): {name} differs"
        # lists uploaded per frame: one camera in modes 2 / 4; in mode 5 two entries only on the frame after a cycle's start
        assert kept == [0, 1, 1, 1, 1, 1, 1, 2, 1, 1, 2, 1, 1]
        assert np.isfinite(res).all() and res[..., :3].max() > 0
    finally:
        lib.ptm_master_destroy(h)


def test_master_mirror_objects_come_and_go():
    """RegisterObject / UnregisterObject / UpdateObject after the first frame (RayTracingMaster.cs:296-313, what
    RayTracingObject.cs:24-38 calls when a GameObject is enabled, disabled or moved): the next Render() rebuilds the mesh
    buffers (:371-413) and the frame equals the oracle's render of the new object list - hit ids included, which number
    the triangles in list order."""
    lib = _ffi.ptmaster(); pt = _ffi.ptrace()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    try:
        s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
        W, H = 96, 54
        assert lib.ptm_master_load_scene(h, s.handle, 64, 32) == 0, lib.ptm_master_error(h)
        lib.ptm_master_configure(h, W, H, 2, 4, 0, pa.PT_ACCUM_NONE, 4)
        lib.ptm_master_set_render_mode(h, 1)
        ctx = lib.ptm_master_context(h)
        # the model's object list: (vertices, indices, matrix) per object, flattened the way RebuildMeshObjectBuffers does
        objs = []
        for o in s.objects:
            idx = s.indices[o["offset"]:o["offset"] + o["count"]]
            lo = int(idx.min())
            objs.append((s.vertices[lo:int(idx.max()) + 1].copy(), (idx - lo).astype(np.int32), o["m"].copy()))
        assert lib.ptm_master_object_count(h) == len(objs)

        def check(what):
            verts, index, table = [], [], np.zeros(len(objs), s.objects.dtype)
            nv = ni = 0
            for k, (v, i, m) in enumerate(objs):
                verts.append(v); index.append(i + nv); table[k] = (m, ni, len(i)); nv += len(v); ni += len(i)
            verts = np.concatenate(verts) if verts else np.zeros((0, 3), np.float32)
            index = np.concatenate(index).astype(np.int32) if index else np.zeros(0, np.int32)
            p = pa.PtFrameParams()
            lib.ptm_master_peek_params(h, C.byref(p))
            assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
            got = np.empty((H, W, 4), np.float32); ids = np.empty((H, W), np.int32)
            assert pt.pt_read_result(ctx, got.ctypes.data) == 0 and pt.pt_read_hit_ids(ctx, ids.ctypes.data) == 0
            want, want_ids, _ = po.Oracle(s.spheres, table, verts, index, sky=sky, accel=False).render(p, W, H, threads=8)
            assert (ids == want_ids).all() and _same(got, want), what
            return ids

        ids0 = check("the scene as loaded")
        # a new object appears: a tetrahedron in front of the camera
        tv = np.array([[0, 0, 0], [1.5, 0, 0], [0, 1.5, 0], [0, 0, 1.5]], np.float32)
        ti = np.array([0, 2, 1, 0, 1, 3, 0, 3, 2, 1, 2, 3], np.int32)
        m = np.eye(4, dtype=np.float32); m[:3, 3] = (-0.5, 0.6, -2.0)
        m = np.ascontiguousarray(m.T.reshape(16))
        k = lib.ptm_master_register_object(h, tv.ctypes.data, len(tv), ti.ctypes.data, len(ti), m.ctypes.data)
        assert k == len(objs)
        objs.append((tv, ti, m))
        ids1 = check("after RegisterObject")
        n_before = 1 + len(s.spheres) + sum(len(i) for _, i, _ in objs[:-1]) // 3
        assert (ids1 >= n_before).any() and not (ids0 >= n_before).any()          # the new triangles are seen, with the last ids
        # it moves (a transform change: matrices only), and a second object sharing its mesh appears
        m2 = m.copy(); m2[12:15] += np.array([0.4, 0.2, 0.3], np.float32)
        assert lib.ptm_master_update_object(h, k, m2.ctypes.data) == 0
        objs[k] = (tv, ti, m2)
        check("after UpdateObject")
        m3 = m.copy(); m3[12:15] += np.array([-1.2, 0.1, 0.5], np.float32)
        assert lib.ptm_master_register_instance(h, k, m3.ctypes.data) == k + 1
        objs.append((tv, ti, m3))
        check("after RegisterInstance")
        # the first object of the scene goes away: later objects move up and their triangles get lower ids
        assert lib.ptm_master_unregister_object(h, 0) == 0
        del objs[0]
        assert lib.ptm_master_object_count(h) == len(objs)
        check("after UnregisterObject")
        assert lib.ptm_master_unregister_object(h, 99) != 0
    finally:
        lib.ptm_master_destroy(h)


def test_master_mirror_render_scale():
    """GetRenderScale (RayTracingMaster.cs:585-597): the render size is Mathf.RoundToInt(screen size * renderScale) - halves
    go to the even neighbour - and the frame is rendered at that size."""
    lib = _ffi.ptmaster(); pt = _ffi.ptrace()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    try:
        s = pa.Scene("fixture")
        assert lib.ptm_master_load_scene(h, s.handle, 64, 32) == 0, lib.ptm_master_error(h)
        lib.ptm_master_configure(h, 1, 1, 1, 4, 0, pa.PT_ACCUM_NONE, 4)
        size = (C.c_int32 * 2)()
        for screen, scale, want in (((64, 36), 1.0, (64, 36)), ((65, 37), 0.5, (32, 18)), ((63, 35), 0.5, (32, 18)), ((64, 36), 1.5, (96, 54))):
            assert lib.ptm_master_set_render_scale(h, screen[0], screen[1], scale) == 0
            lib.ptm_master_render_size(h, size)
            assert tuple(size) == want, (screen, scale, tuple(size))
        W, H = want
        p = pa.PtFrameParams()
        lib.ptm_master_peek_params(h, C.byref(p))
        assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
        got = np.empty((H, W, 4), np.float32)
        assert pt.pt_read_result(lib.ptm_master_context(h), got.ctypes.data) == 0
        c, _, _ = po.Oracle.from_scene(s, pa.procedural_sky(64, 32), accel=False).render(p, W, H, threads=4)
        assert _same(got, c)
        # the history targets of the reference keep the screen size at such a scale (RayTracingMaster.cs:563-571): the modes that
        # read them are refused by name, and accepted again at a scale that gives the render size back
        lib.ptm_master_set_render_mode(h, 2)
        assert lib.ptm_master_render(h) == pa.PT_ERR_UNSUPPORTED and b"renderScale" in lib.ptm_master_error(h)
        assert lib.ptm_master_set_render_scale(h, 64, 36, 1.0) == 0
        assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
        assert lib.ptm_master_set_render_scale(h, 64, 36, 0.0) != 0
    finally:
        lib.ptm_master_destroy(h)


def test_master_mirror_skybox_from_a_radiance_picture(tmp_path):
    """SkyboxTexture (RayTracingMaster.cs:25, bound at :457) from a .hdr file, the format of the reference's asset: the
    frame equals the oracle's with the decoded picture as its sky, and differs from the frame under the procedural sky."""
    rng = np.random.default_rng(12)
    H0, W0 = 32, 64
    rgbe = np.empty((H0, W0, 4), np.uint8)
    rgbe[:, :, :3] = rng.integers(40, 256, (H0, W0, 3))
    rgbe[:, :, 3] = rng.integers(126, 131, (H0, W0))
    path = tmp_path / "sky.hdr"
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {H0} +X {W0}\n".encode() + rgbe.tobytes())     # flat scanlines
    sky = pa.load_hdr(path)
    assert sky.shape == (H0, W0, 4) and sky[..., :3].max() > 2.0
    lib = _ffi.ptmaster(); pt = _ffi.ptrace()
    h = C.c_void_p()
    assert lib.ptm_master_create(0, C.byref(h)) == 0, lib.ptm_master_error(h)
    try:
        s = pa.Scene("fixture")
        assert lib.ptm_master_load_scene(h, s.handle, 64, 32) == 0, lib.ptm_master_error(h)
        lib.ptm_master_configure(h, 1, 1, 2, 4, 0, pa.PT_ACCUM_NONE, 4)
        W, H = 64, 36
        assert lib.ptm_master_set_render_scale(h, W, H, 1.0) == 0
        frames = []
        for use_file in (False, True):
            if use_file:
                assert lib.ptm_master_set_sky_hdr(h, str(path).encode(), 2048) == 0, lib.ptm_master_error(h)
            p = pa.PtFrameParams()
            lib.ptm_master_peek_params(h, C.byref(p))
            assert lib.ptm_master_render(h) == 0, lib.ptm_master_error(h)
            got = np.empty((H, W, 4), np.float32)
            assert pt.pt_read_result(lib.ptm_master_context(h), got.ctypes.data) == 0
            c, _, _ = po.Oracle.from_scene(s, sky if use_file else pa.procedural_sky(64, 32), accel=False).render(p, W, H, threads=4)
            assert _same(got, c), use_file
            frames.append(got)
        assert not _same(frames[0], frames[1])
        assert lib.ptm_master_set_sky_hdr(h, str(tmp_path / "absent.hdr").encode(), 2048) != 0
        assert b"cannot open" in lib.ptm_master_error(h)
    finally:
        lib.ptm_master_destroy(h)


def test_new_targets_are_zero_before_the_next_frame_reads_them():
    """tools/soak_master.py 150 4242, frame 5162: the first PlanerPause frame after a change of size showed colours of
    frames long gone.  pt_resize cleared its new targets with hipMemset - a fill on the null stream that returns before
    it has run - and the context's stream is non-blocking, so the next frame's kernels could read the recycled memory
    between two of the fills.  Here, a few thousand times: targets full of sevens, a change of size (a handful of small
    sizes in turn, so that the allocator hands the old blocks out again), at once a renderMode 2 frame, whose prologue
    copies Result to Result1 (RayTracingShader.compute:728-731) - Result1 must then be zero everywhere, and the frame,
    which blends in what the history shows, must not contain a seven."""
    from reproj_util import frame_params
    s = pa.Scene("fixture"); sky = pa.procedural_sky(64, 32)
    sizes = [(57, 40), (47, 15), (64, 36), (35, 39), (40, 24)]
    with pa.PathTracer(0) as t:
        t.load_scene(s, sky)
        for it in range(3000):
            W, H = sizes[it % len(sizes)]
            t.resize(W, H)
            p, h = frame_params(s, W, H, it % 7, 1, 2, 5, mode=2)
            t.set_camera_history([h])
            t.render(p)
            conv = t.read_converged()
            assert not conv.any(), f"iteration {it}: Result1 holds {int((conv != 0).any(-1).sum())} pixels of recycled memory"
            res = t.read_result()
            assert np.isfinite(res).all() and not (res == 7.0).any(), f"iteration {it}: the frame shows recycled memory"
            sevens = np.full((H, W, 4), 7.0, np.float32)
            for which in (pa.PT_TARGET_RESULT, pa.PT_TARGET_CONVERGED, pa.PT_TARGET_DENOISED):
                t.write_target(which, sevens)


@pytest.mark.parametrize("width", [4, 2, 8])
def test_spheres_seen_from_far_out_on_the_ground_plane(width):
    """tools/soak_params.py 200 20261111, dispatch 153 608: a camera 40 above the ground with a 175 degree field of view
    sees the plane 4 215 units away; a path that bounces there looks back at the spheres from a distance at which
    IntersectSphere's difference of squares (RayTracingShader.compute:170-190) accepts rays passing units beside a sphere -
    the exhaustive loop reports such a hit, the tree, whose sphere boxes are padded for origins near the scene, did not.
    Rays that start beyond that reach now meet the spheres directly (intersect_far_spheres)."""
    s = pa.Scene("xr"); sky = pa.procedural_sky(64, 32)
    W, H = 64, 5
    p = pa.make_params(s, W, H, seed=0.6286261677742004, spp=3, max_bounces=4, fov=175.0,
                       eye=(-4.565361642855158, 40.0, -6.131743939305356), target=(0.8743084992982872, -1.919203517003022, -1.7489561927078008))
    c, cid, cst = po.Oracle.from_scene(s, sky, accel=False).render(p, W, H, threads=4)
    assert c[2, 12, 3] > 4000.0                                   # the pixel whose path comes back from far out
    with pa.PathTracer(0) as t:
        t.set_option("bvh_width", width)
        for backend in (0, 1, 2):
            t.set_option("backend", backend)
            t.load_scene(s, sky); t.resize(W, H)
            t.render(p)
            g = t.read_result()
            assert (t.read_hit_ids() == cid).all() and _same(g, c), (width, backend, np.argwhere(~_same_px(g, c))[:4])
        # the ray queries take the same route
        rng = np.random.default_rng(3)
        o = np.zeros((2000, 3), np.float32); o[:, 0] = rng.uniform(500, 6000, 2000) * rng.choice([-1, 1], 2000); o[:, 1] = rng.uniform(0.001, 3.0, 2000)
        o[:, 2] = rng.uniform(-3000, 3000, 2000)
        aim = rng.uniform(-6, 6, (2000, 3)).astype(np.float32); aim[:, 1] = rng.uniform(0.2, 4, 2000)
        d = aim - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
        od = np.concatenate([o, d.astype(np.float32)], 1)
        ib, tb = t.trace_rays(od); ie, te = t.trace_rays(od, brute_force=True)
        assert (ib == ie).all() and (tb.view(np.uint32) == te.view(np.uint32)).all()
        assert (ib > 0).mean() > 0.02                                 # some of them do find a sphere or the cube


def _same_px(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(-1)


def test_frame_whose_camera_ray_meets_the_ground_plane_39_450_units_out(tracer):
    """A frame a run of tools/soak_session.py recorded as a mismatch (seed 20261112, frame 90 767: scene xr, 48x47,
    renderMode 2, 2 spp, 2 bounces, after 8 964 refits).  Pixel (28, 27) looks along the ground plane and meets it 39 450
    units out; the bounce from there comes back past a sphere, several units beside it, and IntersectSphere's difference
    of squares (RayTracingShader.compute:170-190) calls that a hit.  The exhaustive loop - the reference - shows it and so
    did every kernel; the oracle's own tree had pruned it.  The oracle now tests every sphere for every ray; here the
    frame must be the same on the device, in the exhaustive oracle and in the accelerated one."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "session_far_plane_frame.npz"))
    W, H = int(z["W"]), int(z["H"])
    p = pa.PtFrameParams.from_buffer_copy(z["params"].tobytes())
    sz = C.sizeof(pa.PtCameraHistory)
    old = [pa.PtCameraHistory.from_buffer_copy(z["cams"].tobytes()[i * sz:(i + 1) * sz]) for i in range(int(z["n_cams"]))]
    cams = old + [pa.PtCameraHistory() for _ in range(100 - len(old))]
    scene = pa.Scene(str(z["scene"])); sky = pa.procedural_sky(64, 32)
    objs = z["objs"]
    images = []
    for accel in (False, True):
        orc = po.Oracle(scene.spheres, objs, scene.vertices, scene.indices, sky=sky, accel=accel)
        res, res1, res2 = [a.copy() for a in z["before"]]
        orc.render_history(p, cams, W, H, res, res1, threads=4, result2=res2)
        images.append((res, res1, res2))
    for a, b in zip(*images):
        assert _same(a, b)
    assert images[0][0][27, 28, 3] > 39000.0          # the pixel in question: its primary hit lies that far out
    t = tracer
    t.load_scene(scene, sky)
    t.update_transforms(objs)
    t.resize(W, H)
    t.write_target(pa.PT_TARGET_RESULT, z["before"][0]); t.write_converged(z["before"][1]); t.write_target(pa.PT_TARGET_DENOISED, z["before"][2])
    t.clear_camera_history()
    t.set_camera_history(old)
    t.render(p)
    for got, want in zip((t.read_result(), t.read_converged(), t.read_target(pa.PT_TARGET_DENOISED)), images[0]):
        assert _same(got, want)


@pytest.mark.parametrize("width", [4, 2, 8])
def test_slivers_seen_from_far_away_keep_the_hits_that_only_rounding_produces(width):
    """Moeller-Trumbore's barycentrics (RayTracingShader.compute:192-230) err in proportion to the ray origin's distance and to
    the triangle's thinness: a sliver ten units long and a thousandth wide, seen from 1e3 .. 1e5 units away, reports hits
    units beyond its end.  The exhaustive loop finds them, so the tree must not prune them: a ray origin beyond the reach
    the boxes were padded for asks for a build with a larger reach, in scenes without spheres too (the oracle's own tree
    missed two of 1 500 such rays until it widened its boxes by the triangles' thinness)."""
    rng = np.random.default_rng(11)
    a = rng.uniform(-10, 10, (300, 3)); dirs = rng.normal(size=(300, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    side = np.cross(dirs, rng.normal(size=(300, 3))); side /= np.linalg.norm(side, axis=1, keepdims=True)
    verts = np.stack([a, a + 10 * dirs, a + 10 * dirs + 1e-3 * side], 1).reshape(-1, 3).astype(np.float32)
    index = np.arange(900, dtype=np.int32)
    f = pa.Scene("fixture")
    objs = np.zeros(1, f.objects.dtype); objs[0] = (np.eye(4, dtype=np.float32).reshape(16), 0, 900)
    sph = np.zeros((0, 14), np.float32)
    tris = verts.reshape(-1, 3, 3).astype(np.float64)
    rays = []
    for _ in range(3000):
        t = tris[rng.integers(0, len(tris))]
        dist = 10.0 ** rng.uniform(3, 5)
        k = rng.integers(0, 3); p0, p1, p2 = t[k], t[(k + 1) % 3], t[(k + 2) % 3]
        on_edge = p0 + rng.uniform(0.1, 0.9) * (p1 - p0)
        n = np.cross(t[1] - t[0], t[2] - t[0]); nn = np.linalg.norm(n)
        outward = np.cross(p1 - p0, n); outward /= np.linalg.norm(outward)
        if outward @ (p2 - p0) > 0:
            outward = -outward
        target = on_edge + outward * 1e-3 * rng.uniform(-0.5, 0.6)
        dirn = n / nn + 0.7 * rng.normal(size=3); dirn /= np.linalg.norm(dirn)
        if dirn @ n < 0:
            dirn = -dirn
        o = target + dirn * dist
        d = target - o; d /= np.linalg.norm(d)
        rays.append(np.concatenate([o, d]))
    od = np.array(rays, np.float32)
    want_i, want_t = po.Oracle(sph, objs, verts, index, accel=False).trace_rays(od)
    acc_i, acc_t = po.Oracle(sph, objs, verts, index, accel=True).trace_rays(od)
    assert (acc_i == want_i).all() and (acc_t.view(np.uint32) == want_t.view(np.uint32)).all()
    assert (want_i > 0).sum() > 200
    with pa.PathTracer(0) as t:
        t.set_option("bvh_width", width)
        t.set_spheres(sph); t.set_meshes(objs, verts, index)
        ib, tb = t.trace_rays(od); ie, te = t.trace_rays(od, brute_force=True)
        assert (ie == want_i).all() and (te.view(np.uint32) == want_t.view(np.uint32)).all()       # the device's exhaustive loop
        assert (ib == want_i).all() and (tb.view(np.uint32) == want_t.view(np.uint32)).all()       # and its tree


def test_no_exception_leaves_the_library():
    """A C++ exception inside an entry point (a host allocation that fails) comes back as an error code and a message; the
    context stays usable.  Option "debug_throw" throws from inside pt_set_option for this test."""
    f = pa.Scene("fixture")
    with pa.PathTracer(0) as t:
        lib, ctx = t._lib, t._ctx
        assert lib.pt_set_option(ctx, b"debug_throw", 1) == pa.PT_ERR_NOMEM
        assert b"out of host memory" in lib.pt_last_error(ctx)
        assert lib.pt_set_option(ctx, b"debug_throw", 2) == pa.PT_ERR_HIP
        assert b"debug_throw" in lib.pt_last_error(ctx)
        assert lib.pt_set_option(ctx, b"debug_throw", 0) == pa.PT_OK
        t.load_scene(f, pa.procedural_sky(16, 8)); t.resize(32, 18)
        t.render(pa.make_params(f, 32, 18, spp=1, max_bounces=2))
        assert np.isfinite(t.read_result()).all()
