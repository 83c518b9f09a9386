"""The tree built on the device (option "tree_build" 1: top-down SAH, pt_build_sah.hpp; 3: clustering, pt_build.hpp) in place
of the host's: the image, the hit ids and every ray query must be what the exhaustive loop of RayTracingShader.compute:
232-285 gives - the tree decides how fast a hit is found, never which."""
import re

import numpy as np
import pytest

import pathtracing_amd as pa
from oracle import pt_oracle as po

pytestmark = pytest.mark.gpu

BUILDS = [1, 3]


def _same(a, b):
    return (np.ascontiguousarray(a).view(np.uint32) == np.ascontiguousarray(b).view(np.uint32)).all()


def _rot_y(deg, t, scale=1.0):
    a = np.radians(deg)
    m = np.array([[np.cos(a) * scale, 0, np.sin(a) * scale, t[0]], [0, scale, 0, t[1]],
                  [-np.sin(a) * scale, 0, np.cos(a) * scale, t[2]], [0, 0, 0, 1]], np.float32)
    return m.T.reshape(16)


def _rays(rng, n, extent=6.0):
    o = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1)


def _tree_equals_loop(t, od):
    ib, tb = t.trace_rays(od); ie, te = t.trace_rays(od, brute_force=True)
    assert (ib == ie).all() and (tb.view(np.uint32) == te.view(np.uint32)).all()
    return ie


@pytest.mark.parametrize("build", BUILDS)
@pytest.mark.parametrize("name", ["fixture", "xr", "blob70k"])
def test_device_built_tree_gives_the_oracles_image(build, name, capfd, monkeypatch):
    """Scenes of a few hundred primitives and of 70 k (whole-array levels first, then one wave per small range): the
    oracle's image and hit ids through every backend, the exhaustive loop's answer for random rays, and the same tree from
    a second build of the same scene."""
    monkeypatch.setenv("PTRACE_BUILD_TIMING", "1")
    s = pa.Scene(name); sky = pa.procedural_sky(64, 32)
    W, H = 96, 54
    p = pa.make_params(s, W, H, seed=0.5, spp=2, max_bounces=6)
    want, want_ids, _ = po.Oracle.from_scene(s, sky, accel=True).render(p, W, H, threads=8)
    with pa.PathTracer(0) as t:
        t.set_option("instancing", 0); t.set_option("tree_build", build)
        t.load_scene(s, sky); t.resize(W, H); t.stats(); t.render(p)
        assert _same(t.read_result(), want) and (t.read_hit_ids() == want_ids).all()
        first = t.stats()
        log = capfd.readouterr().err
        m = re.search(r"scene build on the device: (\d+) primitives, (\d+) references, (\d+) nodes", log)
        assert m and int(m.group(1)) == len(s.spheres) + s.n_triangles and int(m.group(3)) == first["bvh_nodes"]     # no fallback to the host
        hit = _tree_equals_loop(t, _rays(np.random.default_rng(5), 4000))
        assert (hit > 0).sum() > 10
        for backend in (1, 2, 0):
            t.set_option("backend", backend); t.render(p)
            assert _same(t.read_result(), want), backend
        # built again (the option is set to the host's builder and back): the same tree
        t.set_option("tree_build", 0); t.render(p)
        assert _same(t.read_result(), want)
        t.set_option("tree_build", build); t.stats(); t.render(p)
        again = t.stats()
        assert _same(t.read_result(), want)
        for key in ("bvh_nodes", "bvh_depth", "nodes_visited", "prims_tested", "rays"):
            assert again[key] == first[key], key


@pytest.mark.parametrize("build", BUILDS)
def test_device_built_tree_is_refitted_like_the_hosts(build, sky_small):
    """pt_update_transforms after a device build: the refit kernels work from the levels the device build recorded.  Three
    placements of one mesh, flattened (instancing 0: 1 076 primitives), moved for four frames - each equals the oracle's
    render of the moved scene, and the tree still answers like the exhaustive loop."""
    f = pa.Scene("fixture")
    objs = np.zeros(3, dtype=f.objects.dtype)
    for k in range(3):
        objs[k] = (_rot_y(0, (-3.0 + 3.0 * k, 0.0, 2.0 * k)), 0, len(f.indices))
    W, H = 96, 54
    p = pa.make_params(f, W, H, spp=1, max_bounces=4, seed=0.3, eye=(0, 5, -12), target=(0, 1, 2), fov=60)
    with pa.PathTracer(0) as t:
        t.set_option("instancing", 0); t.set_option("tree_build", build)
        t.set_spheres(f.spheres); t.set_meshes(objs, f.vertices, f.indices); t.set_sky(sky_small); t.resize(W, H)
        t.render(p); t.sync()
        built = t.stats()["bvh_build_ms"]
        rng = np.random.default_rng(5)
        for frame in range(4):
            moved = objs.copy()
            moved["m"][0] = _rot_y(20.0 * (frame + 1), (-3.0, 0.2 * frame, 0.0))
            moved["m"][1] = _rot_y(-35.0 * frame, (1.5 * frame, 0.0, 2.0 + 10.0 * (frame == 2)))
            moved["m"][2] = _rot_y(10.0, (3.0, 0.0, 4.0), scale=1.0 + 0.25 * frame)
            t.update_transforms(moved); t.render(p)
            c, cid, _ = po.Oracle(f.spheres, moved, f.vertices, f.indices, sky=sky_small).render(p, W, H, threads=4)
            assert (t.read_hit_ids() == cid).all() and _same(t.read_result(), c), frame
            _tree_equals_loop(t, _rays(rng, 300, 8.0))
        assert built > 0


def _scene_of(kind, rng):
    """(spheres, objects, vertices, indices) of an awkward scene."""
    f = pa.Scene("fixture")
    eye4 = np.eye(4, dtype=np.float32).reshape(16)
    sph = f.spheres[:3].copy()
    if kind == "coincident":               # 300 copies of one triangle: every centre in one bin, ranges are halved as they lie
        v = np.array([[0, 1, 0], [2, 1, 0], [0, 3, 1]], np.float32)
        idx = np.tile(np.arange(3, dtype=np.int32), 300)
        objs = np.zeros(1, f.objects.dtype); objs[0] = (eye4, 0, len(idx))
    elif kind == "special":                # NaN, infinite and huge coordinates, triangles without area, among 400 ordinary ones
        v = rng.uniform(-4, 4, (600, 3)).astype(np.float32); v[:, 1] = np.abs(v[:, 1])
        idx = rng.integers(0, 600, (420, 3)).astype(np.int32)
        v[5, 0] = np.nan; v[17, 2] = np.inf; v[29, 1] = -np.inf; v[41, 0] = 1e20; v[53, 1] = 3e18
        idx[7] = (60, 60, 61); idx[8] = (62, 62, 62); idx[9] = idx[10]
        idx[0, 0] = 5; idx[1, 1] = 17; idx[2, 2] = 29; idx[3, 0] = 41; idx[4, 1] = 53
        idx = idx.reshape(-1)
        objs = np.zeros(1, f.objects.dtype); objs[0] = (eye4, 0, len(idx))
    elif kind == "two":                    # two primitives: the smallest scene the device builds (one is the host's business)
        v = np.array([[0, 1, 0], [2, 1, 0], [0, 3, 1], [1, 0.5, -1], [3, 0.5, -1], [1, 2, -2]], np.float32)
        idx = np.arange(6, dtype=np.int32)
        objs = np.zeros(1, f.objects.dtype); objs[0] = (eye4, 0, 6)
        sph = sph[:0]
    elif kind == "ragged":                 # an object without triangles between two with, index counts that are no multiple of three
        v = f.vertices.copy(); idx = f.indices.copy()
        objs = np.zeros(4, f.objects.dtype)
        objs[0] = (_rot_y(10, (-2.5, 0, 0)), 0, 3 * 100 + 2)
        objs[1] = (eye4, 300, 0)
        objs[2] = (_rot_y(-40, (2.5, 0.5, 1), 1.3), 300, len(idx) - 300 - 1)
        objs[3] = (eye4, len(idx), 0)
    elif kind == "spheres":                # no mesh at all
        sph = pa.Scene("spheres").spheres.copy()
        v = np.zeros((0, 3), np.float32); idx = np.zeros(0, np.int32); objs = np.zeros(0, f.objects.dtype)
    else:
        raise ValueError(kind)
    return sph, objs, v, idx


@pytest.mark.parametrize("build", BUILDS)
@pytest.mark.parametrize("kind", ["coincident", "special", "two", "ragged", "spheres"])
def test_device_build_of_awkward_scenes(build, kind, sky_small):
    """What the host builder is tested with, through the device builder: coincident primitives, NaN / infinite / huge
    coordinates, triangles without area, two primitives, empty objects and ragged index counts, spheres only.  The image
    equals the host-built tree's, random rays get the exhaustive loop's answer, ids and distances bit for bit."""
    rng = np.random.default_rng(3)
    sph, objs, v, idx = _scene_of(kind, rng)
    f = pa.Scene("fixture")
    W, H = 64, 36
    p = pa.make_params(f, W, H, spp=2, max_bounces=4, seed=0.7, eye=(0.5, 4, -11), target=(0, 1.5, 0), fov=60)
    images = {}
    for b in (0, build):
        with pa.PathTracer(0) as t:
            t.set_option("instancing", 0); t.set_option("tree_build", b)
            t.set_spheres(sph); t.set_meshes(objs, v, idx); t.set_sky(sky_small); t.resize(W, H)
            t.render(p)
            images[b] = (t.read_result(), t.read_hit_ids())
            if b:
                _tree_equals_loop(t, _rays(rng, 3000, 5.0))
                if len(objs):
                    moved = objs.copy(); moved["m"][:, 12] += 0.75
                    t.update_transforms(moved); t.render(p)
                    moved_image = t.read_result()
                    t.set_option("tree_build", 0); t.set_option("refit", 0); t.update_transforms(moved); t.render(p)
                    assert _same(moved_image, t.read_result())
    g, c = images[build][0], images[0][0]
    assert ((g.view(np.uint32) == c.view(np.uint32)) | (np.isnan(g) & np.isnan(c))).all() and (images[build][1] == images[0][1]).all()


@pytest.mark.parametrize("build", BUILDS)
def test_device_build_cuts_large_triangles_into_references(build, sky_small, capfd, monkeypatch):
    """Option "split_percent" in a device build: the device names the wall-sized triangles and long beams, the host's
    split_references cuts them, and the build works on references - several records naming one triangle, boxes clipped to
    their cuts.  As after a host build: the oracle's image, ids and ray count at every budget, fewer node visits and
    primitive tests with cuts than without, the same primitive count; a refit with unchanged matrices keeps every cut (same
    visit count), a moved beam still equals the oracle and a full build of the moved scene, and the exhaustive ray query -
    which walks records - reports every triangle once."""
    from test_gpu_parity import _room_with_beams
    monkeypatch.setenv("PTRACE_BUILD_TIMING", "1")
    f, objs, verts, index = _room_with_beams()
    W, H = 128, 72
    p = pa.make_params(f, W, H, spp=2, max_bounces=5, seed=0.41, eye=(2.0, 6.0, -13.0), target=(0.0, 4.0, 0.0), fov=60)
    c, cid, cst = po.Oracle(f.spheres, objs, verts, index, sky=sky_small).render(p, W, H, threads=4)
    seen = {}
    with pa.PathTracer(0) as t:
        t.set_option("instancing", 0); t.set_option("tree_build", build)
        t.set_spheres(f.spheres); t.set_meshes(objs, verts, index); t.set_sky(sky_small); t.resize(W, H)
        for percent in (0, 30, 400):
            t.set_option("split_percent", percent)
            t.stats(); capfd.readouterr(); t.render(p)
            st = t.stats()
            m = re.search(r"scene build on the device: (\d+) primitives, (\d+) references", capfd.readouterr().err)
            assert m, percent
            assert (t.read_hit_ids() == cid).all() and _same(t.read_result(), c) and st["rays"] == cst["rays"], percent
            seen[percent] = (st["nodes_visited"], st["prims_tested"], st["bvh_nodes"], st["bvh_prims"], int(m.group(2)) - int(m.group(1)))
            _tree_equals_loop(t, _rays(np.random.default_rng(9), 2000, 10.0))
        nt = int(objs["count"].sum()) // 3                                   # the budget is a share of the scene's triangles
        assert seen[0][4] == 0 and 0 < seen[30][4] <= 0.3 * nt and seen[30][4] < seen[400][4] <= 4.0 * nt, seen
        assert seen[0][3] == seen[30][3] == seen[400][3]                     # primitives, not references
        assert seen[30][2] > seen[0][2] and seen[400][2] > seen[30][2]       # more references, more nodes
        assert seen[30][0] < 0.85 * seen[0][0] and seen[30][1] < 0.6 * seen[0][1], seen
        assert seen[400][0] < 0.7 * seen[0][0] and seen[400][1] < 0.2 * seen[0][1], seen
        t.set_option("split_percent", 30)
        t.stats(); t.render(p); before = t.stats()["nodes_visited"]
        t.update_transforms(objs); t.render(p)                                # the same matrices: a refit that keeps every cut
        assert _same(t.read_result(), c) and t.stats()["nodes_visited"] == before
        moved = objs.copy()
        m4 = moved["m"][5].reshape(4, 4).T.copy(); m4[:3, 3] += (1.5, -0.5, 2.0); moved["m"][5] = m4.T.reshape(16)
        t.update_transforms(moved); t.render(p)
        c2, cid2, _ = po.Oracle(f.spheres, moved, verts, index, sky=sky_small).render(p, W, H, threads=4)
        g2 = t.read_result()
        assert (t.read_hit_ids() == cid2).all() and _same(g2, c2)
        _tree_equals_loop(t, _rays(np.random.default_rng(10), 2000, 10.0))
        t.set_option("refit", 0); t.update_transforms(moved); t.render(p)     # a full device build of the moved scene
        assert _same(t.read_result(), g2)


def test_refitted_tree_that_has_decayed_is_built_anew(sky_small, capfd, monkeypatch):
    """Option "rebuild_percent" (default 30): a refit keeps the last build's topology, so three meshes built on top of one
    another and then pulled apart leave a tree whose every node spans all three.  The summed node area is taken on the device
    after the refit and read by a later render without waiting; beyond the threshold that render builds anew - on the device,
    so it costs milliseconds.  Every frame equals the oracle's whether it comes from the refitted or the rebuilt tree; with
    the option at 0 the tree is only ever refitted."""
    monkeypatch.setenv("PTRACE_BUILD_TIMING", "1")
    f = pa.Scene("fixture")
    objs = np.zeros(3, dtype=f.objects.dtype)
    for k in range(3):
        objs[k] = (_rot_y(7.0 * k, (0.05 * k, 0.0, 0.0)), 0, len(f.indices))
    apart = objs.copy()
    for k in range(3):
        apart["m"][k] = _rot_y(7.0 * k, (-9.0 + 9.0 * k, 0.0, 3.0 * k))
    W, H = 96, 54
    p = pa.make_params(f, W, H, spp=1, max_bounces=4, seed=0.3, eye=(0, 6, -20), target=(0, 1, 2), fov=60)
    want, want_ids, _ = po.Oracle(f.spheres, apart, f.vertices, f.indices, sky=sky_small).render(p, W, H, threads=4)
    last = {}
    for percent, builds_expected in ((30, 2), (0, 1)):
        with pa.PathTracer(0) as t:
            t.set_option("instancing", 0); t.set_option("tree_build", 1); t.set_option("rebuild_percent", percent)
            t.set_spheres(f.spheres); t.set_meshes(objs, f.vertices, f.indices); t.set_sky(sky_small); t.resize(W, H)
            capfd.readouterr()
            t.render(p); t.sync()
            t.update_transforms(apart)
            visits = []
            for frame in range(3):
                t.stats(); t.render(p); t.sync()
                assert (t.read_hit_ids() == want_ids).all() and _same(t.read_result(), want), (percent, frame)
                visits.append(t.stats()["nodes_visited"])
            builds = len(re.findall(r"scene build on the device", capfd.readouterr().err))
            assert builds == builds_expected, (percent, builds)
            last[percent] = visits[-1]
            if not percent:
                assert visits[-1] == visits[0]
    assert last[30] < 0.8 * last[0], last                            # the rebuilt tree is the better one
    with pa.PathTracer(0) as t:
        with pytest.raises(Exception):
            t.set_option("rebuild_percent", -1)


def test_tree_build_option_values():
    with pa.PathTracer(0) as t:
        for v in (0, 1, 2, 3):
            t.set_option("tree_build", v)
        with pytest.raises(Exception):
            t.set_option("tree_build", 4)


def test_device_built_tree_too_deep_for_the_stack_falls_back_to_the_host(monkeypatch, capfd):
    """A four-wide tree whose worst case of stacked siblings exceeds a lane's stack is not walked (pt_device.hpp); the device
    build then hands the scene to the host builder, whose two-child tree always fits.  PTRACE_STACK_LIMIT = 20 makes the
    70 k-triangle scene (worst case 35) such a tree."""
    monkeypatch.setenv("PTRACE_STACK_LIMIT", "20"); monkeypatch.setenv("PTRACE_BUILD_TIMING", "1")
    s = pa.Scene("blob70k"); sky = pa.procedural_sky(64, 32)
    W, H = 96, 54
    p = pa.make_params(s, W, H, seed=0.5, spp=2, max_bounces=6)
    want, want_ids, _ = po.Oracle.from_scene(s, sky, accel=True).render(p, W, H, threads=8)
    with pa.PathTracer(0) as t:
        t.set_option("tree_build", 1)
        t.load_scene(s, sky); t.resize(W, H); t.render(p)
        assert _same(t.read_result(), want) and (t.read_hit_ids() == want_ids).all()
        log = capfd.readouterr().err
        assert "scene build on the device" not in log and "scene build:" in log
        _tree_equals_loop(t, _rays(np.random.default_rng(5), 2000))
