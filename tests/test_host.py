"""Host side above the C ABI: scene procedure (RayTracingMaster.cs:315-369), mesh flattening
(:371-413), camera matrices (:489-490) and the master's per-frame parameters (:455-540)."""
import ctypes as C
import os

import numpy as np
import pytest

import pathtracing_amd as pa
from pathtracing_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sphere_field_procedure():
    s = pa.Scene("spheres")
    sp = s.spheres
    assert 20 < len(sp) <= 100
    pos, r = sp[:, :3], sp[:, 3]
    assert ((r >= 3.0) & (r <= 8.0)).all()
    assert np.array_equal(pos[:, 1], r)                                  # resting on the plane (M:328)
    assert (np.hypot(pos[:, 0], pos[:, 2]) <= 100.0 + 1e-3).all()        # insideUnitCircle * placement radius
    d = np.linalg.norm(pos[:, None] - pos[None], axis=-1) + np.eye(len(sp)) * 1e9
    assert (d >= (r[:, None] + r[None]) - 1e-4).all()                    # rejection of overlaps (M:331-336)
    albedo, spec, smooth, emis = sp[:, 4:7], sp[:, 7:10], sp[:, 10], sp[:, 11:14]
    metal = (albedo == 0).all(1) & (spec > 0).any(1)
    diel = np.isclose(spec, 0.04).all(1)
    emit = (emis > 0).any(1)
    assert (metal | diel | emit).all() and not (emit & (metal | diel)).any()
    assert (emis[emit].max(1) >= 3.0 - 1e-4).all() and (emis[emit].max(1) <= 8.0 + 1e-4).all()   # ColorHSV(..., 3, 8)
    assert ((albedo[emit] == 0).all() and (spec[emit] == 0).all())
    assert ((smooth >= 0) & (smooth <= 1)).all()
    # deterministic for a given seed, different for another
    a = np.zeros((200, 14), np.float32); b = np.zeros((200, 14), np.float32); c = np.zeros((200, 14), np.float32)
    lib = _ffi.ptmaster()
    na = lib.ptm_sphere_field(207407, 100, 3.0, 8.0, 100.0, a.ctypes.data, 200)
    nb = lib.ptm_sphere_field(207407, 100, 3.0, 8.0, 100.0, b.ctypes.data, 200)
    nc = lib.ptm_sphere_field(1, 100, 3.0, 8.0, 100.0, c.ctypes.data, 200)
    assert na == nb == len(sp) and (a == b).all() and (a[:na] == sp).all() and not (a[:min(na, nc)] == c[:min(na, nc)]).all()


def test_scene_sizes():
    assert pa.Scene("blob70k").n_triangles == 69936
    h = pa.Scene("hall260k")
    assert h.n_triangles == 262480
    x = pa.Scene("xr")
    assert x.n_triangles == 12 and len(x.objects) == 1 and len(x.spheres) > 30
    with pytest.raises(ValueError):
        pa.Scene("nope")


def test_mesh_flattening_offsets_indices_like_the_reference():
    """indices are stored already offset by the mesh's first vertex (RayTracingMaster.cs:392-399)."""
    s = pa.Scene("fixture")
    n0 = len(s.vertices)
    import tempfile, os
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "quad.obj")
        open(path, "w").write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
        s.add_obj(path)
    assert len(s.objects) == 2 and len(s.vertices) == n0 + 4
    o = s.objects[1]
    assert o["offset"] == s.objects[0]["count"] and o["count"] == 6      # quad fan-triangulated
    assert s.indices[o["offset"]:].min() == n0 and s.indices[o["offset"]:].max() == n0 + 3
    # groups become separate mesh objects, each with the vertices it uses (in file order) and nothing else
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "two.obj")
        open(path, "w").write("# two groups sharing one vertex pool\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 5 5 5\nv 0 0 2\n"
                              "o first\nf 1/1/1 2/2/1 3/3/1\ng second\nf -1 -3 -4\nf 6//2 3//2 4//2\no empty\n")
        n_obj, n_v = len(s.objects), len(s.vertices)
        first = s.add_obj(path)
    assert first == n_obj and len(s.objects) == n_obj + 2 and len(s.vertices) == n_v + 3 + 3       # vertex 5 is used by nobody
    a, b = s.objects[n_obj], s.objects[n_obj + 1]
    assert a["count"] == 3 and b["count"] == 6 and b["offset"] == a["offset"] + 3
    v = np.asarray(s.vertices).reshape(-1, 3)
    assert (v[s.indices[a["offset"]:a["offset"] + 3]] == [[0, 0, 0], [1, 0, 0], [1, 1, 0]]).all()
    assert (v[s.indices[b["offset"]:b["offset"] + 6]] == [[0, 0, 2], [0, 1, 0], [1, 1, 0], [0, 0, 2], [1, 1, 0], [0, 1, 0]]).all()


def test_blob_is_wound_outward():
    s = pa.Scene("fixture")
    v = s.vertices[s.indices.reshape(-1, 3)]
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    assert (np.einsum("ij,ij->i", n, v.mean(1)) > 0).all()               # blob is centred on the object origin


def test_city_scene_instances_share_geometry():
    s = pa.Scene("city10m")
    assert len(s.objects) == 38 and s.n_triangles == 38 * 262480
    assert len(s.indices) == 262480 * 3                                  # one copy of the geometry
    assert (s.objects["offset"] == 0).all()


def test_camera_matrices():
    eye = np.array([3, 8, -19], np.float32); tgt = np.array([0, 3.8, 0], np.float32)
    c2w, ip = pa.camera_matrices(eye, tgt, 60.0, 16 / 9)
    M = c2w.reshape(4, 4).T
    R = M[:3, :3]
    assert np.allclose(R.T @ R, np.eye(3), atol=1e-6) and np.isclose(np.linalg.det(R), 1.0, atol=1e-6)
    assert np.allclose(M[:3, 3], eye)
    fwd = (tgt - eye) / np.linalg.norm(tgt - eye)
    assert np.allclose(-R[:, 2], fwd, atol=1e-6)                         # camera looks down -Z (OpenGL convention)
    # inverse projection: P^-1 * (u, v, 0, 1) = (u*aspect*tan, v*tan, -1, .)
    P = ip.reshape(4, 4).T
    t = np.tan(np.radians(30.0))
    assert np.allclose(P @ np.array([0.5, -0.25, 0, 1]), [0.5 * 16 / 9 * t, -0.25 * t, -1, (100 + 0.1) / (2 * 100 * 0.1)], rtol=1e-5)
    # and it really is the inverse of the OpenGL perspective matrix
    n, f, a = 0.1, 100.0, 16 / 9
    proj = np.array([[1 / (a * t), 0, 0, 0], [0, 1 / t, 0, 0], [0, 0, (f + n) / (n - f), 2 * f * n / (n - f)], [0, 0, -1, 0]])
    assert np.allclose(proj @ P, np.eye(4), atol=1e-4)


def test_frame_seed():
    assert pa.frame_seed(0) == 0.5
    for k in (1, 2, 17, 1000):
        want = (0.5 + k * 0.6180339) % 1.0
        assert abs(pa.frame_seed(k) - want) < 1e-6
        assert 0.0 <= pa.frame_seed(k) < 1.0


def test_procedural_sky_is_deterministic_and_bright_at_the_sun():
    a = pa.procedural_sky(64, 32); b = pa.procedural_sky(64, 32)
    assert (a == b).all() and np.isfinite(a).all() and (a[..., :3] >= 0).all()
    assert a[..., 0].max() > 2.0 and a[..., :3].min() < 0.5


def test_master_parameters_without_a_gpu():
    """SetShaderParameters of the C++ mirror: per-frame seed, spp rule, depth flag, sample counter."""
    lib = _ffi.ptmaster()
    h = C.c_void_p()
    rc = lib.ptm_master_create(0, C.byref(h))
    try:
        lib.ptm_master_configure(h, 1920, 1080, 4, 8, 0, pa.PT_ACCUM_ADDMAT, 10)
        eye = (C.c_float * 3)(0, 4.6, 0); tgt = (C.c_float * 3)(0, 4.6, 1)
        lib.ptm_master_set_camera(h, eye, tgt, 90.0, -0.011)
        p = pa.PtFrameParams()
        lib.ptm_master_peek_params(h, C.byref(p))
        assert p.samples_per_pixel == 4 and p.max_bounces == 8 and p.depth == 0 and p.render_mode == 1
        assert p.seed == 0.5 and p.sample_index == 0 and p.accumulate == pa.PT_ACCUM_ADDMAT
        c2w = np.array(p.camera_to_world[:]).reshape(4, 4).T
        # left eye: shifted by -0.011 along camera right; looking down +z with up +y, camera right is world -x
        assert np.isclose(abs(c2w[0, 3]), 0.011, atol=1e-6) and np.isclose(c2w[1, 3], 4.6)
        want_c2w, want_ip = pa.camera_matrices((c2w[0, 3], 4.6, 0), (c2w[0, 3], 4.6, 1), 90.0, 1920 / 1080)
        assert np.allclose(p.camera_to_world[:], want_c2w) and np.allclose(p.camera_inverse_projection[:], want_ip)
        lib.ptm_master_configure(h, 1920, 1080, 4, 8, 1, pa.PT_ACCUM_ADDMAT, 10)     # depth on
        lib.ptm_master_peek_params(h, C.byref(p))
        assert p.depth == 1 and p.accumulate == pa.PT_ACCUM_NONE and p.sample_index == 0   # M:805: depth ? 0 : _currentSample
        if rc != 0:
            assert b"pt_create" in lib.ptm_master_error(h)     # no GPU here: the error is reported, nothing rendered
            assert lib.ptm_master_render(h) != 0
    finally:
        lib.ptm_master_destroy(h)


def test_obj_loader_reads_the_reference_bunny_when_present():
    """Assets/Meshes/BunnyLow.obj, the mesh the reference's scene registers (Blender export: v/vt/vn index triples,
    220 vertices, 432 triangular faces).  Only where the reference checkout exists; nothing is copied from it."""
    import os
    path = "/root/reference/Assets/Meshes/BunnyLow.obj"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    s = pa.Scene("spheres")
    assert s.n_triangles == 0
    s.add_obj(path)
    assert s.n_triangles == 432 and len(s.vertices) == 220 and len(s.objects) == 1
    idx = s.indices[s.objects[0]["offset"]:s.objects[0]["offset"] + s.objects[0]["count"]]
    assert idx.min() == 0 and idx.max() == 219
    # the first face "f 4/1/1 1/2/2 5/3/3": position indices only, zero-based
    assert list(idx[:3]) == [3, 0, 4]
    v = np.asarray(s.vertices).reshape(-1, 3)
    assert np.allclose(v[0], [-0.615033, 1.454237, -0.542636], atol=1e-6)
    # and the oracle renders it: the exhaustive loop and the oracle's own tree agree on every primary hit
    from oracle import pt_oracle as po
    sky = pa.procedural_sky(64, 32)
    s.eye = np.array([0.0, 1.2, -4.0], np.float32); s.target = np.array([0.0, 0.8, 0.0], np.float32)
    p = pa.make_params(s, 48, 27, seed=0.5, spp=1, max_bounces=2)
    a = po.Oracle.from_scene(s, sky, accel=False).render(p, 48, 27, threads=4)
    b = po.Oracle.from_scene(s, sky, accel=True).render(p, 48, 27, threads=4)
    assert (a[1] == b[1]).all() and (a[0] == b[0]).all()
    assert (a[1] > len(s.spheres)).sum() > 50            # the bunny is in view


def test_public_headers_are_plain_c_and_keep_the_reference_layouts(tmp_path):
    """include/ptrace.h (and the host and oracle headers) must compile as C99 and as C++11 - the boundary is a C ABI -
    and the structs a C# host marshals must keep the reference's strides (RayTracingMaster.cs:366, 410)."""
    import os, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "include/ptrace.h"\n#include "pathtracing_amd/host/ptmaster.h"\n#include "oracle/pt_oracle.h"\n'
                   '#include <stddef.h>\n'
                   'int main(void) { return sizeof(PtSphere) == 56 && sizeof(PtMeshObject) == 72 && sizeof(PtCameraHistory) == 256 &&\n'
                   '                        sizeof(PtMaterial) == 48 && offsetof(PtFrameParams, seed) == 128 ? 0 : 1; }\n')
    exe = tmp_path / "hdr"
    if shutil.which("gcc"):
        subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + root, str(src), "-o", str(exe)], check=True)
        assert subprocess.run([str(exe)]).returncode == 0
    if shutil.which("g++"):
        subprocess.run(["g++", "-std=c++11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + root, "-x", "c++", "-fsyntax-only", str(src)], check=True)


def test_bvh_builder_invariants_and_stack_bound(tmp_path):
    """tools/bvh_check.cpp (host only, under AddressSanitizer + UBSan): on named scenes and on degenerate inputs every
    primitive sits in exactly one leaf, every dequantised box holds what lies below it, and the builder's worst-case
    traversal-stack figure for the four-wide tree equals an independent walk of the tree - the figure the library uses to
    decide whether a lane's 92-entry stack can take that tree ("4000 points on a line" needs 126 and must not).  The
    eight-wide layout gets the same checks plus a host restatement of the walk by groups over random rays."""
    import subprocess
    exe = str(tmp_path / "bvh_check")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-I", "include", "-I", "pathtracing_amd/csrc",
           "-I", "pathtracing_amd/host", "tools/bvh_check.cpp", "pathtracing_amd/csrc/bvh.cpp", "pathtracing_amd/host/scene_gen.cpp",
           "-lpthread", "-o", exe]
    subprocess.run(cmd, cwd=ROOT, check=True)
    out = subprocess.run([exe], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0 and "bvh_check ok" in out.stdout, out.stdout + out.stderr
    need = {ln.split("  ")[0].strip(): int(ln.split("stack need")[1].split("/")[1].split(",")[0]) for ln in out.stdout.splitlines()
            if "stack need" in ln and "eight-wide" not in ln}
    assert need["4000 points on a line"] > 92 and need["hall260k leaf 2"] < 60
    # the eight-wide tree: one group per level plus one; the bench scenes stay far below the 46 groups a lane's stack holds
    need8 = {ln.split("  ")[0].strip(): int(ln.split("stack need")[1].split("(")[0]) for ln in out.stdout.splitlines() if "eight-wide" in ln}
    assert need8["hall260k leaf 2"] <= 12 and need8["blob70k leaf 2"] <= 12 and need8["4000 points on a line"] <= 46


def test_matrix_inverse_and_stereo_eye_matrices():
    """Matrix4x4.inverse as the stereo branch of SetShaderParameters uses it (RayTracingMaster.cs:463-466), and the per-eye
    view / projection matrices the tests feed it."""
    eye, target, fov, aspect = (1.0, 4.6, -6.0), (0.5, 1.0, 2.0), 90.0, 2064 / 2208
    h = pa.camera_history_entry(eye, target, fov, aspect)
    # inverse(cameraToWorld) = worldToCamera, inverse(projection) = the analytic inverse projection, and back
    assert np.allclose(pa.matrix_inverse(h.camera_to_world), np.array(h.world_to_camera), atol=2e-6)
    assert np.allclose(pa.matrix_inverse(h.camera_projection), np.array(h.camera_inverse_projection), rtol=1e-5, atol=1e-6)
    m = np.array(h.camera_to_world, np.float32).reshape(4, 4).T
    assert np.allclose(pa.matrix_inverse(h.camera_to_world).reshape(4, 4).T @ m, np.eye(4), atol=1e-5)
    with pytest.raises(ValueError):
        pa.matrix_inverse(np.zeros(16, np.float32))
    # the two eyes: 0.022 apart along the camera's right axis (TestScene.unity:870), frusta sheared in opposite directions
    vl, pl = pa.stereo_eye_matrices(eye, target, fov, aspect, -0.011, +0.07)
    vr, pr = pa.stereo_eye_matrices(eye, target, fov, aspect, +0.011, -0.07)
    cl, cr = pa.matrix_inverse(vl), pa.matrix_inverse(vr)
    right = np.array(h.camera_to_world[0:3])
    assert np.allclose(cr[12:15] - cl[12:15], 0.022 * right, atol=1e-5)
    assert np.allclose(cl[0:12], cr[0:12], atol=1e-6)                       # parallel axes
    assert pl[8] == np.float32(0.07) and pr[8] == np.float32(-0.07)
    assert np.allclose(np.delete(pl, 8), np.delete(np.array(h.camera_projection, np.float32), 8))
    # a sheared frustum's centre ray leaves the eye off-axis: clip (0, 0) comes from x = P02 / P00 per unit of distance ahead
    ip = pa.matrix_inverse(pl).reshape(4, 4).T
    centre = ip @ np.array([0, 0, 0, 1], np.float32)
    assert abs(centre[0] / -centre[2] - 0.07 / pl[0]) < 1e-5


def _write_hdr(path, rgbe, rle=True, flip_header=False):
    """A Radiance picture from (H, W, 4) uint8 RGBE texels, top scanline first; new-style run-length coding or flat."""
    h, w, _ = rgbe.shape
    out = bytearray(b"#?RADIANCE\n# written by the test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n")
    out += (f"+Y {h} +X {w}\n" if flip_header else f"-Y {h} +X {w}\n").encode()
    for y in range(h):
        if not rle:
            out += rgbe[y].tobytes()
            continue
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            row = rgbe[y, :, c]
            x = 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, row[x]]); x += run
                else:
                    lit = 1
                    while x + lit < w and lit < 128 and not (x + lit + 2 < w and row[x + lit] == row[x + lit + 1] == row[x + lit + 2]):
                        lit += 1
                    out += bytes([lit]) + row[x:x + lit].tobytes(); x += lit
    with open(path, "wb") as f:
        f.write(out)


def test_radiance_hdr_sky_loader(tmp_path):
    """The format of the reference's SkyboxTexture asset (Assets/HDRI/cape_hill_4k.hdr; the file itself is not in the
    checkout): run-length coded and flat scanlines decode to mantissa * 2^(e - 136), the picture's bottom scanline becomes
    row 0 (v = 0), a picture over the size limit is halved by 2x2 means, and broken files are refused with a reason."""
    rng = np.random.default_rng(4)
    H, W = 16, 32
    rgbe = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    rgbe[:, :, 3] = rng.integers(120, 140, (H, W))
    rgbe[3, 5:20] = rgbe[3, 5]                        # a long run
    rgbe[7, :, :] = (10, 20, 30, 129)                 # a constant scanline
    rgbe[2, 4, 3] = 0                                 # exponent 0: black
    want = rgbe[:, :, :3].astype(np.float32) * np.ldexp(np.float32(1), rgbe[:, :, 3].astype(np.int32) - 136)[..., None]
    want[rgbe[:, :, 3] == 0] = 0
    want = np.concatenate([want, np.ones((H, W, 1), np.float32)], 2)[::-1]      # bottom scanline first
    for rle in (True, False):
        path = tmp_path / f"sky_{int(rle)}.hdr"
        _write_hdr(path, rgbe, rle=rle)
        got = pa.load_hdr(path, max_size=0)
        assert got.shape == (H, W, 4) and (got == want).all()
    _write_hdr(tmp_path / "up.hdr", rgbe, flip_header=True)                      # "+Y": the file already starts at the bottom
    assert (pa.load_hdr(tmp_path / "up.hdr", max_size=0) == want[::-1]).all()
    half = pa.load_hdr(tmp_path / "sky_1.hdr", max_size=16)
    assert half.shape == (8, 16, 4)
    assert np.allclose(half, 0.25 * (want[0::2, 0::2] + want[0::2, 1::2] + want[1::2, 0::2] + want[1::2, 1::2]), rtol=1e-6)
    assert pa.load_hdr(tmp_path / "sky_1.hdr").shape == (H, W, 4)                # default limit 2048: untouched
    # refusals
    raw = (tmp_path / "sky_1.hdr").read_bytes()
    (tmp_path / "short.hdr").write_bytes(raw[:len(raw) // 2])
    (tmp_path / "magic.hdr").write_bytes(b"P6\n" + raw)
    (tmp_path / "xyze.hdr").write_bytes(raw.replace(b"32-bit_rle_rgbe", b"32-bit_rle_xyze"))
    # hostile files: old-style repeat texels (1, 1, 1, n) with a count of zero never advance while their shift grows, and a
    # header may claim 32768 x 32768 texels; both are errors, not undefined behaviour or an allocation of 16 GB
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n"
    (tmp_path / "zero_run.hdr").write_bytes(head + b"-Y 1 +X 4\n" + bytes([9, 9, 9, 128]) + bytes([1, 1, 1, 0]) * 40)
    (tmp_path / "long_run.hdr").write_bytes(head + b"-Y 1 +X 4\n" + bytes([9, 9, 9, 128]) + bytes([1, 1, 1, 1]) * 40)
    (tmp_path / "huge.hdr").write_bytes(head + b"-Y 32768 +X 32768\n" + bytes(64))
    for name, why in (("short.hdr", "ends early"), ("magic.hdr", "not a Radiance"), ("xyze.hdr", "32-bit_rle_rgbe"), ("absent.hdr", "cannot open"),
                      ("zero_run.hdr", "bad run"), ("long_run.hdr", "bad run"), ("huge.hdr", "larger than")):
        with pytest.raises(ValueError, match=why):
            pa.load_hdr(tmp_path / name)


def test_block_row_multiplier_is_exact():
    """pt_render numbers a rectangle's 8x8 pixel blocks row by row and the kernels turn a block number n back into
    (n % tiles_x, n / tiles_x).  Where blocks * tiles_x < 2^32 the host hands them m = floor(2^32 / tiles_x) + 1 and they take
    n / tiles_x = (n * m) >> 32 (csrc/ptrace_api.hip set_tiles, csrc/pt_device.hpp numbered_pixel); this restates the rule and
    checks it at every multiple of tiles_x, one below and one above, up to the largest block number the rule admits."""
    rng = np.random.default_rng(7)
    sizes = [(2, 1), (3, 5), (240, 135), (258, 276), (480, 270), (1000, 1000), (1625, 1625), (65535, 1)]
    sizes += [(int(rng.integers(2, 1500)), int(rng.integers(1, 1500))) for _ in range(200)]
    checked = 0
    for tiles_x, tiles_y in sizes:
        blocks = tiles_x * tiles_y + 64                      # kWorkShards more than the rectangle holds
        if blocks * tiles_x >= 1 << 32:
            continue                                         # the host sets no multiplier: the kernels divide
        m = (1 << 32) // tiles_x + 1
        assert m < 1 << 32
        rows = np.arange(0, blocks // tiles_x + 1, dtype=np.uint64)
        for n in (rows * tiles_x, rows * tiles_x + (tiles_x - 1), np.array([blocks - 1], np.uint64)):
            n = n[n < blocks]
            assert ((n * np.uint64(m)) >> np.uint64(32) == n // np.uint64(tiles_x)).all(), (tiles_x, tiles_y)
        checked += 1
    assert checked > 100
