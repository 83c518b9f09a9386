"""Synthetic scenes and frame parameters (CPU side; wraps host/libptmaster.so)."""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import PtCameraHistory, PtFrameParams, PtMeshObject, PtSphere

SCENE_NAMES = ("spheres", "xr", "fixture", "blob70k", "hall260k", "city10m")


class Scene:
    """The reference's scene inputs as flat arrays: spheres (n,14) f32, mesh objects
    (structured, 72 B each), vertices (n,3) f32, indices int32 - the exact buffer
    layouts of RayTracingMaster.cs:93-108 - plus a default camera."""

    def __init__(self, name):
        lib = _ffi.ptmaster()
        self.name = name
        self._h = lib.ptm_scene_create(name.encode())
        if not self._h:
            raise ValueError(f"unknown scene {name!r}; known: {SCENE_NAMES}")
        self._refresh()

    def _refresh(self):
        lib = _ffi.ptmaster()
        counts = (C.c_int32 * 4)()
        lib.ptm_scene_counts(self._h, counts)
        ns, no, nv, ni = counts
        self.spheres = np.zeros((ns, 14), np.float32)
        self.objects = np.zeros(no, dtype=np.dtype([("m", np.float32, 16), ("offset", np.int32), ("count", np.int32)]))
        self.vertices = np.zeros((nv, 3), np.float32)
        self.indices = np.zeros(ni, np.int32)
        lib.ptm_scene_copy(self._h, self.spheres.ctypes.data, self.objects.ctypes.data,
                           self.vertices.ctypes.data, self.indices.ctypes.data)
        eye = (C.c_float * 3)(); tgt = (C.c_float * 3)(); fov = C.c_float()
        lib.ptm_scene_camera(self._h, eye, tgt, C.byref(fov))
        self.eye = np.array(eye, np.float32); self.target = np.array(tgt, np.float32); self.fov = float(fov.value)

    @property
    def handle(self):
        return self._h

    @property
    def n_triangles(self):
        return int(self.objects["count"].sum() // 3) if len(self.objects) else 0

    def add_obj(self, path, local_to_world=None):
        m = np.eye(4, dtype=np.float32).T.reshape(16) if local_to_world is None else np.asarray(local_to_world, np.float32)
        rc = _ffi.ptmaster().ptm_scene_add_obj(self._h, str(path).encode(), m.ctypes.data)
        if rc < 0:
            raise ValueError(f"cannot load OBJ {path}: code {rc}")
        self._refresh()
        return rc

    def __del__(self):
        try:
            if self._h:
                _ffi.ptmaster().ptm_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass


def procedural_sky(w=256, h=128):
    sky = np.zeros((h, w, 4), np.float32)
    _ffi.ptmaster().ptm_sky_procedural(w, h, sky.ctypes.data)
    return sky


def load_hdr(path, max_size=2048):
    """A Radiance .hdr picture as a sky texture (H, W, 4) float32, bottom row first, halved until neither side exceeds
    max_size (the reference asset's import setting, Assets/HDRI/cape_hill_4k.hdr.meta:34)."""
    lib = _ffi.ptmaster()
    w, h = C.c_int(0), C.c_int(0)
    if lib.ptm_sky_load_hdr(str(path).encode(), int(max_size), C.byref(w), C.byref(h), None, 0) != 0:
        raise ValueError(lib.ptm_sky_error().decode())
    sky = np.zeros((h.value, w.value, 4), np.float32)
    if lib.ptm_sky_load_hdr(str(path).encode(), int(max_size), C.byref(w), C.byref(h), sky.ctypes.data, w.value * h.value) != 0:
        raise ValueError(lib.ptm_sky_error().decode())
    return sky


def frame_seed(k):
    return float(_ffi.ptmaster().ptm_frame_seed(int(k)))


def camera_matrices(eye, target, fov_deg, aspect, up=(0, 1, 0), z_near=0.1, z_far=100.0):
    e = np.asarray(eye, np.float32); t = np.asarray(target, np.float32); u = np.asarray(up, np.float32)
    c2w = np.zeros(16, np.float32); ip = np.zeros(16, np.float32)
    _ffi.ptmaster().ptm_camera_look_at(e.ctypes.data, t.ctypes.data, u.ctypes.data, fov_deg, aspect, z_near, z_far,
                                       c2w.ctypes.data, ip.ctypes.data)
    return c2w, ip


MATERIAL_DTYPE = np.dtype([("albedo", np.float32, 3), ("smoothness", np.float32), ("specular", np.float32, 3),
                           ("lobe", np.int32), ("emission", np.float32, 3), ("reserved", np.float32)])


def materials(entries):
    """PtMaterial array from (albedo, smoothness, specular, lobe[, emission]) tuples, one per mesh object."""
    m = np.zeros(len(entries), MATERIAL_DTYPE)
    for i, e in enumerate(entries):
        m[i]["albedo"] = e[0]; m[i]["smoothness"] = e[1]; m[i]["specular"] = e[2]; m[i]["lobe"] = e[3]
        if len(e) > 4:
            m[i]["emission"] = e[4]
    return m


def camera_history_entry(eye, target, fov_deg, aspect, up=(0, 1, 0), z_near=0.1, z_far=100.0):
    """One PtCameraHistory (cameraToWorld, inverse projection, worldToCamera, projection) for a look-at camera."""
    e = np.asarray(eye, np.float32); t = np.asarray(target, np.float32); u = np.asarray(up, np.float32)
    h = PtCameraHistory()
    _ffi.ptmaster().ptm_camera_history_entry(e.ctypes.data, t.ctypes.data, u.ctypes.data, fov_deg, aspect, z_near, z_far,
                                             C.byref(h))
    return h


def stereo_eye_matrices(eye, target, fov_deg, aspect, offset, shear, up=(0, 1, 0), z_near=0.1, z_far=100.0):
    """(view, projection) of one stereo eye, each 16 floats column-major: what GetStereoViewMatrix / GetStereoProjectionMatrix return."""
    e = np.asarray(eye, np.float32); t = np.asarray(target, np.float32); u = np.asarray(up, np.float32)
    view = np.zeros(16, np.float32); proj = np.zeros(16, np.float32)
    _ffi.ptmaster().ptm_camera_stereo_eye(e.ctypes.data, t.ctypes.data, u.ctypes.data, fov_deg, aspect, z_near, z_far, offset, shear,
                                          view.ctypes.data, proj.ctypes.data)
    return view, proj


def matrix_inverse(m):
    m = np.ascontiguousarray(m, np.float32).reshape(16); out = np.zeros(16, np.float32)
    if not _ffi.ptmaster().ptm_matrix_inverse(m.ctypes.data, out.ctypes.data):
        raise ValueError("singular matrix")
    return out


def stereo_history_entry(view, proj):
    """The PtCameraHistory the stereo branch appends (RayTracingMaster.cs:480-485): inverse view, inverse projection, view, projection."""
    h = PtCameraHistory()
    h.camera_to_world[:] = matrix_inverse(view).tolist(); h.camera_inverse_projection[:] = matrix_inverse(proj).tolist()
    h.world_to_camera[:] = np.asarray(view, np.float32).tolist(); h.camera_projection[:] = np.asarray(proj, np.float32).tolist()
    return h


def make_params(scene, width, height, *, seed=0.5, spp=1, max_bounces=8, depth=0, accumulate=0, sample_index=0,
                tile=None, eye=None, target=None, fov=None, render_mode=1, accumulation_frames=0, divisions=0, counter=0, left_eye=0):
    """PtFrameParams for `scene`'s default (or the given) camera."""
    c2w, ip = camera_matrices(scene.eye if eye is None else eye, scene.target if target is None else target,
                              scene.fov if fov is None else fov, width / height)
    p = PtFrameParams()
    p.camera_to_world[:] = c2w.tolist()
    p.camera_inverse_projection[:] = ip.tolist()
    p.seed = seed; p.samples_per_pixel = spp; p.max_bounces = max_bounces; p.depth = depth
    p.render_mode = render_mode; p.sample_index = sample_index; p.accumulate = accumulate
    p.accumulation_frames = accumulation_frames; p.divisions = divisions; p.counter = counter; p.left_eye = left_eye
    if tile is not None:
        p.tile_x0, p.tile_y0, p.tile_w, p.tile_h = tile
    p.directional_light[1] = -1.0; p.directional_light[3] = 1.0
    return p
