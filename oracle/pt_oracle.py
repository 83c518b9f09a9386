"""ctypes binding of the CPU oracle (oracle/_build/libptoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg - never by the pathtracing_amd package.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from pathtracing_amd._ffi import PtCameraHistory, PtFrameParams  # noqa: E402  (struct layouts only)

LIB_PATH = os.path.join(_HERE, "_build", "libptoracle.so")


class PtoScene(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("n_spheres", C.c_int32), ("objects", C.c_void_p), ("n_objects", C.c_int32),
                ("vertices", C.c_void_p), ("n_vertices", C.c_int32), ("indices", C.c_void_p), ("n_indices", C.c_int32),
                ("sky", C.c_void_p), ("sky_w", C.c_int32), ("sky_h", C.c_int32),
                ("materials", C.c_void_p), ("n_materials", C.c_int32)]


class PtoStats(C.Structure):
    _fields_ = [("trace_calls", C.c_uint64), ("rays", C.c_uint64), ("paths", C.c_uint64), ("prims_tested", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.run(["make", "-C", _HERE], check=True)
        L = C.CDLL(LIB_PATH)
        L.pto_prepare.restype = C.c_void_p
        L.pto_prepare.argtypes = [C.POINTER(PtoScene), C.c_int]
        L.pto_release.argtypes = [C.c_void_p]
        L.pto_render.restype = C.c_int
        L.pto_render.argtypes = [C.c_void_p, C.POINTER(PtFrameParams), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.POINTER(PtoStats), C.c_int]
        L.pto_render_history.restype = C.c_int
        L.pto_render_history.argtypes = [C.c_void_p, C.POINTER(PtFrameParams), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PtoStats), C.c_int]
        L.pto_denoiser.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.pto_taa.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pto_accumulate.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int64]
        L.pto_rand_sequence.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pto_camera_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pto_trace_ray.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        L.pto_sky_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_path_dump.restype = C.c_int
        L.pto_path_dump.argtypes = [C.c_void_p, C.POINTER(PtFrameParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.pto_find_camera_ray.argtypes = [C.POINTER(PtCameraHistory), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.pto_bilinear_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.pto_display_denoise.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.pto_fmath.restype = C.c_float
        L.pto_fmath.argtypes = [C.c_int, C.c_float, C.c_float]
        _lib = L
    return _lib


class Oracle:
    """A prepared scene. accel=False is the reference's exhaustive loop; accel=True
    uses the oracle's own BVH (same winner, checked against accel=False in tests)."""

    def __init__(self, spheres, objects, vertices, indices, sky=None, accel=False, materials=None):
        self._keep = [np.ascontiguousarray(spheres, np.float32).reshape(-1, 14), np.ascontiguousarray(objects),
                      np.ascontiguousarray(vertices, np.float32).reshape(-1, 3), np.ascontiguousarray(indices, np.int32)]
        s, o, v, i = self._keep
        sc = PtoScene()
        sc.spheres = s.ctypes.data if len(s) else None; sc.n_spheres = len(s)
        sc.objects = o.ctypes.data if len(o) else None; sc.n_objects = len(o)
        sc.vertices = v.ctypes.data if len(v) else None; sc.n_vertices = len(v)
        sc.indices = i.ctypes.data if len(i) else None; sc.n_indices = len(i)
        if sky is not None:
            k = np.ascontiguousarray(sky, np.float32)
            self._keep.append(k)
            sc.sky = k.ctypes.data; sc.sky_h, sc.sky_w = k.shape[:2]
        if materials is not None and len(materials):
            m = np.ascontiguousarray(materials)
            assert m.dtype.itemsize == 48 and len(m) == len(o)
            self._keep.append(m)
            sc.materials = m.ctypes.data; sc.n_materials = len(m)
        self._p = lib().pto_prepare(C.byref(sc), int(bool(accel)))
        if not self._p:
            raise MemoryError("pto_prepare failed")

    @classmethod
    def from_scene(cls, scene, sky=None, accel=False, materials=None):
        return cls(scene.spheres, scene.objects, scene.vertices, scene.indices, sky, accel, materials)

    def __del__(self):
        try:
            if self._p:
                lib().pto_release(self._p)
                self._p = None
        except Exception:
            pass

    def render(self, params, width, height, threads=1, out=None):
        img = np.zeros((height, width, 4), np.float32) if out is None else out
        ids = np.full((height, width), -1, np.int32)
        st = PtoStats()
        rc = lib().pto_render(self._p, C.byref(params), width, height, img.ctypes.data, ids.ctypes.data, C.byref(st), threads)
        if rc != 0:
            raise ValueError("pto_render: bad arguments")
        return img, ids, {k: getattr(st, k) for k, _ in PtoStats._fields_}

    def render_history(self, params, history, width, height, result, result1, threads=1, result2=None):
        """Any renderMode with the camera history `history` (list of PtCameraHistory).  `result` (previous frame on
        entry, new frame on return), `result1` (the history image) and, for renderMode 0, 7 and 8, `result2` (the
        accumulated radiance / the frame being assembled) are (H, W, 4) float32 arrays updated in place."""
        for a in (result, result1) + ((result2,) if result2 is not None else ()):
            assert a.dtype == np.float32 and a.flags.c_contiguous and a.shape == (height, width, 4)
        ids = np.full((height, width), -1, np.int32)
        st = PtoStats()
        arr = (PtCameraHistory * max(1, len(history)))(*history)
        rc = lib().pto_render_history(self._p, C.byref(params), C.byref(arr), len(history), width, height, result.ctypes.data,
                                      result1.ctypes.data, result2.ctypes.data if result2 is not None else None,
                                      ids.ctypes.data, C.byref(st), threads)
        if rc != 0:
            raise ValueError("pto_render_history: bad arguments")
        return ids, {k: getattr(st, k) for k, _ in PtoStats._fields_}

    def trace_rays(self, origins_dirs):
        od = np.ascontiguousarray(origins_dirs, np.float32).reshape(-1, 6)
        ids = np.empty(len(od), np.int32); ts = np.empty(len(od), np.float32)
        i = C.c_int32(); t = C.c_float()
        for k in range(len(od)):
            lib().pto_trace_ray(self._p, od[k].ctypes.data, C.byref(i), C.byref(t))
            ids[k] = i.value; ts[k] = t.value
        return ids, ts

    def path_dump(self, params, width, height, x, y, cap=4096):
        log = np.zeros((cap, 8), np.float32)
        n = lib().pto_path_dump(self._p, C.byref(params), width, height, x, y, log.ctypes.data, cap)
        return log[:n]

    def sky_sample(self, d):
        d = np.ascontiguousarray(d, np.float32); out = np.zeros(3, np.float32)
        lib().pto_sky_sample(self._p, d.ctypes.data, out.ctypes.data)
        return out


def accumulate(mode, sample_index, fresh, converged):
    f = np.ascontiguousarray(fresh, np.float32); c = np.ascontiguousarray(converged, np.float32).copy()
    lib().pto_accumulate(mode, sample_index, f.ctypes.data, c.ctypes.data, f.size // 4)
    return c


def rand_sequence(seed, idx, idy, n):
    out = np.zeros(n, np.float32)
    lib().pto_rand_sequence(seed, idx, idy, n, out.ctypes.data)
    return out


def camera_ray(c2w, ip, idx, idy, w, h):
    c2w = np.ascontiguousarray(c2w, np.float32); ip = np.ascontiguousarray(ip, np.float32)
    out = np.zeros(6, np.float32)
    lib().pto_camera_ray(c2w.ctypes.data, ip.ctypes.data, idx, idy, w, h, out.ctypes.data)
    return out


def fmath(which, a, b=0.0):
    return float(lib().pto_fmath({"sin": 0, "cos": 1, "pow": 2, "acos": 3, "atan2": 4, "log2": 5, "exp2": 6, "exp": 7, "frac": 8}[which], a, b))


def find_camera_ray(old_camera, width, height, direction):
    d = np.ascontiguousarray(direction, np.float32); out = np.zeros(2, np.float32)
    lib().pto_find_camera_ray(C.byref(old_camera), width, height, d.ctypes.data, out.ctypes.data)
    return out


def bilinear_sample(result1, px, py):
    img = np.ascontiguousarray(result1, np.float32); out = np.zeros(4, np.float32)
    lib().pto_bilinear_sample(img.ctypes.data, img.shape[1], img.shape[0], px, py, out.ctypes.data)
    return out


def denoiser(result1, x, y, seed):
    img = np.ascontiguousarray(result1, np.float32); out = np.zeros(4, np.float32)
    lib().pto_denoiser(img.ctypes.data, img.shape[1], img.shape[0], x, y, seed, out.ctypes.data)
    return out


def taa(result1, current, x, y):
    a = np.ascontiguousarray(result1, np.float32); b = np.ascontiguousarray(current, np.float32); out = np.zeros(4, np.float32)
    lib().pto_taa(a.ctypes.data, b.ctypes.data, a.shape[1], a.shape[0], x, y, out.ctypes.data)
    return out


def display_denoise(result, threads=4):
    img = np.ascontiguousarray(result, np.float32); out = np.zeros_like(img)
    lib().pto_display_denoise(img.ctypes.data, img.shape[1], img.shape[0], out.ctypes.data, threads)
    return out
