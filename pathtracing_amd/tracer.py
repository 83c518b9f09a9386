"""PathTracer: a libptrace context (one per GPU / eye), driven through the C ABI."""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import PtFrameParams, PtStats


class PtraceError(RuntimeError):
    pass


class PathTracer:
    def __init__(self, device=0, _borrowed_ctx=None):
        self._lib = _ffi.ptrace()
        self._owned = _borrowed_ctx is None
        self.width = self.height = 0
        if not self._owned:                      # a member of a PathTracerGroup: the group destroys it
            self._ctx = C.c_void_p(_borrowed_ctx)
            return
        self._ctx = C.c_void_p()
        rc = self._lib.pt_create(int(device), C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.pt_last_error(None).decode()
            self._ctx = None
            raise PtraceError(f"pt_create({device}) failed ({rc}): {msg}")

    def _check(self, rc, what):
        if rc != 0:
            raise PtraceError(f"{what} failed ({rc}): {self._lib.pt_last_error(self._ctx).decode()}")

    def close(self):
        if getattr(self, "_ctx", None):
            if self._owned:
                self._lib.pt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_option(self, name, value):
        self._check(self._lib.pt_set_option(self._ctx, name.encode(), int(value)), "pt_set_option")

    # ---- scene ------------------------------------------------------------
    def set_spheres(self, spheres):
        s = np.ascontiguousarray(spheres, np.float32).reshape(-1, 14)
        self._check(self._lib.pt_set_spheres(self._ctx, s.ctypes.data if len(s) else None, len(s)), "pt_set_spheres")

    def set_meshes(self, objects, vertices, indices):
        o = np.ascontiguousarray(objects)
        assert o.dtype.itemsize == 72 or len(o) == 0
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, np.int32)
        self._check(self._lib.pt_set_meshes(self._ctx, o.ctypes.data if len(o) else None, len(o),
                                            v.ctypes.data if len(v) else None, len(v),
                                            i.ctypes.data if len(i) else None, len(i)), "pt_set_meshes")

    def set_mesh_materials(self, materials):
        """materials: structured array of 48-byte PtMaterial records (scenes.materials), one per mesh object; None or
        empty goes back to the shader's hard-coded surface."""
        m = None if materials is None else np.ascontiguousarray(materials)
        n = 0 if m is None else len(m)
        assert n == 0 or m.dtype.itemsize == 48
        self._check(self._lib.pt_set_mesh_materials(self._ctx, m.ctypes.data if n else None, n), "pt_set_mesh_materials")

    def update_transforms(self, objects):
        o = np.ascontiguousarray(objects)
        self._check(self._lib.pt_update_transforms(self._ctx, o.ctypes.data if len(o) else None, len(o)), "pt_update_transforms")

    def set_sky(self, sky):
        if sky is None:
            self._check(self._lib.pt_set_sky(self._ctx, None, 0, 0), "pt_set_sky")
            return
        s = np.ascontiguousarray(sky, np.float32)
        h, w = s.shape[:2]
        self._check(self._lib.pt_set_sky(self._ctx, s.ctypes.data, w, h), "pt_set_sky")

    def load_scene(self, scene, sky=None):
        self.set_spheres(scene.spheres)
        self.set_meshes(scene.objects, scene.vertices, scene.indices)
        self.set_sky(sky)

    def set_camera_history(self, entries):
        """entries: sequence of PtCameraHistory, the cameras of earlier frames, most recent upload order as the host keeps it."""
        from ._ffi import PtCameraHistory
        arr = (PtCameraHistory * max(1, len(entries)))(*entries)
        self._check(self._lib.pt_set_camera_history(self._ctx, C.byref(arr), len(entries)), "pt_set_camera_history")

    def clear_camera_history(self):
        """Every element of the four history arrays back to zero, the state of a new context."""
        self._check(self._lib.pt_set_camera_history(self._ctx, None, 100), "pt_set_camera_history")

    # ---- targets / render ---------------------------------------------------
    def resize(self, width, height):
        self._check(self._lib.pt_resize(self._ctx, int(width), int(height)), "pt_resize")
        self.width, self.height = int(width), int(height)

    def bind_target(self, which, device_ptr):
        self._check(self._lib.pt_bind_target(self._ctx, int(which), C.c_void_p(device_ptr)), "pt_bind_target")

    def set_stream(self, stream_ptr):
        """stream_ptr: a hipStream_t as an integer (0 = HIP's null stream); None = the context's own stream."""
        ptr = C.c_void_p(-1) if stream_ptr is None else C.c_void_p(stream_ptr)
        self._check(self._lib.pt_set_stream(self._ctx, ptr), "pt_set_stream")

    def render(self, params):
        self._check(self._lib.pt_render(self._ctx, C.byref(params)), "pt_render")

    def display_denoise(self):
        """Denoiser.shader over Result into the display target (read it with read_target(PT_TARGET_DISPLAY))."""
        self._check(self._lib.pt_display_denoise(self._ctx), "pt_display_denoise")

    def sync(self):
        self._check(self._lib.pt_sync(self._ctx), "pt_sync")

    def read_result(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._lib.pt_read_result(self._ctx, out.ctypes.data), "pt_read_result")
        return out

    def read_converged(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._lib.pt_read_converged(self._ctx, out.ctypes.data), "pt_read_converged")
        return out

    def write_converged(self, img):
        a = np.ascontiguousarray(img, np.float32)
        assert a.shape == (self.height, self.width, 4)
        self._check(self._lib.pt_write_converged(self._ctx, a.ctypes.data), "pt_write_converged")

    def read_target(self, which):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._lib.pt_read_target(self._ctx, int(which), out.ctypes.data), "pt_read_target")
        return out

    def write_target(self, which, img):
        a = np.ascontiguousarray(img, np.float32)
        assert a.shape == (self.height, self.width, 4)
        self._check(self._lib.pt_write_target(self._ctx, int(which), a.ctypes.data), "pt_write_target")

    def read_hit_ids(self):
        out = np.empty((self.height, self.width), np.int32)
        self._check(self._lib.pt_read_hit_ids(self._ctx, out.ctypes.data), "pt_read_hit_ids")
        return out

    def stats(self):
        st = PtStats()
        self._check(self._lib.pt_get_stats(self._ctx, C.byref(st)), "pt_get_stats")
        return {k: getattr(st, k) for k, _ in PtStats._fields_}

    def trace_rays(self, origins_dirs, brute_force=False):
        od = np.ascontiguousarray(origins_dirs, np.float32).reshape(-1, 6)
        ids = np.empty(len(od), np.int32); ts = np.empty(len(od), np.float32)
        self._check(self._lib.pt_trace_rays(self._ctx, od.ctypes.data, len(od), int(bool(brute_force)),
                                            ids.ctypes.data, ts.ctypes.data), "pt_trace_rays")
        return ids, ts

    def trace_rays_static(self, origins_dirs, brute_force=False):
        """pt_trace_rays_static: the same query without any device allocation (at most PT_TRACE_RAYS_STATIC_MAX rays)."""
        od = np.ascontiguousarray(origins_dirs, np.float32).reshape(-1, 6)
        ids = np.empty(len(od), np.int32); ts = np.empty(len(od), np.float32)
        self._check(self._lib.pt_trace_rays_static(self._ctx, od.ctypes.data, len(od), int(bool(brute_force)),
                                                   ids.ctypes.data, ts.ctypes.data), "pt_trace_rays_static")
        return ids, ts


class PathTracerGroup:
    """pt_group_*: one context per listed device, driven from this thread; `members[i]` is a PathTracer."""

    def __init__(self, devices):
        self._lib = _ffi.ptrace()
        self._grp = C.c_void_p()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._lib.pt_group_create(len(devices), arr, C.byref(self._grp))
        if rc != 0:
            msg = self._lib.pt_group_last_error(None).decode()
            self._grp = None
            raise PtraceError(f"pt_group_create({list(devices)}) failed ({rc}): {msg}")
        self.members = [PathTracer(_borrowed_ctx=self._lib.pt_group_context(self._grp, i))
                        for i in range(self._lib.pt_group_size(self._grp))]

    def reduce(self, root=0):
        rc = self._lib.pt_group_reduce(self._grp, int(root))
        if rc != 0:
            raise PtraceError(f"pt_group_reduce failed ({rc}): {self._lib.pt_group_last_error(self._grp).decode()}")

    def close(self):
        if getattr(self, "_grp", None):
            for m in self.members:
                m.close()
            self._lib.pt_group_destroy(self._grp)
            self._grp = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
