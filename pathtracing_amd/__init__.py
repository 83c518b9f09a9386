"""pathtracing_amd - MI355X-native path-tracing hot path of VirtualEngineeringLab/Pathtracing.

The product is csrc/libptrace.so (HIP kernels for gfx950 behind the C ABI of
include/ptrace.h) and host/libptmaster.so (C++ mirror of the reference's C#
host).  This package is the ctypes harness over both.
"""
from ._ffi import (PT_OK, PT_ERR_INVALID, PT_ERR_HIP, PT_ERR_NOMEM, PT_ERR_UNSUPPORTED, PT_LOBE_GGX, PT_LOBE_PHONG, PtMaterial, PT_ACCUM_ADDMAT, PT_ACCUM_MEAN, PT_ACCUM_NONE, PT_ACCUM_SUM, PT_TARGET_CONVERGED,
                   PT_TARGET_DENOISED, PT_TARGET_DISPLAY, PT_TARGET_HIT_IDS, PT_TARGET_RESULT, PtCameraHistory, PtFrameParams, PtMeshObject, PtSphere, PtStats)
from .scenes import (Scene, materials, camera_history_entry, camera_matrices, frame_seed, make_params, matrix_inverse, load_hdr, procedural_sky,
                     stereo_eye_matrices, stereo_history_entry)
from .tracer import PathTracer, PathTracerGroup, PtraceError

__all__ = ["PathTracer", "PathTracerGroup", "PtraceError", "Scene", "camera_matrices", "camera_history_entry", "frame_seed", "make_params", "procedural_sky", "load_hdr",
           "PtFrameParams", "PtCameraHistory", "PtMeshObject", "PtSphere", "PtStats", "PT_ACCUM_NONE", "PT_ACCUM_MEAN", "PT_ACCUM_SUM",
           "PT_ACCUM_ADDMAT", "PT_TARGET_RESULT", "PT_TARGET_CONVERGED", "PT_TARGET_HIT_IDS", "PT_TARGET_DENOISED", "PT_TARGET_DISPLAY", "PtMaterial", "PT_LOBE_PHONG", "PT_LOBE_GGX", "PT_OK", "PT_ERR_INVALID", "PT_ERR_HIP", "PT_ERR_NOMEM", "PT_ERR_UNSUPPORTED", "materials",
           "matrix_inverse", "stereo_eye_matrices", "stereo_history_entry"]
