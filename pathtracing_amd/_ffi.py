"""ctypes bindings of libptrace (include/ptrace.h) and libptmaster (host/ptmaster.h).

There is no fallback: if a library is missing the import raises, naming the
build command.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PTRACE_PATH = os.path.join(_HERE, "csrc", "libptrace.so")
PTMASTER_PATH = os.path.join(_HERE, "host", "libptmaster.so")


class PtSphere(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("radius", C.c_float), ("albedo", C.c_float * 3),
                ("specular", C.c_float * 3), ("smoothness", C.c_float), ("emission", C.c_float * 3)]


class PtMeshObject(C.Structure):
    _fields_ = [("local_to_world", C.c_float * 16), ("indices_offset", C.c_int32), ("indices_count", C.c_int32)]


class PtFrameParams(C.Structure):
    _fields_ = [("camera_to_world", C.c_float * 16), ("camera_inverse_projection", C.c_float * 16),
                ("seed", C.c_float), ("samples_per_pixel", C.c_int32), ("max_bounces", C.c_int32),
                ("depth", C.c_int32), ("render_mode", C.c_int32), ("sample_index", C.c_uint32),
                ("accumulate", C.c_int32), ("tile_x0", C.c_int32), ("tile_y0", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("pixel_offset", C.c_float * 2),
                ("directional_light", C.c_float * 4), ("accumulation_frames", C.c_int32),
                ("divisions", C.c_int32), ("counter", C.c_int32), ("left_eye", C.c_int32)]


class PtCameraHistory(C.Structure):
    _fields_ = [("camera_to_world", C.c_float * 16), ("camera_inverse_projection", C.c_float * 16),
                ("world_to_camera", C.c_float * 16), ("camera_projection", C.c_float * 16)]


class PtMaterial(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("smoothness", C.c_float), ("specular", C.c_float * 3), ("lobe", C.c_int32),
                ("emission", C.c_float * 3), ("reserved", C.c_float)]


class PtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("paths", C.c_uint64), ("kernel_ms", C.c_float), ("bvh_nodes", C.c_int32),
                ("bvh_prims", C.c_int32), ("bvh_depth", C.c_int32), ("bvh_build_ms", C.c_float)]


assert C.sizeof(PtSphere) == 56 and C.sizeof(PtMeshObject) == 72 and C.sizeof(PtMaterial) == 48

PT_LOBE_PHONG, PT_LOBE_GGX = 0, 1
PT_OK, PT_ERR_INVALID, PT_ERR_HIP, PT_ERR_NOMEM, PT_ERR_UNSUPPORTED = 0, 1, 2, 3, 4      # include/ptrace.h:39-43

PT_ACCUM_NONE, PT_ACCUM_MEAN, PT_ACCUM_SUM, PT_ACCUM_ADDMAT = 0, 1, 2, 3
PT_TARGET_RESULT, PT_TARGET_CONVERGED, PT_TARGET_HIT_IDS, PT_TARGET_DENOISED, PT_TARGET_DISPLAY = 0, 1, 2, 3, 4

_P = C.POINTER
_ctx = C.c_void_p

# name -> (restype, argtypes): every entry point include/ptrace.h declares
PTRACE_SYMBOLS = {
    "pt_create": (C.c_int, [C.c_int, _P(_ctx)]),
    "pt_destroy": (None, [_ctx]),
    "pt_last_error": (C.c_char_p, [_ctx]),
    "pt_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "pt_set_option": (C.c_int, [_ctx, C.c_char_p, C.c_int]),
    "pt_set_spheres": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "pt_set_meshes": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "pt_set_mesh_materials": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "pt_update_transforms": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "pt_set_sky": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "pt_set_camera_history": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "pt_resize": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "pt_bind_target": (C.c_int, [_ctx, C.c_int, C.c_void_p]),
    "pt_render": (C.c_int, [_ctx, _P(PtFrameParams)]),
    "pt_display_denoise": (C.c_int, [_ctx]),
    "pt_sync": (C.c_int, [_ctx]),
    "pt_read_result": (C.c_int, [_ctx, C.c_void_p]),
    "pt_read_converged": (C.c_int, [_ctx, C.c_void_p]),
    "pt_write_converged": (C.c_int, [_ctx, C.c_void_p]),
    "pt_read_target": (C.c_int, [_ctx, C.c_int, C.c_void_p]),
    "pt_write_target": (C.c_int, [_ctx, C.c_int, C.c_void_p]),
    "pt_read_hit_ids": (C.c_int, [_ctx, C.c_void_p]),
    "pt_get_stats": (C.c_int, [_ctx, _P(PtStats)]),
    "pt_trace_rays": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pt_trace_rays_static": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pt_allocation_calls": (C.c_ulonglong, []),
    "pt_version": (C.c_char_p, []),
    "pt_group_create": (C.c_int, [C.c_int, _P(C.c_int), _P(C.c_void_p)]),
    "pt_group_destroy": (None, [C.c_void_p]),
    "pt_group_size": (C.c_int, [C.c_void_p]),
    "pt_group_context": (C.c_void_p, [C.c_void_p, C.c_int]),
    "pt_group_reduce": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_group_last_error": (C.c_char_p, [C.c_void_p]),
}

PTMASTER_SYMBOLS = {
    "ptm_sphere_field": (C.c_int, [C.c_int, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int]),
    "ptm_sky_procedural": (None, [C.c_int, C.c_int, C.c_void_p]),
    "ptm_camera_look_at": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_void_p, C.c_void_p]),
    "ptm_camera_history_entry": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                        _P(PtCameraHistory)]),
    "ptm_frame_seed": (C.c_float, [C.c_uint32]),
    "ptm_matrix_inverse": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ptm_camera_stereo_eye": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_void_p, C.c_void_p]),
    "ptm_master_register_object": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "ptm_master_register_instance": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ptm_master_unregister_object": (C.c_int, [C.c_void_p, C.c_int]),
    "ptm_master_update_object": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ptm_master_object_count": (C.c_int, [C.c_void_p]),
    "ptm_master_set_render_scale": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float]),
    "ptm_master_render_size": (None, [C.c_void_p, C.c_void_p]),
    "ptm_master_set_stereo": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ptm_scene_create": (C.c_void_p, [C.c_char_p]),
    "ptm_scene_destroy": (None, [C.c_void_p]),
    "ptm_scene_counts": (None, [C.c_void_p, C.c_void_p]),
    "ptm_scene_copy": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptm_scene_camera": (None, [C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_float)]),
    "ptm_scene_add_obj": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "ptm_sky_load_hdr": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_int]),
    "ptm_sky_error": (C.c_char_p, []),
    "ptm_master_set_sky_hdr": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "ptm_master_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "ptm_master_destroy": (None, [C.c_void_p]),
    "ptm_master_error": (C.c_char_p, [C.c_void_p]),
    "ptm_master_context": (C.c_void_p, [C.c_void_p]),
    "ptm_master_load_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "ptm_master_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]),
    "ptm_master_set_render_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "ptm_master_set_divisions": (C.c_int, [C.c_void_p, C.c_int]),
    "ptm_master_set_camera": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]),
    "ptm_master_render": (C.c_int, [C.c_void_p]),
    "ptm_master_peek_params": (C.c_int, [C.c_void_p, _P(PtFrameParams)]),
    "ptm_master_reset_accumulation": (None, [C.c_void_p]),
    "ptm_master_current_sample": (C.c_uint32, [C.c_void_p]),
    "ptm_master_frame_index": (C.c_uint32, [C.c_void_p]),
}


def _load(path, symbols, what):
    if not os.path.exists(path):
        raise ImportError(f"{what} not built: {path} is missing. Build it with `python -m pathtracing_amd.build` "
                          "(needs hipcc); there is no CPU fallback for the render path.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_ptrace = None
_ptmaster = None


def ptrace():
    global _ptrace
    if _ptrace is None:
        # PTRACE_LIB: another build of libptrace.so (a measurement build, an A/B candidate) for this process only, so that
        # comparisons never have to overwrite the installed library.  libptmaster still links against the installed one.
        path = os.environ.get("PTRACE_LIB") or PTRACE_PATH
        if not os.path.isabs(path):
            path = os.path.join(_HERE, "csrc", path)
        _ptrace = _load(path, PTRACE_SYMBOLS, "libptrace")
    return _ptrace


def ptmaster():
    global _ptmaster
    if _ptmaster is None:
        ptrace()  # libptmaster links against it
        _ptmaster = _load(PTMASTER_PATH, PTMASTER_SYMBOLS, "libptmaster")
    return _ptmaster
