/*
 * pt_oracle.h - CPU oracle for libptrace.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar restatement, in plain C, of kernel CSMain of
 * Assets/Scripts/RayTracingShader.compute in every renderMode the host can
 * select (RayTracingMaster.cs:977-990), of Assets/Shaders/Denoiser.shader and of
 * Assets/Shaders/AddMat.shader, written line by line from those files.  It is
 * the checker the HIP kernels are compared with; nothing in the product links
 * or calls it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer
 * vectors, and its implementation (HLSL run by the Unity editor, driven by C#)
 * cannot be built or run where this was written.  The oracle is therefore
 * pinned only by analytic known-answer tests (tests/test_oracle_kat.py) and by
 * reading it against the cited shader lines; HLSL intrinsics are taken from
 * include/ptrace_fmath.h, which fixes the precision the shader leaves open.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>
#include "../include/ptrace.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtoScene {
    const PtSphere *spheres;      int32_t n_spheres;
    const PtMeshObject *objects;  int32_t n_objects;
    const float *vertices;        int32_t n_vertices;
    const int32_t *indices;       int32_t n_indices;
    const float *sky;             int32_t sky_w, sky_h;   /* RGBA32F; NULL = black sky */
    const PtMaterial *materials;  int32_t n_materials;    /* one per object, or 0 = the shader's hard-coded surface */
} PtoScene;

typedef struct PtoStats {
    uint64_t trace_calls;        /* every Trace() the shader would execute                         */
    uint64_t rays;               /* the same minus the re-trace of an unchanged ray after a primary
                                    miss (RayTracingShader.compute:866-870), which cannot change the
                                    image; this is the count the HIP kernels report                */
    uint64_t paths;              /* camera rays (pixels * samples)                                 */
    uint64_t prims_tested;       /* sphere + triangle tests                                        */
} PtoStats;

/* Opaque prepared scene: world-space triangles (and, with accel != 0, the
 * oracle's own median-split BVH). accel == 0 is the reference's exhaustive
 * loop over every primitive (RayTracingShader.compute:262-285). */
typedef struct PtoPrepared PtoPrepared;
PtoPrepared *pto_prepare(const PtoScene *scene, int accel);
void pto_release(PtoPrepared *p);

/* CSMain, plain mode, over the tile named in `params` (whole image when the
 * tile is empty).  result: W*H*4 floats, only tile pixels are written.
 * hit_ids (may be NULL): W*H primary-hit ids.  threads: OpenMP threads (rows
 * are independent); <= 0 means 1. */
int pto_render(const PtoPrepared *p, const PtFrameParams *params, int width, int height,
               float *result, int32_t *hit_ids, PtoStats *stats, int threads);

/* The same for every renderMode (0 .. 8, 10, 11, 12); `history` = the four *Old camera arrays, elements
 * n_history .. 99 zero.  renderMode 3, 4 and 10 treat `result` / `result1` as renderMode 2 does; 5 / 6 copy the
 * rendered stripe into `result1` last; 7 / 8 write their stripe to `result2` and, when counter is 0, copy the
 * finished pixels of `result2` to `result1` and `result` once every read of `result1` is done; 11 / 12 leave
 * `result1` alone.  Reads of Result1 always see it as it was when the dispatch started (after the prologue copy).
 * renderMode 2 (RayTracingShader.compute:728-731,950-983): `result` holds the previous frame on entry;
 * `result1` (W*H*4) first receives a copy of it over the dispatched rectangle - the whole copy before any
 * pixel is traced - and is the history the pixels reproject from; history[0] is entry 0 of the four *Old
 * camera arrays (zeros when n_history == 0).
 * renderMode 0 (:733-736,885-930): `result1` first receives `result2` (the accumulated radiance of the
 * previous frame); the traced, reprojected radiance of every pixel goes to `result2`; then every pixel of
 * `result` = denoiser(Result1) (:479-615), then every pixel of `result` = TAA (:644-714) of Result1 and the
 * denoised image.  Whole frames only.  The shader runs these steps per thread without any barrier, so
 * which neighbours a thread sees finished is a race there; this is the ordering in which every step is
 * complete before the next one starts. */
int pto_render_history(const PtoPrepared *p, const PtFrameParams *params, const PtCameraHistory *history, int n_history,
                       int width, int height, float *result, float *result1, float *result2, int32_t *hit_ids,
                       PtoStats *stats, int threads);

/* AddMat.shader / running mean / sum: same modes as PtFrameParams.accumulate. */
void pto_accumulate(int mode, uint32_t sample_index, const float *fresh, float *converged, int64_t n_pixels);

/* Pieces exposed for known-answer tests. */
void pto_rand_sequence(float seed, int id_x, int id_y, int n, float *out);           /* SH:47-55,831 */
void pto_camera_ray(const float *cam_to_world, const float *cam_inv_proj,
                    int id_x, int id_y, int width, int height, float *origin_dir6); /* SH:108-120,842 */
void pto_trace_ray(const PtoPrepared *p, const float *origin_dir6, int32_t *id_out, float *t_out);
void pto_sky_sample(const PtoPrepared *p, const float *dir3, float *rgb_out);        /* SH:324-332 */
/* Every Trace() call of one pixel in order, 8 floats each: origin, direction, hit id, distance. Returns the count. */
int pto_path_dump(const PtoPrepared *p, const PtFrameParams *params, int width, int height, int id_x, int id_y,
                  float *log8, int capacity);
void pto_find_camera_ray(const PtCameraHistory *old_camera, int width, int height, const float *dir3, float *uv2); /* SH:394-411 */
void pto_bilinear_sample(const float *result1, int width, int height, float px, float py, float *out4);            /* SH:440-463 */
void pto_denoiser(const float *result1, int width, int height, int x, int y, float seed, float *out4);             /* SH:479-615 */
void pto_taa(const float *result1, const float *current, int width, int height, int x, int y, float *out4);        /* SH:644-714 */
/* Assets/Shaders/Denoiser.shader over a whole image (the display blit, RayTracingMaster.cs:760-764). */
void pto_display_denoise(const float *result, int width, int height, float *display, int threads);
float pto_fmath(int which, float a, float b); /* 0 sin 1 cos 2 pow 3 acos 4 atan2 5 log2 6 exp2 7 exp 8 frac */

#ifdef __cplusplus
}
#endif
#endif
