/*
 * ptmaster.h - host side above the libptrace C ABI: a C++ mirror of the
 * reference's C# host (Assets/Scripts/RayTracingMaster.cs, RayTracingObject.cs)
 * and the synthetic scenes the benchmarks and tests render.  Everything here is
 * plain CPU code; only ptm_master_* touches the GPU, and only through
 * include/ptrace.h.
 *
 * The C functions below exist so that the Python test/bench harness can drive
 * the C++ classes through ctypes.
 */
#ifndef PTMASTER_H
#define PTMASTER_H

#include <stdint.h>
#include "../../include/ptrace.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scene pieces (CPU only) ------------------------------------------- */

/* SetUpScene (RayTracingMaster.cs:315-369) with a PCG32 stream in place of
 * UnityEngine.Random (engine-internal, not reproducible).  Returns the number
 * of spheres written (<= capacity, <= spheres_max). */
int ptm_sphere_field(int sphere_seed, uint32_t spheres_max, float radius_min, float radius_max,
                     float placement_radius, PtSphere *out, int capacity);

/* Deterministic lat-long sky (vertical gradient + sun), RGBA32F, w*h texels. */
void ptm_sky_procedural(int w, int h, float *rgba);

/* A Radiance picture (.hdr, 32-bit_rle_rgbe) as _SkyboxTexture: what the reference's scene binds
 * (Assets/HDRI/cape_hill_4k.hdr, RayTracingMaster.cs:25,457; import setting maxTextureSize 2048,
 * cape_hill_4k.hdr.meta:34).  The picture is halved until neither side exceeds max_size (0: as it is) and stored
 * bottom row first, RGBA32F, as pt_set_sky takes it.  Call with rgba = NULL to learn *w and *h, then with a buffer of
 * capacity_texels >= w*h.  Returns 0, or -1 with the reason in ptm_sky_error() (thread-local). */
int ptm_sky_load_hdr(const char *path, int max_size, int *w, int *h, float *rgba, int capacity_texels);
const char *ptm_sky_error(void);

/* Camera.cameraToWorldMatrix and Camera.projectionMatrix.inverse
 * (RayTracingMaster.cs:489-490) for a camera at `eye` looking at `target`:
 * OpenGL convention, camera looks down -Z. */
void ptm_camera_look_at(const float eye[3], const float target[3], const float up[3], float fov_y_deg,
                        float aspect, float z_near, float z_far, float cam_to_world[16], float cam_inv_proj[16]);

/* The same camera as one element of the shader's history arrays: adds Camera.worldToCameraMatrix and
 * Camera.projectionMatrix (RayTracingMaster.cs:522-525). */
void ptm_camera_history_entry(const float eye[3], const float target[3], const float up[3], float fov_y_deg,
                              float aspect, float z_near, float z_far, PtCameraHistory *out);

/* Per-frame _Seed (replaces Random.value, RayTracingMaster.cs:530): frac(0.5 + k * 0.6180339). */
float ptm_frame_seed(uint32_t k);

/* Matrix4x4.inverse (RayTracingMaster.cs:463,466,490) for a column-major 4x4; returns 0 for a singular matrix. */
int ptm_matrix_inverse(const float m[16], float out[16]);

/* One eye of a stereo camera, as Camera.GetStereoViewMatrix / GetStereoProjectionMatrix hand it to the stereo branch
 * of SetShaderParameters (RayTracingMaster.cs:463-466): the view matrix of the look-at camera moved `offset` along its
 * right axis, and an OpenGL projection whose frustum is sheared sideways by `shear` (an asymmetric, per-eye frustum). */
void ptm_camera_stereo_eye(const float eye[3], const float target[3], const float up[3], float fov_y_deg, float aspect,
                           float z_near, float z_far, float offset, float shear, float view[16], float proj[16]);

/* ---- named synthetic scenes --------------------------------------------
 * "spheres"   Kuri part-3 field: 100 candidates, radius U(3,8), placement radius 100 (RayTracingMaster.cs:32-34)
 * "xr"        the repository's TestScene: 200 candidates, radius U(.5,3), placement 20, one rotated cube
 * "blob70k"   ~70 k-triangle displaced blob + small sphere field
 * "hall260k"  ~260 k-triangle colonnade hall (columns, arches, bumpy walls, statues)
 * "city10m"   ~10 M triangles: the hall instanced on a grid through MeshObject matrices
 * "fixture"   20 spheres + a 320-triangle blob, for brute-force parity tests
 */
typedef struct ptm_scene ptm_scene;
ptm_scene *ptm_scene_create(const char *name);
void ptm_scene_destroy(ptm_scene *s);
/* counts[0..3] = spheres, mesh objects, vertices, indices */
void ptm_scene_counts(const ptm_scene *s, int32_t counts[4]);
void ptm_scene_copy(const ptm_scene *s, PtSphere *spheres, PtMeshObject *objects, float *vertices, int32_t *indices);
/* default camera of the scene: eye, target, fov_y_deg */
void ptm_scene_camera(const ptm_scene *s, float eye[3], float target[3], float *fov_y_deg);

/* Wavefront OBJ (v / f records in every index form, polygons fan-triangulated) appended to a scene: one mesh object per
 * "o" / "g" group that has faces.  Returns the index of the first object added, or < 0 (-1 unreadable, -2 bad index, -3 no faces). */
int ptm_scene_add_obj(ptm_scene *s, const char *path, const float local_to_world[16]);

/* ---- RayTracingMaster mirror ---------------------------------------------- */
typedef struct ptm_master ptm_master;
/* One master = one camera/eye = one libptrace context on `device`. */
int ptm_master_create(int device, ptm_master **out);
void ptm_master_destroy(ptm_master *m);
const char *ptm_master_error(const ptm_master *m);
pt_context *ptm_master_context(ptm_master *m);
/* OnEnable + scene upload: spheres and registered meshes of `s`, procedural sky of sky_w x sky_h. */
int ptm_master_load_scene(ptm_master *m, const ptm_scene *s, int sky_w, int sky_h);
/* SkyboxTexture from a Radiance picture (see ptm_sky_load_hdr; max_size 2048 is the asset's import setting). */
int ptm_master_set_sky_hdr(ptm_master *m, const char *path, int max_size);
/* Inspector fields (RayTracingMaster.cs:73-91, 686, 991) */
int ptm_master_configure(ptm_master *m, int width, int height, int samples_per_pixel, int max_bounces,
                         int depth, int accumulate, uint32_t sample_frames);
/* RayTracingMaster.renderMode (RayTracingMaster.cs:977-991): any value of the RenderMode enum (0 .. 8, 10, 11, 12). */
int ptm_master_set_render_mode(ptm_master *m, int render_mode);
/* RayTracingMaster.divisions (RayTracingMaster.cs:701): stripes per frame cycle in renderMode 5 .. 8. */
int ptm_master_set_divisions(ptm_master *m, int divisions);
/* eye_offset: signed offset along the camera's right axis (stereo: -sep/2 left, +sep/2 right). */
int ptm_master_set_camera(ptm_master *m, const float eye[3], const float target[3], float fov_y_deg, float eye_offset);
/* renderScale + GetRenderScale (RayTracingMaster.cs:86,585-597): the render size becomes Mathf.RoundToInt(screen size *
 * renderScale) in place of the width / height of ptm_master_configure.  All targets get that size; the reference gives
 * _converged and denoisedTex the screen size instead (:563-571), which differs for renderScale != 1 only
 * (TestScene.unity:783 has 1): at such a scale ptm_master_render renders renderMode 1, which reads neither, and returns
 * PT_ERR_UNSUPPORTED for every other renderMode. */
int ptm_master_set_render_scale(ptm_master *m, int screen_width, int screen_height, float render_scale);
void ptm_master_render_size(const ptm_master *m, int32_t size[2]);
/* RayTracingMaster.RegisterObject / UnregisterObject / UpdateObject (RayTracingMaster.cs:296-313), which the reference's
 * RayTracingObject component calls from OnEnable, OnDisable and on a changed transform (RayTracingObject.cs:24-38).  A mesh
 * is object-space vertices (xyz) and the indices of its sub-mesh 0; register returns the object's position in the list
 * (-1 on a bad mesh), unregister moves later objects up.  register_instance adds an object sharing an earlier object's
 * mesh.  All take effect at the next ptm_master_render(), which rebuilds the mesh buffers as :371-413 do: a full upload when
 * the object list changed, the matrices alone when only transforms did. */
int ptm_master_register_object(ptm_master *m, const float *vertices, int n_vertices, const int32_t *indices, int n_indices,
                               const float local_to_world[16]);
int ptm_master_register_instance(ptm_master *m, int of_object, const float local_to_world[16]);
int ptm_master_unregister_object(ptm_master *m, int object);
int ptm_master_update_object(ptm_master *m, int object, const float local_to_world[16]);
int ptm_master_object_count(const ptm_master *m);
/* _camera.stereoEnabled, stereoTargetEye == Left, and that eye's view and projection matrices: selects the stereo branch
 * of SetShaderParameters (RayTracingMaster.cs:459-485) - _leftEye, inverted per-eye matrices, and its own rule for when
 * the camera lists restart (:473).  enabled == 0 goes back to the mono branch (:486-527). */
int ptm_master_set_stereo(ptm_master *m, int enabled, int left_eye, const float view[16], const float proj[16]);
/* Render(): RebuildMeshObjectBuffers, SetShaderParameters, InitRenderTexture, Dispatch (RayTracingMaster.cs:704-747),
 * then the _currentSample bookkeeping of :885-888.  Asynchronous. */
int ptm_master_render(ptm_master *m);
/* The PtFrameParams the next ptm_master_render() would upload (for the oracle to render the same frame). */
int ptm_master_peek_params(ptm_master *m, PtFrameParams *out);
/* _currentSample = 0 (RayTracingMaster.cs:145,753) */
void ptm_master_reset_accumulation(ptm_master *m);
uint32_t ptm_master_current_sample(const ptm_master *m);
uint32_t ptm_master_frame_index(const ptm_master *m);

#ifdef __cplusplus
}
#endif
#endif
