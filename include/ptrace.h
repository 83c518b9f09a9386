/*
 * ptrace.h - C ABI of libptrace: the path-tracing compute kernel of
 * VirtualEngineeringLab/Pathtracing (Assets/Scripts/RayTracingShader.compute,
 * kernel CSMain in every renderMode the host can select: plain path tracing,
 * depth-aware temporal reprojection and its blur / headset / partial-frame /
 * full-frame / pause variants, reprojection + a-trous denoiser + temporal
 * anti-aliasing) plus progressive accumulation
 * (Assets/Shaders/AddMat.shader), implemented as HIP kernels for gfx950.
 *
 * Each entry point replaces one call the reference's C# host
 * (Assets/Scripts/RayTracingMaster.cs) makes on UnityEngine.ComputeShader /
 * ComputeBuffer / RenderTexture; the replaced call is cited beside it.
 *
 * Conventions
 *   - every function returns 0 on success, a PT_ERR_* code otherwise, and never
 *     throws; pt_last_error() gives the text of the last failure of a context
 *     (or of pt_create when passed NULL);
 *   - `const` host pointers are copied during the call (ComputeBuffer.SetData
 *     semantics); the caller may free them on return;
 *   - a context belongs to one device and is not thread-safe (Unity drives its
 *     ComputeShader from the main thread only); different contexts may be used
 *     from different threads;
 *   - pt_render() enqueues on the context's HIP stream and returns; pt_sync()
 *     and the pt_read_* calls wait for it;
 *   - matrices are 16 floats in UnityEngine.Matrix4x4 memory order
 *     (column-major: element (row i, col j) at [j*4+i]);
 *   - images are RGBA float32, row-major, pixel (x,y) at [(y*W + x)*4], y as
 *     SV_DispatchThreadID.y (RayTracingShader.compute:726);
 *   - no C++ exception leaves the library: a host allocation that fails inside an entry point comes back as
 *     PT_ERR_NOMEM with a message (pt_last_error), and the context stays usable.
 */
#ifndef PTRACE_H
#define PTRACE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_ERR_INVALID 1   /* bad argument / bad state           */
#define PT_ERR_HIP 2       /* a HIP runtime call (or the host runtime underneath the library) failed */
#define PT_ERR_NOMEM 3     /* host or device allocation failed   */
#define PT_ERR_UNSUPPORTED 4

typedef struct pt_context pt_context;

/* struct Sphere, RayTracingMaster.cs:100-108 / RayTracingShader.compute:61-69; stride 56 (RayTracingMaster.cs:366). */
typedef struct PtSphere {
    float position[3];
    float radius;
    float albedo[3];
    float specular[3];
    float smoothness;
    float emission[3];
} PtSphere;

/* struct MeshObject, RayTracingMaster.cs:93-98 / RayTracingShader.compute:77-82; stride 72 (RayTracingMaster.cs:410). */
typedef struct PtMeshObject {
    float local_to_world[16];
    int32_t indices_offset;
    int32_t indices_count;
} PtMeshObject;

/* Surface of a mesh object.  The shader hard-codes one for every triangle (albedo 0, specular .65,
 * smoothness .99, emission 0: RayTracingShader.compute:250-253), which is what applies until
 * pt_set_mesh_materials supplies others; the fields are the ones RayHit carries (:126-135).
 * lobe picks the specular sampling: PT_LOBE_PHONG is the shader's (:345-353, modified Phong with
 * alpha = 1000^(smoothness^2)); PT_LOBE_GGX is an extension with no counterpart in the reference:
 * a GGX microfacet lobe of roughness (1 - smoothness)^2 with Smith shadowing and Schlick Fresnel
 * (F0 = specular), sampled through its normal distribution.  The diffuse branch is the shader's. */
enum { PT_LOBE_PHONG = 0, PT_LOBE_GGX = 1 };
typedef struct PtMaterial {
    float albedo[3];
    float smoothness;
    float specular[3];
    int32_t lobe;
    float emission[3];
    float reserved;
} PtMaterial;

/* Accumulation of the fresh frame into the converged image. */
enum {
    PT_ACCUM_NONE = 0,   /* no accumulation (converged = the previous Result, as the shader leaves it) */
    PT_ACCUM_MEAN = 1,   /* converged = converged*(1 - 1/(S+1)) + fresh/(S+1), all four channels   */
    PT_ACCUM_SUM = 2,    /* converged += fresh (S == 0: converged = fresh); for sample sharding     */
    PT_ACCUM_ADDMAT = 3  /* AddMat.shader:44-63 with its SrcAlpha/OneMinusSrcAlpha blend (:11)      */
};

/* Which render target a device pointer is bound to (pt_bind_target). */
enum {
    PT_TARGET_RESULT = 0,     /* "Result"  (RayTracingMaster.cs:740)  RGBA32F W*H */
    PT_TARGET_CONVERGED = 1,  /* "Result1" (RayTracingMaster.cs:741)  RGBA32F W*H */
    PT_TARGET_HIT_IDS = 2,    /* int32 W*H primary-hit ids; no counterpart in the reference (parity aid) */
    PT_TARGET_DENOISED = 3,   /* "Result2" (RayTracingMaster.cs:743, denoisedTex)  RGBA32F W*H; renderMode 0 */
    PT_TARGET_DISPLAY = 4     /* the `destination` of the display blit (RayTracingMaster.cs:764)  RGBA32F W*H   */
};

/* Per-dispatch uniforms: what SetShaderParameters() (RayTracingMaster.cs:455-540)
 * and Render() (:735-738) upload before Dispatch. */
typedef struct PtFrameParams {
    float camera_to_world[16];           /* _CameraToWorld            RayTracingMaster.cs:464,489 */
    float camera_inverse_projection[16]; /* _CameraInverseProjection  RayTracingMaster.cs:467,490 */
    float seed;                          /* _Seed                     RayTracingMaster.cs:530     */
    int32_t samples_per_pixel;           /* _SamplesPerPixel          RayTracingMaster.cs:531     */
    int32_t max_bounces;                 /* path segments per sample; the shader hard-codes 4
                                            (RayTracingShader.compute:866-873); 1..4095          */
    int32_t depth;                       /* _Depth                    RayTracingMaster.cs:532     */
    int32_t render_mode;                 /* renderMode (RayTracingMaster.cs:735,977-990): 0 = Denoise, 1 = Default,
                                            2 = Reproj, 3 = BlurAndReproj, 4 = StereoReproj, 5 = PartialFrameReproj,
                                            6 = PartialFrameReprojDepth, 7 = FullFrameReproj, 8 = FullFrameReprojDepth,
                                            10 = NewRender, 11 = PlanerPause, 12 = DepthPause; anything else is
                                            PT_ERR_UNSUPPORTED */
    uint32_t sample_index;               /* _Sample of AddMat.shader:42 (RayTracingMaster.cs:805) */
    int32_t accumulate;                  /* PT_ACCUM_*                                            */
    int32_t tile_x0, tile_y0;            /* sub-rectangle of the image this dispatch renders;     */
    int32_t tile_w, tile_h;              /*   tile_w == 0 or tile_h == 0 means the whole image    */
    float pixel_offset[2];               /* _PixelOffset      uploaded, unused (RayTracingShader.compute:21,842) */
    float directional_light[4];          /* _DirectionalLight uploaded, unused (RayTracingShader.compute:19)    */
    int32_t accumulation_frames;         /* _AccumulationFrames (RayTracingMaster.cs:738); renderMode 0 and 2   */
    int32_t divisions;                   /* _Divisions (RayTracingMaster.cs:736) and                            */
    int32_t counter;                     /* _Counter = counter++ % divisions (:737); renderMode 5 .. 8 only     */
    int32_t left_eye;                    /* _leftEye (RayTracingMaster.cs:462); renderMode 4 only               */
} PtFrameParams;

/* One element of each of the four history arrays _CameraToWorldOld,
 * _CameraInverseProjectionOld, _WorldToCameraOld, _CameraProjectionOld
 * (RayTracingShader.compute:9-16; filled from RayTracingMaster.cs:450-453,520-526). */
typedef struct PtCameraHistory {
    float camera_to_world[16];
    float camera_inverse_projection[16];
    float world_to_camera[16];
    float camera_projection[16];
} PtCameraHistory;
#define PT_MAX_CAMERA_HISTORY 100        /* the arrays' declared length */

/* Counters over every pt_render() issued since the previous pt_get_stats()
 * (or since pt_create); pt_get_stats() waits for those renders. */
typedef struct PtStats {
    uint64_t rays;            /* Trace() calls executed by the kernels                      */
    uint64_t nodes_visited;   /* BVH nodes fetched                                          */
    uint64_t prims_tested;    /* triangle + sphere intersection tests                       */
    uint64_t paths;           /* camera rays generated (pixels * samples)                   */
    float kernel_ms;          /* device time from the first of those renders' start to the
                                 last one's end (HIP events on the context's stream)        */
    int32_t bvh_nodes;        /* scene statistics                                           */
    int32_t bvh_prims;
    int32_t bvh_depth;
    float bvh_build_ms;       /* host time of the last BVH build                            */
} PtStats;

/* Create / destroy a context on HIP device `device_ordinal` (replaces the
 * ComputeShader instance + its GPU resources; released like
 * RayTracingMaster.cs:154-161). */
int pt_create(int device_ordinal, pt_context **out_ctx);
void pt_destroy(pt_context *ctx);
const char *pt_last_error(const pt_context *ctx);

/* Run on an existing hipStream_t (e.g. the caller's current stream) instead of
 * the context's own.  NULL is HIP's null stream, as everywhere in HIP;
 * PT_STREAM_OWN goes back to the stream the context created for itself.  The stream must stay valid
 * for as long as it is set: pt_set_stream and pt_destroy wait for the work the context has on it. */
#define PT_STREAM_OWN ((void *)(intptr_t)-1)
int pt_set_stream(pt_context *ctx, void *hip_stream);

/* Implementation switches; no counterpart in the reference.
 *   "backend"    0 (default): one persistent kernel whose waves vote between node, leaf and shade phases;
 *                1: wavefront stages - generate / persistent trace / shade+compact+regenerate (renderMode 1 only);
 *                2: the persistent kernel with a pool of rays per wave (more lanes per instruction, not faster).
 *   "bvh_width"  4 (default): four-child nodes with 8-bit boxes; 2: two-child nodes with float boxes; 8: eight-child
 *                nodes with their children in octant order, walked by (first child, hit mask) groups - a third fewer
 *                node visits per ray, each dearer: 0.88x - 1.0x the rays per second of the four-child tree.
 *   "block_order"  1 (default): a dispatch hands its 8x8-pixel blocks out most expensive first, by the cost the last
 *                dispatch over the same rectangle measured; 0: row by row.
 *   "fast_intrinsics"  0 (default): Shade() on the correctly rounded functions of include/ptrace_fmath.h, bit for bit the
 *                oracle's image.  1: reciprocal, square root, exp2, log2, sine and cosine of Shade() on the hardware's
 *                approximate instructions, short polynomials for acos and atan2 - renderMode 1 only.  The random number
 *                stream, and with it every path's length and the primary hit ids, stay as they are; the image differs
 *                from the exact one by a relative RMSE below 5e-3 (measured 7e-4 to 1.6e-3); +3 % to +11 % rays per second.
 *                The only option that changes the image.
 *   "tree_build"  Who builds the tree over whatever is not a copy of a shared mesh ("instancing").  0: the host's
 *                binned-SAH builder (sixteen threads: 75-95 ms for 262 k triangles, 4.3 s for ten million).  1: the device -
 *                Morton sort, the same top-down binned SAH (passes over the whole array while a node has more than 128
 *                primitives, then one wave per node in LDS), collapse, boxes by the refit kernels: 5 ms and 28 ms for the
 *                same scenes, and the same tree quality (14.96 against 14.97 node visits per ray on the first).
 *                2 (default): the device for scenes of 3 072 primitives or more, the host below.  Large triangles
 *                are cut into references ("split_percent") in a device build as in a host build.
 *                3: the device, with the two-child tree made by locally-ordered clustering from
 *                the leaves up - a quarter less build time on the largest scenes (21 ms for ten million), a fifth more
 *                node visits per ray.
 *                Four-child tree only; a tree too deep for a lane's traversal stack is handed to the host builder.
 *   "rebuild_percent"  30 (default): pt_update_transforms refits, which keeps the topology of the last build - after
 *                large motion the boxes overlap and rays visit more nodes.  The summed node area of a refitted
 *                device-built tree is taken on the device after every refit and read by a later pt_render without
 *                waiting for it; when it exceeds the freshly built tree's by this many per cent, that render builds
 *                anew (on the device: milliseconds).  0: never.  Trees built by the host are only ever refitted.
 *   "debug_throw"  a test hook: 1 throws std::bad_alloc, 2 std::runtime_error from inside the call, which must come back
 *                as PT_ERR_NOMEM / PT_ERR_HIP like any exception under an entry point; 0 does nothing.
 *   "blocks_per_cu"  0 (default): the persistent kernel takes every resident-block slot of the GPU; n > 0: at most
 *                n blocks per compute unit, which leaves room for another context's dispatch to run alongside
 *                (two eyes of a stereo pair on one GPU: 2 each).
 *   "reuse_primary_hit"  0 (default): every sample traces its camera ray, as the shader does; 1: a pixel's camera ray
 *                is traced once per dispatch and its hit reused for the pixel's other samples - the shader applies no
 *                jitter (RayTracingShader.compute:842), so the ray and its hit are the same every time.  Same image;
 *                PtStats.rays then counts only the rays actually traced.
 *   "refit"      1 (default): pt_update_transforms refits the four-wide tree on the device; 0: it rebuilds on the host.
 *   "instancing"  What becomes of mesh objects with the same object-space triangles.  RebuildMeshObjectBuffers
 *                (RayTracingMaster.cs:387-408) gives every object its own copy of its mesh's vertices and indices, so
 *                objects sharing a sharedMesh are recognised by the content of their ranges (equal indices relative to
 *                the range's smallest, bit-identical positions), as are objects handed one common range; meshes used
 *                once and meshes under 64 triangles are always flattened.  The image is the same in all three.
 *                2 (default): the shared mesh is built ONCE on the host, as a tree over its object-space triangles;
 *                the device makes a copy of that tree per object and refits each copy under its object's matrix.
 *                What the kernels walk is an ordinary flattened world-space tree (any backend, any localToWorld),
 *                at the flattened tree's speed, that the host never built: 38 copies of 262 k triangles in 82 ms
 *                instead of 4.2 s, and pt_update_transforms refits it like any flattened scene (1 ms there).  When
 *                the shared meshes add up to more than a hundredth of the scene and "tree_build" allows it, the whole
 *                scene is built flat on the device instead, which is then quicker (that scene: 33 ms).
 *                1: the objects - if placed by rotation or mirror x uniform scale + translation; sheared or unevenly
 *                scaled ones are flattened - share ONE tree on the device too, under a top-level tree with a box per
 *                object; triangle tests still run on world-space vertices transformed as the shader transforms them.
 *                Device memory falls with the number of copies (2.6 M -> 69 k nodes there), tracing is slower (0.55x -
 *                0.75x); scheduled kernel only; pt_update_transforms rebuilds the top-level tree on the host (1.2 ms).
 *                0: every mesh object is flattened into world-space triangle records on the host.
 *                Four-child tree only: "bvh_width" 2 and 8 flatten on the host.
 *   "split_percent"  30 (default): before the tree is built, triangles that span a sixteenth of the scene or more are cut
 *                into several references (boxes holding a part of the triangle each; a leaf still names the whole
 *                triangle), at most this many additional references per hundred triangles of the scene; 0: none.  A
 *                scene of evenly small triangles is left as it is; long beams and wall-sized triangles stop inflating
 *                every node above them.  A refit keeps the cuts of objects that did not move.
 * All combinations but "fast_intrinsics" 1 produce the same image bit for bit. */
int pt_set_option(pt_context *ctx, const char *name, int value);

/* _Spheres: new ComputeBuffer(count,56).SetData + SetBuffer (RayTracingMaster.cs:366-367,536).
 * count == 0 is valid (buffer not bound, RayTracingMaster.cs:445). */
int pt_set_spheres(pt_context *ctx, const PtSphere *spheres, int count);

/* _MeshObjects/_Vertices/_Indices: CreateComputeBuffer x3 + SetBuffer
 * (RayTracingMaster.cs:410-412,537-539).  vertices = n_vertices*3 floats, object
 * space; indices already offset by each mesh's first vertex
 * (RayTracingMaster.cs:399).  Transforms every triangle to world space once and
 * (re)builds the acceleration structure.  n_objects == 0 is valid. */
int pt_set_meshes(pt_context *ctx, const PtMeshObject *objects, int n_objects,
                  const float *vertices, int n_vertices,
                  const int32_t *indices, int n_indices);

/* One PtMaterial per mesh object of the last pt_set_meshes, in the same order; count 0 goes back to
 * the shader's hard-coded surface.  No counterpart in the reference (its MeshObject has no material). */
int pt_set_mesh_materials(pt_context *ctx, const PtMaterial *materials, int count);

/* Same geometry, new localToWorld matrices (what a transform change costs the
 * reference: a full re-upload, RayTracingMaster.cs:254-262,371-413).  On a scene
 * that has been rendered this sends only the matrices and refits the
 * acceleration structure on the device, ordered on the context's stream; the
 * tree keeps the topology of the last full build (pt_set_meshes, or option
 * "refit" 0), so large motion costs traversal speed, never correctness. */
int pt_update_transforms(pt_context *ctx, const PtMeshObject *objects, int n_objects);

/* _SkyboxTexture (RayTracingMaster.cs:457): RGBA float32, w*h texels, row 0 at
 * v = 0; sampled bilinearly with Repeat wrap at mip 0 (RayTracingShader.compute:331). */
int pt_set_sky(pt_context *ctx, const float *rgba, int width, int height);

/* SetMatrixArray("_CameraInverseProjectionOld" / "_CameraToWorldOld" / "_WorldToCameraOld" /
 * "_CameraProjectionOld") (RayTracingMaster.cs:468-471,492-495): the cameras of earlier frames,
 * entry 0 first.  Like a write into the shader's uniform arrays, a call overwrites elements
 * 0..count-1 and leaves the others as they were (the partial-frame modes rely on that: early in a
 * cycle the tail still holds the previous cycle's cameras; the pause modes, whose host list is
 * empty, rely on count == 0 changing nothing).  All elements start as zero matrices - renderMode 2
 * then finds no valid reprojection and writes the fresh frame, as on the reference's first frame.
 * entries == NULL with count == PT_MAX_CAMERA_HISTORY puts every element back to zero (no counterpart
 * in the reference).  count 0..PT_MAX_CAMERA_HISTORY. */
int pt_set_camera_history(pt_context *ctx, const PtCameraHistory *entries, int count);

/* InitRenderTexture (RayTracingMaster.cs:542-583): allocate Result / converged
 * / hit-id targets of width*height, zero-filled. */
int pt_resize(pt_context *ctx, int width, int height);

/* SetTexture(0, "Result"/"Result1", rt) (RayTracingMaster.cs:740-741) for a
 * target the CALLER owns in device memory (width*height*16 bytes; hit ids
 * width*height*4).  NULL re-binds the context's own buffer. */
int pt_bind_target(pt_context *ctx, int which, void *device_ptr);

/* SetShaderParameters + Dispatch(0, ceil(W/32), ceil(H/32), 1)
 * (RayTracingMaster.cs:716,745-747).
 *
 * renderMode 1 (RayTracingShader.compute:728-731,822-874,933-936): plain path tracing into Result.  Like the shader,
 * the dispatch first copies the previous content of Result into Result1 (the converged target) over the dispatched
 * rectangle - renderMode 1 never reads it, a later frame in renderMode 5..8, 11 or 12 does.  With accumulate !=
 * PT_ACCUM_NONE the converged target holds the accumulation instead and that copy is not made.
 *
 * renderMode 2 (RayTracingShader.compute:728-731,950-983): Result1 - the converged target - first
 * receives the previous content of Result over the dispatched rectangle, as one copy that is complete
 * before any pixel is traced; every pixel then blends its fresh radiance with the bilinear sample of
 * Result1 at the position its primary hit had in the previous camera (history entry 0), provided that
 * position is on screen and its stored distance agrees within 5 %.  The shader does that copy per
 * thread inside the same dispatch, so which of its threads see old or new history is a race there;
 * the whole-image copy is the ordering in which no thread sees another thread's new output.
 * accumulate must be PT_ACCUM_NONE (the converged target is the history).
 *
 * renderMode 0 (RayTracingShader.compute:733-736,885-930; the reference's start-up mode,
 * RayTracingMaster.cs:991): Result1 first receives Result2 (the previous frame's accumulated radiance);
 * the path tracer's reprojected radiance goes to Result2 (as renderMode 2 writes Result, plus the
 * thread-group row limit of :908-915); then Result = denoiser(Result1) (:479-615) for every pixel, then
 * Result = TAA(Result1, Result) (:644-714) for every pixel.  The shader runs all of it per thread without
 * a barrier; here each step is complete before the next starts.  Whole frames only.
 *
 * renderMode 5 / 6 (RayTracingShader.compute:742-789,933-936,1108-1112): the image is cut into
 * `divisions` horizontal stripes; a dispatch path-traces stripe `counter` (rows y with
 * H - s*(counter+1) <= y <= H - s*counter, s = H / divisions) exactly as renderMode 1 does and then
 * copies it into Result1; every other pixel takes the bilinear sample of Result1 at the position where
 * the camera that rendered ITS stripe (history element divisions - y/s - 1) saw the point 10 units along
 * the pixel's current camera ray (mode 5) or the pixel's current primary hit (mode 6; a primary miss
 * shows the sky), or (0,0,0,1) when that position is off screen.  Result1 is read as it was before the
 * dispatch.  Whole frames only; accumulate must be PT_ACCUM_NONE.
 *
 * renderMode 3 (:986-1019) and 4 (:1021-1046) are renderMode 2 with another blend: 3 mixes the fresh radiance
 * (1/2.5) with the bilinear history sample (1/5) and the history's four diagonal neighbours (1/10 each) and
 * tests the distance against the unfiltered texel; 4 finds the old position with FindCameraRayVR (:418-437:
 * the headset's 2240/2016 aspect ratio and per-eye pixel offsets, `left_eye`).
 *
 * renderMode 10 (:1048-1069): a pixel whose primary hit is found in the history (on screen, distance within
 * 5 % or beyond 80) shows the bilinear history sample, alpha included, and traces no path; every other pixel
 * is path-traced as in renderMode 1.  The host sets samples_per_pixel to 1 from this mode on
 * (RayTracingMaster.cs:531).
 *
 * renderMode 11 / 12 (:1071-1104; "pause"): no path is traced and Result1 is left alone; every pixel shows
 * the bilinear sample of Result1 where history camera 0 saw the point 10 units along the pixel's ray (11) or
 * the pixel's primary hit (12; a miss gives (0,0,0,1), as does a position off screen).  Whole frames only.
 *
 * renderMode 7 / 8 (:751-789,850-853,938-948): like 5 / 6 the frame is rendered one stripe per dispatch, but
 * into Result2 and from a history camera - element counter-1, the previous dispatch's camera - while EVERY
 * pixel of Result shows Result1 reprojected from history camera 0 (the camera of the cycle's first dispatch)
 * through the point 10 units along its ray (7) or its primary hit (8; a miss shows the sky and, in the
 * stripe, is not path-traced).  The dispatch with counter 0 path-traces the whole frame and then copies
 * Result2 into Result1 and Result.  Its camera index, (0-1) % divisions = -1, addresses the uniform declared
 * just before each array (:8-11): the current camera.  History rays are built by CreateCameraRayOld
 * (:374-387: clip z = 1).  Whole frames only. */
int pt_render(pt_context *ctx, const PtFrameParams *params);

/* Graphics.Blit(_target, destination, denoiser) (RayTracingMaster.cs:760-764, the imageBlur switch):
 * Assets/Shaders/Denoiser.shader over Result into the display target.  The shader samples through tex2D;
 * here that is bilinear filtering with float weights, clamp addressing and texel centres at (i + 0.5) / size,
 * and the fixed4 output stays float.  Asynchronous, ordered after the renders issued so far. */
int pt_display_denoise(pt_context *ctx);

int pt_sync(pt_context *ctx);

/* Read-backs (the reference never reads back on the live path; its unused
 * helper is toTexture2D, RayTracingMaster.cs:599-607). */
int pt_read_result(pt_context *ctx, float *rgba_out);
int pt_read_converged(pt_context *ctx, float *rgba_out);
int pt_write_converged(pt_context *ctx, const float *rgba_in);
/* The same for any RGBA target (PT_TARGET_RESULT / _CONVERGED / _DENOISED). */
int pt_read_target(pt_context *ctx, int which, float *rgba_out);
int pt_write_target(pt_context *ctx, int which, const float *rgba_in);
int pt_read_hit_ids(pt_context *ctx, int32_t *ids_out);

int pt_get_stats(pt_context *ctx, PtStats *out);

/* Trace caller-supplied rays (origin xyz, direction xyz per ray) through the
 * scene with the kernels' traversal, or (brute_force != 0) with a plain loop
 * over every primitive on the device in the reference's visiting order
 * (RayTracingShader.compute:262-285).  Outputs per ray: hit id (-1 miss,
 * 0 ground plane, 1+i sphere i, 1+n_spheres+k triangle k) and distance
 * (+inf on miss).  No counterpart in the reference; exists so the
 * acceleration structure can be checked against the exhaustive loop. */
int pt_trace_rays(pt_context *ctx, const float *origins_dirs, int n_rays,
                  int brute_force, int32_t *ids_out, float *t_out);

/* pt_trace_rays for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results as pt_trace_rays,
 * but the call makes no hipMalloc / hipFree: the device copies of the rays and of the results and the traversal
 * stack's overflow area live in one array of 11.5 MB (352 bytes per ray) that is part of the library's device image -
 * every GPU the library runs on holds it until the process ends.  Limits: n_rays <= PT_TRACE_RAYS_STATIC_MAX
 * (PT_ERR_INVALID beyond); the scene must have been built by an earlier pt_render or pt_trace_rays and not changed
 * since, and no ray may start farther from it than that build allowed for (PT_ERR_INVALID says which; the remedy is
 * one ordinary pt_trace_rays call, which rebuilds).  All contexts on a GPU share the array, so calls on one GPU - from
 * different contexts or threads - must not overlap: a call that finds the array in use returns PT_ERR_INVALID ("in
 * use") instead of overwriting the other call's rays; calls on different GPUs do not meet. */
#define PT_TRACE_RAYS_STATIC_MAX 32768
int pt_trace_rays_static(pt_context *ctx, const float *origins_dirs, int n_rays,
                         int brute_force, int32_t *ids_out, float *t_out);

/* Test and debugging aid, no counterpart in the reference: how many device or pinned-host allocation and release calls
 * (hipMalloc, hipFree, hipHostMalloc, hipHostFree) the library has made in this process so far, all contexts together.
 * Unchanged across a call means the call allocated nothing - which free-memory figures cannot show when a buffer is
 * allocated and released inside one call. */
unsigned long long pt_allocation_calls(void);

const char *pt_version(void);

/* Several devices from one host thread.  The reference's host is a single Unity process
 * (RayTracingMaster.cs drives one ComputeShader per eye camera, :47-52 share the scene lists between
 * them); a group gives such a host one context per device: pt_group_create makes them, the caller
 * uploads the scene to each (pt_group_context(i) + the pt_set_* calls) and splits the work by eye, by
 * sample index (accumulate = PT_ACCUM_SUM, one _Seed per sample index) or by rows (tile_*), and
 * pt_group_reduce adds every member's converged target into the root's, member by member in index
 * order (so the float sum is reproducible), ordered after the renders issued so far on each member and
 * before any later render of a member overwrites its image; the call itself does not wait.  Eye and
 * row splits need no reduce.  No counterpart in the reference, which renders on one GPU.  Processes that own one
 * device each use their own communication layer instead (bench.py: torch.distributed over RCCL). */
typedef struct pt_group pt_group;
int pt_group_create(int n, const int *device_ordinals, pt_group **out_group);
void pt_group_destroy(pt_group *group);
int pt_group_size(const pt_group *group);
pt_context *pt_group_context(pt_group *group, int index);
int pt_group_reduce(pt_group *group, int root);
const char *pt_group_last_error(const pt_group *group);

#ifdef __cplusplus
}
#endif
#endif /* PTRACE_H */
