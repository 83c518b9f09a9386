// libptrace: C ABI (include/ptrace.h) over the HIP kernels in pt_device.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <tuple>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <vector>

#include "../../include/ptrace.h"
#include "bvh.hpp"
#include "pt_device.hpp"
#include "pt_wavefront.hpp"
#include "pt_pool.hpp"
#include "pt_screen.hpp"
#include "pt_refit.hpp"
#include "pt_build.hpp"
#include "pt_build_sah.hpp"

using namespace ptrace;

// Every device or pinned-host allocation and release the library makes goes through these four, which count the calls
// (pt_allocation_calls): a test can then see that an entry point which promises to allocate nothing made none - free
// memory before and after would not show a hipMalloc that was followed by its hipFree.
namespace {
std::atomic<unsigned long long> g_allocation_calls{0};
std::atomic<unsigned long long> g_allocation_ns{0};          // time inside hipMalloc / hipFree (reported by PTRACE_BUILD_TIMING)
struct AllocationClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocationClock() { g_allocation_ns += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
inline hipError_t dev_malloc(void **p, size_t bytes) { g_allocation_calls++; AllocationClock c; return hipMalloc(p, bytes); }
inline hipError_t dev_free(void *p) { g_allocation_calls++; AllocationClock c; return hipFree(p); }
inline hipError_t host_malloc(void **p, size_t bytes, unsigned flags) { g_allocation_calls++; return hipHostMalloc(p, bytes, flags); }
inline hipError_t host_free(void *p) { g_allocation_calls++; return hipHostFree(p); }
}  // namespace

// The tree over one shared mesh's object-space triangles (two-level scenes), kept between builds: moving instances
// then costs the top-level tree only.  Valid while the geometry is the one of the last pt_set_meshes and the padding it was
// built with still covers what a build asks for.
struct CachedMesh {
    int32_t indices_offset = 0, indices_count = 0, leaf_size = 0;
    uint64_t version = 0;                 // a new number with every build of this tree: what the device copy is checked against
    float pad = 0.0f, reach = 0.0f;       // object-space box padding and intersection reach the boxes were made for
    BvhResult bvh;
    std::vector<float4> recs;             // (v0, triangle index within the mesh) (v1, 0) (v2, 0) in tree order
    Aabb root;                            // union of the padded boxes
};

struct pt_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    // host copies of the scene inputs (ComputeBuffer contents)
    std::vector<PtSphere> spheres;
    std::vector<PtMeshObject> objects;
    std::vector<float> vertices;
    std::vector<int32_t> indices;
    bool scene_dirty = true;
    std::vector<PtMaterial> materials;             // pt_set_mesh_materials; empty = the shader's hard-coded surface
    float4 *d_mesh_mat = nullptr;
    std::vector<PtCameraHistory> camera_history = std::vector<PtCameraHistory>(PT_MAX_CAMERA_HISTORY);   // the four *Old arrays; zero-initialised
    // device copies for the partial-frame modes: a small ring, so that an upload never waits for the frame
    // that still reads the previous copy (pinned staging buffer + device buffer + "last reader done" event each)
    static constexpr int kCamRing = 4;
    PtCameraHistory *h_cams[kCamRing] = {}, *d_cams[kCamRing] = {};
    hipEvent_t ev_cams[kCamRing] = {};
    bool cams_used[kCamRing] = {};
    int cam_slot = 0;
    bool cams_dirty = true;

    // refit state (pt_refit.hpp): what pt_update_transforms needs to redo boxes without a host build
    int refit = 1;                                 // option "refit"
    bool refit_ready = false;                      // the buffers below describe the current tree
    int4 *d_slot_src = nullptr;
    float *d_vertices = nullptr, *d_matrices = nullptr, *d_prim_box = nullptr, *d_node_box = nullptr;
    float *h_matrices = nullptr; size_t h_matrices_objects = 0;   // pinned staging for pt_update_transforms (free again once ev_refit has passed)
    int32_t *d_level_nodes = nullptr;
    std::vector<int> level_first;                  // level_first[l] .. level_first[l+1]: nodes of level l, deepest level first
    float box_pad = 0.0f, sphere_slack2 = 0.0f;    // what the last full build padded boxes with
    // Sphere boxes are inflated for rays whose origin lies within `sphere_reach` of the sphere (see rebuild_scene); a camera
    // farther away than that asks for a rebuild with a larger reach (ensure_scene).
    float sphere_reach = 0.0f, sphere_reach_request = 0.0f;
    float scene_lo[3] = {0, 0, 0}, scene_hi[3] = {0, 0, 0};   // finite bounds of the scene: last build, or estimate after a refit
    float spheres_lo[3] = {0, 0, 0}, spheres_hi[3] = {0, 0, 0};
    std::vector<Aabb> object_box;                  // object-space bounds of every mesh object (for that estimate)
    hipEvent_t ev_refit = nullptr;
    bool refit_in_flight = false;
    bool bvh2_stale = false;                       // a refit updated only the four-wide tree

    // device scene
    uint4 *d_nodes4 = nullptr;
    uint4 *d_nodes8 = nullptr;        // the eight-wide tree (option bvh_width 8); the records then follow its order, and the other trees are not kept
    int n_nodes8 = 0, stack_need8 = 0;
    bool walk8 = false;               // the last build made the eight-wide tree and a lane's stack holds it
    float4 *d_nodes = nullptr, *d_prims = nullptr, *d_sphere_mat = nullptr, *d_sky = nullptr;
    int n_nodes4 = 0, bvh_width = 4;
    int stack_need4 = 0;                           // most stack entries a ray can hold in the four-wide tree of the last build
    int stack_limit = kStackLds + kStackOverflow;  // entries a lane's stack holds; PTRACE_STACK_LIMIT lowers it (tests of the fallback)
    int n_nodes = 0, n_prims = 0, n_spheres = 0, sky_w = 0, sky_h = 0;
    int n_top_records = 0;            // records of the top-level tree (n_slots plus one per instance)
    int n_slots = 0;                  // primitive records on the device: n_prims plus the extra references of split triangles
    float split_budget = 0.3f;        // option "split_percent": additional references, per triangle of the scene, the build may spend on large triangles
    float *d_slot_clip = nullptr;     // per slot, the unpadded box of its reference (only when some triangle was split)
    float *d_build_matrices = nullptr;   // the matrices those boxes were cut under
    // two-level scenes: mesh objects that share a mesh and are placed by a rotation, a uniform scale and a translation keep
    // one tree over the mesh's object-space triangles; the top-level tree holds a box per such object
    int instancing = 2;               // option "instancing": 0 flatten on the host, 1 two-level tree, 2 (default) shared meshes copied and refitted on the device
    int n_inst = 0;                   // instances in the current build (0: everything flattened)
    bool expanded = false;            // the last build copied shared trees on the device (instancing 2): no two-child tree exists
    // what a transform update of such a scene re-runs: the table of copies, per shared mesh its levels (deepest first) and
    // its copies' range in the table, and the top-level tree's levels
    struct ExpandedGroup { int32_t *d_level_nodes = nullptr; std::vector<int> level_first; size_t first_inst = 0, n_inst = 0; };
    int4 *d_exp_inst = nullptr;
    std::vector<ExpandedGroup> exp_groups;
    int32_t *d_exp_top_levels = nullptr;
    std::vector<int> exp_top_first;
    bool flatten_once = false;        // the two-level tree of the last attempt could overflow a lane's stack: build flat
    // option "tree_build": 0 the host's binned-SAH builder; 1 the device builder (pt_build.hpp; top-down SAH, pt_build_sah.hpp) for
    // whatever is not a copy of a shared mesh; 2 (default) the device builder for scenes of kDeviceBuildMinPrims primitives or
    // more; 3 the device builder with its two-child tree made by
    // clustering - half the build time again, a fifth more node visits per ray
    int tree_build = 2;
    bool device_built = false;        // the last build ran on the device: a four-wide tree only, like `expanded`
    // option "rebuild_percent": a refit keeps the topology of the last build, so large motion leaves boxes that overlap; when
    // the summed node area of a refitted device-built tree exceeds that of the build by this much, the next render builds
    // anew (milliseconds on the device).  The sum is taken on the device after every refit and read without waiting.
    int rebuild_percent = 30;
    double built_cost = 0.0;          // summed node area right after the last device build
    float *d_cost = nullptr, *h_cost = nullptr;       // one partial sum per block of tree_cost_kernel; pinned copy
    size_t cost_blocks = 0;
    hipEvent_t ev_cost = nullptr;
    bool cost_pending = false;        // h_cost is being written by the copy ev_cost follows
    // the device builder's two scratch areas, kept between builds while they are small (a scene that changes every frame
    // would otherwise pay four hipMalloc / hipFree calls of a hundred megabytes per build)
    char *build_scratch[2] = {nullptr, nullptr};
    size_t build_scratch_bytes[2] = {0, 0};
    float4 *d_inst = nullptr;         // SceneDev::inst
    std::vector<CachedMesh> mesh_cache;
    uint64_t mesh_versions = 0;
    // Device layout of a two-level scene: [top-level nodes, room for inst_node_cap | mesh trees] and the same for the records, so
    // that a build which finds the mesh trees of inst_sig already on the device sends the top level only.
    size_t inst_node_cap = 0, inst_rec_cap = 0;
    std::vector<uint64_t> inst_sig;
    int4 *d_objects = nullptr;        // SceneDev::objects
    int32_t *d_indices = nullptr;     // SceneDev::indices

    // targets
    int W = 0, H = 0;
    float4 *own_result = nullptr, *own_conv = nullptr, *own_result2 = nullptr;
    float4 *own_display = nullptr, *display = nullptr;   // the display blit's destination (pt_display_denoise), made on first use
    float4 *result2 = nullptr;                     // "Result2" (denoisedTex): renderMode 0's accumulated radiance
    float4 *d_filtered = nullptr;                  // renderMode 0 scratch: the denoised image, and
    float *d_seeds = nullptr;                      //   every pixel's running _Seed
    size_t screen_pixels = 0;
    int32_t *own_ids = nullptr;
    float4 *result = nullptr, *conv = nullptr;
    int32_t *ids = nullptr;

    uint32_t *d_work = nullptr;
    uint32_t *d_stack_ovf = nullptr;   // traversal stack overflow, kStackOverflow entries per thread
    size_t stack_ovf_threads = 0;
    unsigned long long *d_stats = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool epoch_open = false;  // renders issued since the last pt_get_stats
    PtStats stats{};
    int num_cus = 0;

    // wavefront pipeline
    int refill_threshold = 32;
    int node_vote = 8;
    int vote_shade = 44, vote_node = 16;
    int vote_shade_few = 52;               // the shade vote of dispatches with one or two samples per pixel (launch_path)
    int sched_blocks_per_cu = 0, sched_ext_blocks_per_cu = 0;   // resident blocks per CU of path_sched_kernel / path_sched_mat_kernel
    int sched_inst_blocks_per_cu = 0;                           // and of path_sched_inst_kernel
    int sched8_blocks_per_cu = 0, sched8_ext_blocks_per_cu = 0; // and of the two kernels' eight-wide instantiations
    int blocks_per_cu = 0;                 // option "blocks_per_cu": 0 = as many as fit
    int reuse_primary = 0;                 // option "reuse_primary_hit"
    int fast_intrinsics = 0;               // option "fast_intrinsics": Shade() on the hardware's approximate rcp / rsq / sqrt / exp2 / log2 / sin (plain path tracing only)
    int sched_fast_blocks_per_cu = 0, sched8_fast_blocks_per_cu = 0;
    int backend = 0;                       // 0 = path_sched_kernel (one ray per lane), 1 = wavefront stages, 2 = path_pool_kernel (ray pool per wave)
    int pool_blocks_per_cu = 0, pool_ext_blocks_per_cu = 0;   // resident one-wave blocks per CU of path_pool_kernel / path_pool_mat_kernel
    int pool_vote_shade = 56, pool_vote_node = 56, pool_vote_leaf = 32, pool_cont = 48;
    float4 *d_pool_state = nullptr;        // 3 float4 per slot
    size_t pool_slots = 0;
    // expensive blocks first: the cost of every 8x8 pixel block as the last dispatch over the same rectangle measured it
    int block_order = 1;                   // option "block_order"
    uint32_t *d_block_cost = nullptr, *d_block_order = nullptr;
    size_t order_capacity = 0;             // blocks the two arrays hold
    bool order_ready = false;              // d_block_order is a permutation for order_key
    int order_key[6] = {0, 0, 0, 0, 0, 0}; // W, H, x0, y0, tw, th of the dispatch that produced the costs
    int trace_blocks_per_cu = 0, shade_blocks_per_cu = 0;
    float4 *wf_state = nullptr;            // 5 float4 arrays of wf_slots entries
    uint32_t *wf_queues = nullptr;         // 2 queues of wf_slots entries
    size_t wf_slots = 0;
    uint32_t *wf_counters = nullptr;       // counts[wf_iters + 2] then cursors[wf_iters + 2]
    size_t wf_iters = 0;
};

// Several contexts, one per device, driven by one host thread (a Unity host is one process): the scene is replicated,
// each member renders its share of the eyes / sample indices / rows, and pt_group_reduce adds the members' converged
// images up on one of them.
struct pt_group {
    std::vector<pt_context *> members;
    std::vector<hipEvent_t> ready;     // member i's renders so far are complete (recorded on its stream)
    hipEvent_t taken = nullptr;        // the root has finished reading the members' images
    float4 *staging = nullptr;         // on the root's device
    size_t staging_pixels = 0;
    int staging_device = -1;
    std::string err;
};

namespace ptrace {
__global__ __launch_bounds__(256) void add_image_kernel(float4 *dst, const float4 *src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 a = dst[i], b = src[i];
    dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
}  // namespace ptrace

// ---- pt_trace_rays_static: every device buffer of a ray query in one array that exists from load time on
//   [0, kStaticIds)            origins + directions, 6 floats per ray (staging of the caller's host buffer)
//   [kStaticIds, kStaticT)     hit ids                      [kStaticT, kStaticStack)   hit distances
//   [kStaticStack, end)        traversal-stack overflow, [entry][thread] over the launch's threads
// Each GPU holds its own copy at its own address; calls on one GPU share it, so calls that overlap there race.
constexpr size_t kStaticMaxRays = PT_TRACE_RAYS_STATIC_MAX;
constexpr size_t kStaticIds = kStaticMaxRays * 6 * sizeof(float);
constexpr size_t kStaticT = kStaticIds + kStaticMaxRays * sizeof(int32_t);
constexpr size_t kStaticStack = kStaticT + kStaticMaxRays * sizeof(float);
constexpr size_t kStaticBytes = kStaticStack + kStaticMaxRays * (size_t)kStackOverflow * sizeof(uint32_t);
static_assert(kStaticMaxRays % kBlock == 0 && kStaticIds % 256 == 0 && kStaticT % 256 == 0 && kStaticStack % 256 == 0, "scratch layout");
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

namespace {

// The array's address on each device (hipGetSymbolAddress with that device current), looked up by the first
// pt_create on the device.  Host code uses these pointers only, never the array's name.
constexpr int kMaxScratchDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxScratchDevices] = {};
// 1 while a pt_trace_rays_static call is between its first copy into the device's array and its last copy out of it:
// a second call on the same GPU - another context, another thread - is refused instead of overwriting the first one's rays
std::atomic<int> g_scratch_busy[kMaxScratchDevices] = {};
struct ScratchLease {
    int device; bool held;
    explicit ScratchLease(int dev) : device(dev), held(false)
    {
        int expected = 0;
        held = g_scratch_busy[dev].compare_exchange_strong(expected, 1);
    }
    ~ScratchLease() { if (held) g_scratch_busy[device].store(0); }
};

// Call with `device` current.
hipError_t resolve_scratch(int device)
{
    if (device < 0 || device >= kMaxScratchDevices) return hipSuccess;     // pt_trace_rays_static reports it when asked
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return hipSuccess;
    void *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
    return hipSuccess;
}

unsigned char *scratch_of(int device)
{
    if (device < 0 || device >= kMaxScratchDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    return g_scratch_ptr[device];
}

thread_local std::string g_create_error;

// The tree the kernels walk: the four-wide one unless a ray could need more stack entries in it than a lane has (a lopsided
// tree after the collapse: up to three stacked siblings per level).  The two-child tree stacks one entry per level, and its
// depth is bounded by kMaxDepth, so it always fits.
int tree_width(const pt_context *ctx)
{
    if (ctx->walk8) return 8;         // set by a build under option bvh_width 8 whose tree fits a lane's stack of groups
    if (ctx->expanded || ctx->device_built) return 4;      // trees made on the device exist four-wide only (their stack need was checked by the build)
    return ctx->bvh_width != 2 && ctx->stack_need4 <= ctx->stack_limit ? 4 : 2;
}
// Groups a lane's stack holds when it walks the eight-wide tree (pt_device.hpp, GroupStackT); PTRACE_STACK_LIMIT lowers it too.
int group_stack_limit(const pt_context *ctx)
{
    return std::min(ctx->stack_limit, kStack8Lds + kStackOverflow / 2);
}

int fail(pt_context *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

#define PT_HIP(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(ctx, e_ == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP,              \
                        std::string(#call) + ": " + hipGetErrorString(e_));                      \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

template <class T>
void free_dev(T *&p)
{
    if (p) { (void)dev_free(p); p = nullptr; }
}

// Host to device on the context's stream, complete on return.  (hipMemcpy and hipMemset work on the null stream, which the
// context's non-blocking stream does not wait for, and may return before a small transfer or a fill has reached the device:
// a kernel launched right after could still see the old contents.)
int h2d(pt_context *ctx, void *dst, const void *src, size_t bytes)
{
    PT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

template <class T>
int upload(pt_context *ctx, T *&dst, const void *src, size_t bytes)
{
    free_dev(dst);
    if (bytes == 0) return PT_OK;
    PT_HIP(ctx, dev_malloc((void **)&dst, bytes));
    return h2d(ctx, dst, src, bytes);
}

static inline void box_reset_inf(Aabb &b)
{
    for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
}

// ---- two-level scenes ----------------------------------------------------------------------------------------------
// Mesh objects with the same object-space triangles are instances of one mesh.  RebuildMeshObjectBuffers appends a COPY of
// mesh.vertices and of the indices, offset by the copy's first vertex, for every object (RayTracingMaster.cs:387-408), so
// objects sharing a sharedMesh arrive with different index ranges and different vertex ranges of equal content: they are
// recognised by that content - equal index counts, indices equal after subtracting each range's smallest index, and
// bit-identical positions at the referenced vertices (a hash, then a full comparison) - and hosts that keep one copy and
// hand every object the same range are covered by the same rule.  A group of two or more gets one tree over the mesh's
// object-space triangles if every member is placed by a similarity: localToWorld = rotation (or mirror) * uniform
// scale + translation, finite, scale within [1e-3, 1e3].  Under a similarity a box padded by p in object space holds
// everything within p * scale of the triangle in world space, whichever way the object is turned; shears and uneven
// scales have no such bound and stay flattened, as do meshes of fewer than kMinInstanceTris triangles.
struct MeshGroup {
    int32_t indices_offset = 0, indices_count = 0;
    std::vector<int> objects;      // members, in scene order
    double scale_min = 0.0;        // smallest uniform scale among them
    double inv_norm = 0.0;         // largest infinity norm of an inverse linear part
};
constexpr int kMinInstanceTris = 64;      // PTRACE_INSTANCE_MIN_TRIS lowers it (tests with tiny meshes)

// The uniform scale of a similarity, or 0.
double similarity_scale(const float *m)
{
    double a[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            const float v = m[j * 4 + i];
            if (!(std::fabs(v) < 1e6f)) return 0.0;          // NaN, infinite or absurd
            if (j < 3) a[i][j] = v;
        }
    double g[3][3], s2 = 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { g[i][j] = a[0][i] * a[0][j] + a[1][i] * a[1][j] + a[2][i] * a[2][j]; if (i == j) s2 += g[i][j] / 3.0; }
    if (!(s2 > 1e-6 && s2 < 1e6)) return 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            if (std::fabs(g[i][j] - (i == j ? s2 : 0.0)) > 1e-4 * s2) return 0.0;
    return std::sqrt(s2);
}

// Inverse of the affine map x -> A x + b held in rows 0..2 of m (column-major 4x4), as rows (Ainv | -Ainv b).
void affine_inverse_rows(const float *m, double inv[3][4])
{
    double a[3][3], b[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) a[i][j] = m[j * 4 + i]; b[i] = m[12 + i]; }
    const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    const double c[3][3] = {{a[1][1] * a[2][2] - a[1][2] * a[2][1], a[0][2] * a[2][1] - a[0][1] * a[2][2], a[0][1] * a[1][2] - a[0][2] * a[1][1]},
                            {a[1][2] * a[2][0] - a[1][0] * a[2][2], a[0][0] * a[2][2] - a[0][2] * a[2][0], a[0][2] * a[1][0] - a[0][0] * a[1][2]},
                            {a[1][0] * a[2][1] - a[1][1] * a[2][0], a[0][1] * a[2][0] - a[0][0] * a[2][1], a[0][0] * a[1][1] - a[0][1] * a[1][0]}};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) inv[i][j] = c[i][j] / det;
        inv[i][3] = -(inv[i][0] * b[0] + inv[i][1] * b[1] + inv[i][2] * b[2]);
    }
}

// fn(begin, end) over [0, n) cut into at most sixteen pieces of `grain` or more, a thread each (the caller's among them).
template <class F> void parallel_pieces(size_t n, size_t grain, F fn)
{
    const size_t cores = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    const size_t pieces = std::max<size_t>(1, std::min(cores, n / std::max<size_t>(1, grain)));
    const size_t piece = (n + pieces - 1) / pieces;
    std::vector<std::thread> others;
    others.reserve(pieces);
    for (size_t t = 1; t < pieces; t++) {
        try { others.emplace_back([=, &fn] { fn(std::min(n, t * piece), std::min(n, (t + 1) * piece)); }); }
        catch (const std::system_error &) { fn(std::min(n, t * piece), std::min(n, (t + 1) * piece)); }      // no thread to be had: here, then
    }
    fn(0, std::min(n, piece));
    for (std::thread &t : others) t.join();
}

// An index range as a mesh: its smallest index, and a hash of the indices relative to the range's first index together with
// the positions they name - a sum over the places, so that any cut of the range into pieces gives the same number.
struct RangeContent { int32_t offset, count, base; uint64_t hash; };

// Hash and smallest index of every listed range, the ranges laid end to end and shared out over the host's threads (ten
// million triangles as 38 copies of a mesh: 150 ms on one thread).
void range_contents(std::vector<RangeContent> &rcs, const std::vector<int32_t> &indices, const std::vector<float> &vertices)
{
    const size_t R = rcs.size();
    std::vector<size_t> first(R + 1, 0);
    for (size_t r = 0; r < R; r++) first[r + 1] = first[r] + (size_t)rcs[r].count;
    std::vector<std::atomic<uint64_t>> hash(R);
    std::vector<std::atomic<int32_t>> base(R);
    for (size_t r = 0; r < R; r++) { hash[r].store(0); base[r].store(std::numeric_limits<int32_t>::max()); }
    parallel_pieces(first[R], (size_t)1 << 16, [&](size_t b, size_t e) {
        size_t r = (size_t)(std::upper_bound(first.begin(), first.end(), b) - first.begin()) - 1;
        while (b < e) {
            while (first[r + 1] <= b) r++;
            const size_t stop = std::min(e, first[r + 1]);
            const int32_t *ix = &indices[(size_t)rcs[r].offset];
            const int32_t head = ix[0];
            uint64_t sum = 0; int32_t smallest = std::numeric_limits<int32_t>::max();
            for (size_t k = b - first[r]; k < stop - first[r]; k++) {
                uint32_t w[4];
                w[0] = (uint32_t)(ix[k] - head);
                std::memcpy(&w[1], &vertices[3 * (size_t)ix[k]], 12);
                uint64_t h = (uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull;
                for (uint32_t x : w) { h = (h ^ x) * 0xff51afd7ed558ccdull; h ^= h >> 32; }
                sum += h;
                smallest = std::min(smallest, ix[k]);
            }
            hash[r].fetch_add(sum, std::memory_order_relaxed);
            int32_t seen = base[r].load(std::memory_order_relaxed);
            while (smallest < seen && !base[r].compare_exchange_weak(seen, smallest, std::memory_order_relaxed)) {}
            b = stop;
        }
    });
    for (size_t r = 0; r < R; r++) { rcs[r].hash = hash[r].load(); rcs[r].base = base[r].load(); }
}

// same[p]: whether the two ranges of pair p (equal lengths) hold the same mesh - the same indices relative to their smallest
// one, naming the same positions.  All pairs end to end, shared out like the hashes.
void same_contents(const std::vector<std::pair<RangeContent, RangeContent>> &pairs, const std::vector<int32_t> &indices, const std::vector<float> &vertices, std::vector<char> &same)
{
    const size_t P = pairs.size();
    std::vector<size_t> first(P + 1, 0);
    for (size_t p = 0; p < P; p++) first[p + 1] = first[p] + (size_t)pairs[p].first.count;
    std::vector<std::atomic<char>> differs(P);
    for (size_t p = 0; p < P; p++) differs[p].store(0);
    parallel_pieces(first[P], (size_t)1 << 16, [&](size_t b, size_t e) {
        size_t p = (size_t)(std::upper_bound(first.begin(), first.end(), b) - first.begin()) - 1;
        while (b < e) {
            while (first[p + 1] <= b) p++;
            const size_t stop = std::min(e, first[p + 1]);
            const RangeContent &x = pairs[p].first, &y = pairs[p].second;
            const int32_t *ix = &indices[(size_t)x.offset], *iy = &indices[(size_t)y.offset];
            bool diff = false;
            for (size_t k = b - first[p]; k < stop - first[p] && !diff; k++)
                diff = ix[k] - x.base != iy[k] - y.base || std::memcmp(&vertices[3 * (size_t)ix[k]], &vertices[3 * (size_t)iy[k]], 12) != 0;
            if (diff) differs[p].store(1, std::memory_order_relaxed);
            b = stop;
        }
    });
    same.resize(P);
    for (size_t p = 0; p < P; p++) same[p] = !differs[p].load();
}

std::vector<MeshGroup> find_instances(const std::vector<PtMeshObject> &objects, const std::vector<int32_t> &indices, const std::vector<float> &vertices,
                                      std::vector<int> &obj_group, std::vector<int32_t> &obj_base, bool any_affine)
{
    obj_base.assign(objects.size(), 0);
    std::vector<MeshGroup> groups;
    int min_tris = kMinInstanceTris;
    if (const char *env = std::getenv("PTRACE_INSTANCE_MIN_TRIS")) min_tris = std::max(1, std::atoi(env));
    // every distinct index range is read once; ranges of equal content then share a class, named by its first range
    std::vector<std::pair<std::pair<int32_t, int32_t>, int>> by_range;   // ((offset, count), object)
    for (size_t o = 0; o < objects.size(); o++)
        if (objects[o].indices_count / 3 >= min_tris && objects[o].indices_offset >= 0) by_range.push_back({{objects[o].indices_offset, objects[o].indices_count}, (int)o});
    std::sort(by_range.begin(), by_range.end());
    std::vector<std::pair<std::pair<int32_t, int32_t>, int>> keyed;      // ((offset, count) of the object's class, object)
    // a range whose length no other range has can share its content with nobody: it is not read at all when one object
    // uses it (the usual scene of distinct meshes costs nothing here)
    std::vector<int32_t> lengths;
    for (size_t i = 0; i < by_range.size(); i++)
        if (i == 0 || by_range[i].first != by_range[i - 1].first) lengths.push_back(by_range[i].first.second);
    std::sort(lengths.begin(), lengths.end());
    std::vector<RangeContent> ranges;                                    // the distinct ranges that are read
    std::vector<std::pair<size_t, size_t>> users;                        // and the stretch of by_range that uses each
    for (size_t i = 0; i < by_range.size();) {
        size_t j = i;
        while (j < by_range.size() && by_range[j].first == by_range[i].first) j++;
        const int32_t count = by_range[i].first.second;
        const bool length_shared = std::upper_bound(lengths.begin(), lengths.end(), count) - std::lower_bound(lengths.begin(), lengths.end(), count) > 1;
        if (length_shared || j - i >= 2) { ranges.push_back(RangeContent{by_range[i].first.first, count, 0, 0}); users.push_back({i, j}); }
        i = j;
    }
    range_contents(ranges, indices, vertices);
    // ranges of one length and hash are compared with the first of them (the one that starts first); what differs after all -
    // two meshes under one hash - goes the slow way, against every class of its length
    std::vector<size_t> order(ranges.size());
    for (size_t r = 0; r < order.size(); r++) order[r] = r;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) {
        return std::make_tuple(ranges[x].count, ranges[x].hash, ranges[x].offset) < std::make_tuple(ranges[y].count, ranges[y].hash, ranges[y].offset);
    });
    std::vector<size_t> leader(ranges.size());
    std::vector<std::pair<RangeContent, RangeContent>> pairs; std::vector<size_t> pair_range;
    for (size_t k = 0; k < order.size(); k++) {
        const size_t r = order[k];
        const bool heads = k == 0 || ranges[order[k - 1]].count != ranges[r].count || ranges[order[k - 1]].hash != ranges[r].hash;
        leader[r] = heads ? r : leader[order[k - 1]];
        if (!heads) { pairs.push_back({ranges[leader[r]], ranges[r]}); pair_range.push_back(r); }
    }
    std::vector<char> same;
    same_contents(pairs, indices, vertices, same);
    std::vector<int> cls_of(ranges.size(), -1);
    std::vector<size_t> classes;                                         // a class is named by its first range
    for (size_t r = 0; r < ranges.size(); r++)
        if (leader[r] == r) { cls_of[r] = (int)classes.size(); classes.push_back(r); }
    for (size_t p = 0; p < pairs.size(); p++) {
        const size_t r = pair_range[p];
        if (same[p]) { cls_of[r] = cls_of[leader[r]]; continue; }
        std::vector<char> one;
        for (size_t c = 0; c < classes.size() && cls_of[r] < 0; c++) {
            if (ranges[classes[c]].count != ranges[r].count) continue;
            same_contents({{ranges[classes[c]], ranges[r]}}, indices, vertices, one);
            if (one[0]) cls_of[r] = (int)c;
        }
        if (cls_of[r] < 0) { cls_of[r] = (int)classes.size(); classes.push_back(r); }
    }
    for (size_t r = 0; r < ranges.size(); r++) {
        const RangeContent &rep = ranges[classes[(size_t)cls_of[r]]];
        for (size_t j = users[r].first; j < users[r].second; j++) {
            keyed.push_back({{rep.offset, rep.count}, by_range[j].second});
            obj_base[(size_t)by_range[j].second] = ranges[r].base;
        }
    }
    std::sort(keyed.begin(), keyed.end());
    for (size_t i = 0; i < keyed.size();) {
        size_t j = i;
        MeshGroup g;
        g.indices_offset = keyed[i].first.first; g.indices_count = keyed[i].first.second;
        g.scale_min = 1e30;
        for (; j < keyed.size() && keyed[j].first == keyed[i].first; j++) {
            const float *m = objects[(size_t)keyed[j].second].local_to_world;
            if (any_affine) {       // copies refitted in world space need no bound through the inverse map: any matrix will do
                g.objects.push_back(keyed[j].second);
                g.scale_min = 1.0; g.inv_norm = 1.0;
                continue;
            }
            const double sc = similarity_scale(m);
            if (!(sc >= 1e-3 && sc <= 1e3)) continue;
            double inv[3][4];
            affine_inverse_rows(m, inv);
            double norm = 0.0;
            for (int r = 0; r < 3; r++) norm = std::max(norm, std::fabs(inv[r][0]) + std::fabs(inv[r][1]) + std::fabs(inv[r][2]));
            if (!std::isfinite(norm)) continue;
            g.objects.push_back(keyed[j].second);
            g.scale_min = std::min(g.scale_min, sc); g.inv_norm = std::max(g.inv_norm, norm);
        }
        if (g.objects.size() >= 2) {
            for (int o : g.objects) obj_group[(size_t)o] = (int)groups.size();
            groups.push_back(std::move(g));
        }
        i = j;
    }
    return groups;
}

void free_expanded(pt_context *ctx)
{
    free_dev(ctx->d_exp_inst); free_dev(ctx->d_exp_top_levels);
    for (pt_context::ExpandedGroup &g : ctx->exp_groups) free_dev(g.d_level_nodes);
    ctx->exp_groups.clear(); ctx->exp_top_first.clear();
}

// Every record and its box from the current matrices, then the copies' nodes level by level, then the top-level tree's:
// the second half of a build with copies made on the device, and all of a transform update of such a scene.
int refit_expanded(pt_context *ctx, const RefitArgs &ra)
{
    hipLaunchKernelGGL(refit_prims_kernel, dim3((unsigned)(((size_t)ra.n_prims + kRefitBlock - 1) / kRefitBlock)), dim3(kRefitBlock), 0, ctx->stream, ra);
    for (const pt_context::ExpandedGroup &g : ctx->exp_groups) {
        RefitArgs rg = ra; rg.level_nodes = g.d_level_nodes;
        for (size_t l = 0; l + 1 < g.level_first.size(); l++) {
            const int first = g.level_first[l], count = g.level_first[l + 1] - first;
            const size_t tn = g.n_inst * (size_t)count;
            if (count > 0)
                hipLaunchKernelGGL(refit_nodes_copies_kernel, dim3((unsigned)((tn + kRefitBlock - 1) / kRefitBlock)), dim3(kRefitBlock), 0, ctx->stream, rg,
                                   (const int4 *)(ctx->d_exp_inst + g.first_inst), (int)g.n_inst, first, count);
        }
    }
    RefitArgs rt = ra; rt.level_nodes = ctx->d_exp_top_levels;
    for (size_t l = 0; l + 1 < ctx->exp_top_first.size(); l++) {
        const int first = ctx->exp_top_first[l], count = ctx->exp_top_first[l + 1] - first;
        if (count > 0) hipLaunchKernelGGL(refit_nodes_kernel, dim3((unsigned)((count + kRefitBlock - 1) / kRefitBlock)), dim3(kRefitBlock), 0, ctx->stream, rt, first, count);
    }
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// ---- the tree built on the device (pt_build.hpp) ----------------------------------------------------------------
constexpr size_t kDeviceBuildMinPrims = 3072;        // option tree_build 2: scenes below this build on the host (2 000 triangles: 1.7 ms against 1.8; 4 000: 3.4 against 2.0)
constexpr int kDeviceBuildFallback = -1000;          // not an error: the caller goes on with the host build
constexpr int kDeviceBuildMaxLevels = 2048;
constexpr size_t kDeviceBuildKeepBytes = (size_t)256 << 20;       // scratch areas up to this size stay with the context between builds
constexpr int kDeviceBuildTopClusters = 8192;        // clusters at which the clustering rounds stop and the host builds the top of the tree

// One device allocation cut into 256-byte-aligned parts: first a dry run for the size, then the same calls for the pointers.
struct DeviceArena {
    char *base = nullptr;
    size_t used = 0;
    template <class T> void take(T *&p, size_t n)
    {
        p = base ? reinterpret_cast<T *>(base + used) : nullptr;
        used += (n * sizeof(T) + 255) & ~(size_t)255;
    }
};

// The scene as pt_set_spheres / pt_set_meshes left it, every mesh object flattened (RayTracingShader.compute:236-240 applied
// once instead of per ray), as a four-wide tree made on the device.  The host touches the objects and the spheres, never a
// triangle.  Leaves the context exactly as the host build of a flat scene does, minus the two-child tree.
int build_scene_on_device(pt_context *ctx, const std::vector<uint32_t> &obj_first_tri, size_t nt, std::chrono::steady_clock::time_point t_called)
{
    const auto t0 = std::chrono::steady_clock::now();
    const unsigned long long alloc_ns0 = g_allocation_ns.load(), alloc_calls0 = g_allocation_calls.load();
    const int nS = (int)ctx->spheres.size();
    const size_t nO = ctx->objects.size();
    const size_t nP = (size_t)nS + nt;
    if (nP < 2 || nP >= (1u << 27)) return kDeviceBuildFallback;
    int n = (int)nP;                   // references: the primitives, plus the extra pieces of cut triangles (below)
    int rc;
    // ---- inputs: geometry and matrices stay on the device for refits, the rest lives in the arena
    std::vector<int4> otab(nO + 1);
    for (size_t o = 0; o < nO; o++) otab[o] = make_int4((int)obj_first_tri[o], ctx->objects[o].indices_offset, 0, 0);
    otab[nO] = make_int4((int)nt, 0, 0, 0);
    std::vector<float> mats(nO * 16);
    for (size_t o = 0; o < nO; o++) std::memcpy(&mats[o * 16], ctx->objects[o].local_to_world, 64);
    std::vector<float4> smat((size_t)nS * 4);
    for (int i = 0; i < nS; i++) {
        const PtSphere &sp = ctx->spheres[i];
        smat[i * 4 + 0] = make_float4(sp.position[0], sp.position[1], sp.position[2], sp.radius);
        smat[i * 4 + 1] = make_float4(sp.albedo[0], sp.albedo[1], sp.albedo[2], sp.smoothness);
        smat[i * 4 + 2] = make_float4(sp.specular[0], sp.specular[1], sp.specular[2], 0.0f);
        smat[i * 4 + 3] = make_float4(sp.emission[0], sp.emission[1], sp.emission[2], 0.0f);
    }
    free_dev(ctx->d_inst); free_dev(ctx->d_objects); free_dev(ctx->d_indices); free_dev(ctx->d_nodes); free_dev(ctx->d_nodes8);
    free_dev(ctx->d_nodes4); free_dev(ctx->d_prims); free_dev(ctx->d_slot_src); free_dev(ctx->d_prim_box); free_dev(ctx->d_node_box);
    free_dev(ctx->d_slot_clip); free_dev(ctx->d_build_matrices); free_dev(ctx->d_level_nodes);
    free_expanded(ctx);
    ctx->expanded = false; ctx->device_built = false; ctx->refit_ready = false;
    ctx->n_nodes = ctx->n_nodes4 = 0; ctx->n_inst = 0; ctx->inst_sig.clear(); ctx->inst_node_cap = ctx->inst_rec_cap = 0;
    if ((rc = upload(ctx, ctx->d_vertices, ctx->vertices.data(), ctx->vertices.size() * sizeof(float)))) return rc;
    if ((rc = upload(ctx, ctx->d_matrices, mats.data(), mats.size() * sizeof(float)))) return rc;
    if ((rc = upload(ctx, ctx->d_sphere_mat, smat.data(), smat.size() * sizeof(float4)))) return rc;

    struct Parts {
        int32_t *indices; int4 *objects; uint32_t *bounds, *large;
        int4 *src; float4 *prims; float *box, *clip; int2 *cut_entry; float *cut_box;
        uint64_t *keys[2]; uint32_t *vals[2];
        float4 *cl_lo[2], *cl_hi[2]; int32_t *nn; uint64_t *flags, *offs, *totals;
        int2 *bn_child; float4 *bn_lo, *bn_hi;
        int32_t *items[2], *stacked[2]; int4 *kids;
        uint4 *nodes4; uint32_t *slot_prim, *slot_prim_dfs; int32_t *stack_need;
        int2 *below, *place;
        SahRange *act[2], *cand, *small; int32_t *slot; int4 *split; uint32_t *bins, *centres[2];
        char *temp;
    } d{};
    const size_t n_bounds = 6 * (nO + 1);
    // the first arena: what it takes to know the scene's bounds and its large triangles; the second, once the number of
    // references is known, holds everything else
    auto layout_first = [&](DeviceArena &a) {
        a.take(d.indices, ctx->indices.size()); a.take(d.objects, nO + 1); a.take(d.bounds, n_bounds); a.take(d.large, nP + 1);
        a.take(d.src, nP); a.take(d.prims, nP * 3); a.take(d.box, nP * 6);
    };
    // scratch area `which` of at least `bytes`: the context's if it is large enough, else a new one in its place; areas
    // beyond kDeviceBuildKeepBytes go when this function returns
    struct Scratch {
        pt_context *ctx; int which; char *p = nullptr;
        ~Scratch()
        {
            if (ctx->build_scratch_bytes[which] > kDeviceBuildKeepBytes) { free_dev(ctx->build_scratch[which]); ctx->build_scratch_bytes[which] = 0; }
        }
    } first_arena{ctx, 0}, second_arena{ctx, 1};
    const auto scratch = [&](Scratch &a, size_t bytes) -> hipError_t {
        if (ctx->build_scratch_bytes[a.which] < bytes) {
            free_dev(ctx->build_scratch[a.which]); ctx->build_scratch_bytes[a.which] = 0;
            const hipError_t e = dev_malloc((void **)&ctx->build_scratch[a.which], bytes);
            if (e != hipSuccess) { ctx->build_scratch[a.which] = nullptr; return e; }
            ctx->build_scratch_bytes[a.which] = bytes;
        }
        a.p = ctx->build_scratch[a.which];
        return hipSuccess;
    };
    DeviceArena arena;
    layout_first(arena);
    size_t arena_bytes = arena.used;
    PT_HIP(ctx, scratch(first_arena, arena_bytes));
    arena = DeviceArena(); arena.base = first_arena.p;
    layout_first(arena);
    const auto grid = [](size_t count) { return dim3((unsigned)((count + kBuildBlock - 1) / kBuildBlock)); };
    hipStream_t st = ctx->stream;

    // ---- bounds
    std::vector<uint32_t> bounds(n_bounds);
    for (size_t b = 0; b <= nO; b++)
        for (int k = 0; k < 3; k++) { bounds[b * 6 + k] = 0xffffffffu; bounds[b * 6 + 3 + k] = 0u; }
    if (!ctx->indices.empty()) PT_HIP(ctx, hipMemcpyAsync(d.indices, ctx->indices.data(), ctx->indices.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d.objects, otab.data(), otab.size() * sizeof(int4), hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d.bounds, bounds.data(), n_bounds * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    BuildScene bs;
    bs.vertices = ctx->d_vertices; bs.indices = d.indices; bs.matrices = ctx->d_matrices; bs.objects = d.objects; bs.spheres = ctx->d_sphere_mat;
    bs.n_objects = (int32_t)nO; bs.n_spheres = nS; bs.n_tris = (int32_t)nt;
    if (nt) hipLaunchKernelGGL(build_bounds_kernel, dim3((unsigned)wave_chunk_blocks(nt)), dim3(kBuildBlock), 0, st, bs, d.bounds + 6 * nO, d.bounds);
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipMemcpyAsync(bounds.data(), d.bounds, n_bounds * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    const auto t_bounds = std::chrono::steady_clock::now();
    float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
    ctx->object_box.assign(nO, Aabb{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}});
    for (size_t b = 0; b <= nO; b++)
        for (int k = 0; k < 3; k++) {
            if (bounds[b * 6 + k] == 0xffffffffu) continue;            // nothing finite on this axis
            const float l = ordered_float(bounds[b * 6 + k]), h = ordered_float(bounds[b * 6 + 3 + k]);
            if (b < nO) { ctx->object_box[b].lo[k] = l; ctx->object_box[b].hi[k] = h; } else { slo[k] = l; shi[k] = h; }
        }
    for (int k = 0; k < 3; k++) { ctx->spheres_lo[k] = INFINITY; ctx->spheres_hi[k] = -INFINITY; }
    for (int i = 0; i < nS; i++) {
        const PtSphere &sp = ctx->spheres[i];
        const float r = std::fabs(sp.radius);
        for (int k = 0; k < 3; k++) {
            const float l = sp.position[k] - r, h = sp.position[k] + r;
            if (std::fabs(l) < kHugeCoord) { slo[k] = std::min(slo[k], l); ctx->spheres_lo[k] = std::min(ctx->spheres_lo[k], l); }
            if (std::fabs(h) < kHugeCoord) { shi[k] = std::max(shi[k], h); ctx->spheres_hi[k] = std::max(ctx->spheres_hi[k], h); }
        }
    }
    // the paddings: the same figures, for the same reasons, as the host build's (rebuild_scene)
    float diag = std::sqrt((shi[0] - slo[0]) * (shi[0] - slo[0]) + (shi[1] - slo[1]) * (shi[1] - slo[1]) + (shi[2] - slo[2]) * (shi[2] - slo[2]));
    if (!std::isfinite(diag)) diag = 0.0f;
    float pad = 3e-6f * diag + 1e-6f;
    if (const char *env = std::getenv("PTRACE_PAD_SCALE")) pad *= std::max(0.0f, (float)std::atof(env));
    const float reach = std::max(2.0f * diag, ctx->sphere_reach_request);
    const float sphere_slack2 = 8.0f * 5.9604645e-8f * reach * reach;
    ctx->sphere_reach = reach;
    for (int k = 0; k < 3; k++) { ctx->scene_lo[k] = slo[k] <= shi[k] ? slo[k] : 0.0f; ctx->scene_hi[k] = slo[k] <= shi[k] ? shi[k] : 0.0f; }

    // ---- records and padded boxes per primitive, Morton order
    hipLaunchKernelGGL(build_prims_kernel, grid(nP), dim3(kBuildBlock), 0, st, bs, d.src, d.prims);
    RefitArgs ra;
    ra.slot_src = d.src; ra.vertices = ctx->d_vertices; ra.matrices = ctx->d_matrices; ra.prims = d.prims; ra.prim_box = d.box;
    ra.slot_clip = nullptr; ra.build_matrices = nullptr; ra.nodes4 = nullptr; ra.nodes8 = nullptr; ra.node_box = nullptr; ra.level_nodes = nullptr;
    ra.pad = pad; ra.reach = reach; ra.sphere_slack2 = sphere_slack2; ra.n_prims = n;
    hipLaunchKernelGGL(refit_prims_kernel, grid(nP), dim3(kRefitBlock), 0, st, ra);
    // ---- large triangles: cut into references by the host's split_references (pt_build.hpp, list_large_kernel)
    std::vector<int2> cut_entry; std::vector<float> cut_box;
    if (ctx->split_budget > 0.0f && nt) {
        const float ext_max = std::max(std::max(ctx->scene_hi[0] - ctx->scene_lo[0], ctx->scene_hi[1] - ctx->scene_lo[1]), ctx->scene_hi[2] - ctx->scene_lo[2]);
        uint32_t n_large = 0;
        PT_HIP(ctx, hipMemsetAsync(d.large + nP, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(list_large_kernel, grid(nt), dim3(kBuildBlock), 0, st, d.box, nS, n, 0.99f * (ext_max / 16.0f) - 2.0f * pad, d.large, d.large + nP);
        PT_HIP(ctx, hipMemcpyAsync(&n_large, d.large + nP, sizeof n_large, hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        if (n_large) {
            std::vector<uint32_t> large(n_large);
            PT_HIP(ctx, hipMemcpyAsync(large.data(), d.large, (size_t)n_large * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipStreamSynchronize(st));
            std::sort(large.begin(), large.end());
            std::vector<float> tri9((size_t)n_large * 9);
            std::vector<Aabb> exact(n_large);
            for (size_t e = 0; e < n_large; e++) {
                const uint32_t gid = large[e] - (uint32_t)nS;
                const size_t o = (size_t)(std::upper_bound(obj_first_tri.begin(), obj_first_tri.begin() + (ptrdiff_t)nO, gid) - obj_first_tri.begin()) - 1;
                const PtMeshObject &mo = ctx->objects[o];
                const int32_t *ix = &ctx->indices[(size_t)mo.indices_offset + 3 * (size_t)(gid - obj_first_tri[o])];
                for (int c = 0; c < 3; c++) {
                    const float *v = &ctx->vertices[3 * (size_t)ix[c]];
                    const pt_f3 w = pt_mul_point(mo.local_to_world, pt_make3(v[0], v[1], v[2]));      // RayTracingShader.compute:238-240
                    tri9[e * 9 + 3 * c] = w.x; tri9[e * 9 + 3 * c + 1] = w.y; tri9[e * 9 + 3 * c + 2] = w.z;
                }
                for (int k = 0; k < 3; k++) {
                    exact[e].lo[k] = std::min(tri9[e * 9 + k], std::min(tri9[e * 9 + 3 + k], tri9[e * 9 + 6 + k]));
                    exact[e].hi[k] = std::max(tri9[e * 9 + k], std::max(tri9[e * 9 + 3 + k], tri9[e * 9 + 6 + k]));
                }
            }
            // the budget is a share of ALL the scene's triangles (bvh.hpp), the function counts the ones it is given
            std::vector<uint32_t> ref_tri; std::vector<Aabb> ref_box;
            split_references(tri9.data(), exact.data(), n_large, ctx->scene_lo, ctx->scene_hi, ctx->split_budget * (float)nt / (float)n_large, ref_tri, ref_box);
            std::vector<uint32_t> pieces(n_large, 0);
            for (uint32_t t : ref_tri) pieces[t]++;
            std::vector<uint32_t> seen(n_large, 0);
            size_t extra = 0;
            for (size_t j = 0; j < ref_tri.size(); j++) {
                const uint32_t e = ref_tri[j];
                if (pieces[e] < 2) continue;                                  // not cut: the reference is the triangle
                const int r = seen[e]++ == 0 ? (int)large[e] : (int)(nP + extra++);
                cut_entry.push_back(make_int2(r, (int)large[e]));
                for (int k = 0; k < 3; k++) cut_box.push_back(ref_box[j].lo[k]);
                for (int k = 0; k < 3; k++) cut_box.push_back(ref_box[j].hi[k]);
            }
            if (nP + extra >= (1u << 27)) return kDeviceBuildFallback;
            n = (int)(nP + extra);
        }
    }
    const size_t nR = (size_t)n;
    const bool cut = !cut_entry.empty();

    // ---- the second arena
    size_t sort_bytes = 0, scan_bytes = 0, scan32_bytes = 0;
    {
        uint64_t *k = nullptr; uint32_t *v = nullptr;
        PT_HIP(ctx, rocprim::radix_sort_pairs(nullptr, sort_bytes, k, k, v, v, nR, 0u, 63u, ctx->stream));
        PT_HIP(ctx, rocprim::exclusive_scan(nullptr, scan_bytes, k, k, (uint64_t)0, nR, rocprim::plus<uint64_t>(), ctx->stream));
        PT_HIP(ctx, rocprim::exclusive_scan(nullptr, scan32_bytes, v, v, 0u, nR, rocprim::plus<uint32_t>(), ctx->stream));
    }
    const size_t temp_bytes = std::max(std::max(sort_bytes, scan_bytes), scan32_bytes) + 256;
    const size_t sah_nodes = nR / kSahSmall + 2;          // nodes of more than kSahSmall references one level can hold
    const size_t tree_cap = 2 * nR + 1024;                // two-child node numbers (the top-down build leaves gaps in them)
    auto layout_second = [&](DeviceArena &a) {
        if (cut) { a.take(d.src, nR); a.take(d.prims, nR * 3); a.take(d.box, nR * 6); a.take(d.clip, nR * 6); a.take(d.cut_entry, cut_entry.size()); a.take(d.cut_box, cut_box.size()); }
        for (int k = 0; k < 2; k++) { a.take(d.keys[k], nR); a.take(d.vals[k], nR); a.take(d.cl_lo[k], nR); a.take(d.cl_hi[k], nR); a.take(d.items[k], nR); a.take(d.stacked[k], nR); }
        a.take(d.nn, nR); a.take(d.flags, nR); a.take(d.offs, nR); a.take(d.totals, 32);
        a.take(d.bn_child, tree_cap); a.take(d.bn_lo, tree_cap); a.take(d.bn_hi, tree_cap);
        a.take(d.kids, nR); a.take(d.nodes4, nR * 4); a.take(d.slot_prim, nR + 1); a.take(d.slot_prim_dfs, nR + 1); a.take(d.stack_need, 64);
        a.take(d.below, nR); a.take(d.place, nR);
        a.take(d.act[0], sah_nodes); a.take(d.act[1], sah_nodes); a.take(d.cand, 2 * sah_nodes); a.take(d.small, nR); a.take(d.slot, 2 * sah_nodes); a.take(d.split, sah_nodes);
        a.take(d.bins, sah_nodes * kSahBinWords); a.take(d.centres[0], sah_nodes * 6); a.take(d.centres[1], sah_nodes * 6);
        a.take(d.temp, temp_bytes);
    };
    {
        const Parts first = d;                             // (a dry run of the layout overwrites the pointers it hands out)
        arena = DeviceArena();
        layout_second(arena);
        arena_bytes += arena.used;
        PT_HIP(ctx, scratch(second_arena, arena.used));
        d = first;
        arena = DeviceArena(); arena.base = second_arena.p;
        layout_second(arena);
    }
    PT_HIP(ctx, hipMemsetAsync(d.stack_need, 0, 64 * sizeof(int32_t), st));
    if (cut) {
        // records and boxes again, now per reference: the primitives' as before, the extra pieces' copied from their
        // primitive, every cut piece's box clipped to its cut (refit_prims_kernel, slot_clip)
        hipLaunchKernelGGL(build_prims_kernel, grid(nP), dim3(kBuildBlock), 0, st, bs, d.src, d.prims);
        hipLaunchKernelGGL(fill_clip_kernel, grid(nR), dim3(kBuildBlock), 0, st, d.clip, nR);
        PT_HIP(ctx, hipMemcpyAsync(d.cut_entry, cut_entry.data(), cut_entry.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        PT_HIP(ctx, hipMemcpyAsync(d.cut_box, cut_box.data(), cut_box.size() * sizeof(float), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(apply_split_kernel, grid(cut_entry.size()), dim3(kBuildBlock), 0, st, d.cut_entry, d.cut_box, (int)cut_entry.size(), d.src, d.prims, d.clip);
        ra.slot_src = d.src; ra.prims = d.prims; ra.prim_box = d.box; ra.slot_clip = d.clip; ra.build_matrices = ctx->d_matrices; ra.n_prims = n;
        hipLaunchKernelGGL(refit_prims_kernel, grid(nR), dim3(kRefitBlock), 0, st, ra);
        PT_HIP(ctx, hipGetLastError());
        PT_HIP(ctx, hipStreamSynchronize(st));               // cut_entry and cut_box are read by the copies above
    }
    MortonGrid mg;
    for (int k = 0; k < 3; k++) {
        mg.lo[k] = ctx->scene_lo[k];
        const float ext = ctx->scene_hi[k] - ctx->scene_lo[k];
        mg.scale[k] = ext > 0.0f && std::isfinite(ext) ? 2097152.0f / ext : 0.0f;
    }
    hipLaunchKernelGGL(build_morton_kernel, grid(nR), dim3(kBuildBlock), 0, st, d.box, n, mg, d.keys[0], d.vals[0]);
    PT_HIP(ctx, hipGetLastError());
    {
        size_t bytes = temp_bytes;
        PT_HIP(ctx, rocprim::radix_sort_pairs(d.temp, bytes, d.keys[0], d.keys[1], d.vals[0], d.vals[1], (size_t)n, 0u, 63u, st));
    }
    hipLaunchKernelGGL(ploc_init_kernel, grid(nR), dim3(kBuildBlock), 0, st, d.vals[1], d.box, n, d.cl_lo[0], d.cl_hi[0]);
    PT_HIP(ctx, hipGetLastError());
    std::chrono::steady_clock::time_point t_sorted;
    if (std::getenv("PTRACE_BUILD_TIMING")) PT_HIP(ctx, hipStreamSynchronize(st));
    t_sorted = std::chrono::steady_clock::now();

    TwoChildTree tree{d.bn_child, d.bn_lo, d.bn_hi};
    int made = 0, rounds = 0;
    int32_t root = 0;
    const bool top_down = ctx->tree_build != 3;
    if (top_down) {
        // ---- the two-child tree, top-down by SAH (pt_build_sah.hpp)
        uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
        PT_HIP(ctx, hipMemcpyAsync(d.bounds, init, sizeof init, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(sah_bounds_kernel, dim3((unsigned)wave_chunk_blocks(nR)), dim3(kBuildBlock), 0, st, d.cl_lo[0], d.cl_hi[0], n, d.bounds);
        PT_HIP(ctx, hipMemcpyAsync(init, d.bounds, sizeof init, hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipMemsetAsync(d.bn_hi, 0, tree_cap * sizeof(float4), st));          // numbers never given out hold no primitives
        PT_HIP(ctx, hipStreamSynchronize(st));
        SahRange whole;
        for (int k = 0; k < 3; k++) { whole.lo[k] = ordered_float(init[k]); whole.hi[k] = ordered_float(init[3 + k]); }
        whole.begin = 0; whole.count = n; whole.node = 0; whole.parent = -1; whole.side = 0; whole.pad = 0;
        int n_act = 0, n_small = 0, node_base = 1, cur = 0, level = 0;
        int32_t *seg[2] = {d.items[0], d.items[1]};
        uint32_t *left = reinterpret_cast<uint32_t *>(d.nn), *rank = reinterpret_cast<uint32_t *>(d.stacked[0]);
        if (n > kSahSmall) {
            PT_HIP(ctx, hipMemcpyAsync(d.act[0], &whole, sizeof whole, hipMemcpyHostToDevice, st));
            PT_HIP(ctx, hipMemsetAsync(seg[0], 0, nR * sizeof(int32_t), st));
            n_act = 1;
        } else {
            PT_HIP(ctx, hipMemcpyAsync(d.small, &whole, sizeof whole, hipMemcpyHostToDevice, st));
            n_small = 1;
        }
        PT_HIP(ctx, hipStreamSynchronize(st));               // `whole` and `init` are on the stack
        while (n_act > 0) {
            if (level >= kDeviceBuildMaxLevels || (size_t)node_base + 2 * (size_t)n_act + nR > tree_cap) return kDeviceBuildFallback;
            const SahRange *act = d.act[level & 1];
            uint32_t *centres = d.centres[level & 1], *centres_next = d.centres[(level + 1) & 1];
            if (level == 0) {                 // below the root, the level above has collected them (sah_side_kernel)
                hipLaunchKernelGGL(sah_fill_centres_kernel, grid((size_t)n_act * 6), dim3(kBuildBlock), 0, st, centres, n_act * 6);
                hipLaunchKernelGGL(sah_centres_kernel, dim3((unsigned)wave_chunk_blocks(nR)), dim3(kBuildBlock), 0, st, d.cl_lo[cur], d.cl_hi[cur], seg[cur], n, centres);
            }
            hipLaunchKernelGGL(sah_fill_kernel, grid((size_t)n_act * kSahBinWords), dim3(kBuildBlock), 0, st, d.bins, (size_t)n_act * kSahBinWords);
            hipLaunchKernelGGL(sah_bin_kernel, dim3((unsigned)((nR + (size_t)kBinChunk * (kBuildBlock / 64) - 1) / ((size_t)kBinChunk * (kBuildBlock / 64)))), dim3(kBuildBlock), 0, st, d.cl_lo[cur], d.cl_hi[cur], seg[cur], n, centres, d.bins);
            hipLaunchKernelGGL(sah_split_kernel, grid((size_t)n_act * 64), dim3(kBuildBlock), 0, st, act, n_act, d.bins, node_base, tree, d.split, d.cand, d.flags);      // a wave per node
            size_t bytes = temp_bytes;
            PT_HIP(ctx, rocprim::exclusive_scan(d.temp, bytes, d.flags, d.offs, (uint64_t)0, (size_t)(2 * n_act), rocprim::plus<uint64_t>(), st));
            hipLaunchKernelGGL(sah_next_kernel, grid((size_t)(2 * n_act)), dim3(kBuildBlock), 0, st, d.cand, 2 * n_act, d.flags, d.offs, d.act[(level + 1) & 1], d.small, n_small, d.slot,
                               d.totals);
            const int next_words = 6 * (int)std::min<size_t>(2 * (size_t)n_act, sah_nodes);      // (the next level has at most sah_nodes - 2 nodes)
            hipLaunchKernelGGL(sah_fill_centres_kernel, grid((size_t)next_words), dim3(kBuildBlock), 0, st, centres_next, next_words);
            const int side_places = n_act >= 64 ? 256 : kWaveChunk;
            const size_t side_blocks = (nR + (size_t)side_places * (kBuildBlock / 64) - 1) / ((size_t)side_places * (kBuildBlock / 64));
            hipLaunchKernelGGL(sah_side_kernel, dim3((unsigned)side_blocks), dim3(kBuildBlock), 0, st, d.cl_lo[cur], d.cl_hi[cur], seg[cur], n, act, centres, d.split, d.slot, left,
                               centres_next, side_places);
            bytes = temp_bytes;
            PT_HIP(ctx, rocprim::exclusive_scan(d.temp, bytes, left, rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
            hipLaunchKernelGGL(sah_partition_kernel, grid(nR), dim3(kBuildBlock), 0, st, d.cl_lo[cur], d.cl_hi[cur], seg[cur], n, act, d.split, left, rank, d.slot, d.cl_lo[cur ^ 1],
                               d.cl_hi[cur ^ 1], seg[cur ^ 1]);
            uint64_t tot = 0;
            PT_HIP(ctx, hipMemcpyAsync(&tot, d.totals, sizeof tot, hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipStreamSynchronize(st));
            node_base += 2 * n_act; n_small += (int)(tot >> 32); n_act = (int)(uint32_t)tot;
            if ((size_t)n_act > sah_nodes - 2 || (size_t)n_small > nR) return fail(ctx, PT_ERR_HIP, "device tree build: more ranges than primitives");
            cur ^= 1; level++;
        }
        rounds = level;
        hipLaunchKernelGGL(sah_extra_kernel, grid((size_t)n_small), dim3(kBuildBlock), 0, st, d.small, n_small, d.flags);
        size_t bytes = temp_bytes;
        PT_HIP(ctx, rocprim::exclusive_scan(d.temp, bytes, d.flags, d.offs, (uint64_t)0, (size_t)n_small, rocprim::plus<uint64_t>(), st));
        uint64_t last[2] = {0, 0};
        PT_HIP(ctx, hipMemcpyAsync(&last[0], d.flags + (n_small - 1), 8, hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipMemcpyAsync(&last[1], d.offs + (n_small - 1), 8, hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        const size_t numbers = (size_t)node_base + (size_t)(last[0] + last[1]);
        if (numbers > tree_cap) return kDeviceBuildFallback;
        hipLaunchKernelGGL(sah_small_kernel, dim3((unsigned)n_small), dim3(64), 0, st, d.small, d.offs, node_base, d.cl_lo[cur], d.cl_hi[cur], tree);
        PT_HIP(ctx, hipGetLastError());
        made = (int)numbers; root = 0;
    } else {
    // ---- the two-child tree by clustering: clusters merge with their best neighbour until few are left
    int c = n, cur = 0;
    int radius = kPlocRadius;
    if (const char *env = std::getenv("PTRACE_PLOC_RADIUS")) radius = std::max(1, std::min(kPlocRadiusMax, std::atoi(env)));      // measurement knob
    float lambda = 0.0f;
    if (const char *env = std::getenv("PTRACE_PLOC_LAMBDA")) lambda = (float)std::atof(env);                                          // measurement knob
    int top_clusters = kDeviceBuildTopClusters;
    if (const char *env = std::getenv("PTRACE_BUILD_TOP")) top_clusters = std::max(1, std::atoi(env));                               // measurement knob
    while (c > top_clusters) {
        hipLaunchKernelGGL(ploc_nn_kernel, grid((size_t)c), dim3(kBuildBlock), 0, st, d.cl_lo[cur], d.cl_hi[cur], c, d.nn, radius, lambda);
        hipLaunchKernelGGL(ploc_flag_kernel, grid((size_t)c), dim3(kBuildBlock), 0, st, d.nn, c, d.flags);
        size_t bytes = temp_bytes;
        PT_HIP(ctx, rocprim::exclusive_scan(d.temp, bytes, d.flags, d.offs, (uint64_t)0, (size_t)c, rocprim::plus<uint64_t>(), st));
        hipLaunchKernelGGL(ploc_merge_kernel, grid((size_t)c), dim3(kBuildBlock), 0, st, d.cl_lo[cur], d.cl_hi[cur], d.nn, d.flags, d.offs, c, made, tree,
                           d.cl_lo[cur ^ 1], d.cl_hi[cur ^ 1], d.totals);
        uint64_t tot = 0;
        PT_HIP(ctx, hipMemcpyAsync(&tot, d.totals, sizeof tot, hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        const int kept = (int)(uint32_t)tot, merged = (int)(tot >> 32);
        if (merged < 1 || kept != c - merged) return fail(ctx, PT_ERR_HIP, "device tree build: a clustering round merged nothing");
        c = kept; made += merged; cur ^= 1; rounds++;
    }
    root = made - 1;
    if (c > 1) {
        // ---- the top of the tree: the host's binned-SAH builder over the clusters that are left (a few thousand boxes, a
        // millisecond or two).  Clustering is at its best near the leaves, where a good partner is always close by in the sorted
        // order; near the root it pairs whatever is left, and every ray pays for overlap there.
        std::vector<float4> clo((size_t)c), chi((size_t)c);
        PT_HIP(ctx, hipMemcpyAsync(clo.data(), d.cl_lo[cur], (size_t)c * sizeof(float4), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipMemcpyAsync(chi.data(), d.cl_hi[cur], (size_t)c * sizeof(float4), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        std::vector<Aabb> cbox((size_t)c);
        for (int i = 0; i < c; i++) cbox[(size_t)i] = Aabb{{clo[(size_t)i].x, clo[(size_t)i].y, clo[(size_t)i].z}, {chi[(size_t)i].x, chi[(size_t)i].y, chi[(size_t)i].z}};
        const BvhResult top = build_bvh(cbox, 1, 0, 1.0f);
        const size_t T = top.nodes.size();
        if (T != (size_t)c - 1) return fail(ctx, PT_ERR_HIP, "device tree build: the top-level tree is incomplete");
        std::vector<int2> tchild(T); std::vector<float4> tlo(T), thi(T);
        std::vector<int32_t> below(T);
        const auto bits = [](float f) { int32_t i; std::memcpy(&i, &f, 4); return i; };
        for (size_t j = T; j-- > 0;) {                       // children come after their parents
            const BvhNode &nd = top.nodes[j];
            int32_t ref[2], cnt[2];
            const int32_t link[2] = {nd.left, nd.right};
            for (int k = 0; k < 2; k++) {
                if (link[k] >= 0) { ref[k] = made + link[k]; cnt[k] = below[(size_t)link[k]]; }
                else {
                    const uint32_t enc = (uint32_t)~link[k];
                    if (enc & 7u) return fail(ctx, PT_ERR_HIP, "device tree build: a top-level leaf of several clusters");
                    const uint32_t cl = top.order[enc >> 3];
                    ref[k] = bits(clo[cl].w); cnt[k] = bits(chi[cl].w);
                }
            }
            below[j] = cnt[0] + cnt[1];
            float bf; std::memcpy(&bf, &below[j], 4);
            tchild[j] = make_int2(ref[0], ref[1]);
            tlo[j] = make_float4(std::min(nd.llo[0], nd.rlo[0]), std::min(nd.llo[1], nd.rlo[1]), std::min(nd.llo[2], nd.rlo[2]), 0.0f);
            thi[j] = make_float4(std::max(nd.lhi[0], nd.rhi[0]), std::max(nd.lhi[1], nd.rhi[1]), std::max(nd.lhi[2], nd.rhi[2]), bf);
        }
        PT_HIP(ctx, hipMemcpyAsync(d.bn_child + made, tchild.data(), T * sizeof(int2), hipMemcpyHostToDevice, st));
        PT_HIP(ctx, hipMemcpyAsync(d.bn_lo + made, tlo.data(), T * sizeof(float4), hipMemcpyHostToDevice, st));
        PT_HIP(ctx, hipMemcpyAsync(d.bn_hi + made, thi.data(), T * sizeof(float4), hipMemcpyHostToDevice, st));
        PT_HIP(ctx, hipStreamSynchronize(st));               // the vectors go out of scope
        root = made;
        made += (int)T;
    }
    if (made != n - 1) return fail(ctx, PT_ERR_HIP, "device tree build: the two-child tree is incomplete");
    }
    if (!top_down && !std::getenv("PTRACE_BUILD_NO_BOTTOM"))                                                                         // (measurement knob)
        hipLaunchKernelGGL(bottom_rebuild_kernel, dim3((unsigned)((2 * (size_t)made + 63) / 64)), dim3(64), 0, st, tree, made, d.box);
    if (std::getenv("PTRACE_BUILD_TIMING")) PT_HIP(ctx, hipStreamSynchronize(st));
    const auto t_tree = std::chrono::steady_clock::now();

    // ---- four children per node, one level per round from the root; level l's nodes are level_base[l] .. + level_count[l]
    std::vector<int> level_base, level_count;
    {
        const int32_t zero = 0;
        PT_HIP(ctx, hipMemcpyAsync(d.items[0], &root, 4, hipMemcpyHostToDevice, st));
        PT_HIP(ctx, hipMemcpyAsync(d.stacked[0], &zero, 4, hipMemcpyHostToDevice, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
    }
    int items = 1, nodes_done = 0, recs_done = 0, side = 0;
    bool open_pairs = true;
    if (const char *env = std::getenv("PTRACE_BUILD_OPEN_PAIRS")) open_pairs = std::atoi(env) != 0;                                  // measurement knob
    while (items > 0) {
        if ((int)level_base.size() >= kDeviceBuildMaxLevels) return kDeviceBuildFallback;
        CollapseArgs ca;
        ca.tree = tree; ca.items = d.items[side]; ca.stacked = d.stacked[side]; ca.n_items = items;
        ca.passes = open_pairs ? 2 : 1; ca.prim_box = d.box;
        ca.level_base = nodes_done; ca.next_base = nodes_done + items; ca.rec_base = recs_done; ca.zero_rec = n;
        ca.kids = d.kids; ca.counts = d.flags; ca.offs = d.offs; ca.items_next = d.items[side ^ 1]; ca.stacked_next = d.stacked[side ^ 1];
        ca.nodes4 = d.nodes4; ca.slot_prim = d.slot_prim; ca.stack_need = d.stack_need; ca.totals = d.totals;
        hipLaunchKernelGGL(collapse_pick_kernel, grid((size_t)items), dim3(kBuildBlock), 0, st, ca);
        size_t bytes = temp_bytes;
        PT_HIP(ctx, rocprim::exclusive_scan(d.temp, bytes, d.flags, d.offs, (uint64_t)0, (size_t)items, rocprim::plus<uint64_t>(), st));
        hipLaunchKernelGGL(collapse_emit_kernel, grid((size_t)items), dim3(kBuildBlock), 0, st, ca);
        uint64_t tot = 0;
        PT_HIP(ctx, hipMemcpyAsync(&tot, d.totals, sizeof tot, hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        level_base.push_back(nodes_done); level_count.push_back(items);
        nodes_done += items; recs_done += (int)(tot >> 32);
        items = (int)(uint32_t)tot; side ^= 1;
        if ((size_t)nodes_done + (size_t)items > nR) return fail(ctx, PT_ERR_HIP, "device tree build: more nodes than primitives");
    }
    if (recs_done != n) return fail(ctx, PT_ERR_HIP, "device tree build: the collapse lost a primitive");
    int32_t stack_need = 0;
    PT_HIP(ctx, hipMemcpyAsync(&stack_need, d.stack_need, 4, hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    if (stack_need > ctx->stack_limit) return kDeviceBuildFallback;
    const int n4 = nodes_done, levels = (int)level_base.size();
    const auto t_wide = std::chrono::steady_clock::now();

    // ---- the arrays the kernels read: records in tree order, nodes, boxes from the leaves up
    PT_HIP(ctx, dev_malloc((void **)&ctx->d_nodes4, (size_t)n4 * sizeof(Bvh4Node)));
    PT_HIP(ctx, dev_malloc((void **)&ctx->d_prims, (nR + 1) * 3 * sizeof(float4)));
    PT_HIP(ctx, dev_malloc((void **)&ctx->d_slot_src, (nR + 1) * sizeof(int4)));
    PT_HIP(ctx, dev_malloc((void **)&ctx->d_prim_box, (nR + 1) * 6 * sizeof(float)));
    if (cut) {
        // what a later refit needs to keep the cuts of objects that have not moved (pt_refit.hpp, slot_clip)
        PT_HIP(ctx, dev_malloc((void **)&ctx->d_slot_clip, (nR + 1) * 6 * sizeof(float)));
        PT_HIP(ctx, dev_malloc((void **)&ctx->d_build_matrices, std::max<size_t>(1, nO) * 16 * sizeof(float)));
        PT_HIP(ctx, hipMemcpyAsync(ctx->d_build_matrices, ctx->d_matrices, nO * 16 * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    PT_HIP(ctx, dev_malloc((void **)&ctx->d_node_box, (size_t)n4 * 6 * sizeof(float)));
    PT_HIP(ctx, dev_malloc((void **)&ctx->d_level_nodes, (size_t)n4 * sizeof(int32_t)));
    ctx->level_first.assign(1, 0);
    for (int l = levels - 1; l >= 0; l--) ctx->level_first.push_back(ctx->level_first.back() + level_count[l]);       // deepest level first
    {
        ReorderArgs ro;
        ro.nodes_in = d.nodes4; ro.slot_prim_in = d.slot_prim; ro.below = d.below; ro.place = d.place; ro.nodes_out = ctx->d_nodes4; ro.slot_prim_out = d.slot_prim_dfs;
        for (int l = levels - 1; l >= 0; l--) {
            ro.first = level_base[l]; ro.count = level_count[l]; ro.level_nodes_out = nullptr;
            hipLaunchKernelGGL(reorder_count_kernel, grid((size_t)ro.count), dim3(kBuildBlock), 0, st, ro);
        }
        PT_HIP(ctx, hipMemsetAsync(d.place, 0, sizeof(int2), st));             // the root: node 0, record 0
        for (int l = 0; l < levels; l++) {
            ro.first = level_base[l]; ro.count = level_count[l]; ro.level_nodes_out = ctx->d_level_nodes + ctx->level_first[(size_t)(levels - 1 - l)];
            hipLaunchKernelGGL(reorder_place_kernel, grid((size_t)ro.count), dim3(kBuildBlock), 0, st, ro);
        }
    }
    hipLaunchKernelGGL(build_gather_kernel, grid(nR + 1), dim3(kBuildBlock), 0, st, d.slot_prim_dfs, n, d.src, d.prims, d.box, cut ? d.clip : nullptr, ctx->d_slot_src, ctx->d_prims,
                       ctx->d_prim_box, ctx->d_slot_clip);
    ra.slot_clip = ctx->d_slot_clip; ra.build_matrices = ctx->d_build_matrices;
    ra.slot_src = ctx->d_slot_src; ra.prims = ctx->d_prims; ra.prim_box = ctx->d_prim_box; ra.nodes4 = ctx->d_nodes4; ra.node_box = ctx->d_node_box; ra.level_nodes = ctx->d_level_nodes;
    for (size_t l = 0; l + 1 < ctx->level_first.size(); l++) {
        const int first = ctx->level_first[l], count = ctx->level_first[l + 1] - first;
        hipLaunchKernelGGL(refit_nodes_kernel, grid((size_t)count), dim3(kRefitBlock), 0, st, ra, first, count);
    }
    PT_HIP(ctx, hipGetLastError());
    {
        // the tree's summed node area, against which refits are measured (option rebuild_percent)
        const size_t nb = ((size_t)n4 + kBuildBlock - 1) / kBuildBlock;
        if (nb != ctx->cost_blocks) {
            free_dev(ctx->d_cost);
            if (ctx->h_cost) { (void)host_free(ctx->h_cost); ctx->h_cost = nullptr; }
            ctx->cost_blocks = 0;
            PT_HIP(ctx, dev_malloc((void **)&ctx->d_cost, nb * sizeof(float)));
            PT_HIP(ctx, host_malloc((void **)&ctx->h_cost, nb * sizeof(float), hipHostMallocDefault));
            ctx->cost_blocks = nb;
        }
        hipLaunchKernelGGL(tree_cost_kernel, dim3((unsigned)nb), dim3(kBuildBlock), 0, st, ctx->d_node_box, n4, 0, ctx->d_cost);
        PT_HIP(ctx, hipMemcpyAsync(ctx->h_cost, ctx->d_cost, nb * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    PT_HIP(ctx, hipStreamSynchronize(st));               // the arenas go when this returns
    ctx->built_cost = 0.0;
    for (size_t b = 0; b < ctx->cost_blocks; b++) ctx->built_cost += ctx->h_cost[b];
    ctx->cost_pending = false;
    ctx->device_built = true;
    ctx->refit_ready = ctx->refit != 0 && nO > 0;
    ctx->box_pad = pad; ctx->sphere_slack2 = sphere_slack2;
    ctx->walk8 = false; ctx->n_nodes8 = 0; ctx->stack_need8 = 0;
    ctx->n_nodes4 = n4; ctx->n_nodes = n4;                // no two-child tree: n_nodes only says "there is a tree"
    ctx->n_prims = (int)nP; ctx->n_slots = n; ctx->n_top_records = n; ctx->n_spheres = nS;
    ctx->stack_need4 = stack_need;
    ctx->stats.bvh_nodes = n4; ctx->stats.bvh_prims = (int)nP; ctx->stats.bvh_depth = levels;
    // from the moment the scene's rebuild began: the search for shared meshes before this function is part of the figure
    ctx->stats.bvh_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_called).count();
    if (std::getenv("PTRACE_BUILD_TIMING")) {
        const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<float, std::milli>(b - a).count(); };
        std::fprintf(stderr, "scene build on the device: %d primitives, %d references, %d nodes in %d levels, stack need %d, %d clustering rounds, %.0f MB of scratch; before the build %.1f ms, upload + bounds %.1f ms, "
                             "records + sort %.1f ms, two-child tree %.1f ms, collapse %.1f ms, gather + boxes %.1f ms; %.2f ms of it in %llu hipMalloc / hipFree calls\n", (int)nP, n, n4, levels, stack_need, rounds, arena_bytes / 1048576.0,
                     ms(t_called, t0), ms(t0, t_bounds), ms(t_bounds, t_sorted), ms(t_sorted, t_tree), ms(t_tree, t_wide), ms(t_wide, std::chrono::steady_clock::now()),
                     (g_allocation_ns.load() - alloc_ns0) * 1e-6, g_allocation_calls.load() - alloc_calls0);
    }
    ctx->scene_dirty = false;
    ctx->bvh2_stale = false;
    return PT_OK;
}

// Flatten meshes to world space (RayTracingShader.compute:236-240 applied once
// instead of per ray), box every primitive, build the BVH, upload.
int rebuild_scene(pt_context *ctx)
{
    auto t0 = std::chrono::steady_clock::now();
    const int nS = (int)ctx->spheres.size();
    struct Tri { pt_f3 v0, v1, v2; };
    std::vector<Tri> tris;                 // the triangles kept as world-space records ("flattened")
    std::vector<int32_t> tri_object;       // their mesh object
    std::vector<uint32_t> tri_gid;         // and their index among all triangles of the scene in the reference's visiting order
    size_t nt = 0;
    std::vector<uint32_t> obj_first_tri(ctx->objects.size() + 1, 0);
    for (size_t oi = 0; oi < ctx->objects.size(); oi++) {
        obj_first_tri[oi] = (uint32_t)nt;
        nt += (size_t)std::max(0, ctx->objects[oi].indices_count) / 3;
    }
    obj_first_tri[ctx->objects.size()] = (uint32_t)nt;
    // ---- which mesh objects become instances of a shared tree
    std::vector<int> obj_group(ctx->objects.size(), -1);
    std::vector<int32_t> obj_base;          // smallest vertex index of every candidate object's range
    std::vector<MeshGroup> groups;
    // instancing 2: shared meshes are built once and copied per object on the device (expand_instances below); the scene the
    // kernels see is flat, so every backend and the exhaustive ray query work on it unchanged
    const bool expand = ctx->instancing == 2;
#if !defined(PT_NODES_SOA) && !defined(PT_NODES_3LOAD)
    if (ctx->instancing && !ctx->flatten_once && ctx->bvh_width == 4) groups = find_instances(ctx->objects, ctx->indices, ctx->vertices, obj_group, obj_base, expand);
#endif
    ctx->flatten_once = false;
#if !defined(PT_NODES_SOA) && !defined(PT_NODES_3LOAD)
    // Copies of shared meshes (instancing 2) spare the host the big tree, not the small one: the mesh's own tree is still the
    // host builder's, 70 ms for 262 k triangles.  The device builds ALL ten million of that scene's triangles in 28 ms, so
    // copying only pays when the meshes are small against the scene: by the two builders' rates, a hundredth of it or less.
    bool flat_is_quicker = groups.empty();
    if (!groups.empty() && expand) {
        size_t mesh_tris = 0;
        for (const MeshGroup &g : groups) mesh_tris += (size_t)g.indices_count / 3;
        flat_is_quicker = 100 * mesh_tris > nt;
    }
    if (flat_is_quicker && ctx->bvh_width == 4 &&
        (ctx->tree_build == 1 || ctx->tree_build == 3 || (ctx->tree_build == 2 && (size_t)nS + nt >= kDeviceBuildMinPrims))) {
        const int rc = build_scene_on_device(ctx, obj_first_tri, nt, t0);
        if (rc != kDeviceBuildFallback) return rc;
    }
#endif
    ctx->device_built = false;
    for (int k = 0; k < 2; k++) { free_dev(ctx->build_scratch[k]); ctx->build_scratch_bytes[k] = 0; }      // the host builds this scene: nothing to keep scratch for
    std::vector<Aabb> group_bounds(groups.size()); std::vector<char> group_bounds_done(groups.size(), 0);
    {
        size_t n_flat = 0;
        for (size_t oi = 0; oi < ctx->objects.size(); oi++)
            if (obj_group[oi] < 0) n_flat += obj_first_tri[oi + 1] - obj_first_tri[oi];
        tris.reserve(n_flat); tri_object.reserve(n_flat); tri_gid.reserve(n_flat);
    }
    ctx->object_box.assign(ctx->objects.size(), Aabb{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}});
    for (size_t oi = 0; oi < ctx->objects.size(); oi++) {
        const PtMeshObject &mo = ctx->objects[oi];
        const int32_t end = mo.indices_offset + mo.indices_count;
        Aabb &ob = ctx->object_box[oi];
        uint32_t gid = obj_first_tri[oi];
        if (obj_group[oi] >= 0) {          // instanced: only the object-space bounds are needed here, once per shared mesh
            const size_t gi = (size_t)obj_group[oi];
            if (!group_bounds_done[gi]) {
                Aabb gb; box_reset_inf(gb);
                for (int32_t i = mo.indices_offset; i + 2 < end; i += 3)
                    for (int c = 0; c < 3; c++) {
                        const float *v = &ctx->vertices[3 * (size_t)ctx->indices[i + c]];
                        for (int k = 0; k < 3; k++)
                            if (std::fabs(v[k]) < kHugeCoord) { gb.lo[k] = std::min(gb.lo[k], v[k]); gb.hi[k] = std::max(gb.hi[k], v[k]); }
                    }
                group_bounds[gi] = gb; group_bounds_done[gi] = true;
            }
            ob = group_bounds[gi];
            continue;
        }
        for (int32_t i = mo.indices_offset; i + 2 < end; i += 3, gid++) {
            tri_object.push_back((int32_t)oi);
            tri_gid.push_back(gid);
            const float *a = &ctx->vertices[3 * (size_t)ctx->indices[i]];
            const float *b = &ctx->vertices[3 * (size_t)ctx->indices[i + 1]];
            const float *c = &ctx->vertices[3 * (size_t)ctx->indices[i + 2]];
            for (const float *v : {a, b, c})
                for (int k = 0; k < 3; k++)
                    if (std::fabs(v[k]) < kHugeCoord) { ob.lo[k] = std::min(ob.lo[k], v[k]); ob.hi[k] = std::max(ob.hi[k], v[k]); }
            Tri t;
            t.v0 = pt_mul_point(mo.local_to_world, pt_make3(a[0], a[1], a[2]));
            t.v1 = pt_mul_point(mo.local_to_world, pt_make3(b[0], b[1], b[2]));
            t.v2 = pt_mul_point(mo.local_to_world, pt_make3(c[0], c[1], c[2]));
            tris.push_back(t);
        }
    }
    const size_t nP = (size_t)nS + tris.size();        // primitives with a world-space record
    const size_t nIds = (size_t)nS + nt;               // primitives of the scene (ids 1 .. nIds)
    if (nIds >= (1u << 27)) return fail(ctx, PT_ERR_UNSUPPORTED, "more than 2^27 primitives");

    std::vector<Aabb> boxes(nP);
    float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < nS; i++) {
        const PtSphere &s = ctx->spheres[i];
        float r = std::fabs(s.radius);
        for (int k = 0; k < 3; k++) { boxes[i].lo[k] = s.position[k] - r; boxes[i].hi[k] = s.position[k] + r; }
    }
    for (size_t j = 0; j < tris.size(); j++) {
        const float *a = &tris[j].v0.x, *b = &tris[j].v1.x, *c = &tris[j].v2.x;
        Aabb &bx = boxes[nS + j];
        for (int k = 0; k < 3; k++) {
            bx.lo[k] = std::min(a[k], std::min(b[k], c[k]));
            bx.hi[k] = std::max(a[k], std::max(b[k], c[k]));
        }
    }
    for (size_t i = 0; i < nP; i++)
        for (int k = 0; k < 3; k++) {
            if (std::fabs(boxes[i].lo[k]) < kHugeCoord) slo[k] = std::min(slo[k], boxes[i].lo[k]);
            if (std::fabs(boxes[i].hi[k]) < kHugeCoord) shi[k] = std::max(shi[k], boxes[i].hi[k]);
        }
    for (size_t oi = 0; oi < ctx->objects.size(); oi++) {       // instanced objects: the corners of their object-space bounds
        const Aabb &ob = ctx->object_box[oi];
        if (obj_group[oi] < 0 || !(ob.lo[0] <= ob.hi[0])) continue;
        for (int corner = 0; corner < 8; corner++) {
            const pt_f3 w = pt_mul_point(ctx->objects[oi].local_to_world, pt_make3((corner & 1) ? ob.hi[0] : ob.lo[0], (corner & 2) ? ob.hi[1] : ob.lo[1],
                                                                                   (corner & 4) ? ob.hi[2] : ob.lo[2]));
            const float wc[3] = {w.x, w.y, w.z};
            for (int k = 0; k < 3; k++)
                if (std::fabs(wc[k]) < kHugeCoord) { slo[k] = std::min(slo[k], wc[k]); shi[k] = std::max(shi[k], wc[k]); }
        }
    }
    // Pad every box: the intersection tests accept hits a few ulps outside the exact primitive.
    float diag = 0.0f;
    if (nP) diag = std::sqrt((shi[0] - slo[0]) * (shi[0] - slo[0]) + (shi[1] - slo[1]) * (shi[1] - slo[1]) +
                             (shi[2] - slo[2]) * (shi[2] - slo[2]));
    if (!std::isfinite(diag)) diag = 0.0f;
    // A general margin under the specific ones below (a triangle's own fuzz, the spheres' slack, the relative term for the slab
    // arithmetic): about what a well-shaped triangle's own fuzz comes to.  It was 1e-5 * diag; rays that graze a tessellated
    // floor then cross the margin of every triangle on their way, and on the 10 M-triangle scene (diag 600) that cost 6 % of
    // the rays per second flattened and 20 % two-level.  With a hundredth of the old value (PTRACE_PAD_SCALE=0.0333 on this
    // one) the benchmark scenes still render bit-identical images and a million soak frames still match the exhaustive loop.
    float pad = 3e-6f * diag + 1e-6f;
    if (const char *env = std::getenv("PTRACE_PAD_SCALE")) pad *= std::max(0.0f, (float)std::atof(env));      // measurement knob
    // IntersectSphere (SH:170-190) decides hit or miss by the sign of p1*p1 - dot(d,d) + r*r with d = origin - centre:
    // a difference of squares of the ray's distance to the sphere, so rounding lets it accept rays that pass up to
    // sqrt(r*r + k*eps*|d|^2) from the centre (for r = 0 that is |d| * 7e-4, far more than `pad`).  The box must hold
    // that fuzzy sphere, or the tree would cull hits the exhaustive loop finds: radius sqrt(r*r + slack2) with k = 8 and
    // |d| <= `reach`.
    // rays that start inside the scene are at most one diagonal away; cameras farther out raise the request (ensure_scene)
    const float reach = std::max(2.0f * diag, ctx->sphere_reach_request);
    const float sphere_slack2 = 8.0f * 5.9604645e-8f * reach * reach;
    ctx->sphere_reach = reach;
    for (int k = 0; k < 3; k++) {
        ctx->scene_lo[k] = slo[k] <= shi[k] ? slo[k] : 0.0f; ctx->scene_hi[k] = slo[k] <= shi[k] ? shi[k] : 0.0f;
        ctx->spheres_lo[k] = INFINITY; ctx->spheres_hi[k] = -INFINITY;
    }
    for (int i = 0; i < nS; i++)
        for (int k = 0; k < 3; k++) {
            if (std::fabs(boxes[i].lo[k]) < kHugeCoord) ctx->spheres_lo[k] = std::min(ctx->spheres_lo[k], boxes[i].lo[k]);
            if (std::fabs(boxes[i].hi[k]) < kHugeCoord) ctx->spheres_hi[k] = std::max(ctx->spheres_hi[k], boxes[i].hi[k]);
        }
    for (int i = 0; i < nS; i++) {
        const PtSphere &s = ctx->spheres[i];
        const float r = std::sqrt(s.radius * s.radius + sphere_slack2);
        for (int k = 0; k < 3; k++) { boxes[i].lo[k] = s.position[k] - r; boxes[i].hi[k] = s.position[k] + r; }
    }
    // ---- references: every sphere is one; a large triangle may be several (bvh.hpp, split_references)
    static_assert(sizeof(Tri) == 36, "nine floats per triangle");
    std::vector<uint32_t> ref_prim;
    {
        std::vector<Aabb> rboxes;
        ref_prim.reserve(nP); rboxes.reserve(nP);
        for (int i = 0; i < nS; i++) { ref_prim.push_back((uint32_t)i); rboxes.push_back(boxes[i]); }
        std::vector<uint32_t> ref_tri;
        split_references(tris.empty() ? nullptr : &tris[0].v0.x, boxes.data() + nS, tris.size(), ctx->scene_lo, ctx->scene_hi,
                         ctx->split_budget, ref_tri, rboxes);
        for (uint32_t t : ref_tri) ref_prim.push_back((uint32_t)nS + t);
        boxes.swap(rboxes);
    }
    const size_t nR = ref_prim.size();
    if (nR >= (1u << 28)) return fail(ctx, PT_ERR_UNSUPPORTED, "more than 2^28 primitive references");
    std::vector<Aabb> clip;            // the unpadded boxes: what a refit may cut a static triangle's new box down to
    if (nR != nP) clip = boxes;
    for (size_t i = 0; i < nR; i++) {
        float own = 0.0f;      // what this primitive's own intersection arithmetic needs beyond the general padding
        if (ref_prim[i] >= (uint32_t)nS) {
            const Tri &t = tris[ref_prim[i] - nS];
            own = triangle_fuzz(pt_sub3(t.v1, t.v0), pt_sub3(t.v2, t.v0), reach);
            if (!(own < kHugeCoord)) own = kHugeCoord;           // zero area, NaN: the huge box below
        }
        for (int k = 0; k < 3; k++) {
            float l = boxes[i].lo[k], h = boxes[i].hi[k];
            float rel = 4e-7f * std::max(std::fabs(l), std::fabs(h));
            const float p = std::max(pad, own);
            if (!(std::fabs(l) + p < kHugeCoord) || !(std::fabs(h) + p < kHugeCoord)) { boxes[i].lo[k] = -kHugeCoord; boxes[i].hi[k] = kHugeCoord; continue; }
            boxes[i].lo[k] = l - p - rel; boxes[i].hi[k] = h + p + rel;
        }
    }

    int leaf_size = 2;   // small leaves keep the lanes of a leaf phase in step; measured best of 1..4
    if (const char *env = std::getenv("PTRACE_LEAF")) leaf_size = std::max(1, std::min(4, std::atoi(env)));
    float ct = 1.0f;
    if (const char *env = std::getenv("PTRACE_SAH_CT")) ct = (float)std::atof(env);
    auto t_prep = std::chrono::steady_clock::now();

    // ---- two-level scenes: one tree per shared mesh over its object-space triangles, one top-level reference per instance
    typedef CachedMesh MeshTree;
    std::vector<const MeshTree *> mesh_trees(groups.size(), nullptr);
    std::vector<int> inst_object;          // instance -> mesh object
    int blas_stack_need = 0;
    if (groups.empty()) ctx->mesh_cache.clear();
    ctx->mesh_cache.reserve(ctx->mesh_cache.size() + groups.size());      // pointers into it stay valid through the loop
    for (size_t gi = 0; gi < groups.size(); gi++) {
        const MeshGroup &g = groups[gi];
        const size_t n = (size_t)g.indices_count / 3;
        // How far, in object space, the ray's point at the hit's parameter can lie from the object-space triangle when the
        // world-space test accepts the hit: the world padding of the flattened tree, the rounding of the world vertices and
        // of the ray's transform (coordinates up to `coord`, a few ulps each), all taken through the inverse map.
        float coord = reach;
        for (int k = 0; k < 3; k++) coord = std::max(coord, std::max(std::fabs(ctx->scene_lo[k]), std::fabs(ctx->scene_hi[k])) + reach);
        const float pad_need = (float)(g.inv_norm * 1.25 * ((double)pad + 4.8e-7 * (3.0 * coord + reach)));
        const float reach_need = (float)(reach / g.scale_min);
        // a kept tree serves if it was padded for at least this much and not for absurdly more; a new one gets some room, so
        // that instances can move and the scene can grow a little before the mesh has to be built again
        CachedMesh *hit = nullptr;
        for (CachedMesh &cm : ctx->mesh_cache)
            if (cm.indices_offset == g.indices_offset && cm.indices_count == g.indices_count) hit = &cm;
        // (expanded copies are refitted in world space: any padding the kept tree was built with will do)
        if (hit && hit->leaf_size == leaf_size && (expand || (hit->pad >= pad_need && hit->pad <= 1.3f * pad_need && hit->reach >= reach_need && hit->reach <= 2.0f * reach_need))) {
            mesh_trees[gi] = hit;
        } else {
        if (!hit) { ctx->mesh_cache.emplace_back(); hit = &ctx->mesh_cache.back(); }
        CachedMesh &mt = *hit;
        mt.indices_offset = g.indices_offset; mt.indices_count = g.indices_count; mt.leaf_size = leaf_size;
        mt.version = ++ctx->mesh_versions;
        const float pad_o = 1.1f * pad_need, reach_o = 1.25f * reach_need;      // wider boxes cost rays: 1.5 x the padding is 20 % fewer rays per second
        mt.pad = pad_o; mt.reach = reach_o;
        std::vector<Aabb> ob(n);
        box_reset_inf(mt.root);
        for (size_t t = 0; t < n; t++) {
            const int32_t *ix = &ctx->indices[(size_t)g.indices_offset + 3 * t];
            const float *a = &ctx->vertices[3 * (size_t)ix[0]], *b = &ctx->vertices[3 * (size_t)ix[1]], *c = &ctx->vertices[3 * (size_t)ix[2]];
            float own = 2.0f * triangle_fuzz(pt_make3(b[0] - a[0], b[1] - a[1], b[2] - a[2]), pt_make3(c[0] - a[0], c[1] - a[1], c[2] - a[2]), reach_o);
            if (!(own < kHugeCoord)) own = kHugeCoord;
            for (int k = 0; k < 3; k++) {
                const float l = std::min(a[k], std::min(b[k], c[k])), h = std::max(a[k], std::max(b[k], c[k]));
                const float rel = 4e-7f * std::max(std::fabs(l), std::fabs(h));
                const float pp = std::max(pad_o, own);
                if (!(std::fabs(l) + pp < kHugeCoord) || !(std::fabs(h) + pp < kHugeCoord)) { ob[t].lo[k] = -kHugeCoord; ob[t].hi[k] = kHugeCoord; }
                else { ob[t].lo[k] = l - pp - rel; ob[t].hi[k] = h + pp + rel; }
                mt.root.lo[k] = std::min(mt.root.lo[k], ob[t].lo[k]); mt.root.hi[k] = std::max(mt.root.hi[k], ob[t].hi[k]);
            }
        }
        mt.bvh = build_bvh(ob, leaf_size, 0, ct);
        mt.bvh.nodes.clear(); mt.bvh.nodes.shrink_to_fit();      // the two-child tree is not used for instances
        mt.recs.assign((n + 1) * 3, make_float4(0, 0, 0, 0));      // the last record stays zero (see rec_cap below)
        for (size_t slot = 0; slot < n; slot++) {
            const uint32_t t = mt.bvh.order[slot];
            const int32_t *ix = &ctx->indices[(size_t)g.indices_offset + 3 * t];
            const float *a = &ctx->vertices[3 * (size_t)ix[0]], *b = &ctx->vertices[3 * (size_t)ix[1]], *c = &ctx->vertices[3 * (size_t)ix[2]];
            float tf; const int32_t ti = (int32_t)t; std::memcpy(&tf, &ti, 4);
            mt.recs[slot * 3 + 0] = make_float4(a[0], a[1], a[2], tf);
            mt.recs[slot * 3 + 1] = make_float4(b[0], b[1], b[2], 0.0f);
            mt.recs[slot * 3 + 2] = make_float4(c[0], c[1], c[2], 0.0f);
        }
        mesh_trees[gi] = &mt;
        }
        const MeshTree &mt = *mesh_trees[gi];
        blas_stack_need = std::max(blas_stack_need, mt.bvh.stack_need4);
        // a top-level reference per member: the world box of the mesh's root box under the member's matrix
        for (int o : g.objects) {
            const float *m = ctx->objects[(size_t)o].local_to_world;
            Aabb wb; box_reset_inf(wb);
            for (int corner = 0; corner < 8; corner++) {
                const double v[3] = {(corner & 1) ? mt.root.hi[0] : mt.root.lo[0], (corner & 2) ? mt.root.hi[1] : mt.root.lo[1], (corner & 4) ? mt.root.hi[2] : mt.root.lo[2]};
                for (int k = 0; k < 3; k++) {
                    const double w = (double)m[k] * v[0] + (double)m[4 + k] * v[1] + (double)m[8 + k] * v[2] + (double)m[12 + k];
                    wb.lo[k] = std::min(wb.lo[k], (float)w); wb.hi[k] = std::max(wb.hi[k], (float)w);
                }
            }
            for (int k = 0; k < 3; k++) {
                const float p2 = pad + 1e-6f * std::max(std::fabs(wb.lo[k]), std::fabs(wb.hi[k]));
                if (!(std::fabs(wb.lo[k]) + p2 < kHugeCoord) || !(std::fabs(wb.hi[k]) + p2 < kHugeCoord)) { wb.lo[k] = -kHugeCoord; wb.hi[k] = kHugeCoord; }
                else { wb.lo[k] -= p2; wb.hi[k] += p2; }
            }
            ref_prim.push_back((uint32_t)(nP + inst_object.size()));
            boxes.push_back(wb);
            inst_object.push_back(o);
        }
    }
    const size_t nI = inst_object.size();
    const size_t nRefs = ref_prim.size();                  // top-level references: nR of primitives, then one per instance

    const bool want8 = ctx->bvh_width == 8 && nI == 0;
    // expanded copies hang off the top-level tree as subtrees: every instance reference needs a leaf of its own
    BvhResult bvh = build_bvh(boxes, expand && nI ? 1 : leaf_size, 0, ct, want8);
    // the eight-wide tree is walked if a lane's stack holds it; a tree too deep for that (a degenerate chain) is walked as the
    // two-child tree of the same build, whose records follow the other order
    const bool walk8 = want8 && !bvh.nodes8.empty() && bvh.stack_need8 <= group_stack_limit(ctx);
#ifdef PT_NODES_3LOAD
    // measurement build: the four-wide tree renumbered for nodes without links (before anything reads `order`)
    const Consecutive4 c4 = consecutive_layout(bvh);
    if (!c4.ok) return fail(ctx, PT_ERR_UNSUPPORTED, "PT_NODES_3LOAD: this tree has no link-free layout");
#endif
    const std::vector<uint32_t> &order = walk8 ? bvh.order8 : bvh.order;
    if (walk8) { bvh.nodes.clear(); bvh.nodes4.clear(); }
    else { bvh.nodes8.clear(); }
    auto t_built = std::chrono::steady_clock::now();
    if (nI && expand) {
        const int need = bvh.stack_need4 + blas_stack_need;
        if (need > ctx->stack_limit) { ctx->flatten_once = true; return rebuild_scene(ctx); }
        bvh.stack_need4 = need;
    } else if (nI) {
        // entering an instance costs the exit marker, and a top-level leaf can stack a link per instance record it holds
        const int need = bvh.stack_need4 + leaf_size + 1 + blas_stack_need;
        if (need > ctx->stack_limit) { ctx->flatten_once = true; return rebuild_scene(ctx); }
        bvh.stack_need4 = need;
    }

    // records on the device: the top level's (in two-level scenes with room to grow), then every mesh tree's
    std::vector<uint64_t> sig;
    for (const MeshTree *mt : mesh_trees) sig.push_back(mt->version);
    const bool keep_mesh_trees = nI && sig == ctx->inst_sig && bvh.nodes4.size() <= ctx->inst_node_cap && nRefs < ctx->inst_rec_cap &&
                                 ctx->d_nodes4 && ctx->d_prims;
    const size_t node_cap = !nI ? bvh.nodes4.size() : keep_mesh_trees ? ctx->inst_node_cap : std::max<size_t>(256, 2 * bvh.nodes4.size());
    // + 1: the record after the last one is all zeros, a triangle nothing hits - where the links of unused node slots point (bvh.hpp)
    const size_t rec_cap = !nI ? nRefs + 1 : keep_mesh_trees ? ctx->inst_rec_cap : std::max<size_t>(256, 2 * (nRefs + 1));
    size_t n_recs = rec_cap;
    for (const MeshTree *mt : mesh_trees) n_recs += mt->recs.size() / 3;
    if (n_recs >= (1u << 27)) return fail(ctx, PT_ERR_UNSUPPORTED, "more than 2^27 primitive records");
    std::vector<float4> recs((keep_mesh_trees ? nRefs + 1 : n_recs) * 3, make_float4(0, 0, 0, 0));
    for (size_t slot = 0; slot < nRefs; slot++) {
        uint32_t prim = ref_prim[order[slot]];
        if (prim >= nP) {                                  // an instance: kind 2, b.x = its index
            const int32_t ii = (int32_t)(prim - nP), two = 2;
            float iif, kind; std::memcpy(&iif, &ii, 4); std::memcpy(&kind, &two, 4);
            recs[slot * 3 + 0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            recs[slot * 3 + 1] = make_float4(iif, 0.0f, 0.0f, kind);
            recs[slot * 3 + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            continue;
        }
        int32_t id = (int)prim < nS ? 1 + (int32_t)prim : 1 + nS + (int32_t)tri_gid[prim - nS];
        float idf; std::memcpy(&idf, &id, 4);
        if ((int)prim < nS) {
            const PtSphere &s = ctx->spheres[prim];
            int32_t one = 1; float kind; std::memcpy(&kind, &one, 4);
            recs[slot * 3 + 0] = make_float4(s.position[0], s.position[1], s.position[2], idf);
            recs[slot * 3 + 1] = make_float4(s.radius, 0.0f, 0.0f, kind);
            recs[slot * 3 + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            const Tri &t = tris[prim - nS];
            pt_f3 e1 = pt_sub3(t.v1, t.v0), e2 = pt_sub3(t.v2, t.v0);
            recs[slot * 3 + 0] = make_float4(t.v0.x, t.v0.y, t.v0.z, idf);
            recs[slot * 3 + 1] = make_float4(e1.x, e1.y, e1.z, 0.0f);
            float objf; std::memcpy(&objf, &tri_object[prim - nS], 4);
            recs[slot * 3 + 2] = make_float4(e2.x, e2.y, e2.z, objf);
        }
    }
    if (nI && !expand) {
        // a top-level leaf that holds one instance and nothing else: its parent links to the instance itself
        for (Bvh4Node &nd : bvh.nodes4)
            for (int k = 0; k < 4; k++) {
                if (nd.child[k] >= 0 || !((nd.valid >> k) & 1)) continue;      // an unused slot links to the record past the last one
                const uint32_t enc = (uint32_t)~nd.child[k];
                if ((enc & 7u) != 0) continue;
                const uint32_t prim = ref_prim[order[enc >> 3]];
                if (prim >= nP) nd.child[k] = (int32_t)~(kInstanceBit | (uint32_t)(prim - nP));
            }
    }
    std::vector<float4> smat((size_t)nS * 4);
    for (int i = 0; i < nS; i++) {
        const PtSphere &s = ctx->spheres[i];
        smat[i * 4 + 0] = make_float4(s.position[0], s.position[1], s.position[2], s.radius);
        smat[i * 4 + 1] = make_float4(s.albedo[0], s.albedo[1], s.albedo[2], s.smoothness);
        smat[i * 4 + 2] = make_float4(s.specular[0], s.specular[1], s.specular[2], 0.0f);
        smat[i * 4 + 3] = make_float4(s.emission[0], s.emission[1], s.emission[2], 0.0f);
    }
    int rc;
    if (nI && expand) {
        // ---- option "instancing" 2: one copy of the shared tree per object, made and refitted on the device (pt_refit.hpp)
        const size_t n_top = bvh.nodes4.size();
        std::vector<int4> inst_tab(nI);                 // (first node, first record, object, smallest vertex index)
        std::vector<int32_t> inst_id0(nI);
        std::vector<size_t> group_first_inst(groups.size() + 1, 0);
        size_t node_next = n_top, rec_next = nRefs + 1;
        {
            // instances were appended group by group, members in scene order
            size_t ii = 0;
            for (size_t gi = 0; gi < groups.size(); gi++) {
                group_first_inst[gi] = ii;
                const MeshTree &mt = *mesh_trees[gi];
                for (size_t m = 0; m < groups[gi].objects.size(); m++, ii++) {
                    const int o = inst_object[ii];
                    inst_tab[ii] = make_int4((int)node_next, (int)rec_next, o, obj_base[(size_t)o]);
                    inst_id0[ii] = 1 + nS + (int32_t)obj_first_tri[(size_t)o];
                    node_next += mt.bvh.nodes4.size();
                    rec_next += (size_t)groups[gi].indices_count / 3 + 1;
                }
            }
            group_first_inst[groups.size()] = ii;
        }
        const size_t total_nodes = node_next, total_recs = rec_next;
        if (total_recs >= (1u << 27) || total_nodes >= (1u << 30)) return fail(ctx, PT_ERR_UNSUPPORTED, "more than 2^27 primitive records");
        // top-level levels (an instance reference is a leaf here; its link is patched to the copy's root afterwards)
        std::vector<int> level(n_top, 0);
        int max_level = 0;
        for (size_t n = 0; n < n_top; n++)
            for (int k = 0; k < 4; k++)
                if (((bvh.nodes4[n].valid >> k) & 1) && bvh.nodes4[n].child[k] >= 0) {
                    level[(size_t)bvh.nodes4[n].child[k]] = level[n] + 1;
                    max_level = std::max(max_level, level[n] + 1);
                }
        std::vector<std::vector<int32_t>> top_levels((size_t)max_level + 1);
        for (size_t n = 0; n < n_top; n++) top_levels[(size_t)level[n]].push_back((int32_t)n);
        for (Bvh4Node &nd : bvh.nodes4)
            for (int k = 0; k < 4; k++) {
                if (nd.child[k] >= 0 || !((nd.valid >> k) & 1)) continue;
                const uint32_t enc = (uint32_t)~nd.child[k];
                const uint32_t prim = ref_prim[order[enc >> 3]];             // leaves hold one reference each
                if (prim >= nP) nd.child[k] = inst_tab[prim - nP].x;          // the copy's root: an inner node like any other
            }
        // top-level slots: where a refit takes each record from
        std::vector<int4> slot_src(nRefs + 1, make_int4(-1, 0, 0, 0));
        {
            size_t j = 0;
            std::vector<int4> tri_src(tris.size());
            for (size_t o = 0; o < ctx->objects.size(); o++) {
                if (obj_group[o] >= 0) continue;
                const PtMeshObject &mo = ctx->objects[o];
                const int32_t end = mo.indices_offset + mo.indices_count;
                for (int32_t i = mo.indices_offset; i + 2 < end; i += 3, j++) tri_src[j] = make_int4((int)o, ctx->indices[i], ctx->indices[i + 1], ctx->indices[i + 2]);
            }
            for (size_t slot = 0; slot < nRefs; slot++) {
                const uint32_t prim = ref_prim[order[slot]];
                if (prim < nP && (int)prim >= nS) slot_src[slot] = tri_src[prim - nS];
            }
        }
        free_dev(ctx->d_inst); free_dev(ctx->d_objects); free_dev(ctx->d_indices); free_dev(ctx->d_nodes); free_dev(ctx->d_nodes8);
        free_dev(ctx->d_nodes4); free_dev(ctx->d_prims); free_dev(ctx->d_slot_src); free_dev(ctx->d_prim_box); free_dev(ctx->d_node_box);
        free_dev(ctx->d_slot_clip); free_dev(ctx->d_build_matrices); free_dev(ctx->d_level_nodes);
        PT_HIP(ctx, dev_malloc((void **)&ctx->d_nodes4, total_nodes * sizeof(Bvh4Node)));
        PT_HIP(ctx, dev_malloc((void **)&ctx->d_prims, total_recs * 3 * sizeof(float4)));
        PT_HIP(ctx, dev_malloc((void **)&ctx->d_slot_src, total_recs * sizeof(int4)));
        PT_HIP(ctx, dev_malloc((void **)&ctx->d_prim_box, total_recs * 6 * sizeof(float)));
        PT_HIP(ctx, dev_malloc((void **)&ctx->d_node_box, total_nodes * 6 * sizeof(float)));
        PT_HIP(ctx, hipMemcpyAsync(ctx->d_nodes4, bvh.nodes4.data(), n_top * sizeof(Bvh4Node), hipMemcpyHostToDevice, ctx->stream));
        PT_HIP(ctx, hipMemcpyAsync(ctx->d_prims, recs.data(), (nRefs + 1) * 3 * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
        PT_HIP(ctx, hipMemcpyAsync(ctx->d_slot_src, slot_src.data(), (nRefs + 1) * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<float> mats(ctx->objects.size() * 16);
        for (size_t o = 0; o < ctx->objects.size(); o++) std::memcpy(&mats[o * 16], ctx->objects[o].local_to_world, 64);
        if ((rc = upload(ctx, ctx->d_vertices, ctx->vertices.data(), ctx->vertices.size() * sizeof(float)))) return rc;
        if ((rc = upload(ctx, ctx->d_matrices, mats.data(), mats.size() * sizeof(float)))) return rc;
        if ((rc = upload(ctx, ctx->d_sphere_mat, smat.data(), smat.size() * sizeof(float4)))) return rc;
        int32_t *d_id0 = nullptr;
        struct Temp { std::vector<void *> p; ~Temp() { for (void *q : p) (void)dev_free(q); } } temp;
        free_expanded(ctx);
        if ((rc = upload(ctx, ctx->d_exp_inst, inst_tab.data(), nI * sizeof(int4)))) return rc;
        int4 *const d_inst_tab = ctx->d_exp_inst;
        if ((rc = upload(ctx, d_id0, inst_id0.data(), nI * sizeof(int32_t)))) return rc;
        temp.p.push_back(d_id0);
        RefitArgs ra;
        ra.slot_src = ctx->d_slot_src; ra.vertices = ctx->d_vertices; ra.matrices = ctx->d_matrices; ra.prims = ctx->d_prims;
        ra.prim_box = ctx->d_prim_box; ra.nodes4 = ctx->d_nodes4; ra.nodes8 = nullptr; ra.node_box = ctx->d_node_box; ra.level_nodes = nullptr;
        ra.pad = pad; ra.sphere_slack2 = sphere_slack2; ra.reach = reach; ra.n_prims = (int32_t)total_recs;
        ra.slot_clip = nullptr; ra.build_matrices = nullptr;
        ctx->exp_groups.assign(groups.size(), pt_context::ExpandedGroup());
        for (size_t gi = 0; gi < groups.size(); gi++) {
            const MeshTree &mt = *mesh_trees[gi];
            const size_t n_mesh_nodes = mt.bvh.nodes4.size(), n_mesh_slots = (size_t)groups[gi].indices_count / 3;
            const size_t gI = group_first_inst[gi + 1] - group_first_inst[gi];
            std::vector<int4> mslots(n_mesh_slots);
            const int32_t base_rep = obj_base[(size_t)groups[gi].objects[0]];
            for (size_t sl = 0; sl < n_mesh_slots; sl++) {
                const uint32_t t = mt.bvh.order[sl];
                const int32_t *ix = &ctx->indices[(size_t)groups[gi].indices_offset + 3 * (size_t)t];
                mslots[sl] = make_int4(ix[0] - base_rep, ix[1] - base_rep, ix[2] - base_rep, (int)t);
            }
            // the mesh tree's levels, deepest first (nodes are stored parents first)
            std::vector<int> mlevel(n_mesh_nodes, 0);
            int mmax = 0;
            for (size_t n = 0; n < n_mesh_nodes; n++)
                for (int k = 0; k < 4; k++)
                    if (((mt.bvh.nodes4[n].valid >> k) & 1) && mt.bvh.nodes4[n].child[k] >= 0) {
                        mlevel[(size_t)mt.bvh.nodes4[n].child[k]] = mlevel[n] + 1;
                        mmax = std::max(mmax, mlevel[n] + 1);
                    }
            std::vector<int> count((size_t)mmax + 2, 0);
            for (size_t n = 0; n < n_mesh_nodes; n++) count[(size_t)(mmax - mlevel[n]) + 1]++;
            for (int l = 0; l <= mmax; l++) count[(size_t)l + 1] += count[(size_t)l];
            std::vector<int32_t> lnodes(n_mesh_nodes);
            std::vector<int> cursor(count.begin(), count.end() - 1);
            for (size_t n = 0; n < n_mesh_nodes; n++) lnodes[(size_t)cursor[(size_t)(mmax - mlevel[n])]++] = (int32_t)n;
            ctx->exp_groups[gi].level_first = count;
            ctx->exp_groups[gi].first_inst = group_first_inst[gi]; ctx->exp_groups[gi].n_inst = gI;
            uint4 *d_mnodes = nullptr; int4 *d_mslots = nullptr;
            if ((rc = upload(ctx, d_mnodes, mt.bvh.nodes4.data(), n_mesh_nodes * sizeof(Bvh4Node)))) return rc;
            temp.p.push_back(d_mnodes);
            if ((rc = upload(ctx, d_mslots, mslots.data(), n_mesh_slots * sizeof(int4)))) return rc;
            temp.p.push_back(d_mslots);
            if ((rc = upload(ctx, ctx->exp_groups[gi].d_level_nodes, lnodes.data(), lnodes.size() * sizeof(int32_t)))) return rc;
            ExpandArgs ea;
            ea.mesh_nodes = d_mnodes; ea.mesh_slots = d_mslots; ea.inst = d_inst_tab + group_first_inst[gi]; ea.inst_id0 = d_id0 + group_first_inst[gi];
            ea.n_mesh_nodes = (int32_t)n_mesh_nodes; ea.n_mesh_slots = (int32_t)n_mesh_slots; ea.n_inst = (int32_t)gI;
            ea.nodes4 = ctx->d_nodes4; ea.prims = ctx->d_prims; ea.slot_src = ctx->d_slot_src;
            const size_t tn = gI * n_mesh_nodes, ts = gI * (n_mesh_slots + 1);
            hipLaunchKernelGGL(expand_nodes_kernel, dim3((unsigned)((tn + kRefitBlock - 1) / kRefitBlock)), dim3(kRefitBlock), 0, ctx->stream, ea);
            hipLaunchKernelGGL(expand_slots_kernel, dim3((unsigned)((ts + kRefitBlock - 1) / kRefitBlock)), dim3(kRefitBlock), 0, ctx->stream, ea);
        }
        PT_HIP(ctx, hipGetLastError());
        {
            std::vector<int32_t> tl;
            ctx->exp_top_first.clear();
            for (int l = max_level; l >= 0; l--) { ctx->exp_top_first.push_back((int)tl.size()); tl.insert(tl.end(), top_levels[(size_t)l].begin(), top_levels[(size_t)l].end()); }
            ctx->exp_top_first.push_back((int)tl.size());
            if ((rc = upload(ctx, ctx->d_exp_top_levels, tl.data(), tl.size() * sizeof(int32_t)))) return rc;
        }
        if ((rc = refit_expanded(ctx, ra))) return rc;
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));       // the temporaries go when this returns
        ctx->n_inst = 0; ctx->inst_sig.clear(); ctx->inst_node_cap = ctx->inst_rec_cap = 0;
        ctx->refit_ready = ctx->refit != 0;                    // a transform update re-runs the refit passes over the copies (refit_expanded)
        ctx->walk8 = false; ctx->n_nodes8 = 0; ctx->stack_need8 = 0;
        ctx->n_nodes4 = (int)total_nodes; ctx->n_nodes = (int)total_nodes;     // no two-child tree: n_nodes only says "there is a tree"
        ctx->n_prims = (int)nIds; ctx->n_slots = (int)total_recs; ctx->n_top_records = (int)total_recs; ctx->n_spheres = nS;
        ctx->stack_need4 = bvh.stack_need4;
        ctx->box_pad = pad; ctx->sphere_slack2 = sphere_slack2;
        ctx->expanded = true;
        ctx->stats.bvh_nodes = (int)total_nodes; ctx->stats.bvh_prims = ctx->n_prims;
        ctx->stats.bvh_depth = bvh.depth4 + [&] { int d = 0; for (const MeshTree *mt : mesh_trees) d = std::max(d, mt->bvh.depth4); return d; }();
        ctx->stats.bvh_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (std::getenv("PTRACE_BUILD_TIMING"))
            std::fprintf(stderr, "scene build (copies made on the device): %zu primitives, %zu copies of %zu meshes, %zu nodes; prepare %.1f ms, trees %.1f ms, device %.1f ms\n", nIds, nI, groups.size(), total_nodes,
                         std::chrono::duration<float, std::milli>(t_prep - t0).count(), std::chrono::duration<float, std::milli>(t_built - t_prep).count(),
                         std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_built).count());
        ctx->scene_dirty = false;
        ctx->bvh2_stale = false;
        return PT_OK;
    }
    ctx->expanded = false;
    free_expanded(ctx);
    // ---- two-level scenes: the mesh trees follow the top-level tree in the node and record arrays; instance table
    ctx->n_inst = (int)nI;
    size_t real_nodes4 = bvh.nodes4.size();                // without the room left after the top level of a two-level scene
    free_dev(ctx->d_inst); free_dev(ctx->d_objects);
    if (!keep_mesh_trees) free_dev(ctx->d_indices);
    if (nI) {
        std::vector<uint32_t> group_root(groups.size());
        size_t rec_base = rec_cap, node_next = node_cap;
        const size_t top_nodes = bvh.nodes4.size();
        if (!keep_mesh_trees) bvh.nodes4.resize(node_cap, Bvh4Node{});        // the room between the top level and the mesh trees
        for (size_t gi = 0; gi < groups.size(); gi++) {
            const MeshTree &mt = *mesh_trees[gi];
            const uint32_t node_base = (uint32_t)node_next;
            group_root[gi] = node_base;
            node_next += mt.bvh.nodes4.size();
            bvh.depth4 = std::max(bvh.depth4, mt.bvh.depth4);
            if (keep_mesh_trees) { rec_base += mt.recs.size() / 3; continue; }
            for (Bvh4Node nd : mt.bvh.nodes4) {
                for (int k = 0; k < 4; k++) {
                    if (nd.child[k] >= 0) nd.child[k] += (int32_t)node_base;
                    else { const uint32_t enc = (uint32_t)~nd.child[k]; nd.child[k] = ~(int32_t)((((enc >> 3) + (uint32_t)rec_base) << 3) | (enc & 7u)); }
                }
                bvh.nodes4.push_back(nd);
            }
            std::copy(mt.recs.begin(), mt.recs.end(), recs.begin() + rec_base * 3);
            rec_base += mt.recs.size() / 3;
        }
        real_nodes4 = top_nodes + (node_next - node_cap);
        std::vector<float4> inst(nI * 8, make_float4(0, 0, 0, 0));
        for (size_t ii = 0; ii < nI; ii++) {
            const int o = inst_object[ii];
            const float *m = ctx->objects[(size_t)o].local_to_world;
            double inv[3][4];
            affine_inverse_rows(m, inv);
            for (int r = 0; r < 3; r++) {
                inst[ii * 8 + r] = make_float4(m[r], m[4 + r], m[8 + r], m[12 + r]);
                inst[ii * 8 + 3 + r] = make_float4((float)inv[r][0], (float)inv[r][1], (float)inv[r][2], (float)inv[r][3]);
            }
            const int32_t root = (int32_t)group_root[(size_t)obj_group[(size_t)o]], id0 = 1 + nS + (int32_t)obj_first_tri[(size_t)o], oi = o;
            float rf, idf, of; std::memcpy(&rf, &root, 4); std::memcpy(&idf, &id0, 4); std::memcpy(&of, &oi, 4);
            inst[ii * 8 + 6] = make_float4(rf, idf, of, 0.0f);
        }
        std::vector<int4> otab(ctx->objects.size() + 1);
        for (size_t o = 0; o < ctx->objects.size(); o++) otab[o] = make_int4((int)obj_first_tri[o], ctx->objects[o].indices_offset, 0, 0);
        otab[ctx->objects.size()] = make_int4((int)nt, 0, 0, 0);
        if ((rc = upload(ctx, ctx->d_inst, inst.data(), inst.size() * sizeof(float4)))) return rc;
        if ((rc = upload(ctx, ctx->d_objects, otab.data(), otab.size() * sizeof(int4)))) return rc;
        std::vector<float> mats(ctx->objects.size() * 16);
        for (size_t o = 0; o < ctx->objects.size(); o++) std::memcpy(&mats[o * 16], ctx->objects[o].local_to_world, 64);
        if (!keep_mesh_trees || !ctx->d_indices || !ctx->d_vertices) {        // geometry: unchanged while the mesh trees are
            if ((rc = upload(ctx, ctx->d_indices, ctx->indices.data(), ctx->indices.size() * sizeof(int32_t)))) return rc;
            if ((rc = upload(ctx, ctx->d_vertices, ctx->vertices.data(), ctx->vertices.size() * sizeof(float)))) return rc;
        }
        if ((rc = upload(ctx, ctx->d_matrices, mats.data(), mats.size() * sizeof(float)))) return rc;
    }
    ctx->inst_sig = nI ? sig : std::vector<uint64_t>();
    ctx->inst_node_cap = nI ? node_cap : 0; ctx->inst_rec_cap = nI ? rec_cap : 0;
    // ---- what a later refit needs: per slot its source triangle, the object-space vertices, the tree's levels
    ctx->refit_ready = false;
    if (ctx->refit && (!bvh.nodes4.empty() || walk8) && nI == 0) {
        std::vector<int32_t> tri_obj(tris.size()), tri_i0(tris.size()), tri_i1(tris.size()), tri_i2(tris.size());
        size_t j = 0;
        for (size_t o = 0; o < ctx->objects.size(); o++) {
            const PtMeshObject &mo = ctx->objects[o];
            const int32_t end = mo.indices_offset + mo.indices_count;
            for (int32_t i = mo.indices_offset; i + 2 < end; i += 3, j++) {
                tri_obj[j] = (int32_t)o; tri_i0[j] = ctx->indices[i]; tri_i1[j] = ctx->indices[i + 1]; tri_i2[j] = ctx->indices[i + 2];
            }
        }
        std::vector<int4> slot_src(nR);
        std::vector<float> slot_clip(clip.empty() ? 0 : nR * 6);
        for (size_t slot = 0; slot < nR; slot++) {
            uint32_t prim = ref_prim[order[slot]];
            if ((int)prim < nS) slot_src[slot] = make_int4(-1, 0, 0, 0);
            else { size_t t = prim - nS; slot_src[slot] = make_int4(tri_obj[t], tri_i0[t], tri_i1[t], tri_i2[t]); }
            if (!clip.empty()) { std::memcpy(&slot_clip[slot * 6], clip[order[slot]].lo, 12); std::memcpy(&slot_clip[slot * 6 + 3], clip[order[slot]].hi, 12); }
        }
        // nodes are stored parents first, so one forward pass gives every node its level
        const size_t nN = walk8 ? bvh.nodes8.size() : bvh.nodes4.size();
        std::vector<int> level(nN, 0);
        int max_level = 0;
        for (size_t n = 0; n < nN; n++) {
            if (walk8) {
                const Bvh8Node &nd = bvh.nodes8[n];
                for (int k = 0, kids = __builtin_popcount(nd.imask); k < kids; k++) {
                    level[nd.child_base + (uint32_t)k] = level[n] + 1;
                    max_level = std::max(max_level, level[n] + 1);
                }
                continue;
            }
            for (int k = 0; k < 4; k++)
                if (((bvh.nodes4[n].valid >> k) & 1) && bvh.nodes4[n].child[k] >= 0) {
                    level[bvh.nodes4[n].child[k]] = level[n] + 1;
                    max_level = std::max(max_level, level[n] + 1);
                }
        }
        std::vector<int> count(max_level + 2, 0);
        for (size_t n = 0; n < nN; n++) count[max_level - level[n] + 1]++;       // deepest level first
        for (int l = 0; l <= max_level; l++) count[l + 1] += count[l];
        ctx->level_first.assign(count.begin(), count.end());
        std::vector<int32_t> level_nodes(nN);
        std::vector<int> cursor(count.begin(), count.end() - 1);
        for (size_t n = 0; n < nN; n++) level_nodes[cursor[max_level - level[n]]++] = (int32_t)n;
        std::vector<float> mats(ctx->objects.size() * 16);
        for (size_t o = 0; o < ctx->objects.size(); o++) std::memcpy(&mats[o * 16], ctx->objects[o].local_to_world, 64);
        if ((rc = upload(ctx, ctx->d_slot_src, slot_src.data(), slot_src.size() * sizeof(int4)))) return rc;
        if ((rc = upload(ctx, ctx->d_vertices, ctx->vertices.data(), ctx->vertices.size() * sizeof(float)))) return rc;
        if ((rc = upload(ctx, ctx->d_matrices, mats.data(), mats.size() * sizeof(float)))) return rc;
        if ((rc = upload(ctx, ctx->d_slot_clip, slot_clip.data(), slot_clip.size() * sizeof(float)))) return rc;
        if ((rc = upload(ctx, ctx->d_build_matrices, slot_clip.empty() ? nullptr : mats.data(), slot_clip.empty() ? 0 : mats.size() * sizeof(float)))) return rc;
        if ((rc = upload(ctx, ctx->d_level_nodes, level_nodes.data(), level_nodes.size() * sizeof(int32_t)))) return rc;
        free_dev(ctx->d_prim_box); free_dev(ctx->d_node_box);
        PT_HIP(ctx, dev_malloc((void **)&ctx->d_prim_box, nR * 6 * sizeof(float)));
        PT_HIP(ctx, dev_malloc((void **)&ctx->d_node_box, nN * 6 * sizeof(float)));
        ctx->box_pad = pad; ctx->sphere_slack2 = sphere_slack2;
        ctx->refit_ready = !ctx->objects.empty();
    }
    if (keep_mesh_trees) {
        // the mesh trees and their records stay where they are: only the top level goes to the device
        if ((rc = h2d(ctx, ctx->d_nodes4, bvh.nodes4.data(), bvh.nodes4.size() * sizeof(Bvh4Node)))) return rc;
        if ((rc = h2d(ctx, ctx->d_prims, recs.data(), recs.size() * sizeof(float4)))) return rc;
    }
    if (!nI && (rc = upload(ctx, ctx->d_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode)))) return rc;   // a two-level scene is never walked as a two-child tree
    if ((rc = upload(ctx, ctx->d_nodes8, bvh.nodes8.data(), bvh.nodes8.size() * sizeof(Bvh8Node)))) return rc;
    ctx->walk8 = walk8; ctx->n_nodes8 = (int)bvh.nodes8.size(); ctx->stack_need8 = bvh.stack_need8;
    if (nI) free_dev(ctx->d_nodes);
#ifdef PT_NODES_SOA
    {
        const size_t n4 = bvh.nodes4.size();
        std::vector<uint4> cols(n4 * 4);
        for (size_t n = 0; n < n4; n++)
            for (int c = 0; c < 4; c++) std::memcpy(&cols[(size_t)c * n4 + n], reinterpret_cast<const char *>(&bvh.nodes4[n]) + 16 * c, 16);
        if ((rc = upload(ctx, ctx->d_nodes4, cols.data(), cols.size() * sizeof(uint4)))) return rc;
        ctx->refit_ready = false;          // the refit kernels write whole records
    }
#elif defined(PT_NODES_3LOAD)
    {
        const size_t n4 = bvh.nodes4.size();
        std::vector<uint4> packed(n4 * 3);
        for (size_t n = 0; n < n4; n++) {
            const Bvh4Node &nd = bvh.nodes4[n];
            uint32_t w[12];
            std::memcpy(&w[0], nd.org, 12);
            w[3] = (uint32_t)nd.ex | ((uint32_t)nd.ey << 8) | ((uint32_t)nd.ez << 16) | ((uint32_t)c4.kinds[n] << 24);
            w[4] = c4.first_child[n]; w[5] = c4.first_record[n];
            std::memcpy(&w[6], nd.qlo, 12); std::memcpy(&w[9], nd.qhi, 12);
            std::memcpy(&packed[n * 3], w, 48);
        }
        if ((rc = upload(ctx, ctx->d_nodes4, packed.data(), packed.size() * sizeof(uint4)))) return rc;
        ctx->refit_ready = false;          // the refit kernels write 64-byte records
    }
#else
    if (!keep_mesh_trees && (rc = upload(ctx, ctx->d_nodes4, bvh.nodes4.data(), bvh.nodes4.size() * sizeof(Bvh4Node)))) return rc;
#endif
    ctx->n_nodes4 = (int)real_nodes4;
    if (!keep_mesh_trees && (rc = upload(ctx, ctx->d_prims, recs.data(), recs.size() * sizeof(float4)))) return rc;
    if ((rc = upload(ctx, ctx->d_sphere_mat, smat.data(), smat.size() * sizeof(float4)))) return rc;
    ctx->n_nodes = (int)bvh.nodes.size();
    ctx->n_prims = (int)nIds;
    ctx->n_slots = (int)nR;
    ctx->n_top_records = (int)nRefs;
    ctx->n_spheres = nS;
    ctx->stack_need4 = bvh.stack_need4;
    static_assert(kMaxDepth + 1 <= kStackLds + kStackOverflow, "the two-child tree must always fit the traversal stack");
    ctx->stats.bvh_nodes = walk8 ? ctx->n_nodes8 : tree_width(ctx) == 4 ? ctx->n_nodes4 : ctx->n_nodes;
    ctx->stats.bvh_prims = ctx->n_prims;
    ctx->stats.bvh_depth = walk8 ? bvh.depth8 : tree_width(ctx) == 4 ? bvh.depth4 : bvh.depth;
    ctx->stats.bvh_build_ms =
        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (std::getenv("PTRACE_BUILD_TIMING"))
        std::fprintf(stderr, "scene build: %zu primitives, %zu references, %zu instances, %zu nodes%s; prepare %.1f ms, tree %.1f ms, pack + upload %.1f ms\n", nIds, nR, nI, real_nodes4, keep_mesh_trees ? " (mesh trees kept on the device)" : "",
                     std::chrono::duration<float, std::milli>(t_prep - t0).count(), std::chrono::duration<float, std::milli>(t_built - t_prep).count(),
                     std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_built).count());
    ctx->scene_dirty = false;
    ctx->bvh2_stale = false;
    return PT_OK;
}

// How far from a ray origin (`n` points, `stride` floats apart) the far side of the scene lies: what the sphere boxes must
// have been inflated for (pt_context::sphere_reach).
float reach_needed(const pt_context *ctx, const float *origins, int n, int stride)
{
    float c[3], half = 0.0f, need = 0.0f;
    for (int k = 0; k < 3; k++) { c[k] = 0.5f * (ctx->scene_lo[k] + ctx->scene_hi[k]); const float e = ctx->scene_hi[k] - ctx->scene_lo[k]; half += e * e; }
    half = 0.5f * std::sqrt(half);
    for (int i = 0; i < n; i++) {
        const float *o = origins + (size_t)i * stride;
        const float d = std::sqrt((o[0] - c[0]) * (o[0] - c[0]) + (o[1] - c[1]) * (o[1] - c[1]) + (o[2] - c[2]) * (o[2] - c[2])) + half;
        if (std::isfinite(d)) need = std::max(need, d);
    }
    return std::max(need, 2.0f * half);          // rays that start inside the (possibly refitted) scene
}

// The scene as the next launch needs it: rebuilt if something changed, and rebuilt once more with a larger reach if a ray
// origin (a camera; `n` points, `stride` floats apart) lies farther from the scene than the sphere boxes were inflated for.
int rebuild_scene(pt_context *ctx);
int ensure_scene(pt_context *ctx, const float *origins, int n, int stride)
{
    if (ctx->cost_pending && hipEventQuery(ctx->ev_cost) == hipSuccess) {
        // what the last refit left of the tree (option rebuild_percent): read when it has arrived, never waited for
        ctx->cost_pending = false;
        double cost = 0.0;
        for (size_t b = 0; b < ctx->cost_blocks; b++) cost += ctx->h_cost[b];
        if (ctx->device_built && cost > ctx->built_cost * (1.0 + 0.01 * ctx->rebuild_percent)) ctx->scene_dirty = true;
    }
    if (tree_width(ctx) == 2 && (ctx->bvh2_stale || ctx->n_inst)) ctx->scene_dirty = true;   // the two-child tree is flat and never refitted
    if ((ctx->expanded || ctx->device_built) && ctx->bvh_width != 4) ctx->scene_dirty = true;  // another width: a host build of the flattened scene
    for (int pass = 0; pass < 2; pass++) {
        if (ctx->scene_dirty) {
            PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            int rc = rebuild_scene(ctx);
            if (rc) return rc;
        }
        // every box is padded for ray origins within the reach: the spheres' by the slack of IntersectSphere's difference of
        // squares, the triangles' by their own fuzz (triangle_fuzz: a sliver seen from 2 10^4 units away reports hits units
        // beyond its end), the instanced meshes' by the rounding of the ray's transform
        if (ctx->n_prims == 0) break;
        const float need = reach_needed(ctx, origins, n, stride);
        if (!(need > ctx->sphere_reach)) break;
        ctx->sphere_reach_request = 2.0f * need;
        ctx->scene_dirty = true;
    }
    return PT_OK;
}

SceneDev scene_dev(const pt_context *ctx)
{
    SceneDev sc;
    sc.nodes = ctx->d_nodes; sc.nodes4 = ctx->d_nodes4; sc.nodes8 = ctx->d_nodes8; sc.prims = ctx->d_prims; sc.sphere_mat = ctx->d_sphere_mat; sc.sky = ctx->d_sky;
    sc.mesh_mat = ctx->d_mesh_mat;
    sc.sky_w = ctx->sky_w; sc.sky_h = ctx->sky_h;
    sc.n_nodes = ctx->walk8 ? ctx->n_nodes8 : ctx->n_nodes; sc.n_prims = ctx->n_prims; sc.n_spheres = ctx->n_spheres; sc.n_nodes4 = ctx->n_nodes4;
    sc.inst = ctx->d_inst; sc.objects = ctx->d_objects; sc.indices = ctx->d_indices; sc.vertices = ctx->d_vertices; sc.matrices = ctx->d_matrices;
    sc.n_inst = ctx->n_inst; sc.n_top_records = ctx->n_top_records;
    float half = 0.0f;
    for (int k = 0; k < 3; k++) { sc.centre[k] = 0.5f * (ctx->scene_lo[k] + ctx->scene_hi[k]); const float e = ctx->scene_hi[k] - ctx->scene_lo[k]; half += e * e; }
    sc.half_diag = 0.5f * std::sqrt(half);
    sc.far_reach = ctx->sphere_reach;
    for (int k = 0; k < 3; k++) { sc.spheres_lo[k] = ctx->spheres_lo[k]; sc.spheres_hi[k] = ctx->spheres_hi[k]; }
    return sc;
}

int validate_meshes(pt_context *ctx, const PtMeshObject *objects, int n_objects, int n_vertices,
                    const int32_t *indices, int n_indices)
{
    for (int o = 0; o < n_objects; o++) {
        const PtMeshObject &mo = objects[o];
        if (mo.indices_offset < 0 || mo.indices_count < 0 ||
            (int64_t)mo.indices_offset + mo.indices_count > n_indices)
            return fail(ctx, PT_ERR_INVALID, "mesh object " + std::to_string(o) + ": index range outside _Indices");
    }
    std::atomic<int64_t> first_bad(std::numeric_limits<int64_t>::max());         // (thirty million indices take their time on one thread)
    parallel_pieces((size_t)n_indices, (size_t)1 << 18, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++)
            if ((uint32_t)indices[i] >= (uint32_t)n_vertices) {
                int64_t seen = first_bad.load();
                while ((int64_t)i < seen && !first_bad.compare_exchange_weak(seen, (int64_t)i)) {}
                return;
            }
    });
    if (first_bad.load() != std::numeric_limits<int64_t>::max())
        return fail(ctx, PT_ERR_INVALID, "index " + std::to_string(first_bad.load()) + " outside _Vertices");
    return PT_OK;
}

// dst = src[0 .. n), the copy shared out over the host's threads when it is large.
template <class T> void assign_parallel(std::vector<T> &dst, const T *src, size_t n)
{
    dst.resize(n);
    parallel_pieces(n, ((size_t)4 << 20) / sizeof(T), [&](size_t b, size_t e) { if (e > b) std::memcpy(dst.data() + b, src + b, (e - b) * sizeof(T)); });
}

// The traversal stack's overflow area for a launch of `threads` threads with kStackOverflow entries each.
int ensure_stack_overflow(pt_context *ctx, size_t threads)
{
    if (threads <= ctx->stack_ovf_threads) return PT_OK;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_dev(ctx->d_stack_ovf);
    ctx->stack_ovf_threads = 0;
    PT_HIP(ctx, dev_malloc((void **)&ctx->d_stack_ovf, threads * kStackOverflow * sizeof(uint32_t)));
    ctx->stack_ovf_threads = threads;
    return PT_OK;
}

int ensure_wavefront_state(pt_context *ctx, size_t slots, size_t iters)
{
    if (slots > ctx->wf_slots) {
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        free_dev(ctx->wf_state); free_dev(ctx->wf_queues);
        ctx->wf_slots = 0;
        PT_HIP(ctx, dev_malloc((void **)&ctx->wf_state, slots * 5 * sizeof(float4)));
        PT_HIP(ctx, dev_malloc((void **)&ctx->wf_queues, slots * 2 * sizeof(uint32_t)));
        ctx->wf_slots = slots;
    }
    if (iters > ctx->wf_iters) {
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        free_dev(ctx->wf_counters);
        ctx->wf_iters = 0;
        PT_HIP(ctx, dev_malloc((void **)&ctx->wf_counters, (iters + 2) * 2 * sizeof(uint32_t)));
        ctx->wf_iters = iters;
    }
    return PT_OK;
}

// Wavefront backend: generate, then (trace, shade) per iteration.  A pixel traces at most
// spp * max_bounces rays, one per iteration, which bounds the iteration count; iterations whose
// queue is empty return at once.
int render_wavefront(pt_context *ctx, RenderArgs &a)
{
    const size_t iters = (size_t)a.spp * (size_t)a.maxb;
    int rc;
    if ((rc = ensure_wavefront_state(ctx, a.total_items, iters))) return rc;
    const int trace_grid = ctx->num_cus * ctx->trace_blocks_per_cu;
    const int64_t item_blocks = ((int64_t)a.total_items + kBlock - 1) / kBlock;
    const int shade_grid = (int)std::min<int64_t>(item_blocks, (int64_t)ctx->num_cus * ctx->shade_blocks_per_cu);
    if ((rc = ensure_stack_overflow(ctx, (size_t)trace_grid * kBlock))) return rc;
    a.stack_overflow = ctx->d_stack_ovf;
    const size_t S = ctx->wf_slots;
    a.ray_o = ctx->wf_state; a.ray_d = ctx->wf_state + S; a.hit = ctx->wf_state + 2 * S;
    a.path_e = ctx->wf_state + 3 * S; a.path_r = ctx->wf_state + 4 * S;
    uint32_t *qa = ctx->wf_queues, *qb = ctx->wf_queues + S;
    a.counts = ctx->wf_counters; a.cursors = ctx->wf_counters + (ctx->wf_iters + 2);
    PT_HIP(ctx, hipMemsetAsync(ctx->wf_counters, 0, (ctx->wf_iters + 2) * 2 * sizeof(uint32_t), ctx->stream));
    a.iter = 0; a.q_in = nullptr; a.q_out = qa;
    a.refill_threshold = ctx->refill_threshold;
    a.node_vote = ctx->node_vote;
    hipLaunchKernelGGL(wf_generate, dim3((unsigned)item_blocks), dim3(kBlock), 0, ctx->stream, a);
    for (size_t it = 0; it < iters; it++) {
        a.iter = (int32_t)it;
        a.q_in = (it & 1) ? qb : qa;
        a.q_out = (it & 1) ? qa : qb;
        if (tree_width(ctx) == 4) hipLaunchKernelGGL(wf_trace<4>, dim3(trace_grid), dim3(kBlock), 0, ctx->stream, a);
        else hipLaunchKernelGGL(wf_trace<2>, dim3(trace_grid), dim3(kBlock), 0, ctx->stream, a);
        hipLaunchKernelGGL(wf_shade, dim3(shade_grid), dim3(kBlock), 0, ctx->stream, a);
    }
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// Close the statistics epoch: wait for the last render, read its counters.
int finish_epoch(pt_context *ctx)
{
    if (!ctx->epoch_open) return PT_OK;
    PT_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.0f;
    PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    unsigned long long part[kStatsShards * kStatsStride], h[4] = {0, 0, 0, 0};
    PT_HIP(ctx, hipMemcpy(part, ctx->d_stats, sizeof part, hipMemcpyDeviceToHost));
    for (int sh = 0; sh < kStatsShards; sh++)
        for (int k = 0; k < 4; k++) h[k] += part[sh * kStatsStride + k];
#ifdef PT_PROFILE_PHASES
    {
        unsigned long long ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int sh = 0; sh < kStatsShards; sh++)
            for (int k = 0; k < 9; k++) ph[k] += part[sh * kStatsStride + 4 + k];
        unsigned long long life_sum = 0, life_max = 0;
        for (int sh = 0; sh < kStatsShards; sh++) { life_sum += part[sh * kStatsStride + 13]; life_max = std::max(life_max, part[sh * kStatsStride + 14]); }
        std::fprintf(stderr, "waves: summed lifetime %.3f ms x waves, longest %.3f ms (100 MHz counter); kernels of the epoch %.3f ms\n", (double)life_sum * 1e-5, (double)life_max * 1e-5, ms);
        const char *names[3] = {"shade", "node", "leaf"};
        for (int k = 0; k < 3; k++)
            std::fprintf(stderr, "phase %-5s rounds %12llu  lanes/round %6.2f  clocks %14llu (%5.1f%%)  clocks/round %8.1f\n", names[k], ph[2 * k],
                         ph[2 * k] ? (double)ph[2 * k + 1] / (double)ph[2 * k] : 0.0, ph[6 + k], 100.0 * (double)ph[6 + k] / (double)(ph[6] + ph[7] + ph[8] + 1),
                         ph[2 * k] ? (double)ph[6 + k] / (double)ph[2 * k] : 0.0);
    }
#endif
    ctx->stats.kernel_ms = ms;
    ctx->stats.rays = h[0]; ctx->stats.nodes_visited = h[1]; ctx->stats.prims_tested = h[2]; ctx->stats.paths = h[3];
    ctx->epoch_open = false;
    return PT_OK;
}

}  // namespace

// No C++ exception leaves the library: a host allocation that fails (or a thread that cannot be started) inside an entry point
// becomes an error code and a message, as every other failure does.
void note_error(pt_context *ctx, const char *msg) noexcept
{
    try { if (ctx) ctx->err = msg; else g_create_error = msg; } catch (...) {}
}
template <class F> int guarded(pt_context *ctx, F body) noexcept
{
    try { return body(); }
    catch (const std::bad_alloc &) { note_error(ctx, "out of host memory"); return PT_ERR_NOMEM; }
    catch (const std::exception &e) { note_error(ctx, e.what()); return PT_ERR_HIP; }
    catch (...) { note_error(ctx, "unknown exception"); return PT_ERR_HIP; }
}

extern "C" {

const char *pt_version(void) { return "libptrace 0.1 (gfx950)"; }

const char *pt_last_error(const pt_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

unsigned long long pt_allocation_calls(void) { return g_allocation_calls.load(); }

int pt_create(int device_ordinal, pt_context **out_ctx)
{
    return guarded(nullptr, [&]() -> int {
    if (!out_ctx) return fail(nullptr, PT_ERR_INVALID, "out_ctx is NULL");
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, PT_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(nullptr, PT_ERR_INVALID, "device ordinal out of range");
    pt_context *ctx = new (std::nothrow) pt_context();
    if (!ctx) return fail(nullptr, PT_ERR_NOMEM, "out of host memory");
    ctx->device = device_ordinal;
    DeviceGuard g(device_ordinal);
    auto bail = [&](const char *what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        pt_destroy(ctx);
        return PT_ERR_HIP;
    };
    if (!g.ok) return bail("hipSetDevice", hipErrorInvalidDevice);
    if ((e = resolve_scratch(device_ordinal)) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    ctx->stream = ctx->own_stream;
    if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_refit, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_cost, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = dev_malloc((void **)&ctx->d_work, kWorkShards * kWorkStride * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = dev_malloc((void **)&ctx->d_stats, kStatsShards * kStatsStride * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_ordinal)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    ctx->num_cus = prop.multiProcessorCount;
    int bpc = 0;
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, wf_trace<4>, kBlock, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->trace_blocks_per_cu = std::max(1, bpc);
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, wf_shade, kBlock, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->shade_blocks_per_cu = std::max(1, bpc);
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, path_sched_kernel<4>, kBlock, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->sched_blocks_per_cu = std::max(1, bpc);
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, path_sched_mat_kernel<4>, kBlock, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->sched_ext_blocks_per_cu = std::max(1, bpc);
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, path_sched_inst_kernel, kBlock, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->sched_inst_blocks_per_cu = std::max(1, bpc);
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, path_sched_kernel<8>, kBlock, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->sched8_blocks_per_cu = std::max(1, bpc);
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, path_sched_mat_kernel<8>, kBlock, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->sched8_ext_blocks_per_cu = std::max(1, bpc);
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, path_sched_fast_kernel<4>, kBlock, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->sched_fast_blocks_per_cu = std::max(1, bpc);
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, path_sched_fast_kernel<8>, kBlock, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->sched8_fast_blocks_per_cu = std::max(1, bpc);
    if (std::getenv("PTRACE_BUILD_TIMING"))
        std::fprintf(stderr, "resident blocks per CU: path_sched_kernel %d, path_sched_mat_kernel %d, path_sched_inst_kernel %d, eight-wide %d and %d\n", ctx->sched_blocks_per_cu, ctx->sched_ext_blocks_per_cu, ctx->sched_inst_blocks_per_cu, ctx->sched8_blocks_per_cu, ctx->sched8_ext_blocks_per_cu);
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, path_pool_kernel<4>, 64, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->pool_blocks_per_cu = std::max(1, bpc);
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, path_pool_mat_kernel<4>, 64, 0)) != hipSuccess) return bail("occupancy query", e);
    ctx->pool_ext_blocks_per_cu = std::max(1, bpc);
    if (const char *env = std::getenv("PTRACE_BACKEND")) {
        if (std::strcmp(env, "sched") == 0 || std::strcmp(env, "mega") == 0) ctx->backend = 0;
        else if (std::strcmp(env, "wavefront") == 0 || std::strcmp(env, "wave") == 0) ctx->backend = 1;
        else if (std::strcmp(env, "pool") == 0) ctx->backend = 2;
        else std::fprintf(stderr, "libptrace: PTRACE_BACKEND=%s ignored (sched | wavefront | pool)\n", env);
    }
    if (const char *env = std::getenv("PTRACE_BLOCKS_PER_CU")) ctx->blocks_per_cu = std::max(0, std::min(16, std::atoi(env)));
    if (const char *env = std::getenv("PTRACE_BVH_WIDTH")) ctx->bvh_width = std::atoi(env) == 2 ? 2 : std::atoi(env) == 8 ? 8 : 4;
    if (const char *env = std::getenv("PTRACE_BLOCK_ORDER")) ctx->block_order = std::atoi(env) != 0;
    if (const char *env = std::getenv("PTRACE_FAST_INTRINSICS")) ctx->fast_intrinsics = std::atoi(env) != 0;
    if (const char *env = std::getenv("PTRACE_TREE_BUILD")) ctx->tree_build = std::max(0, std::min(3, std::atoi(env)));
    if (const char *env = std::getenv("PTRACE_SPLIT")) ctx->split_budget = std::max(0.0f, std::min(4.0f, (float)std::atof(env)));
    if (const char *env = std::getenv("PTRACE_STACK_LIMIT")) ctx->stack_limit = std::max(1, std::min(kStackLds + kStackOverflow, std::atoi(env)));
    if (const char *env = std::getenv("PTRACE_VOTE_SHADE")) ctx->vote_shade = ctx->vote_shade_few = std::max(1, std::min(64, std::atoi(env)));
    if (const char *env = std::getenv("PTRACE_VOTE_SHADE_FEW")) ctx->vote_shade_few = std::max(1, std::min(64, std::atoi(env)));
    if (const char *env = std::getenv("PTRACE_VOTE_NODE")) ctx->vote_node = std::max(1, std::min(64, std::atoi(env)));
    if (const char *env = std::getenv("PTRACE_POOL_VOTES")) {     // shade,node,leaf,cont
        int v[4] = {ctx->pool_vote_shade, ctx->pool_vote_node, ctx->pool_vote_leaf, ctx->pool_cont};
        if (std::sscanf(env, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) >= 1) {
            for (int &x : v) x = std::max(1, std::min(64, x));
            ctx->pool_vote_shade = v[0]; ctx->pool_vote_node = v[1]; ctx->pool_vote_leaf = v[2]; ctx->pool_cont = v[3];
        }
    }
    if (const char *env = std::getenv("PTRACE_VOTE")) ctx->node_vote = std::max(1, std::min(64, std::atoi(env)));
    if (const char *env = std::getenv("PTRACE_REFILL")) ctx->refill_threshold = std::max(1, std::min(64, std::atoi(env)));
    *out_ctx = ctx;
    return PT_OK;
    });
}

void pt_destroy(pt_context *ctx)
{
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    // renders may still be running on the caller's stream (pt_set_stream): they use the buffers freed below
    if (ctx->stream != ctx->own_stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    free_dev(ctx->d_nodes); free_dev(ctx->d_nodes4); free_dev(ctx->d_nodes8); free_dev(ctx->d_prims); free_dev(ctx->d_sphere_mat); free_dev(ctx->d_sky);
    free_dev(ctx->own_result); free_dev(ctx->own_conv); free_dev(ctx->own_ids); free_dev(ctx->own_result2);
    free_dev(ctx->d_filtered); free_dev(ctx->d_seeds); free_dev(ctx->own_display);
    free_dev(ctx->d_work); free_dev(ctx->d_stats); free_dev(ctx->d_stack_ovf);
    free_dev(ctx->d_slot_clip); free_dev(ctx->d_build_matrices);
    free_dev(ctx->d_inst); free_dev(ctx->d_objects); free_dev(ctx->d_indices);
    free_expanded(ctx);
    if (ctx->h_matrices) { (void)host_free(ctx->h_matrices); ctx->h_matrices = nullptr; }
    free_dev(ctx->d_slot_src); free_dev(ctx->d_vertices); free_dev(ctx->d_matrices); free_dev(ctx->d_prim_box);
    free_dev(ctx->d_node_box); free_dev(ctx->d_level_nodes); free_dev(ctx->d_mesh_mat);
    for (int i = 0; i < pt_context::kCamRing; i++) {
        free_dev(ctx->d_cams[i]);
        if (ctx->h_cams[i]) (void)host_free(ctx->h_cams[i]);
        if (ctx->ev_cams[i]) (void)hipEventDestroy(ctx->ev_cams[i]);
    }
    if (ctx->ev_refit) (void)hipEventDestroy(ctx->ev_refit);
    free_dev(ctx->build_scratch[0]); free_dev(ctx->build_scratch[1]);
    if (ctx->ev_cost) (void)hipEventDestroy(ctx->ev_cost);
    free_dev(ctx->d_cost);
    if (ctx->h_cost) (void)host_free(ctx->h_cost);
    free_dev(ctx->wf_state); free_dev(ctx->wf_queues); free_dev(ctx->wf_counters); free_dev(ctx->d_pool_state);
    free_dev(ctx->d_block_cost); free_dev(ctx->d_block_order);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int pt_set_stream(pt_context *ctx, void *hip_stream)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    DeviceGuard g(ctx->device);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream == PT_STREAM_OWN ? ctx->own_stream : (hipStream_t)hip_stream;
    return PT_OK;
    });
}

int pt_set_spheres(pt_context *ctx, const PtSphere *spheres, int count)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (count < 0 || (count > 0 && !spheres)) return fail(ctx, PT_ERR_INVALID, "pt_set_spheres: bad arguments");
    ctx->spheres.assign(spheres, spheres + count);
    ctx->scene_dirty = true;
    return PT_OK;
    });
}

int pt_set_meshes(pt_context *ctx, const PtMeshObject *objects, int n_objects, const float *vertices,
                  int n_vertices, const int32_t *indices, int n_indices)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (n_objects < 0 || n_vertices < 0 || n_indices < 0 || (n_objects > 0 && !objects) ||
        (n_vertices > 0 && !vertices) || (n_indices > 0 && !indices))
        return fail(ctx, PT_ERR_INVALID, "pt_set_meshes: bad arguments");
    int rc = validate_meshes(ctx, objects, n_objects, n_vertices, indices, n_indices);
    if (rc) return rc;
    ctx->objects.assign(objects, objects + n_objects);
    assign_parallel(ctx->vertices, vertices, (size_t)n_vertices * 3);
    assign_parallel(ctx->indices, indices, (size_t)n_indices);
    ctx->mesh_cache.clear();
    ctx->scene_dirty = true;
    if (!ctx->materials.empty()) {            // materials belong to the object list they were given for
        DeviceGuard g(ctx->device);
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->materials.clear();
        free_dev(ctx->d_mesh_mat);
    }
    return PT_OK;
    });
}

int pt_set_mesh_materials(pt_context *ctx, const PtMaterial *materials, int count)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (count < 0 || (count > 0 && !materials)) return fail(ctx, PT_ERR_INVALID, "pt_set_mesh_materials: bad arguments");
    if (count != 0 && count != (int)ctx->objects.size())
        return fail(ctx, PT_ERR_INVALID, "pt_set_mesh_materials: one material per mesh object of pt_set_meshes (or none)");
    for (int i = 0; i < count; i++)
        if (materials[i].lobe != PT_LOBE_PHONG && materials[i].lobe != PT_LOBE_GGX)
            return fail(ctx, PT_ERR_INVALID, "pt_set_mesh_materials: unknown lobe");
    DeviceGuard g(ctx->device);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->materials.assign(materials, materials + count);
    std::vector<float4> table((size_t)count * 3);
    for (int i = 0; i < count; i++) {
        const PtMaterial &m = materials[i];
        float lobef; std::memcpy(&lobef, &m.lobe, 4);
        table[(size_t)i * 3 + 0] = make_float4(m.albedo[0], m.albedo[1], m.albedo[2], m.smoothness);
        table[(size_t)i * 3 + 1] = make_float4(m.specular[0], m.specular[1], m.specular[2], lobef);
        table[(size_t)i * 3 + 2] = make_float4(m.emission[0], m.emission[1], m.emission[2], 0.0f);
    }
    return upload(ctx, ctx->d_mesh_mat, table.data(), table.size() * sizeof(float4));
    });
}

int pt_update_transforms(pt_context *ctx, const PtMeshObject *objects, int n_objects)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (n_objects != (int)ctx->objects.size() || (n_objects > 0 && !objects))
        return fail(ctx, PT_ERR_INVALID, "pt_update_transforms: object count differs from pt_set_meshes");
    for (int o = 0; o < n_objects; o++)
        if (objects[o].indices_offset != ctx->objects[o].indices_offset ||
            objects[o].indices_count != ctx->objects[o].indices_count)
            return fail(ctx, PT_ERR_INVALID, "pt_update_transforms: index ranges differ from pt_set_meshes");
    ctx->objects.assign(objects, objects + n_objects);
    if (!ctx->refit || !ctx->refit_ready || ctx->scene_dirty || tree_width(ctx) == 2 || n_objects == 0) {
        ctx->scene_dirty = true;      // full build at the next render
        return PT_OK;
    }
    // ---- refit on the device: new matrices in, boxes redone bottom-up, all ordered on the context's stream
    DeviceGuard g(ctx->device);
    auto t0 = std::chrono::steady_clock::now();
    if (ctx->refit_in_flight) { PT_HIP(ctx, hipEventSynchronize(ctx->ev_refit)); ctx->refit_in_flight = false; }   // d_matrices is free again
    // the matrices go through a pinned buffer of the context's, on its stream: nothing waits here for the frame in flight
    if (ctx->h_matrices_objects < (size_t)n_objects) {
        if (ctx->h_matrices) { (void)host_free(ctx->h_matrices); ctx->h_matrices = nullptr; ctx->h_matrices_objects = 0; }
        PT_HIP(ctx, host_malloc((void **)&ctx->h_matrices, (size_t)n_objects * 64, hipHostMallocDefault));
        ctx->h_matrices_objects = (size_t)n_objects;
    }
    for (int o = 0; o < n_objects; o++) std::memcpy(ctx->h_matrices + (size_t)o * 16, objects[o].local_to_world, 64);
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_matrices, ctx->h_matrices, (size_t)n_objects * 64, hipMemcpyHostToDevice, ctx->stream));
    RefitArgs ra;
    ra.slot_src = ctx->d_slot_src; ra.vertices = ctx->d_vertices; ra.matrices = ctx->d_matrices; ra.prims = ctx->d_prims;
    ra.prim_box = ctx->d_prim_box; ra.nodes4 = ctx->d_nodes4; ra.nodes8 = ctx->d_nodes8; ra.node_box = ctx->d_node_box; ra.level_nodes = ctx->d_level_nodes;
    ra.pad = ctx->box_pad; ra.sphere_slack2 = ctx->sphere_slack2; ra.reach = ctx->sphere_reach; ra.n_prims = ctx->n_slots;
    ra.slot_clip = ctx->d_slot_clip; ra.build_matrices = ctx->d_build_matrices;
    if (ctx->expanded) {
        ra.slot_clip = nullptr; ra.build_matrices = nullptr;
        int rc = refit_expanded(ctx, ra);
        if (rc) return rc;
    } else {
    hipLaunchKernelGGL(refit_prims_kernel, dim3((ctx->n_slots + kRefitBlock - 1) / kRefitBlock), dim3(kRefitBlock), 0, ctx->stream, ra);
    for (size_t l = 0; l + 1 < ctx->level_first.size(); l++) {
        const int first = ctx->level_first[l], count = ctx->level_first[l + 1] - first;
        if (count > 0 && ctx->walk8)
            hipLaunchKernelGGL(refit_nodes8_kernel, dim3((count + kRefitBlock - 1) / kRefitBlock), dim3(kRefitBlock), 0, ctx->stream, ra, first, count);
        else if (count > 0)
            hipLaunchKernelGGL(refit_nodes_kernel, dim3((count + kRefitBlock - 1) / kRefitBlock), dim3(kRefitBlock), 0, ctx->stream, ra, first, count);
    }
    }
    PT_HIP(ctx, hipGetLastError());
    if (ctx->device_built && ctx->rebuild_percent > 0 && ctx->cost_blocks == ((size_t)ctx->n_nodes4 + kBuildBlock - 1) / kBuildBlock) {
        hipLaunchKernelGGL(tree_cost_kernel, dim3((unsigned)ctx->cost_blocks), dim3(kBuildBlock), 0, ctx->stream, ctx->d_node_box, ctx->n_nodes4, 0, ctx->d_cost);
        PT_HIP(ctx, hipMemcpyAsync(ctx->h_cost, ctx->d_cost, ctx->cost_blocks * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        PT_HIP(ctx, hipEventRecord(ctx->ev_cost, ctx->stream));
        ctx->cost_pending = true;
    }
    if (std::getenv("PTRACE_BUILD_TIMING") && !ctx->walk8 && ctx->n_nodes4 > 0) {
        // node 0 is the root (of the top-level tree when copies hang below it); every node's box is fresh
        const int nb = (ctx->n_nodes4 + kBuildBlock - 1) / kBuildBlock;
        float *d_partial = nullptr;
        PT_HIP(ctx, dev_malloc((void **)&d_partial, (size_t)nb * sizeof(float)));
        hipLaunchKernelGGL(tree_cost_kernel, dim3((unsigned)nb), dim3(kBuildBlock), 0, ctx->stream, ctx->d_node_box, ctx->n_nodes4, 1, d_partial);
        std::vector<float> partial((size_t)nb);
        PT_HIP(ctx, hipMemcpyAsync(partial.data(), d_partial, (size_t)nb * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)dev_free(d_partial);
        double cost = 0.0;
        for (float v : partial) cost += v;
        std::fprintf(stderr, "tree after the refit: %d four-wide nodes, surface-area cost %.3f (expected node visits of a random ray through the root's box)\n", ctx->n_nodes4, cost);
    }
    PT_HIP(ctx, hipEventRecord(ctx->ev_refit, ctx->stream));
    ctx->refit_in_flight = true;
    ctx->bvh2_stale = !ctx->walk8;
    // scene bounds after the move, from the objects' object-space boxes (an over-estimate is fine: it feeds ensure_scene)
    for (int k = 0; k < 3; k++) { ctx->scene_lo[k] = ctx->spheres_lo[k]; ctx->scene_hi[k] = ctx->spheres_hi[k]; }
    for (int o = 0; o < n_objects && o < (int)ctx->object_box.size(); o++) {
        const Aabb &ob = ctx->object_box[(size_t)o];
        if (!(ob.lo[0] <= ob.hi[0])) continue;
        for (int corner = 0; corner < 8; corner++) {
            const pt_f3 w = pt_mul_point(objects[o].local_to_world, pt_make3((corner & 1) ? ob.hi[0] : ob.lo[0], (corner & 2) ? ob.hi[1] : ob.lo[1],
                                                                              (corner & 4) ? ob.hi[2] : ob.lo[2]));
            const float wc[3] = {w.x, w.y, w.z};
            for (int k = 0; k < 3; k++)
                if (std::fabs(wc[k]) < kHugeCoord) { ctx->scene_lo[k] = std::min(ctx->scene_lo[k], wc[k]); ctx->scene_hi[k] = std::max(ctx->scene_hi[k], wc[k]); }
        }
    }
    for (int k = 0; k < 3; k++)
        if (!(ctx->scene_lo[k] <= ctx->scene_hi[k])) ctx->scene_lo[k] = ctx->scene_hi[k] = 0.0f;
    ctx->stats.bvh_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
    });
}

int pt_set_sky(pt_context *ctx, const float *rgba, int width, int height)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    DeviceGuard g(ctx->device);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!rgba || width <= 0 || height <= 0) {
        free_dev(ctx->d_sky); ctx->sky_w = ctx->sky_h = 0;
        return PT_OK;
    }
    int rc = upload(ctx, ctx->d_sky, rgba, (size_t)width * height * sizeof(float4));
    if (rc) return rc;
    ctx->sky_w = width; ctx->sky_h = height;
    return PT_OK;
    });
}

int pt_resize(pt_context *ctx, int width, int height)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (width <= 0 || height <= 0 || width > 32767 || height > 32767 || (int64_t)width * height > (int64_t)1 << 28)
        return fail(ctx, PT_ERR_INVALID, "pt_resize: bad size (1..32767 per side, at most 2^28 pixels)");
    DeviceGuard g(ctx->device);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const bool rebind_r = ctx->result == ctx->own_result, rebind_c = ctx->conv == ctx->own_conv,
               rebind_i = ctx->ids == ctx->own_ids, rebind_2 = ctx->result2 == ctx->own_result2;
    // Until every new buffer exists the context has no targets: a failure below leaves it in the state "before pt_resize"
    // (W == 0), not pointing at freed memory.  Caller-owned targets stay bound.
    if (rebind_r) ctx->result = nullptr;
    if (rebind_c) ctx->conv = nullptr;
    if (rebind_i) ctx->ids = nullptr;
    if (rebind_2) ctx->result2 = nullptr;
    ctx->W = ctx->H = 0;
    ctx->order_ready = false;
    free_dev(ctx->own_result); free_dev(ctx->own_conv); free_dev(ctx->own_ids); free_dev(ctx->own_result2);
    free_dev(ctx->d_filtered); free_dev(ctx->d_seeds);
    if (ctx->display == ctx->own_display) ctx->display = nullptr;
    free_dev(ctx->own_display);
    ctx->screen_pixels = 0;
    size_t n = (size_t)width * height;
    PT_HIP(ctx, dev_malloc((void **)&ctx->own_result2, n * sizeof(float4)));
    PT_HIP(ctx, dev_malloc((void **)&ctx->own_result, n * sizeof(float4)));
    PT_HIP(ctx, dev_malloc((void **)&ctx->own_conv, n * sizeof(float4)));
    PT_HIP(ctx, dev_malloc((void **)&ctx->own_ids, n * sizeof(int32_t)));
    // Cleared on the context's stream and waited for: hipMemset returns before a device fill has run and fills on the null
    // stream, which a non-blocking stream does not wait for - the next frame's kernels could read the new targets (recycled
    // memory) before the zeros arrived.
    PT_HIP(ctx, hipMemsetAsync(ctx->own_result2, 0, n * sizeof(float4), ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->own_result, 0, n * sizeof(float4), ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->own_conv, 0, n * sizeof(float4), ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->own_ids, 0xff, n * sizeof(int32_t), ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->W = width; ctx->H = height;
    if (rebind_r || !ctx->result) ctx->result = ctx->own_result;
    if (rebind_c || !ctx->conv) ctx->conv = ctx->own_conv;
    if (rebind_i || !ctx->ids) ctx->ids = ctx->own_ids;
    if (rebind_2 || !ctx->result2) ctx->result2 = ctx->own_result2;
    return PT_OK;
    });
}

int pt_bind_target(pt_context *ctx, int which, void *device_ptr)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (ctx->W == 0) return fail(ctx, PT_ERR_INVALID, "pt_bind_target before pt_resize");
    DeviceGuard g(ctx->device);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    switch (which) {
    case PT_TARGET_RESULT: ctx->result = device_ptr ? (float4 *)device_ptr : ctx->own_result; break;
    case PT_TARGET_CONVERGED: ctx->conv = device_ptr ? (float4 *)device_ptr : ctx->own_conv; break;
    case PT_TARGET_HIT_IDS: ctx->ids = device_ptr ? (int32_t *)device_ptr : ctx->own_ids; break;
    case PT_TARGET_DENOISED: ctx->result2 = device_ptr ? (float4 *)device_ptr : ctx->own_result2; break;
    case PT_TARGET_DISPLAY: ctx->display = device_ptr ? (float4 *)device_ptr : ctx->own_display; break;
    default: return fail(ctx, PT_ERR_INVALID, "pt_bind_target: unknown target");
    }
    return PT_OK;
    });
}

// The rectangle's 8x8 pixel blocks: blocks per row, work items, and the multiplier that stands in for the division by the
// blocks per row (pt_device.hpp, numbered_pixel): with m = floor(2^32 / d) + 1, floor(n m / 2^32) = floor(n / d) whenever n d < 2^32.
static void set_tiles(RenderArgs &a, int tw, int th)
{
    a.tiles_x = (tw + 7) / 8;
    const int tiles_y = (th + 7) / 8;
    a.total_items = (uint32_t)a.tiles_x * (uint32_t)tiles_y * 64u;
    // the counters hand out up to one block per counter beyond the rectangle's last: kWorkShards more than it holds
    const uint64_t blocks = (uint64_t)a.tiles_x * (uint64_t)tiles_y + (uint64_t)kWorkShards;
    a.tiles_x_magic = a.tiles_x > 1 && blocks * (uint64_t)a.tiles_x < (1ull << 32) ? (uint32_t)((1ull << 32) / (uint64_t)a.tiles_x + 1ull) : 0u;
}

// Path-trace the rectangle (x0, y0, tw, th) with the arguments in `a` (everything but the tile fields filled in).
static int launch_path(pt_context *ctx, RenderArgs &a, int x0, int y0, int tw, int th)
{
    if (tw <= 0 || th <= 0) return PT_OK;
    int rc;
    a.tx0 = x0; a.ty0 = y0; a.tw = tw; a.th = th;
    set_tiles(a, tw, th);
    const bool instanced = ctx->n_inst > 0;        // two-level scene: path_sched_inst_kernel, whatever the backend option says
    const bool extended = instanced || ctx->d_mesh_mat || a.render_mode == 3 || a.render_mode == 4 || a.render_mode >= 7;
    // the wavefront stages pack a path's bounce count into 8 bits (pt_wavefront.hpp): longer paths go to the scheduled kernel
    const int width = tree_width(ctx);                // 8: walked by the scheduled kernel only, whatever the backend option says
    if (ctx->backend == 1 && width != 8 && !ctx->fast_intrinsics && a.depth == 0 && a.render_mode == 1 && !extended && !a.reuse_primary && a.maxb <= 255) return render_wavefront(ctx, a);
    PT_HIP(ctx, hipMemsetAsync(ctx->d_work, 0, kWorkShards * kWorkStride * sizeof(uint32_t), ctx->stream));
    {
        // Few samples per pixel: pixels are claimed often, and one counter would serialise the claims at the L2.  Many
        // samples: claims are rare, and one counter keeps all waves on neighbouring pixels (cache-friendly).
        const uint32_t blocks = a.total_items / 64u;     // 8x8 pixel blocks
        // Measured at 1080p: 1 spp 0.74 -> 0.42 ms (small scene) and 1.39 -> 1.02 ms (262 k triangles) going from 1 to 16
        // counters, slower again beyond 32.  Eight - a wave starts on counter blockIdx % 8, so the workgroups of one XCD share
        // one - is the best count for one-sample frames (4 and 16: 2 - 5 % slower) and for cheap scenes at 16 spp (the
        // 70 k-triangle scene: 9 900 -> 11 000 Mrays/s against four counters); the 262 k-triangle scene does not care.
        int shards = a.depth ? 16 : 8;
        if (const char *env = std::getenv("PTRACE_SHARDS")) shards = std::atoi(env);
        a.n_shards = std::max(1, std::min(kWorkShards, shards));
        a.shard_items = ((blocks + a.n_shards - 1) / a.n_shards) * 64u;
    }
    // ---- expensive blocks first.  Worth it when a pixel is long (many samples x bounces): then the dispatch ends with
    // every lane inside its last pixel, and it matters which pixels those are.  The costs come from the previous dispatch
    // over the same rectangle; the order kernel after this launch prepares the next one's.
    a.block_order = nullptr; a.block_cost = nullptr; a.n_blocks = a.total_items / 64u;
    const bool want_order = ctx->block_order && a.n_blocks >= 1024 && (int64_t)a.spp * a.maxb >= 32;
    const int key[6] = {ctx->W, ctx->H, x0, y0, tw, th};
    if (want_order) {
        if (a.n_blocks > ctx->order_capacity) {
            PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            free_dev(ctx->d_block_cost); free_dev(ctx->d_block_order); ctx->order_capacity = 0; ctx->order_ready = false;
            PT_HIP(ctx, dev_malloc((void **)&ctx->d_block_cost, a.n_blocks * sizeof(uint32_t)));
            PT_HIP(ctx, dev_malloc((void **)&ctx->d_block_order, a.n_blocks * sizeof(uint32_t)));
            ctx->order_capacity = a.n_blocks;
        }
        if (!ctx->order_ready || std::memcmp(key, ctx->order_key, sizeof key) != 0) {
            ctx->order_ready = false;
            PT_HIP(ctx, hipMemsetAsync(ctx->d_block_cost, 0, a.n_blocks * sizeof(uint32_t), ctx->stream));
        } else {
            a.block_order = ctx->d_block_order;
            a.total_items = (uint32_t)a.n_shards * a.shard_items;     // every counter hands out whole ranks; ranks past the last block are skipped
        }
        a.block_cost = ctx->d_block_cost;
    }
    struct AfterLaunch {                    // queue the order kernel behind whichever path kernel is launched below
        pt_context *ctx; const RenderArgs &a; bool want; const int *key;
        int done() const
        {
            if (!want) return PT_OK;
            hipLaunchKernelGGL(order_blocks_kernel, dim3(1), dim3(kOrderThreads), 0, ctx->stream, ctx->d_block_cost, ctx->d_block_order, a.n_blocks, (uint32_t)a.tiles_x);
            if (hipGetLastError() != hipSuccess) return PT_ERR_HIP;
            std::memcpy(ctx->order_key, key, sizeof ctx->order_key);
            ctx->order_ready = true;
            return PT_OK;
        }
    } after{ctx, a, want_order, key};
    if (ctx->backend == 2 && !instanced && width != 8 && !ctx->fast_intrinsics) {
        // one wave per block, kPoolK * 64 ray slots per wave
        constexpr int P = 64 * kPoolK;
        const int fit = extended ? ctx->pool_ext_blocks_per_cu : ctx->pool_blocks_per_cu;
        int per_cu = fit;
        if (ctx->blocks_per_cu > 0) per_cu = std::max(1, fit * std::min(ctx->blocks_per_cu, ctx->sched_blocks_per_cu) / ctx->sched_blocks_per_cu);
        const int grid = (int)std::min<int64_t>(((int64_t)a.total_items + 63) / 64, (int64_t)ctx->num_cus * per_cu);
        const size_t slots = (size_t)grid * P;
        // overflow entries: what a slot's stack may hold beyond the kPoolSE entries in LDS
        const size_t ovf_entries = slots * (size_t)(kStackLds + kStackOverflow - kPoolSE);
        if ((rc = ensure_stack_overflow(ctx, (ovf_entries + kStackOverflow - 1) / kStackOverflow))) return rc;
        if (slots > ctx->pool_slots) {
            PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            free_dev(ctx->d_pool_state); ctx->pool_slots = 0;
            PT_HIP(ctx, dev_malloc((void **)&ctx->d_pool_state, slots * 3 * sizeof(float4)));
            ctx->pool_slots = slots;
        }
        a.stack_overflow = ctx->d_stack_ovf; a.pool_state = ctx->d_pool_state;
        a.pool_vote_shade = ctx->pool_vote_shade; a.pool_vote_node = ctx->pool_vote_node; a.pool_vote_leaf = ctx->pool_vote_leaf; a.pool_cont = ctx->pool_cont;
        if (extended) {
            if (tree_width(ctx) == 4) hipLaunchKernelGGL(path_pool_mat_kernel<4>, dim3(grid), dim3(64), 0, ctx->stream, a);
            else hipLaunchKernelGGL(path_pool_mat_kernel<2>, dim3(grid), dim3(64), 0, ctx->stream, a);
        } else {
            if (tree_width(ctx) == 4) hipLaunchKernelGGL(path_pool_kernel<4>, dim3(grid), dim3(64), 0, ctx->stream, a);
            else hipLaunchKernelGGL(path_pool_kernel<2>, dim3(grid), dim3(64), 0, ctx->stream, a);
        }
        PT_HIP(ctx, hipGetLastError());
        return after.done();
    }
    int64_t want = ((int64_t)a.total_items + kBlock - 1) / kBlock;
    // the approximate-arithmetic kernel stands in for the default one only: plain path tracing, no materials, no instances
    const bool fast = ctx->fast_intrinsics && !extended && !instanced && a.depth == 0;
    const int fit = instanced ? ctx->sched_inst_blocks_per_cu : fast ? (width == 8 ? ctx->sched8_fast_blocks_per_cu : ctx->sched_fast_blocks_per_cu)
                              : width == 8 ? (extended ? ctx->sched8_ext_blocks_per_cu : ctx->sched8_blocks_per_cu)
                              : extended ? ctx->sched_ext_blocks_per_cu : ctx->sched_blocks_per_cu;   // a persistent grid: exactly what is resident at once
    const int per_cu = ctx->blocks_per_cu > 0 ? std::min(ctx->blocks_per_cu, fit) : fit;
    int grid = (int)std::min<int64_t>(want, (int64_t)ctx->num_cus * per_cu);
    if ((rc = ensure_stack_overflow(ctx, (size_t)grid * kBlock))) return rc;
    a.stack_overflow = ctx->d_stack_ovf;
    // With one or two samples per pixel nearly every shade phase also ends pixels, claims new ones and writes them - work the
    // whole wave runs for however few lanes - so it pays to wait for more lanes: 52 instead of 44 is 2.5 - 5 % off the
    // 1080p frame times of renderMode 1, 2, 0 and 10, while at 16 spp it costs the 70 k-triangle scene 8 %.
    a.vote_shade = a.depth == 0 && a.spp <= 2 ? ctx->vote_shade_few : ctx->vote_shade; a.vote_node = ctx->vote_node;
    if (instanced) {
        hipLaunchKernelGGL(path_sched_inst_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    } else if (extended) {
        if (width == 8) hipLaunchKernelGGL(path_sched_mat_kernel<8>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
        else if (width == 4) hipLaunchKernelGGL(path_sched_mat_kernel<4>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
        else hipLaunchKernelGGL(path_sched_mat_kernel<2>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    } else if (fast) {
        if (width == 8) hipLaunchKernelGGL(path_sched_fast_kernel<8>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
        else if (width == 4) hipLaunchKernelGGL(path_sched_fast_kernel<4>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
        else hipLaunchKernelGGL(path_sched_fast_kernel<2>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    } else {
        if (width == 8) hipLaunchKernelGGL(path_sched_kernel<8>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
        else if (width == 4) hipLaunchKernelGGL(path_sched_kernel<4>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
        else hipLaunchKernelGGL(path_sched_kernel<2>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    }
    PT_HIP(ctx, hipGetLastError());
    return after.done();
}

// Primary rays only over the rectangle: what == 1 the _Depth dispatch, what == 2 t of the hits into d_seeds.
static int launch_primary(pt_context *ctx, RenderArgs &a, int what, int x0, int y0, int tw, int th, int skip_lo, int skip_hi)
{
    if (tw <= 0 || th <= 0) return PT_OK;
    int rc;
    a.tx0 = x0; a.ty0 = y0; a.tw = tw; a.th = th;
    set_tiles(a, tw, th);
    const int grid = (int)std::min<int64_t>(((int64_t)a.total_items + kBlock - 1) / kBlock, (int64_t)ctx->num_cus * 8 * 4);
    if ((rc = ensure_stack_overflow(ctx, (size_t)grid * kBlock))) return rc;
    a.stack_overflow = ctx->d_stack_ovf;
    if (tree_width(ctx) == 8) hipLaunchKernelGGL(primary_kernel<8>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, what, skip_lo, skip_hi);
    else if (tree_width(ctx) == 4) hipLaunchKernelGGL(primary_kernel<4>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, what, skip_lo, skip_hi);
    else hipLaunchKernelGGL(primary_kernel<2>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, what, skip_lo, skip_hi);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// Scratch of the screen-space passes: one float4 and one float per pixel.
static int ensure_screen_scratch(pt_context *ctx)
{
    if (ctx->screen_pixels >= (size_t)ctx->W * ctx->H) return PT_OK;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_dev(ctx->d_filtered); free_dev(ctx->d_seeds);
    ctx->screen_pixels = 0;
    PT_HIP(ctx, dev_malloc((void **)&ctx->d_filtered, (size_t)ctx->W * ctx->H * sizeof(float4)));
    PT_HIP(ctx, dev_malloc((void **)&ctx->d_seeds, (size_t)ctx->W * ctx->H * sizeof(float)));
    ctx->screen_pixels = (size_t)ctx->W * ctx->H;
    return PT_OK;
}

static int render_frame(pt_context *ctx, const PtFrameParams *p);

int pt_render(pt_context *ctx, const PtFrameParams *p)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    const int rc = render_frame(ctx, p);
    if (rc != PT_OK && ctx->epoch_open) {
        // the statistics epoch must end in a recorded event whatever happened after it opened
        DeviceGuard g(ctx->device);
        (void)hipEventRecord(ctx->ev1, ctx->stream);
    }
    return rc;
    });
}

static int render_frame(pt_context *ctx, const PtFrameParams *p)
{
    if (!p) return fail(ctx, PT_ERR_INVALID, "pt_render: params is NULL");
    if (ctx->W == 0) return fail(ctx, PT_ERR_INVALID, "pt_render before pt_resize");
    const int mode = p->render_mode;
    const bool stripes = mode == 5 || mode == 6;
    const bool full_reproj = mode == 7 || mode == 8;
    const bool pause = mode == 11 || mode == 12;
    if (mode < 0 || mode == 9 || mode > 12)
        return fail(ctx, PT_ERR_UNSUPPORTED, "pt_render: renderMode must be one of RayTracingMaster.RenderMode's values: 0..8, 10, 11, 12");
    if (mode != 1 && p->accumulate != PT_ACCUM_NONE)
        return fail(ctx, PT_ERR_INVALID, "pt_render: every renderMode but 1 uses the converged target as its history; accumulate must be PT_ACCUM_NONE");
    if (p->max_bounces < 1 || p->max_bounces > 4095) return fail(ctx, PT_ERR_INVALID, "pt_render: max_bounces must be 1..4095");
    if (p->samples_per_pixel >= (1 << 19)) return fail(ctx, PT_ERR_INVALID, "pt_render: samples_per_pixel must be below 2^19 per dispatch");
    if (p->accumulate < PT_ACCUM_NONE || p->accumulate > PT_ACCUM_ADDMAT) return fail(ctx, PT_ERR_INVALID, "pt_render: bad accumulate mode");
    int x0 = 0, y0 = 0, tw = ctx->W, th = ctx->H;
    if (p->tile_w > 0 && p->tile_h > 0) { x0 = p->tile_x0; y0 = p->tile_y0; tw = p->tile_w; th = p->tile_h; }
    if (x0 < 0 || y0 < 0 || (int64_t)x0 + tw > ctx->W || (int64_t)y0 + th > ctx->H)
        return fail(ctx, PT_ERR_INVALID, "pt_render: tile outside the image");
    if ((mode == 0 || stripes || full_reproj || pause) && (tw != ctx->W || th != ctx->H))
        return fail(ctx, PT_ERR_UNSUPPORTED, "pt_render: renderMode 0, 5..8, 11 and 12 render whole frames only");
    if ((stripes || full_reproj) && (p->divisions < 1 || p->divisions > PT_MAX_CAMERA_HISTORY || p->counter < 0 || p->counter >= p->divisions))
        return fail(ctx, PT_ERR_INVALID, "pt_render: renderMode 5..8 need 1 <= divisions <= 100 and 0 <= counter < divisions");
    DeviceGuard g(ctx->device);
    int rc;
    {
        // ray origins of this dispatch: the camera, and in renderMode 7 / 8 the history camera the stripe is traced from
        float cams[6] = {p->camera_to_world[12], p->camera_to_world[13], p->camera_to_world[14], 0, 0, 0};
        int n_cams = 1;
        if (full_reproj && (p->counter - 1) % p->divisions >= 0) {
            const float *h = ctx->camera_history[(size_t)((p->counter - 1) % p->divisions)].camera_to_world;
            cams[3] = h[12]; cams[4] = h[13]; cams[5] = h[14]; n_cams = 2;
        }
        if ((rc = ensure_scene(ctx, cams, n_cams, 3))) return rc;
    }
    const size_t pitch = (size_t)ctx->W * sizeof(float4);
    if (mode == 0 || mode == 2 || mode == 3 || mode == 4 || mode == 10 || (mode == 1 && p->accumulate == PT_ACCUM_NONE)) {
        // SH:728-736, Result1 = Result (renderMode 1, 2, 3, 4, 10) or Result2 (renderMode 0) over the dispatched
        // rectangle, finished before any pixel is traced.  renderMode 1 does not read Result1, but a later switch to a
        // mode that shows it without copying first (5..8, 11, 12) does.  With an accumulation mode the converged target
        // belongs to the accumulation (an extension: the shader has none) and is not overwritten.
        const float4 *src = mode == 0 ? ctx->result2 : ctx->result;
        PT_HIP(ctx, hipMemcpy2DAsync(ctx->conv + (size_t)y0 * ctx->W + x0, pitch, src + (size_t)y0 * ctx->W + x0, pitch,
                                     (size_t)tw * sizeof(float4), (size_t)th, hipMemcpyDeviceToDevice, ctx->stream));
    }
    const bool no_samples = p->depth != 1 && p->samples_per_pixel <= 0;   // the shader's sample loop does not run
    if (no_samples && !stripes && !full_reproj) return PT_OK;
    const bool screen_passes = mode == 0 && p->depth != 1;
    if ((screen_passes || mode == 6 || mode == 8 || mode == 12) && (rc = ensure_screen_scratch(ctx))) return rc;

    RenderArgs a;
    a.sc = scene_dev(ctx);
    std::memcpy(a.c2w, p->camera_to_world, sizeof a.c2w);
    std::memcpy(a.ip, p->camera_inverse_projection, sizeof a.ip);
    a.seed = p->seed; a.spp = p->samples_per_pixel; a.maxb = p->max_bounces; a.depth = p->depth == 1 ? 1 : 0;
    a.W = ctx->W; a.H = ctx->H;
    a.tx0 = x0; a.ty0 = y0; a.tw = tw; a.th = th; a.tiles_x = 0; a.tiles_x_magic = 0; a.total_items = 0;
    a.result = ctx->result; a.converged = ctx->conv; a.hit_ids = ctx->ids;
    a.accumulate = p->accumulate; a.sample_index = p->sample_index;
    a.work_counter = ctx->d_work; a.stats = ctx->d_stats;
    a.reuse_primary = ctx->reuse_primary;
    a.render_mode = stripes ? 1 : mode;   // the stripe being rendered is written as renderMode 1 writes (SH:933-936)
    a.left_eye = p->left_eye; a.cam_old = 0; a.skip_t = nullptr; a.copy_all = 0;
    a.alpha_origin[0] = p->camera_to_world[12]; a.alpha_origin[1] = p->camera_to_world[13]; a.alpha_origin[2] = p->camera_to_world[14];
    a.history = ctx->conv;
    a.accumulation_frames = (float)p->accumulation_frames;
    std::memcpy(a.old_c2w, ctx->camera_history[0].camera_to_world, sizeof a.old_c2w);
    std::memcpy(a.old_w2c, ctx->camera_history[0].world_to_camera, sizeof a.old_w2c);
    std::memcpy(a.old_proj, ctx->camera_history[0].camera_projection, sizeof a.old_proj);
    a.result2 = ctx->result2; a.seed_out = ctx->d_seeds;
    a.block_order = nullptr; a.block_cost = nullptr; a.n_blocks = 0; a.pool_state = nullptr;

    if (!ctx->epoch_open) {
        PT_HIP(ctx, hipMemsetAsync(ctx->d_stats, 0, kStatsShards * kStatsStride * sizeof(unsigned long long), ctx->stream));
        PT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        ctx->epoch_open = true;
    }
    if (stripes || full_reproj || pause) {
        // rows of the stripe being rendered: the shader's own float comparisons (SH:742,751), row by row
        int s_lo = ctx->H, s_hi = -1;
        if (!pause) {
            const float fh = (float)ctx->H, stripe = fh / (float)p->divisions;
            for (int y = 0; y < ctx->H; y++) {
                const float fy = (float)y;
                if (!(fy > fh - stripe * (float)p->counter || fy < fh - stripe * (float)(p->counter + 1))) { s_lo = std::min(s_lo, y); s_hi = std::max(s_hi, y); }
            }
        }
        int s_rows = s_hi >= s_lo ? s_hi - s_lo + 1 : 0;
        constexpr size_t cam_bytes = sizeof(PtCameraHistory) * PT_MAX_CAMERA_HISTORY;
        if (ctx->cams_dirty || !ctx->d_cams[ctx->cam_slot]) {
            const int i = ctx->cam_slot = (ctx->cam_slot + 1) % pt_context::kCamRing;
            if (!ctx->d_cams[i]) {
                PT_HIP(ctx, dev_malloc((void **)&ctx->d_cams[i], cam_bytes));
                PT_HIP(ctx, host_malloc((void **)&ctx->h_cams[i], cam_bytes, hipHostMallocDefault));
                PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_cams[i], hipEventDisableTiming));
            }
            if (ctx->cams_used[i]) PT_HIP(ctx, hipEventSynchronize(ctx->ev_cams[i]));   // its last reader, kCamRing uploads ago
            std::memcpy(ctx->h_cams[i], ctx->camera_history.data(), cam_bytes);
            PT_HIP(ctx, hipMemcpyAsync(ctx->d_cams[i], ctx->h_cams[i], cam_bytes, hipMemcpyHostToDevice, ctx->stream));
            ctx->cams_dirty = false;
        }
        const bool have_samples = !no_samples;      // the pause modes write inside the sample loop (SH:1071-1104)
        if (mode % 2 == 0 && (!pause || (have_samples && a.depth != 1))) {
            // SH:757-759, SH:858-863: the pixels' current primary hits (renderMode 6: of the pixels outside the stripe)
            RenderArgs prim = a;
            const bool whole = mode != 6 || !s_rows;
            if ((rc = launch_primary(ctx, prim, 2, 0, 0, ctx->W, ctx->H, whole ? 1 : s_lo, whole ? 0 : s_hi))) return rc;
        }
        StripeArgs sa;
        sa.cams = ctx->d_cams[ctx->cam_slot]; sa.t_primary = ctx->d_seeds; sa.divisions = pause ? 1 : p->divisions; sa.counter = p->counter; sa.mode = mode;
        const dim3 blk(kScreenBlockX, kScreenBlockY), grd((ctx->W + kScreenBlockX - 1) / kScreenBlockX, (ctx->H + kScreenBlockY - 1) / kScreenBlockY);
        if (!pause || (have_samples && a.depth != 1)) {
            hipLaunchKernelGGL(stripe_reproject_kernel, grd, blk, 0, ctx->stream, a, sa);
            PT_HIP(ctx, hipGetLastError());
            PT_HIP(ctx, hipEventRecord(ctx->ev_cams[ctx->cam_slot], ctx->stream));
            ctx->cams_used[ctx->cam_slot] = true;
        }
        if (pause) {
            if (a.depth == 1 && (rc = launch_primary(ctx, a, 1, 0, 0, ctx->W, ctx->H, 1, 0))) return rc;   // SH:799-809 comes first
        } else if (stripes) {
            if (s_rows && a.depth == 1) { if ((rc = launch_primary(ctx, a, 1, 0, s_lo, ctx->W, s_rows, 1, 0))) return rc; }
            else if (s_rows && !no_samples && (rc = launch_path(ctx, a, 0, s_lo, ctx->W, s_rows))) return rc;
            if (s_rows && p->depth != 1)   // SH:1108-1112, after every read of Result1 above
                PT_HIP(ctx, hipMemcpy2DAsync(ctx->conv + (size_t)s_lo * ctx->W, pitch, ctx->result + (size_t)s_lo * ctx->W, pitch,
                                             (size_t)ctx->W * sizeof(float4), (size_t)s_rows, hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            // renderMode 7 / 8 (SH:786-788, 850-853, 938-948): the first dispatch of a cycle goes on for every pixel, the others for the stripe
            const bool first = p->counter % p->divisions == 0;
            if (first) { s_lo = 0; s_rows = ctx->H; }
            if (mode == 8) a.skip_t = ctx->d_seeds;                      // SH:760-763: a primary miss has returned
            if (s_rows && a.depth == 1) {
                if ((rc = launch_primary(ctx, a, mode == 8 ? 3 : 1, 0, s_lo, ctx->W, s_rows, 1, 0))) return rc;
            } else if (s_rows && !no_samples) {
                const int fb = (p->counter - 1) % p->divisions;          // -1 on the first dispatch: the current camera (SH:8-11)
                if (fb >= 0) {
                    std::memcpy(a.c2w, ctx->camera_history[fb].camera_to_world, sizeof a.c2w);
                    std::memcpy(a.ip, ctx->camera_history[fb].camera_inverse_projection, sizeof a.ip);
                }
                a.cam_old = 1;
                a.copy_all = first ? 1 : 0;
                if ((rc = launch_path(ctx, a, 0, s_lo, ctx->W, s_rows))) return rc;
            }
        }
    } else if (a.depth == 1) {
        if ((rc = launch_primary(ctx, a, 1, x0, y0, tw, th, 1, 0))) return rc;
    } else {
        if ((rc = launch_path(ctx, a, x0, y0, tw, th))) return rc;
    }
    if (screen_passes) {
        // SH:928-929 as two launches: filter every pixel, then resolve every pixel
        ScreenArgs sa;
        sa.history = ctx->conv; sa.seeds = ctx->d_seeds; sa.filtered = ctx->d_filtered; sa.result = ctx->result;
        sa.W = ctx->W; sa.H = ctx->H;
        const dim3 blk(kScreenBlockX, kScreenBlockY), grd((ctx->W + kScreenBlockX - 1) / kScreenBlockX, (ctx->H + kScreenBlockY - 1) / kScreenBlockY);
        hipLaunchKernelGGL(denoise_kernel, grd, blk, 0, ctx->stream, sa);
        hipLaunchKernelGGL(taa_kernel, grd, blk, 0, ctx->stream, sa, a);
        PT_HIP(ctx, hipGetLastError());
    }
    PT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    return PT_OK;
}

int pt_set_camera_history(pt_context *ctx, const PtCameraHistory *entries, int count)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (count < 0 || count > PT_MAX_CAMERA_HISTORY) return fail(ctx, PT_ERR_INVALID, "pt_set_camera_history: count must be 0..100");
    if (count == 0) return PT_OK;   // an empty list writes nothing into the uniform arrays
    if (!entries && count != PT_MAX_CAMERA_HISTORY) return fail(ctx, PT_ERR_INVALID, "pt_set_camera_history: entries is NULL");
    if (!entries) std::memset(ctx->camera_history.data(), 0, sizeof(PtCameraHistory) * PT_MAX_CAMERA_HISTORY);
    else std::memcpy(ctx->camera_history.data(), entries, sizeof(PtCameraHistory) * (size_t)count);
    ctx->cams_dirty = true;
    return PT_OK;
    });
}

int pt_set_option(pt_context *ctx, const char *name, int value)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (!name) return fail(ctx, PT_ERR_INVALID, "pt_set_option: name is NULL");
    if (std::strcmp(name, "backend") == 0) {
        if (value < 0 || value > 2) return fail(ctx, PT_ERR_INVALID, "pt_set_option: backend must be 0, 1 or 2");
        ctx->backend = value;
        return PT_OK;
    }
    if (std::strcmp(name, "bvh_width") == 0) {
        if (value != 2 && value != 4 && value != 8) return fail(ctx, PT_ERR_INVALID, "pt_set_option: bvh_width must be 2, 4 or 8");
        if ((value == 8) != (ctx->bvh_width == 8)) ctx->scene_dirty = true;     // the eight-wide tree has its own record order
        ctx->bvh_width = value;
        return PT_OK;
    }
    if (std::strcmp(name, "block_order") == 0) {
        if (value != 0 && value != 1) return fail(ctx, PT_ERR_INVALID, "pt_set_option: block_order must be 0 or 1");
        ctx->block_order = value; ctx->order_ready = false;
        return PT_OK;
    }
    if (std::strcmp(name, "reuse_primary_hit") == 0) {
        if (value != 0 && value != 1) return fail(ctx, PT_ERR_INVALID, "pt_set_option: reuse_primary_hit must be 0 or 1");
        ctx->reuse_primary = value;
        return PT_OK;
    }
    if (std::strcmp(name, "fast_intrinsics") == 0) {
        if (value != 0 && value != 1) return fail(ctx, PT_ERR_INVALID, "pt_set_option: fast_intrinsics must be 0 or 1");
        ctx->fast_intrinsics = value;
        return PT_OK;
    }
    if (std::strcmp(name, "blocks_per_cu") == 0) {
        if (value < 0 || value > 16) return fail(ctx, PT_ERR_INVALID, "pt_set_option: blocks_per_cu must be 0..16");
        ctx->blocks_per_cu = value;
        return PT_OK;
    }
    if (std::strcmp(name, "refit") == 0) {
        if (value != 0 && value != 1) return fail(ctx, PT_ERR_INVALID, "pt_set_option: refit must be 0 or 1");
        if (value && !ctx->refit) ctx->scene_dirty = true;   // the refit buffers are made by a full build
        ctx->refit = value;
        return PT_OK;
    }
    if (std::strcmp(name, "instancing") == 0) {
        if (value < 0 || value > 2) return fail(ctx, PT_ERR_INVALID, "pt_set_option: instancing must be 0, 1 or 2");
        if (ctx->instancing != value) ctx->scene_dirty = true;
        ctx->instancing = value;
        return PT_OK;
    }
    if (std::strcmp(name, "debug_throw") == 0) {          // exercises the guard around every entry point (tests): 1 bad_alloc, 2 runtime_error
        if (value == 1) throw std::bad_alloc();
        if (value == 2) throw std::runtime_error("debug_throw");
        return PT_OK;
    }
    if (std::strcmp(name, "rebuild_percent") == 0) {
        if (value < 0 || value > 10000) return fail(ctx, PT_ERR_INVALID, "pt_set_option: rebuild_percent must be 0..10000");
        ctx->rebuild_percent = value;
        return PT_OK;
    }
    if (std::strcmp(name, "tree_build") == 0) {
        if (value < 0 || value > 3) return fail(ctx, PT_ERR_INVALID, "pt_set_option: tree_build must be 0, 1, 2 or 3");
        if (ctx->tree_build != value) ctx->scene_dirty = true;
        ctx->tree_build = value;
        return PT_OK;
    }
    if (std::strcmp(name, "split_percent") == 0) {
        if (value < 0 || value > 400) return fail(ctx, PT_ERR_INVALID, "pt_set_option: split_percent must be 0..400");
        if (ctx->split_budget != 0.01f * (float)value) ctx->scene_dirty = true;
        ctx->split_budget = 0.01f * (float)value;
        return PT_OK;
    }
    if (std::strcmp(name, "refill_threshold") == 0) {
        if (value < 1 || value > 64) return fail(ctx, PT_ERR_INVALID, "pt_set_option: refill_threshold must be 1..64");
        ctx->refill_threshold = value;
        return PT_OK;
    }
    return fail(ctx, PT_ERR_INVALID, std::string("pt_set_option: unknown option ") + name);
    });
}

int pt_display_denoise(pt_context *ctx)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (ctx->W == 0) return fail(ctx, PT_ERR_INVALID, "pt_display_denoise before pt_resize");
    DeviceGuard g(ctx->device);
    if (!ctx->display) {
        if (!ctx->own_display) {
            PT_HIP(ctx, dev_malloc((void **)&ctx->own_display, (size_t)ctx->W * ctx->H * sizeof(float4)));
            PT_HIP(ctx, hipMemsetAsync(ctx->own_display, 0, (size_t)ctx->W * ctx->H * sizeof(float4), ctx->stream));
        }
        ctx->display = ctx->own_display;
    }
    const dim3 blk(kScreenBlockX, kScreenBlockY), grd((ctx->W + kScreenBlockX - 1) / kScreenBlockX, (ctx->H + kScreenBlockY - 1) / kScreenBlockY);
    hipLaunchKernelGGL(display_denoise_kernel, grd, blk, 0, ctx->stream, (const float4 *)ctx->result, ctx->display, ctx->W, ctx->H);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
    });
}

int pt_sync(pt_context *ctx)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    DeviceGuard g(ctx->device);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
    });
}

static int read_back(pt_context *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!dst) return fail(ctx, PT_ERR_INVALID, "output pointer is NULL");
    if (ctx->W == 0 || !src) return fail(ctx, PT_ERR_INVALID, "read before pt_resize");
    DeviceGuard g(ctx->device);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_read_result(pt_context *ctx, float *out)
{
    return guarded(ctx, [&]() -> int {
    return read_back(ctx, out, ctx ? ctx->result : nullptr, ctx ? (size_t)ctx->W * ctx->H * 16 : 0);
    });
}
int pt_read_converged(pt_context *ctx, float *out)
{
    return guarded(ctx, [&]() -> int {
    return read_back(ctx, out, ctx ? ctx->conv : nullptr, ctx ? (size_t)ctx->W * ctx->H * 16 : 0);
    });
}
int pt_read_hit_ids(pt_context *ctx, int32_t *out)
{
    return guarded(ctx, [&]() -> int {
    return read_back(ctx, out, ctx ? ctx->ids : nullptr, ctx ? (size_t)ctx->W * ctx->H * 4 : 0);
    });
}
int pt_write_converged(pt_context *ctx, const float *in)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (!in || ctx->W == 0) return fail(ctx, PT_ERR_INVALID, "pt_write_converged: bad arguments");
    DeviceGuard g(ctx->device);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return h2d(ctx, ctx->conv, in, (size_t)ctx->W * ctx->H * 16);
    });
}

static float4 *target_of(pt_context *ctx, int which)
{
    switch (which) {
    case PT_TARGET_RESULT: return ctx->result;
    case PT_TARGET_CONVERGED: return ctx->conv;
    case PT_TARGET_DENOISED: return ctx->result2;
    case PT_TARGET_DISPLAY: return ctx->display;
    default: return nullptr;
    }
}
int pt_read_target(pt_context *ctx, int which, float *out)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    float4 *src = target_of(ctx, which);
    if (!src && ctx->W) return fail(ctx, PT_ERR_INVALID, "pt_read_target: not an RGBA target");
    return read_back(ctx, out, src, (size_t)ctx->W * ctx->H * 16);
    });
}
int pt_write_target(pt_context *ctx, int which, const float *in)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    float4 *dst = target_of(ctx, which);
    if (!in || ctx->W == 0 || !dst) return fail(ctx, PT_ERR_INVALID, "pt_write_target: bad arguments");
    DeviceGuard g(ctx->device);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return h2d(ctx, dst, in, (size_t)ctx->W * ctx->H * 16);
    });
}

int pt_get_stats(pt_context *ctx, PtStats *out)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (!out) return fail(ctx, PT_ERR_INVALID, "pt_get_stats: out is NULL");
    DeviceGuard g(ctx->device);
    int rc = finish_epoch(ctx);
    if (rc) return rc;
    *out = ctx->stats;
    return PT_OK;
    });
}

int pt_trace_rays(pt_context *ctx, const float *origins_dirs, int n_rays, int brute_force, int32_t *ids_out, float *t_out)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (n_rays < 0 || (n_rays > 0 && (!origins_dirs || !ids_out || !t_out)))
        return fail(ctx, PT_ERR_INVALID, "pt_trace_rays: bad arguments");
    if (n_rays == 0) return PT_OK;
    DeviceGuard g(ctx->device);
    int rc;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = ensure_scene(ctx, origins_dirs, n_rays, 6))) return rc;
    float *d_od = nullptr, *d_t = nullptr; int32_t *d_id = nullptr;
    auto cleanup = [&] { free_dev(d_od); free_dev(d_t); free_dev(d_id); };
    hipError_t e;
    if ((e = dev_malloc((void **)&d_od, (size_t)n_rays * 24)) != hipSuccess ||
        (e = dev_malloc((void **)&d_t, (size_t)n_rays * 4)) != hipSuccess ||
        (e = dev_malloc((void **)&d_id, (size_t)n_rays * 4)) != hipSuccess ||
        (e = hipMemcpyAsync(d_od, origins_dirs, (size_t)n_rays * 24, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) {
        cleanup();
        return fail(ctx, PT_ERR_HIP, std::string("pt_trace_rays: ") + hipGetErrorString(e));
    }
    int grid = (n_rays + kBlock - 1) / kBlock;
    if ((rc = ensure_stack_overflow(ctx, (size_t)grid * kBlock))) { cleanup(); return rc; }
    SceneDev sc = scene_dev(ctx);
    if (brute_force)
        hipLaunchKernelGGL(trace_rays_brute_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, sc, d_od, n_rays, d_id, d_t);
    else if (tree_width(ctx) == 8)
        hipLaunchKernelGGL(trace_rays_bvh_kernel<8>, dim3(grid), dim3(kBlock), 0, ctx->stream, sc, ctx->d_stack_ovf, d_od, n_rays, d_id, d_t);
    else if (tree_width(ctx) == 4)
        hipLaunchKernelGGL(trace_rays_bvh_kernel<4>, dim3(grid), dim3(kBlock), 0, ctx->stream, sc, ctx->d_stack_ovf, d_od, n_rays, d_id, d_t);
    else
        hipLaunchKernelGGL(trace_rays_bvh_kernel<2>, dim3(grid), dim3(kBlock), 0, ctx->stream, sc, ctx->d_stack_ovf, d_od, n_rays, d_id, d_t);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(ctx->stream)) != hipSuccess ||
        (e = hipMemcpy(ids_out, d_id, (size_t)n_rays * 4, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(t_out, d_t, (size_t)n_rays * 4, hipMemcpyDeviceToHost)) != hipSuccess) {
        cleanup();
        return fail(ctx, PT_ERR_HIP, std::string("pt_trace_rays: ") + hipGetErrorString(e));
    }
    cleanup();
    return PT_OK;
    });
}

int pt_trace_rays_static(pt_context *ctx, const float *origins_dirs, int n_rays, int brute_force, int32_t *ids_out, float *t_out)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return PT_ERR_INVALID;
    if (n_rays < 0 || (n_rays > 0 && (!origins_dirs || !ids_out || !t_out)))
        return fail(ctx, PT_ERR_INVALID, "pt_trace_rays_static: bad arguments");
    if ((size_t)n_rays > kStaticMaxRays)
        return fail(ctx, PT_ERR_INVALID, "pt_trace_rays_static: at most " + std::to_string(kStaticMaxRays) + " rays per call (PT_TRACE_RAYS_STATIC_MAX)");
    if (n_rays == 0) return PT_OK;
    // Building or re-inflating the scene allocates; this call does not, so the scene must be ready.
    if (ctx->scene_dirty || (tree_width(ctx) == 2 && (ctx->bvh2_stale || ctx->n_inst)))
        return fail(ctx, PT_ERR_INVALID, "pt_trace_rays_static: the scene has changed since it was last built; call pt_render or pt_trace_rays once first");
    if (ctx->n_prims > 0 && reach_needed(ctx, origins_dirs, n_rays, 6) > ctx->sphere_reach)
        return fail(ctx, PT_ERR_INVALID, "pt_trace_rays_static: a ray starts farther from the scene than its tree was built for; call pt_trace_rays once with such rays first");
    unsigned char *base = scratch_of(ctx->device);
    if (!base) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_trace_rays_static: no scratch array on this device (ordinal beyond " + std::to_string(kMaxScratchDevices - 1) + ")");
    ScratchLease lease(ctx->device);
    if (!lease.held) return fail(ctx, PT_ERR_INVALID, "pt_trace_rays_static: the scratch array of this GPU is in use by another call (calls on one GPU must not overlap)");
    DeviceGuard g(ctx->device);
    float *d_od = reinterpret_cast<float *>(base);
    int32_t *d_id = reinterpret_cast<int32_t *>(base + kStaticIds);
    float *d_t = reinterpret_cast<float *>(base + kStaticT);
    uint32_t *d_stack = reinterpret_cast<uint32_t *>(base + kStaticStack);
    const int grid = (n_rays + kBlock - 1) / kBlock;      // grid * kBlock <= kStaticMaxRays threads, kStackOverflow entries each
    PT_HIP(ctx, hipMemcpyAsync(d_od, origins_dirs, (size_t)n_rays * 24, hipMemcpyHostToDevice, ctx->stream));
    const SceneDev sc = scene_dev(ctx);
    if (brute_force)
        hipLaunchKernelGGL(trace_rays_brute_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, sc, d_od, n_rays, d_id, d_t);
    else if (tree_width(ctx) == 8)
        hipLaunchKernelGGL(trace_rays_bvh_kernel<8>, dim3(grid), dim3(kBlock), 0, ctx->stream, sc, d_stack, d_od, n_rays, d_id, d_t);
    else if (tree_width(ctx) == 4)
        hipLaunchKernelGGL(trace_rays_bvh_kernel<4>, dim3(grid), dim3(kBlock), 0, ctx->stream, sc, d_stack, d_od, n_rays, d_id, d_t);
    else
        hipLaunchKernelGGL(trace_rays_bvh_kernel<2>, dim3(grid), dim3(kBlock), 0, ctx->stream, sc, d_stack, d_od, n_rays, d_id, d_t);
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipMemcpyAsync(ids_out, d_id, (size_t)n_rays * 4, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipMemcpyAsync(t_out, d_t, (size_t)n_rays * 4, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
    });
}

// ---------------------------------------------------------------- device groups

int pt_group_create(int n, const int *device_ordinals, pt_group **out_group)
{
    return guarded(nullptr, [&]() -> int {
    if (!out_group) return fail(nullptr, PT_ERR_INVALID, "pt_group_create: out_group is NULL");
    *out_group = nullptr;
    if (n < 1 || n > 64 || !device_ordinals) return fail(nullptr, PT_ERR_INVALID, "pt_group_create: need 1..64 device ordinals");
    pt_group *g = new (std::nothrow) pt_group();
    if (!g) return fail(nullptr, PT_ERR_NOMEM, "out of host memory");
    for (int i = 0; i < n; i++) {
        pt_context *c = nullptr;
        int rc = pt_create(device_ordinals[i], &c);
        hipEvent_t ev = nullptr;
        if (rc == PT_OK) {
            DeviceGuard dg(device_ordinals[i]);
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) { g_create_error = std::string("hipEventCreate: ") + hipGetErrorString(e); pt_destroy(c); rc = PT_ERR_HIP; }
        }
        if (rc != PT_OK) { pt_group_destroy(g); return rc; }      // g_create_error holds the reason
        g->members.push_back(c); g->ready.push_back(ev);
    }
    *out_group = g;
    return PT_OK;
    });
}

void pt_group_destroy(pt_group *g)
{
    if (!g) return;
    if (g->staging) { DeviceGuard dg(g->staging_device); (void)dev_free(g->staging); }
    if (g->taken) { DeviceGuard dg(g->staging_device); (void)hipEventDestroy(g->taken); }
    for (size_t i = 0; i < g->members.size(); i++) {
        if (g->ready[i]) { DeviceGuard dg(g->members[i]->device); (void)hipEventDestroy(g->ready[i]); }
        pt_destroy(g->members[i]);
    }
    delete g;
}

int pt_group_size(const pt_group *g) { return g ? (int)g->members.size() : 0; }

pt_context *pt_group_context(pt_group *g, int index)
{
    return g && index >= 0 && index < (int)g->members.size() ? g->members[(size_t)index] : nullptr;
}

const char *pt_group_last_error(const pt_group *g) { return g ? g->err.c_str() : g_create_error.c_str(); }

int pt_group_reduce(pt_group *g, int root)
{
    const auto body = [&]() -> int {
    if (!g) return PT_ERR_INVALID;
    auto gfail = [&](int code, const std::string &msg) { g->err = msg; return code; };
    const int n = (int)g->members.size();
    if (root < 0 || root >= n) return gfail(PT_ERR_INVALID, "pt_group_reduce: root is not a member index");
    pt_context *r = g->members[(size_t)root];
    if (r->W == 0) return gfail(PT_ERR_INVALID, "pt_group_reduce before pt_resize");
    for (pt_context *c : g->members)
        if (c->W != r->W || c->H != r->H) return gfail(PT_ERR_INVALID, "pt_group_reduce: the members' image sizes differ");
    if (n == 1) return PT_OK;
    const size_t pixels = (size_t)r->W * r->H, bytes = pixels * sizeof(float4);
#define PT_GHIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return gfail(e_ == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
    // every member marks the end of its renders so far
    for (int i = 0; i < n; i++) {
        if (i == root) continue;
        DeviceGuard dg(g->members[(size_t)i]->device);
        PT_GHIP(hipEventRecord(g->ready[(size_t)i], g->members[(size_t)i]->stream));
    }
    {
        DeviceGuard dg(r->device);
        if (g->staging_pixels < pixels || g->staging_device != r->device) {
            PT_GHIP(hipStreamSynchronize(r->stream));
            if (g->staging) { DeviceGuard old(g->staging_device); (void)dev_free(g->staging); g->staging = nullptr; }
            if (g->taken) { DeviceGuard old(g->staging_device); (void)hipEventDestroy(g->taken); g->taken = nullptr; }
            g->staging_pixels = 0;
            PT_GHIP(dev_malloc((void **)&g->staging, bytes));
            PT_GHIP(hipEventCreateWithFlags(&g->taken, hipEventDisableTiming));
            g->staging_pixels = pixels; g->staging_device = r->device;
        }
        // members in index order, each image copied over the fabric into the staging buffer and added: a fixed order,
        // so the sum is reproducible from run to run
        for (int i = 0; i < n; i++) {
            if (i == root) continue;
            pt_context *c = g->members[(size_t)i];
            PT_GHIP(hipStreamWaitEvent(r->stream, g->ready[(size_t)i], 0));
            PT_GHIP(hipMemcpyPeerAsync(g->staging, r->device, c->conv, c->device, bytes, r->stream));
            hipLaunchKernelGGL(ptrace::add_image_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, r->stream, r->conv, g->staging, pixels);
            PT_GHIP(hipGetLastError());
        }
        PT_GHIP(hipEventRecord(g->taken, r->stream));
    }
    // a member's next render may overwrite its image: it waits until the root has read it
    for (int i = 0; i < n; i++) {
        if (i == root) continue;
        DeviceGuard dg(g->members[(size_t)i]->device);
        PT_GHIP(hipStreamWaitEvent(g->members[(size_t)i]->stream, g->taken, 0));
    }
#undef PT_GHIP
    return PT_OK;
    };
    try { return body(); }
    catch (const std::bad_alloc &) { try { if (g) g->err = "out of host memory"; } catch (...) {} return PT_ERR_NOMEM; }
    catch (...) { try { if (g) g->err = "exception in pt_group_reduce"; } catch (...) {} return PT_ERR_HIP; }
}

}  // extern "C"
