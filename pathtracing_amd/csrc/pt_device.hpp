// Device code of libptrace (gfx950, wave64).
//
// SH = Assets/Scripts/RayTracingShader.compute of the reference.  The shading
// and intersection arithmetic follows SH operation by operation on top of the
// intrinsics in include/ptrace_fmath.h (compiled with -ffp-contract=off), so
// that it produces the same bits as any other implementation written on those
// intrinsics.  The traversal around it (BVH, work distribution) is free to use
// fast arithmetic: it only has to be conservative, never exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ptrace.h"
#include "../../include/ptrace_fmath.h"

namespace ptrace {

typedef pt_f3 f3;

struct SceneDev {
    const float4 *nodes;       // 4 x float4 per BvhNode (two children per node)
    const uint4 *nodes4;       // 4 x uint4 per Bvh4Node (four children, 8-bit boxes); see bvh.hpp
    const uint4 *nodes8;       // 5 x uint4 per Bvh8Node (eight children, 8-bit boxes, bases and masks instead of links)
    const float4 *prims;       // 3 x float4 per primitive record, BVH order
    const float4 *sphere_mat;  // 4 x float4 per sphere: (pos,r) (albedo,smoothness) (specular,0) (emission,0)
    const float4 *sky;         // RGBA32F texels, row 0 at v = 0
    const float4 *mesh_mat;    // 3 x float4 per mesh object: (albedo, smoothness) (specular, bits(lobe)) (emission, 0);
                               // NULL = the shader's hard-coded surface for every triangle (SH:250-253)
    int32_t sky_w, sky_h;
    int32_t n_nodes, n_prims, n_spheres;
    int32_t n_top_records;     // records of the top-level tree: the first n_top_records entries of `prims`
    int32_t n_nodes4;          // four-wide nodes (the stride between the node arrays of a PT_NODES_SOA build)
    // two-level scenes (instances): NULL / 0 when every mesh object is flattened into world-space records
    const float4 *inst;        // 8 x float4 per instance: rows 0..2 of localToWorld (x = row . (v, 1)), rows 0..2 of its inverse,
                               // (bits(root node of the mesh's tree), bits(id of the instance's first triangle), bits(mesh object index), 0), spare
    const int4 *objects;       // per mesh object: (first triangle in scene order, first index, 0, 0); one more entry holds the triangle total
    const int32_t *indices;    // the scene's index buffer (RayTracingMaster.cs:107)
    const float *vertices;     // object-space positions, 3 floats per vertex (:106)
    const float *matrices;     // 16 floats per mesh object (:105)
    int32_t n_inst;
    // rays that start farther from the scene than the boxes were padded for (far_reach): see intersect_far_spheres()
    float centre[3], half_diag, far_reach;
    float spheres_lo[3], spheres_hi[3];   // bounds of the spheres (centre -+ |radius|), inside-out without spheres
};

// How far outside a triangle the Moeller-Trumbore arithmetic (SH:192-230) can still report a hit.  Its barycentric
// coordinates come out of cross and dot products of the edges with the ray's offset from vertex 0, so their rounding error
// grows with the edge length L, with the offset's length (at most `reach`) and with the triangle's thinness L / altitude
// = L * L / |e1 x e2|; in world units: about 8 eps * (L * L / |e1 x e2|) * max(L, reach).  Ordinary meshes stay below the
// general padding; a sliver that runs out to 1e6 gets a box as wide as its bogus hits reach.  Infinite for zero area.
__host__ __device__ inline float triangle_fuzz(pt_f3 e1, pt_f3 e2, float reach)
{
    const pt_f3 e3 = pt_sub3(e2, e1);
    const float l2 = fmaxf(pt_dot3(e1, e1), fmaxf(pt_dot3(e2, e2), pt_dot3(e3, e3)));
    const pt_f3 n = pt_cross3(e1, e2);
    const float area2 = sqrtf(pt_dot3(n, n));
    return 8.0f * 5.9604645e-8f * (l2 / area2) * fmaxf(sqrtf(l2), reach);
}

// Primitive record (48 B): a = (v0.xyz, bits(id)), b = (e1.xyz, bits(kind)), c = (e2.xyz, bits(mesh object index))
// with e1 = v1 - v0, e2 = v2 - v0 in world space (kind 0), or for a sphere
// a = (centre.xyz, bits(id)), b = (radius, 0, 0, bits(1)).
// id: 1 + i for sphere i, 1 + n_spheres + k for triangle k in the reference's
// visiting order; the ground plane is id 0 and a miss is -1.

struct RenderArgs {
    SceneDev sc;
    float c2w[16];
    float ip[16];
    float seed;
    int32_t spp, maxb, depth;     // depth: _Depth (0 / 1), or 2 = write the primary hit's t to seed_out and nothing else
    int32_t W, H;
    int32_t tx0, ty0, tw, th;     // tile rectangle
    int32_t tiles_x;              // 8x8 pixel blocks per row of the tile rectangle
    uint32_t tiles_x_magic;       // 0, or floor(2^32 / tiles_x) + 1: then block / tiles_x = umulhi(block, magic) for every block of the rectangle
    uint32_t total_items;         // tiles_x * tiles_y * 64
    float4 *result;
    float4 *converged;
    int32_t *hit_ids;
    int32_t accumulate;
    uint32_t sample_index;
    uint32_t *work_counter;       // kWorkShards counters, kWorkStride words apart; counter s hands out the work items
                                  // s * shard_items ... (s + 1) * shard_items - 1
    uint32_t shard_items;         // a multiple of 64 (whole 8x8 pixel blocks)
    int32_t n_shards;             // 1 .. kWorkShards counters in use
    uint32_t *stack_overflow;     // kStackOverflow * (threads in the grid) entries
    unsigned long long *stats;    // [0] rays [1] nodes [2] prims [3] paths
    // wavefront pipeline state, one slot per work item (pt_wavefront.hpp)
    float4 *ray_o;                // (origin.xyz, _Seed)
    float4 *ray_d;                // (direction.xyz, bits(sample | bounce << 24))
    float4 *hit;                  // (t, bits(id), bits(record), 0)
    float4 *path_e;               // (energy.xyz, primary-hit distance)
    float4 *path_r;               // (result.xyz, bits(primary id))
    const uint32_t *q_in;         // slots with a ray to trace this iteration
    uint32_t *q_out;              // slots with a ray for the next iteration
    uint32_t *counts;             // counts[i] = entries of the queue of iteration i
    uint32_t *cursors;            // cursors[i] = next unclaimed queue index of iteration i
    int32_t iter;
    int32_t refill_threshold;     // wf_trace refills a wave once this many lanes are idle
    int32_t node_vote;            // wf_trace leaves the inner-node loop when fewer lanes than this hold an inner node
    int32_t vote_shade, vote_node; // path_sched_kernel phase thresholds
    // renderMode 2 (temporal reprojection, SH:950-983)
    int32_t reuse_primary;        // option: trace a pixel's primary ray once and reuse its hit for the later samples
    int32_t render_mode;
    float accumulation_frames;    // (float)_AccumulationFrames
    float old_c2w[16];            // _CameraToWorldOld[0]
    float old_w2c[16];            // _WorldToCameraOld[0]
    float old_proj[16];           // _CameraProjectionOld[0]
    const float4 *history;        // Result1: the previous frame's Result (renderMode 2) or Result2 (renderMode 0)
    // renderMode 0 (SH:885-925): the reprojected radiance goes to Result2, and the pixel's running _Seed is kept
    // for the denoiser's hash (SH:564)
    float4 *result2;
    float *seed_out;
    // the remaining renderModes; read by the extended kernel only (path_sched_mat_kernel)
    int32_t left_eye;             // _leftEye, renderMode 4
    int32_t cam_old;              // renderMode 7 / 8: c2w / ip hold a history camera and rays are built as CreateCameraRayOld does
    float alpha_origin[3];        // renderMode 7 / 8: origin of the CURRENT camera, which the alpha channel measures from (SH:940-941)
    const float *skip_t;          // renderMode 8: t of every pixel's current primary hit; pixels with a miss are not traced (SH:760-763)
    int32_t copy_all;             // renderMode 7 / 8, first dispatch of a cycle: the pixel also goes to Result1 and Result (SH:944-947)
    // expensive blocks first (both path kernels)
    const uint32_t *block_order;  // NULL, or the 8x8 pixel blocks of the tile rectangle by falling cost of the previous frame
    uint32_t *block_cost;         // NULL, or one counter per block: the most work a first sample of its pixels took, for the next frame's order
    uint32_t n_blocks;            // blocks in the tile rectangle
    // path_pool_kernel (pt_pool.hpp)
    float4 *pool_state;           // 3 x float4 per slot of the grid: what only Shade() touches
    int32_t pool_vote_shade, pool_vote_node, pool_vote_leaf;   // how many slots each phase wants to see before it is worth running
    int32_t pool_cont;            // the node phase keeps stepping while at least pool_cont / 64 of the gathered rays hold an inner node
};

constexpr int kBlock = 256;            // 4 waves
constexpr int kWorkShards = 64;        // most counters a frame is split over: one hot counter serialises the pixel claims
                                       // of a 1-spp frame at the L2, many counters scatter the waves over the image
constexpr int kWorkStride = 16;        // u32 words between counters: one cache line each
constexpr int kStackLds = 12;          // traversal stack entries kept in LDS per lane: with the path kernel's other LDS arrays that
                                       // is 31 KB per block, so five blocks fit the 160 KB of a CU
constexpr int kStackOverflow = 80;     // further entries in a global buffer.  Capacity 92: a two-child tree stacks one entry per
                                       // level (<= kMaxDepth + 1 = 61); a four-child tree up to three per level, which a lopsided
                                       // collapse can push past 92 - the host measures each tree's worst case (BvhResult::stack_need4)
                                       // and walks the two-child tree instead when it does not fit
constexpr int kDoneNode = (int)0x80000000;   // traversal state "stack empty: ray finished"
// Two-level scenes: a leaf link with this bit set in its complement names an instance (~link = kInstanceBit | instance)
// instead of primitive records, and kExitInstance on the stack marks where a ray leaves the instance it is in.
constexpr uint32_t kInstanceBit = 0x40000000u;
constexpr int kExitInstance = (int)0x80000007;   // between the lane states below and the smallest leaf link

#ifndef PT_STACK_T
#define PT_STACK_T 0
#endif
__device__ __forceinline__ float as_f(int32_t i) { return __int_as_float(i); }
__device__ __forceinline__ int32_t as_i(float f) { return __float_as_int(f); }
__device__ __forceinline__ f3 xyz(float4 q) { return pt_make3(q.x, q.y, q.z); }

// ------------------------------------------------------------------ traversal

// Traversal stack of one lane: the first kStackLds entries live in LDS (entry e of
// thread t at lds[e * kBlock + t]: consecutive lanes hit consecutive banks), deeper
// entries in a global overflow buffer laid out [entry][global thread] so that a
// wave's accesses coalesce.  bvh.hpp's kMaxDepth bounds the total.
// WITH_T: the entries in LDS also keep the distance at which the ray enters their box (lds_t, same layout), so that a
// popped entry that lies beyond the hit found meanwhile is dropped without a visit; entries in the overflow buffer carry
// no distance and are always visited.
template <int LDS_ENTRIES, int LDS_STRIDE, bool WITH_T = false>
struct StackT {
    uint32_t *lds;        // entry 0 of this lane's stack; entry e lies LDS_STRIDE words further per entry
    uint32_t *overflow;   // entry LDS_ENTRIES of this lane's stack in the global buffer
    uint32_t stride;      // words between consecutive overflow entries (stacks in the grid)
    uint32_t *lds_t = nullptr;   // WITH_T: the entry distances' bits
    static constexpr int kLds = LDS_ENTRIES, kStride = LDS_STRIDE;
    static constexpr bool kWithT = WITH_T;
    static constexpr int kCapacity = WITH_T ? LDS_ENTRIES + kStackOverflow : kStackLds + kStackOverflow;
};
typedef StackT<kStackLds, kBlock> Stack;   // one stack per thread of a kBlock-thread block: &lds_stack[threadIdx.x], &overflow[global thread id]

template <class ST>
__device__ __forceinline__ void stack_push(const ST &s, int &sp, int32_t v, uint32_t tbits = 0)
{
    if (sp < ST::kLds) { s.lds[sp * ST::kStride] = (uint32_t)v; if constexpr (ST::kWithT) s.lds_t[sp * ST::kStride] = tbits; }
    else if (sp < ST::kCapacity) s.overflow[(size_t)(sp - ST::kLds) * s.stride] = (uint32_t)v;
    sp++;  // beyond capacity the entry would be dropped; the host never launches on a tree that can get there
}
template <class ST>
__device__ __forceinline__ int32_t stack_pop(const ST &s, int &sp)
{
    sp--;
    if (sp < ST::kLds) return (int32_t)s.lds[sp * ST::kStride];
    if (sp < ST::kCapacity) return (int32_t)s.overflow[(size_t)(sp - ST::kLds) * s.stride];
    return 0;
}
// The next stacked node the ray still has to visit, or kDoneNode: with distances on the stack, entries whose box the ray
// enters beyond `tbest` are skipped (strictly beyond: an equally near hit with a smaller id must still be found).
template <class ST>
__device__ __forceinline__ int32_t stack_next(const ST &s, int &sp, float tbest)
{
    if constexpr (ST::kWithT) {
        while (sp) {
            sp--;
            if (sp < ST::kLds) {
                const int32_t v = (int32_t)s.lds[sp * ST::kStride];
                if (!(__uint_as_float(s.lds_t[sp * ST::kStride]) > tbest)) return v;
            } else {
                return sp < ST::kCapacity ? (int32_t)s.overflow[(size_t)(sp - ST::kLds) * s.stride] : 0;
            }
        }
        return kDoneNode;
    } else {
        return sp ? stack_pop(s, sp) : kDoneNode;
    }
}

// One primitive record against the ray; keeps the (t, id)-smallest hit, which is
// what the reference's strict `t < bestHit.distance` loop in id order keeps
// (SH:180, SH:245).  Sphere: SH:170-190.  Triangle: SH:192-230 + SH:243-246.
__device__ __forceinline__ void intersect_record(float4 a, float4 b, float4 c, uint32_t rec, f3 o, f3 d,
                                                 float &tbest, int32_t &idbest, uint32_t &recbest)
{
    int32_t id = as_i(a.w);
    float t;
    if (as_i(b.w) == 0) {
        f3 vert0 = xyz(a), edge1 = xyz(b), edge2 = xyz(c);
        f3 pvec = pt_cross3(d, edge2);
        float det = pt_dot3(edge1, pvec);
        if (det < PT_EPSILON) return;
        float inv_det = 1.0f / det;
        f3 tvec = pt_sub3(o, vert0);
        float u = pt_dot3(tvec, pvec) * inv_det;
        if (u < 0.0f || u > 1.0f) return;
        f3 qvec = pt_cross3(tvec, edge1);
        float v = pt_dot3(d, qvec) * inv_det;
        if (v < 0.0f || u + v > 1.0f) return;
        t = pt_dot3(edge2, qvec) * inv_det;
    } else {
        f3 dd = pt_sub3(o, xyz(a));
        float radius = b.x;
        float p1 = -pt_dot3(d, dd);
        float p2sqr = p1 * p1 - pt_dot3(dd, dd) + radius * radius;
        if (p2sqr < 0) return;
        float p2 = pt_sqrt(p2sqr);
        t = p1 - p2 > 0 ? p1 - p2 : p1 + p2;
    }
    if (t > 0 && (t < tbest || (t == tbest && id < idbest))) {
        tbest = t; idbest = id; recbest = rec;
    }
}

// All three words of a record are fetched at once (a sphere ignores the third), so that a test
// waits for one memory round trip, not two.
__device__ __forceinline__ void intersect_prim(const float4 *prims, uint32_t rec, f3 o, f3 d,
                                               float &tbest, int32_t &idbest, uint32_t &recbest)
{
    const float4 *p = prims + (size_t)rec * 3;
    intersect_record(p[0], p[1], p[2], rec, o, d, tbest, idbest, recbest);
}

// Reciprocal direction for the slab tests, with |d| clamped away from zero: an infinite reciprocal
// would turn lo*inv - o*inv into inf - inf.  Moving an axis-parallel ray by 1e-18 per unit length is
// far inside the padding of the boxes.
struct RaySlabs { float ix, iy, iz, ox, oy, oz; };
__device__ __forceinline__ RaySlabs ray_slabs(f3 o, f3 d)
{
    const float kTiny = 1e-18f;
    RaySlabs r;
    r.ix = 1.0f / (pt_abs(d.x) < kTiny ? __builtin_copysignf(kTiny, d.x) : d.x);
    r.iy = 1.0f / (pt_abs(d.y) < kTiny ? __builtin_copysignf(kTiny, d.y) : d.y);
    r.iz = 1.0f / (pt_abs(d.z) < kTiny ? __builtin_copysignf(kTiny, d.z) : d.z);
    r.ox = -o.x * r.ix; r.oy = -o.y * r.iy; r.oz = -o.z * r.iz;
    return r;
}

// One inner node: test the children, push the ones to visit later (farther ones deeper), return
// the nearest one entered - or the next stacked node, or kDoneNode.  The slab tests use a widened
// exit distance and non-strict comparisons: they may enter a box needlessly, never skip one.
template <class ST>
__device__ __forceinline__ int32_t node_step2(const SceneDev &sc, int32_t node, const RaySlabs &r, float tbest,
                                              const ST &st, int &sp)
{
    const float4 *q = sc.nodes + (size_t)node * 4;
    float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    float l0x = __builtin_fmaf(q0.x, r.ix, r.ox), l1x = __builtin_fmaf(q1.x, r.ix, r.ox);
    float l0y = __builtin_fmaf(q0.y, r.iy, r.oy), l1y = __builtin_fmaf(q1.y, r.iy, r.oy);
    float l0z = __builtin_fmaf(q0.z, r.iz, r.oz), l1z = __builtin_fmaf(q1.z, r.iz, r.oz);
    float r0x = __builtin_fmaf(q2.x, r.ix, r.ox), r1x = __builtin_fmaf(q3.x, r.ix, r.ox);
    float r0y = __builtin_fmaf(q2.y, r.iy, r.oy), r1y = __builtin_fmaf(q3.y, r.iy, r.oy);
    float r0z = __builtin_fmaf(q2.z, r.iz, r.oz), r1z = __builtin_fmaf(q3.z, r.iz, r.oz);
    float ln = fmaxf(fmaxf(fminf(l0x, l1x), fminf(l0y, l1y)), fmaxf(fminf(l0z, l1z), 0.0f));
    float lf = fminf(fminf(fmaxf(l0x, l1x), fmaxf(l0y, l1y)), fmaxf(l0z, l1z));
    float rn = fmaxf(fmaxf(fminf(r0x, r1x), fminf(r0y, r1y)), fmaxf(fminf(r0z, r1z), 0.0f));
    float rf = fminf(fminf(fmaxf(r0x, r1x), fmaxf(r0y, r1y)), fmaxf(r0z, r1z));
    bool hl = ln <= lf * 1.000002f && ln <= tbest;
    bool hr = rn <= rf * 1.000002f && rn <= tbest;
    int32_t cl = as_i(q0.w), cr = as_i(q1.w);
    if (hl && hr) {
        bool left_first = ln <= rn;
        stack_push(st, sp, left_first ? cr : cl);
        return left_first ? cl : cr;
    }
    if (hl) return cl;
    if (hr) return cr;
    return stack_next(st, sp, tbest);
}

__device__ __forceinline__ float ubyte_f(uint32_t v, int k) { return (float)((v >> (8 * k)) & 0xffu); }

typedef float float2v __attribute__((ext_vector_type(2)));
template <class ST>
__device__ __forceinline__ int32_t node_step4(const SceneDev &sc, int32_t node, const RaySlabs &r, float tbest,
                                              const ST &st, int &sp, uint32_t *empty = nullptr)
{
#ifdef PT_NODES_SOA
    // measurement build: the same nodes as four arrays of 16-byte columns (origin + exponents | links | low planes and
    // high x | high y, z), so that a visit touches four cache lines instead of one
    const uint4 *q = sc.nodes4 + node;
    const size_t col = (size_t)sc.n_nodes4;
    const uint4 q0 = q[0], q1 = q[col], q2 = q[2 * col];
    const uint2 q3 = *reinterpret_cast<const uint2 *>(q + 3 * col);
#elif defined(PT_NODES_3LOAD)
    // measurement build: 48-byte nodes without links (bvh.hpp, consecutive_layout):
    //   (org, ex | ey << 8 | ez << 16 | kinds << 24)  (first child, first record, qlo.x, qlo.y)  (qlo.z, qhi.x, qhi.y, qhi.z)
    const uint4 *q = sc.nodes4 + (size_t)node * 3;
    const uint4 q0 = q[0], qa = q[1], qb = q[2];
    uint4 q1;
    {
        uint32_t ci = qa.x, li = qa.y, lk[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t k = (q0.w >> (24 + 2 * c)) & 3u;
            lk[c] = k == 1u ? ci : ~((li << 3) | (k - 2u));
            ci += k == 1u ? 1u : 0u; li += k >= 2u ? k - 1u : 0u;
        }
        q1 = make_uint4(lk[0], lk[1], lk[2], lk[3]);
    }
    const uint4 q2 = make_uint4(qa.z, qa.w, qb.x, qb.y);
    const uint2 q3 = make_uint2(qb.z, qb.w);
#else
    const uint4 *q = sc.nodes4 + (size_t)node * 4;
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2];
    const uint2 q3 = *reinterpret_cast<const uint2 *>(q + 3);
#endif
    const uint32_t meta = q0.w;
    // plane position along the ray: ((org + k * cell) - o) * inv = k * (cell * inv) + (org - o) * inv
    // cell * inv: the reciprocal direction scaled by the cell's power of two (one byte-select-and-add, one ldexp per axis)
    const float sx = __builtin_ldexpf(r.ix, (int)(meta & 0xffu) - 127);
    const float sy = __builtin_ldexpf(r.iy, (int)((meta >> 8) & 0xffu) - 127);
    const float sz = __builtin_ldexpf(r.iz, (int)((meta >> 16) & 0xffu) - 127);
    const float bx = __builtin_fmaf(__uint_as_float(q0.x), r.ix, r.ox);
    const float by = __builtin_fmaf(__uint_as_float(q0.y), r.iy, r.oy);
    const float bz = __builtin_fmaf(__uint_as_float(q0.z), r.iz, r.oz);
    // per axis: the plane the ray meets first is the low one when it travels in +axis, else the high one
    const uint32_t nx4 = r.ix >= 0.0f ? q2.x : q2.w, fx4 = r.ix >= 0.0f ? q2.w : q2.x;
    const uint32_t ny4 = r.iy >= 0.0f ? q2.y : q3.x, fy4 = r.iy >= 0.0f ? q3.x : q2.y;
    const uint32_t nz4 = r.iz >= 0.0f ? q2.z : q3.y, fz4 = r.iz >= 0.0f ? q3.y : q2.z;
    uint32_t key[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
#ifdef PT_PACKED_SLABS
        // measurement build: the near and the far plane of an axis in one packed instruction (v_pk_fma_f32: two IEEE fmas, the
        // same bits); 12 vector instructions fewer per visit and 2-3 % slower on the 262 k-triangle scene
        const float2v px = __builtin_elementwise_fma(float2v{ubyte_f(nx4, c), ubyte_f(fx4, c)}, float2v{sx, sx}, float2v{bx, bx});
        const float2v py = __builtin_elementwise_fma(float2v{ubyte_f(ny4, c), ubyte_f(fy4, c)}, float2v{sy, sy}, float2v{by, by});
        const float2v pz = __builtin_elementwise_fma(float2v{ubyte_f(nz4, c), ubyte_f(fz4, c)}, float2v{sz, sz}, float2v{bz, bz});
        const float nx = px.x, fx = px.y, ny = py.x, fy = py.y, nz = pz.x, fz = pz.y;
#else
        float nx = __builtin_fmaf(ubyte_f(nx4, c), sx, bx), fx = __builtin_fmaf(ubyte_f(fx4, c), sx, bx);
        float ny = __builtin_fmaf(ubyte_f(ny4, c), sy, by), fy = __builtin_fmaf(ubyte_f(fy4, c), sy, by);
        float nz = __builtin_fmaf(ubyte_f(nz4, c), sz, bz), fz = __builtin_fmaf(ubyte_f(fz4, c), sz, bz);
#endif
        float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
        float tf = fminf(fminf(fx, fy), fz);
        bool hit = tn <= tf * 1.000002f && tn <= tbest;   // an unused slot holds an inside-out box: never entered
        key[c] = hit ? __float_as_uint(tn) : 0xffffffffu;   // entry distance (>= 0, so its bits order like an unsigned)
    }
    // sorting network over (distance, link) pairs, ascending: nearest child first, misses last.  Carrying the links through
    // the five exchanges (a compare and four selects each) is cheaper than sorting keys with the slot in their low bits and
    // picking each link out of the four afterwards (two tests and three selects per link): 17 vector instructions less.
    uint32_t lnk[4] = {q1.x, q1.y, q1.z, q1.w};
#define PT_CSWAP(i, j) { const bool s_ = key[j] < key[i]; const uint32_t ki_ = key[i], li_ = lnk[i]; \
                         key[i] = s_ ? key[j] : ki_; lnk[i] = s_ ? lnk[j] : li_; key[j] = s_ ? ki_ : key[j]; lnk[j] = s_ ? li_ : lnk[j]; }
    PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3) PT_CSWAP(1, 2)
#undef PT_CSWAP
    const int32_t c0 = (int32_t)lnk[0], c1 = (int32_t)lnk[1], c2 = (int32_t)lnk[2], c3 = (int32_t)lnk[3];
    // children to stack: keys 1 .. hits-1, farthest first so that the nearest of them is popped first
    const int np = key[3] != 0xffffffffu ? 3 : key[2] != 0xffffffffu ? 2 : key[1] != 0xffffffffu ? 1 : 0;
    if (sp + 3 <= ST::kLds) {
        // room for three in LDS: store three unconditionally (entries past np are never read), advance by np
        const int32_t e0 = np == 3 ? c3 : np == 2 ? c2 : c1;
        const int32_t e1 = np == 3 ? c2 : c1;
        st.lds[sp * ST::kStride] = (uint32_t)e0;
        st.lds[(sp + 1) * ST::kStride] = (uint32_t)e1;
        st.lds[(sp + 2) * ST::kStride] = (uint32_t)c1;
        if constexpr (ST::kWithT) {
            st.lds_t[sp * ST::kStride] = np == 3 ? key[3] : np == 2 ? key[2] : key[1];
            st.lds_t[(sp + 1) * ST::kStride] = np == 3 ? key[2] : key[1];
            st.lds_t[(sp + 2) * ST::kStride] = key[1];
        }
        sp += np;
    } else {
#ifdef PT_COUNT_OVERFLOW   // measurement build: the node statistic counts the stack entries that went to the global overflow area
        if (empty) (*empty) += (uint32_t)max(0, min(np, sp + np - ST::kLds));
#endif
        if (np == 3) stack_push(st, sp, c3, key[3]);
        if (np >= 2) stack_push(st, sp, c2, key[2]);
        if (np >= 1) stack_push(st, sp, c1, key[1]);
    }
#ifndef PT_COUNT_OVERFLOW
    if (empty && key[0] == 0xffffffffu) (*empty)++;   // -DPT_COUNT_EMPTY only
#endif
    if (key[0] == 0xffffffffu) return stack_next(st, sp, tbest);
    return c0;
}

// ---- the eight-wide tree (bvh.hpp, Bvh8Node) ------------------------------------------------------------------------
// A ray does not hold links but GROUPS: (base, mask) with mask = kind << 31 | member slots << 8 | hits, where `member
// slots` is the node's imask (kind 1: inner children, base = child_base) or lmask (kind 0: leaves of one record each,
// base = leaf_base) and `hits` says which of them the ray still has to visit - bit p standing for slot p ^ oct, so that
// the lowest set bit is the child nearest along the ray's octant.  One visit tests eight boxes, keeps the hit leaves as
// the current group (they are tested next), and leaves ONE stack entry at most behind: the inner children that were hit.
// Nothing is sorted: the order comes from the slots the builder gave the children.
struct Group { uint32_t base, mask; };
constexpr uint32_t kInnerGroup = 0x80000000u;

// Direction signs of the ray, bit a set = travelling towards -a.
__device__ __forceinline__ uint32_t ray_octant(const RaySlabs &r)
{
    return (__float_as_uint(r.ix) >> 31) | ((__float_as_uint(r.iy) >> 31) << 1) | ((__float_as_uint(r.iz) >> 31) << 2);
}
// The group a new ray starts with: the root (node 0) as the only child of a node that does not exist.
__device__ __forceinline__ Group root_group(uint32_t oct) { Group g = {0u, kInnerGroup | (1u << 8) | (1u << oct)}; return g; }

// Stack of groups: entry e of thread t at lds[e * LDS_STRIDE + t] (one 8-byte access per entry), deeper entries as two
// consecutive words of the global overflow area.
template <int LDS_ENTRIES, int LDS_STRIDE>
struct GroupStackT {
    uint2 *lds;
    uint32_t *overflow;
    uint32_t stride;
    static constexpr int kLds = LDS_ENTRIES, kStride = LDS_STRIDE;
    static constexpr int kCapacity = LDS_ENTRIES + kStackOverflow / 2;
};
template <class ST>
__device__ __forceinline__ void group_push(const ST &s, int &sp, Group g)
{
    if (sp < ST::kLds) s.lds[sp * ST::kStride] = make_uint2(g.base, g.mask);
    else if (sp < ST::kCapacity) {
        s.overflow[(size_t)(2 * (sp - ST::kLds)) * s.stride] = g.base;
        s.overflow[(size_t)(2 * (sp - ST::kLds) + 1) * s.stride] = g.mask;
    }
    sp++;
}
// The next group to work on, or mask 0 when the stack is empty (the ray is finished).
template <class ST>
__device__ __forceinline__ Group group_pop(const ST &s, int &sp)
{
    Group g = {0u, 0u};
    if (sp == 0) return g;
    sp--;
    if (sp < ST::kLds) { const uint2 v = s.lds[sp * ST::kStride]; g.base = v.x; g.mask = v.y; }
    else if (sp < ST::kCapacity) {
        g.base = s.overflow[(size_t)(2 * (sp - ST::kLds)) * s.stride];
        g.mask = s.overflow[(size_t)(2 * (sp - ST::kLds) + 1) * s.stride];
    }
    return g;
}
// Takes the nearest member out of the group: its slot and its index (node or record).
__device__ __forceinline__ uint32_t group_take(Group &g, uint32_t oct)
{
    const uint32_t p = (uint32_t)__builtin_ctz(g.mask);          // a group at hand always has a hit in its low byte
    g.mask &= g.mask - 1u;
    const uint32_t slot = p ^ oct;
    return g.base + (uint32_t)__popc((g.mask >> 8) & 0xffu & ((1u << slot) - 1u));
}
__device__ __forceinline__ bool group_empty(const Group &g) { return (g.mask & 0xffu) == 0u; }

// bit i -> bit i ^ oct in both bytes of a 16-bit word: three butterfly stages, each a shift by 0 for a clear bit of oct.
__device__ __forceinline__ uint32_t octant_order(uint32_t x, uint32_t oct)
{
    const uint32_t s1 = oct & 1u, s2 = oct & 2u, s4 = oct & 4u;
    x = ((x & 0x5555u) << s1) | ((x & 0xaaaau) >> s1);
    x = ((x & 0x3333u) << s2) | ((x & 0xccccu) >> s2);
    x = ((x & 0x0f0fu) << s4) | ((x & 0xf0f0u) >> s4);
    return x;
}

// One inner node of the eight-wide tree: `cur` is an inner group with a hit.  On return `cur` is what the ray works on
// next - the visited node's hit leaves, else its hit inner children, else the top of the stack - or has mask 0 (done).
template <class ST>
__device__ __forceinline__ void node_step8(const SceneDev &sc, Group &cur, const RaySlabs &r, float tbest, const ST &st, int &sp)
{
    const uint32_t oct = ray_octant(r);
    const uint32_t node = group_take(cur, oct);
    const uint4 *q = sc.nodes8 + (size_t)node * 5;
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    // what is left of the group the node came from: stored unconditionally, kept if it still has a hit
    const bool keep = !group_empty(cur);
    if (sp + 2 <= ST::kLds) { st.lds[sp * ST::kStride] = make_uint2(cur.base, cur.mask); sp += keep ? 1 : 0; }
    else if (keep) group_push(st, sp, cur);
    const uint32_t meta = q0.w;
    const float sx = __builtin_ldexpf(r.ix, (int)(meta & 0xffu) - 127);
    const float sy = __builtin_ldexpf(r.iy, (int)((meta >> 8) & 0xffu) - 127);
    const float sz = __builtin_ldexpf(r.iz, (int)((meta >> 16) & 0xffu) - 127);
    const float bx = __builtin_fmaf(__uint_as_float(q0.x), r.ix, r.ox);
    const float by = __builtin_fmaf(__uint_as_float(q0.y), r.iy, r.oy);
    const float bz = __builtin_fmaf(__uint_as_float(q0.z), r.iz, r.oz);
    // per axis the planes the ray meets first: the low ones when it travels in +axis, else the high ones (two words: slots 0..3, 4..7)
    const bool px = r.ix >= 0.0f, py = r.iy >= 0.0f, pz = r.iz >= 0.0f;
    const uint32_t nxw[2] = {px ? q2.x : q3.z, px ? q2.y : q3.w}, fxw[2] = {px ? q3.z : q2.x, px ? q3.w : q2.y};
    const uint32_t nyw[2] = {py ? q2.z : q4.x, py ? q2.w : q4.y}, fyw[2] = {py ? q4.x : q2.z, py ? q4.y : q2.w};
    const uint32_t nzw[2] = {pz ? q3.x : q4.z, pz ? q3.y : q4.w}, fzw[2] = {pz ? q4.z : q3.x, pz ? q4.w : q3.y};
    uint32_t hits = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int w = c >> 2, k = c & 3;
        const float nx = __builtin_fmaf(ubyte_f(nxw[w], k), sx, bx), fx = __builtin_fmaf(ubyte_f(fxw[w], k), sx, bx);
        const float ny = __builtin_fmaf(ubyte_f(nyw[w], k), sy, by), fy = __builtin_fmaf(ubyte_f(fyw[w], k), sy, by);
        const float nz = __builtin_fmaf(ubyte_f(nzw[w], k), sz, bz), fz = __builtin_fmaf(ubyte_f(fzw[w], k), sz, bz);
        const float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
        const float tf = fminf(fminf(fminf(fx, fy), fz) * 1.000002f, tbest);   // an absent child holds an inside-out box: never entered
        hits |= tn <= tf ? 1u << c : 0u;
    }
    const uint32_t imask = meta >> 24, lmask = q1.z & 0xffu;
    const uint32_t ordered = octant_order((hits & imask) | ((hits & lmask) << 8), oct);
    const uint32_t ih = ordered & 0xffu, lh = ordered >> 8;
    const Group inner = {q1.x, kInnerGroup | (imask << 8) | ih}, leaves = {q1.y, (lmask << 8) | lh};
    if (lh) {
        if (sp + 1 <= ST::kLds) { st.lds[sp * ST::kStride] = make_uint2(inner.base, inner.mask); sp += ih ? 1 : 0; }
        else if (ih) group_push(st, sp, inner);
        cur = leaves;
    } else if (ih) {
        cur = inner;
    } else {
        cur = group_pop(st, sp);
    }
}

// The records of a leaf group against the ray, two per round trip; then the next group (or mask 0).
template <class ST>
__device__ __forceinline__ void leaf_step8(const SceneDev &sc, Group &cur, f3 o, f3 d, uint32_t oct, float &tbest, int32_t &idbest,
                                           uint32_t &recbest, const ST &st, int &sp, uint32_t &n_prims)
{
    const uint32_t r0 = group_take(cur, oct);
    const bool two = !group_empty(cur);
    const uint32_t r1 = two ? group_take(cur, oct) : r0;
    const float4 *p = sc.prims + (size_t)r0 * 3, *q = sc.prims + (size_t)r1 * 3;
    const float4 a0 = p[0], b0 = p[1], c0 = p[2], a1 = q[0], b1 = q[1], c1 = q[2];
    intersect_record(a0, b0, c0, r0, o, d, tbest, idbest, recbest);
    if (two) intersect_record(a1, b1, c1, r1, o, d, tbest, idbest, recbest);
    n_prims += two ? 2u : 1u;
    if (group_empty(cur)) cur = group_pop(st, sp);
}

template <int WIDE, class ST>
__device__ __forceinline__ int32_t node_step(const SceneDev &sc, int32_t node, const RaySlabs &r, float tbest,
                                             const ST &st, int &sp)
{
    if constexpr (WIDE == 2) return node_step2(sc, node, r, tbest, st, sp);
    else return node_step4(sc, node, r, tbest, st, sp);
}

// One leaf: every primitive of it against the ray, then the next stacked node (or kDoneNode).
template <class ST>
__device__ __forceinline__ int32_t leaf_step(const SceneDev &sc, int32_t node, f3 o, f3 d, float &tbest, int32_t &idbest,
                                             uint32_t &recbest, const ST &st, int &sp, uint32_t &n_prims)
{
    uint32_t enc = (uint32_t)~node;
    uint32_t first = enc >> 3, count = (enc & 7u) + 1u;
    // two records per round trip: the second one's words are fetched before the first is tested
    // (an odd tail fetches the same record twice and tests it once)
    for (uint32_t k = 0; k < count; k += 2) {
        const bool two = k + 1 < count;
        const float4 *p = sc.prims + (size_t)(first + k) * 3;
        const float4 *q = two ? p + 3 : p;
        const float4 a0 = p[0], b0 = p[1], c0 = p[2], a1 = q[0], b1 = q[1], c1 = q[2];
        intersect_record(a0, b0, c0, first + k, o, d, tbest, idbest, recbest);
        if (two) intersect_record(a1, b1, c1, first + k + 1, o, d, tbest, idbest, recbest);
    }
    n_prims += count;
    return stack_next(st, sp, tbest);
}

// ---- two-level scenes: mesh objects that share a mesh keep one tree over the mesh's object-space triangles, and the
// top-level tree holds one box per such object ("instance").  The ray is taken into object space for the walk through
// the boxes only (conservative: the mesh's boxes are padded for the rounding of that transform); every triangle test
// runs in world space on vertices transformed as RayTracingShader.compute:238-240 transforms them, so the hit found
// is bit for bit the one of the flattened scene.

// row . (v, 1) in pt_mul_point's order of operations (include/ptrace_fmath.h)
__device__ __forceinline__ f3 rows_point(float4 r0, float4 r1, float4 r2, f3 v)
{
    f3 r;
    r.x = pt_fma(r0.z, v.z, pt_fma(r0.y, v.y, r0.x * v.x)) + r0.w;
    r.y = pt_fma(r1.z, v.z, pt_fma(r1.y, v.y, r1.x * v.x)) + r1.w;
    r.z = pt_fma(r2.z, v.z, pt_fma(r2.y, v.y, r2.x * v.x)) + r2.w;
    return r;
}
__device__ __forceinline__ f3 rows_dir(float4 r0, float4 r1, float4 r2, f3 v)
{
    f3 r;
    r.x = pt_fma(r0.z, v.z, pt_fma(r0.y, v.y, r0.x * v.x));
    r.y = pt_fma(r1.z, v.z, pt_fma(r1.y, v.y, r1.x * v.x));
    r.z = pt_fma(r2.z, v.z, pt_fma(r2.y, v.y, r2.x * v.x));
    return r;
}

// The slab constants of the ray as instance `inst` sees it.  The ray's parameter is the same in both spaces.
__device__ __forceinline__ RaySlabs instance_slabs(const SceneDev &sc, int inst, f3 o, f3 d)
{
    const float4 *q = sc.inst + (size_t)inst * 8 + 3;
    const float4 i0 = q[0], i1 = q[1], i2 = q[2];
    return ray_slabs(rows_point(i0, i1, i2, o), rows_dir(i0, i1, i2, d));
}

// The two links only two-level scenes have, resolved as soon as a step returns them, so that they never cost a phase of
// their own: an instance link (enter: the exit marker goes on the stack, the ray's slab constants become those of its
// object-space image, the walk continues at the mesh's root, an inner node) and the exit marker (back to world space,
// next stacked node).  `cur_inst` is the instance the ray is in (-1: top level), `r` its slab constants there.
__device__ __forceinline__ bool is_instance_link(int32_t node) { return node < (int32_t)0xc0000000 && node > kExitInstance; }
template <class ST>
__device__ __forceinline__ void resolve_instance_links(const SceneDev &sc, int32_t &node, f3 o, f3 d, int &cur_inst, RaySlabs &r,
                                                       const ST &st, int &sp)
{
    while (node == kExitInstance) {
        cur_inst = -1;
        r = ray_slabs(o, d);
        node = sp ? stack_pop(st, sp) : kDoneNode;
    }
    if (is_instance_link(node)) {
        cur_inst = (int)((uint32_t)~node & (kInstanceBit - 1u));
        r = instance_slabs(sc, cur_inst, o, d);
        stack_push(st, sp, kExitInstance);
        node = as_i(sc.inst[(size_t)cur_inst * 8 + 6].x);
    }
}

// leaf_step for two-level scenes.  At the top level a record of kind 2 stands for an instance whose box shares a leaf
// with other records: its link is stacked.  Inside an instance the records hold object-space vertices; the test runs on
// their world-space images.  For a hit inside an instance recbest = 0x80000000 | mesh object index.
template <class ST>
__device__ __forceinline__ int32_t leaf_step_inst(const SceneDev &sc, int32_t node, f3 o, f3 d, int &cur_inst, RaySlabs &r,
                                                  float &tbest, int32_t &idbest, uint32_t &recbest, const ST &st, int &sp, uint32_t &n_prims)
{
    const uint32_t enc = (uint32_t)~node;
    const uint32_t first = enc >> 3, count = (enc & 7u) + 1u;
    if (cur_inst < 0) {
        for (uint32_t k = 0; k < count; k++) {
            const float4 *p = sc.prims + (size_t)(first + k) * 3;
            const float4 a = p[0], b = p[1], c = p[2];
            if (as_i(b.w) == 2) stack_push(st, sp, (int32_t)~(kInstanceBit | (uint32_t)as_i(b.x)));
            else intersect_record(a, b, c, first + k, o, d, tbest, idbest, recbest);
        }
    } else {
        const float4 *q = sc.inst + (size_t)cur_inst * 8;
        const float4 m0 = q[0], m1 = q[1], m2 = q[2], meta = q[6];
        // as in leaf_step: two records per round trip, their words (and the matrix rows) all in flight together
        for (uint32_t k = 0; k < count; k += 2) {
            const bool two = k + 1 < count;
            const float4 *p = sc.prims + (size_t)(first + k) * 3;
            const float4 *p2 = two ? p + 3 : p;
            const float4 a0 = p[0], b0 = p[1], c0 = p[2], a1 = p2[0], b1 = p2[1], c1 = p2[2];   // object-space vertices; a.w = triangle index within the mesh
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (h == 1 && !two) break;
                const float4 a = h ? a1 : a0, b = h ? b1 : b0, c = h ? c1 : c0;
                const f3 v0 = rows_point(m0, m1, m2, xyz(a)), v1 = rows_point(m0, m1, m2, xyz(b)), v2 = rows_point(m0, m1, m2, xyz(c));   // SH:238-240
                const f3 e1 = pt_sub3(v1, v0), e2 = pt_sub3(v2, v0);
                intersect_record(make_float4(v0.x, v0.y, v0.z, as_f(as_i(meta.y) + as_i(a.w))), make_float4(e1.x, e1.y, e1.z, 0.0f),
                                 make_float4(e2.x, e2.y, e2.z, 0.0f), 0x80000000u | (uint32_t)as_i(meta.z), o, d, tbest, idbest, recbest);
            }
        }
    }
    n_prims += count;
    int32_t next = sp ? stack_pop(st, sp) : kDoneNode;
    resolve_instance_links(sc, next, o, d, cur_inst, r, st, sp);
    return next;
}

// World-space edges of triangle `id` (a hit inside an instance of mesh object `obj`), from the reference's own buffers
// (RayTracingShader.compute:236-240): what the flattened scene keeps in the primitive record.
__device__ __forceinline__ void instanced_edges(const SceneDev &sc, int32_t id, int obj, f3 &e1, f3 &e2)
{
    const int4 ob = sc.objects[obj];
    const int32_t *ix = sc.indices + ob.y + 3 * (id - 1 - sc.n_spheres - ob.x);
    const float *m = sc.matrices + 16 * (size_t)obj;
    float mm[16];
#pragma unroll
    for (int k = 0; k < 16; k++) mm[k] = m[k];
    const float *pa = sc.vertices + 3 * (size_t)ix[0], *pb = sc.vertices + 3 * (size_t)ix[1], *pc = sc.vertices + 3 * (size_t)ix[2];
    const f3 v0 = pt_mul_point(mm, pt_make3(pa[0], pa[1], pa[2]));
    const f3 v1 = pt_mul_point(mm, pt_make3(pb[0], pb[1], pb[2]));
    const f3 v2 = pt_mul_point(mm, pt_make3(pc[0], pc[1], pc[2]));
    e1 = pt_sub3(v1, v0); e2 = pt_sub3(v2, v0);
}

// The per-lane stack a tree of WIDE children per node is walked with, over kWords * kBlock words of LDS per block.
constexpr int kStack8Lds = 6;          // groups (8 bytes each) kept in LDS per lane: the 12 KB that kStackLds links take
template <int WIDE> struct StackFor {
    typedef Stack type;
    static constexpr int kWords = kStackLds;
    static __device__ __forceinline__ type make(uint32_t *lds_words, uint32_t *overflow, uint32_t stride)
    {
        type st = {lds_words + threadIdx.x, overflow, stride};
        return st;
    }
};
template <> struct StackFor<8> {
    typedef GroupStackT<kStack8Lds, kBlock> type;
    static constexpr int kWords = 2 * kStack8Lds;
    static __device__ __forceinline__ type make(uint32_t *lds_words, uint32_t *overflow, uint32_t stride)
    {
        type st = {reinterpret_cast<uint2 *>(lds_words) + threadIdx.x, overflow, stride};
        return st;
    }
};

// Closest hit over spheres and triangles through the BVH.  tbest/idbest come in
// holding the ground-plane result (or +inf / -1).
template <int WIDE, class ST>
__device__ __forceinline__ void traverse(const SceneDev &sc, f3 o, f3 d, float &tbest, int32_t &idbest,
                                         uint32_t &recbest, const ST &st, uint32_t &n_nodes, uint32_t &n_prims)
{
    if (sc.n_nodes == 0) return;
    RaySlabs r = ray_slabs(o, d);
    int sp = 0;
    if constexpr (WIDE == 8) {
        const uint32_t oct = ray_octant(r);
        Group cur = root_group(oct);
        while (cur.mask) {
            if (cur.mask & kInnerGroup) { node_step8(sc, cur, r, tbest, st, sp); n_nodes++; }
            else leaf_step8(sc, cur, o, d, oct, tbest, idbest, recbest, st, sp, n_prims);
        }
        return;
    } else {
    int32_t node = 0;
    if (sc.n_inst) {
        int cur_inst = -1;
        while (node != kDoneNode) {
            if (node >= 0) { node = node_step<WIDE>(sc, node, r, tbest, st, sp); n_nodes++; resolve_instance_links(sc, node, o, d, cur_inst, r, st, sp); }
            else node = leaf_step_inst(sc, node, o, d, cur_inst, r, tbest, idbest, recbest, st, sp, n_prims);
        }
        return;
    }
    while (node != kDoneNode) {
        if (node >= 0) { node = node_step<WIDE>(sc, node, r, tbest, st, sp); n_nodes++; }
        else node = leaf_step(sc, node, o, d, tbest, idbest, recbest, st, sp, n_prims);
    }
    }
}

// SH:154-168 (distance and id only; the material is filled in by shade()).
__device__ __forceinline__ void intersect_ground_plane(f3 o, f3 d, float &tbest, int32_t &idbest)
{
    float t = -o.y / d.y;
    if (t > 0 && t < tbest) { tbest = t; idbest = 0; }
}

// A ray that starts beyond far_reach meets every sphere directly, as the shader's loop does (SH:273-277), if it comes
// anywhere near them: from a distance D IntersectSphere's difference of squares accepts rays that pass within
// sqrt(r * r + 8 eps D * D) of a centre, about 7e-4 D beyond the sphere - too much to give every box of the tree.  The
// tree still holds the spheres; finding one twice changes nothing.
__device__ __forceinline__ void intersect_far_spheres(const SceneDev &sc, f3 o, f3 d, float &tbest, int32_t &idbest)
{
    if (sc.n_spheres == 0) return;
    const float dx = o.x - sc.centre[0], dy = o.y - sc.centre[1], dz = o.z - sc.centre[2];
    // |o - centre| + half_diag > far_reach, asked of the squares: every new ray of every shade phase comes through here,
    // and the root is only worth taking for the few that pass (where exactly the line lies does not matter: the tree's
    // boxes cover rays from anywhere near it as well)
    const float d2 = dx * dx + dy * dy + dz * dz, lim = sc.far_reach - sc.half_diag;
    if (!(lim < 0.0f || d2 > lim * lim)) return;
    const float dist = sqrtf(d2) + sc.half_diag;
    // the spheres' bounds, widened by what the spheres look wider by from there, against the ray (conservative slab test)
    const float m = 7.5e-4f * dist + 1e-3f * sc.half_diag;
    float tn = 0.0f, tf = PT_INF;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float inv = 1.0f / (pt_abs(dd[k]) < 1e-18f ? __builtin_copysignf(1e-18f, dd[k]) : dd[k]);
        const float a = (sc.spheres_lo[k] - m - oo[k]) * inv, b = (sc.spheres_hi[k] + m - oo[k]) * inv;
        tn = fmaxf(tn, fminf(a, b)); tf = fminf(tf, fmaxf(a, b));
    }
    if (!(tn <= tf * 1.0001f)) return;
    for (int i = 0; i < sc.n_spheres; i++) {
        const float4 c = sc.sphere_mat[(size_t)i * 4];      // (centre, radius)
        const f3 q = pt_sub3(o, xyz(c));                    // SH:170-190
        const float p1 = -pt_dot3(d, q);
        const float p2sqr = p1 * p1 - pt_dot3(q, q) + c.w * c.w;
        if (p2sqr < 0) continue;
        const float p2 = pt_sqrt(p2sqr);
        const float t = p1 - p2 > 0 ? p1 - p2 : p1 + p2;
        if (t > 0 && (t < tbest || (t == tbest && i + 1 < idbest))) { tbest = t; idbest = i + 1; }
    }
}

// ------------------------------------------------------------------- shading

// The arithmetic Shade() is written on.  Fm<false>: the intrinsics of include/ptrace_fmath.h - correctly rounded division
// and square root, software sin / cos / pow / acos / atan2 with fixed algorithms - which the oracle shares, so that
// images agree bit for bit.  Fm<true> (option "fast_intrinsics"): the hardware's approximate reciprocal, square root,
// reciprocal square root, exp2, log2 and sine (about 1 ulp each) and short polynomials for acos and atan2.  The random
// numbers do not go through it (rng_next keeps pt_sin): every roulette decision and every (u1, u2) pair is the same in both,
// only the directions and weights computed from them differ in their last bits.
template <bool FAST> struct Fm {
    static __device__ __forceinline__ float rcp(float x) { return 1.0f / x; }
    static __device__ __forceinline__ float div(float a, float b) { return a / b; }
    static __device__ __forceinline__ float sqrt(float x) { return pt_sqrt(x); }
    static __device__ __forceinline__ f3 normalize(f3 a) { return pt_normalize3(a); }
    static __device__ __forceinline__ float pow(float x, float y) { return pt_pow(x, y); }
    static __device__ __forceinline__ float sin(float x) { return pt_sin(x); }
    static __device__ __forceinline__ float cos(float x) { return pt_cos(x); }
    static __device__ __forceinline__ float acos(float x) { return pt_acos(x); }
    static __device__ __forceinline__ float atan2(float y, float x) { return pt_atan2(y, x); }
};
template <> struct Fm<true> {
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ f3 normalize(f3 a) { return pt_scale3(a, __builtin_amdgcn_rsqf(pt_dot3(a, a))); }
    // x >= 0: 2^(y log2 x); log2(0) = -inf gives 0 for y > 0 and 1 for y == 0 as pow does
    static __device__ __forceinline__ float pow(float x, float y) { return y == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
    // v_sin_f32 / v_cos_f32 take revolutions and are specified for |x| <= 256 of them; callers pass angles of a few turns
    static __device__ __forceinline__ float sin(float x) { return __builtin_amdgcn_sinf(x * 0.15915494f); }
    static __device__ __forceinline__ float cos(float x) { return __builtin_amdgcn_cosf(x * 0.15915494f); }
    // acos x = sqrt(1 - |x|) * p(|x|), mirrored for x < 0 (Abramowitz & Stegun 4.4.45: absolute error below 7e-5)
    static __device__ __forceinline__ float acos(float x)
    {
        const float a = pt_abs(x) < 1.0f ? pt_abs(x) : 1.0f;
        const float p = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-0.0187293f, a, 0.0742610f), a, -0.2121144f), a, 1.5707288f);
        const float r = __builtin_amdgcn_sqrtf(1.0f - a) * p;
        return x < 0.0f ? 3.14159265f - r : r;
    }
    // atan of the smaller over the larger magnitude by an odd polynomial (absolute error below 2e-5), then the octant
    static __device__ __forceinline__ float atan2(float y, float x)
    {
        const float ax = pt_abs(x), ay = pt_abs(y);
        const float hi = ax > ay ? ax : ay, lo = ax > ay ? ay : ax;
        const float z = hi > 0.0f ? lo * __builtin_amdgcn_rcpf(hi) : 0.0f, z2 = z * z;
        float r = z * __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-0.0117212f, z2, 0.05265332f), z2, -0.11643287f), z2, 0.19354346f) * z2 - 0.33262347f, z2, 0.99997726f);
        if (ay > ax) r = 1.57079633f - r;
        if (x < 0.0f) r = 3.14159265f - r;
        return y < 0.0f ? -r : r;
    }
};

struct Rng { float seed; float dotp; };  // _Seed and dot(_Pixel, float2(12.9898, 78.233))  SH:47-52

__device__ __forceinline__ float rng_next(Rng &r)  // SH:50-55
{
    float result = pt_frac(pt_sin(r.seed / 100.0f * r.dotp) * 43758.5453f);
    r.seed += 1.0f;
    return result;
}

// GetTangentSpace + mul(tangentSpaceDir, basis), SH:291-302, SH:313
template <bool FAST = false>
__device__ __forceinline__ f3 tangent_to_world(f3 normal, f3 ts);

// SH:304-314
template <bool FAST = false>
__device__ __forceinline__ f3 sample_hemisphere(Rng &rs, f3 normal, float alpha)
{
    typedef Fm<FAST> M;
    float cosTheta = M::pow(rng_next(rs), M::rcp(alpha + 1.0f));
    float sinTheta = M::sqrt(FAST ? pt_max(1.0f - cosTheta * cosTheta, 0.0f) : 1.0f - cosTheta * cosTheta);
    float phi = 2 * PT_PI * rng_next(rs);
    f3 ts = pt_make3(M::cos(phi) * sinTheta, M::sin(phi) * sinTheta, cosTheta);
    return tangent_to_world<FAST>(normal, ts);
}

template <bool FAST>
__device__ __forceinline__ f3 tangent_to_world(f3 normal, f3 ts)
{
    typedef Fm<FAST> M;
    f3 helper = pt_make3(1, 0, 0);
    if (pt_abs(normal.x) > 0.99f) helper = pt_make3(0, 0, 1);
    f3 tangent = M::normalize(pt_cross3(normal, helper));
    f3 binormal = M::normalize(pt_cross3(normal, tangent));
    f3 r;
    r.x = pt_fma(ts.z, normal.x, pt_fma(ts.y, binormal.x, ts.x * tangent.x));
    r.y = pt_fma(ts.z, normal.y, pt_fma(ts.y, binormal.y, ts.x * tangent.y));
    r.z = pt_fma(ts.z, normal.z, pt_fma(ts.y, binormal.z, ts.x * tangent.z));
    return r;
}

// SH:324-332: bilinear, Repeat wrap, texel centres at (i+0.5)/N, lerp(a,b,t) = a + t*(b-a).
template <bool FAST = false>
__device__ __forceinline__ f3 sky_sample(const SceneDev &sc, f3 dir)
{
    typedef Fm<FAST> M;
    if (!sc.sky) return pt_make3(0, 0, 0);
    float theta = FAST ? M::acos(dir.y) * (-1.0f / PT_PI) : pt_acos(dir.y) / -PT_PI;
    float phi = FAST ? M::atan2(dir.x, -dir.z) * (-0.5f / PT_PI) : pt_atan2(dir.x, -dir.z) / -PT_PI * 0.5f;
    int W = sc.sky_w, H = sc.sky_h;
    float x = phi * (float)W - 0.5f, y = theta * (float)H - 0.5f;
    float x0f = pt_floor(x), y0f = pt_floor(y);
    float fx = x - x0f, fy = y - y0f;
    int x0, y0;
    if (((W & (W - 1)) | (H & (H - 1))) == 0) {     // sizes that are powers of two (wave-uniform): the non-negative residue is a mask
        x0 = (int)x0f & (W - 1); y0 = (int)y0f & (H - 1);
    } else {
        x0 = (int)x0f % W; if (x0 < 0) x0 += W;
        y0 = (int)y0f % H; if (y0 < 0) y0 += H;
    }
    int x1 = x0 + 1 == W ? 0 : x0 + 1, y1 = y0 + 1 == H ? 0 : y0 + 1;
    float4 c00 = sc.sky[(size_t)y0 * W + x0], c10 = sc.sky[(size_t)y0 * W + x1];
    float4 c01 = sc.sky[(size_t)y1 * W + x0], c11 = sc.sky[(size_t)y1 * W + x1];
    f3 a = pt_make3(c00.x + fx * (c10.x - c00.x), c00.y + fx * (c10.y - c00.y), c00.z + fx * (c10.z - c00.z));
    f3 b = pt_make3(c01.x + fx * (c11.x - c01.x), c01.y + fx * (c11.y - c01.y), c01.z + fx * (c11.z - c01.z));
    f3 c = pt_make3(a.x + fy * (b.x - a.x), a.y + fy * (b.y - a.y), a.z + fy * (b.z - a.z));
    return pt_scale3(c, 2.0f);
}

// The GGX extension of the specular branch (PtMaterial.lobe == PT_LOBE_GGX; no counterpart in the shader).
// Draws cos(theta_h) first and phi second, as SampleHemisphere does, from the GGX normal distribution of
// roughness a = max((1 - smoothness)^2, 1e-3):  cos^2 = (1 - u1) / (1 + (a^2 - 1) u1);  the new direction is the
// mirror image of the incoming one about that half-vector.  Returns the throughput f * cos / pdf =
// F * G * (v.h) / ((n.v)(n.h)) per channel with Schlick's F (F0 = specular) and Smith's separable G, or zero
// when the sample leaves the upper hemisphere.
template <bool FAST = false>
__device__ __forceinline__ f3 sample_ggx(Rng &rs, f3 normal, f3 specular, float smoothness, f3 &d)
{
    typedef Fm<FAST> M;
    const float rough = 1.0f - smoothness;
    const float a = pt_max(rough * rough, 1e-3f), a2 = a * a;
    const float u1 = rng_next(rs);
    const float cos2 = M::div(1.0f - u1, 1.0f + (a2 - 1.0f) * u1);
    const float cosT = M::sqrt(cos2), sinT = M::sqrt(FAST ? pt_max(1.0f - cos2, 0.0f) : 1.0f - cos2);
    const float phi = 2 * PT_PI * rng_next(rs);
    const f3 h = tangent_to_world<FAST>(normal, pt_make3(M::cos(phi) * sinT, M::sin(phi) * sinT, cosT));
    const f3 v = pt_make3(-d.x, -d.y, -d.z);
    const float vh = pt_dot3(v, h);
    d = pt_reflect3(d, h);
    const float nl = pt_dot3(normal, d), nv = pt_dot3(normal, v), nh = pt_dot3(normal, h);
    if (!(nl > 0.0f && nv > 0.0f && vh > 0.0f && nh > 0.0f)) return pt_make3(0, 0, 0);
    const float m = 1.0f - vh, m2 = m * m, m5 = m2 * m2 * m;
    const float g1v = M::div(2.0f * nv, nv + M::sqrt(a2 + (1.0f - a2) * nv * nv));
    const float g1l = M::div(2.0f * nl, nl + M::sqrt(a2 + (1.0f - a2) * nl * nl));
    const float w = M::div(g1v * g1l * vh, nv * nh);
    return pt_make3((specular.x + (1.0f - specular.x) * m5) * w, (specular.y + (1.0f - specular.y) * m5) * w,
                    (specular.z + (1.0f - specular.z) * m5) * w);
}

// Shade (SH:334-373) for the hit (t, id, rec) together with the accumulation of SH:866/870:
// result += energy_before * Shade() / spp.  Updates origin/direction/energy in place.
// `hit_pos` receives bestHit.position ((0,0,0) on a miss, SH:140).  Values are consumed as early
// as the arithmetic allows (emission right after the material fetch, the material folded into one
// weight before the hemisphere sampling) to keep few registers live across the sampling code.
// MAT: mesh objects carry their own PtMaterial (SceneDev::mesh_mat), possibly with the GGX lobe.
template <bool MAT = false, bool INST = false, bool FAST = false>
__device__ __forceinline__ void shade(const SceneDev &sc, Rng &rs, f3 &o, f3 &d, f3 &energy, f3 &result, float fspp,
                                      float t, int32_t id, uint32_t rec, f3 &hit_pos)
{
    typedef Fm<FAST> M;
    int lobe = PT_LOBE_PHONG;
    // What Shade() returns - the hit's emission, or the sky after a miss (SH:371-372) - is added once for all cases
    // below: result += energy * value / spp (SH:866/870) is three correctly rounded divisions, and a wave with mixed
    // hits would otherwise run them once per case.  Likewise the hit normal: every case leaves a vector that one
    // normalize() turns into the shader's normal ((0,1,0) stays (0,1,0) exactly).
    const bool hit = t < PT_INF;
    f3 emitted, position = pt_make3(0, 0, 0), nraw = pt_make3(0, 1, 0), albedo = pt_make3(0, 0, 0), specular = pt_make3(0, 0, 0);
    float smoothness = 0.0f;
    if (!hit) {
        emitted = sky_sample<FAST>(sc, d);
    } else {
        position = pt_along3(o, t, d);
        if (id == 0) {                                  // SH:162-166
            albedo = pt_make3(0.5f, 0.5f, 0.5f);
            specular = pt_make3(0.03f, 0.03f, 0.03f);
            smoothness = 0.2f;
            emitted = pt_make3(0, 0, 0);
        } else if (id <= sc.n_spheres) {                // SH:184-188
            const float4 *m = sc.sphere_mat + (size_t)(id - 1) * 4;
            float4 m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3];
            nraw = pt_sub3(position, xyz(m0));
            albedo = xyz(m1); smoothness = m1.w;
            specular = xyz(m2);
            emitted = xyz(m3);
        } else {                                        // SH:249-253
            int obj;
            if (INST && (rec & 0x80000000u)) {
                obj = (int)(rec & 0x7fffffffu);
                f3 e1, e2;
                instanced_edges(sc, id, obj, e1, e2);
                nraw = pt_cross3(e1, e2);
            } else {
                float4 b = sc.prims[rec * 3 + 1], c = sc.prims[rec * 3 + 2];
                nraw = pt_cross3(xyz(b), xyz(c));
                obj = as_i(c.w);
            }
            if (MAT && sc.mesh_mat) {
                const float4 *m = sc.mesh_mat + (size_t)obj * 3;
                const float4 m0 = m[0], m1 = m[1], m2 = m[2];
                albedo = xyz(m0); smoothness = m0.w;
                specular = xyz(m1); lobe = as_i(m1.w);
                emitted = xyz(m2);
            } else {
                specular = pt_make3(0.65f, 0.65f, 0.65f);
                smoothness = 0.99f;
                emitted = pt_make3(0, 0, 0);
            }
        }
    }
    hit_pos = position;
    if (FAST || (__float_as_uint(fspp) & 0x007fffffu) == 0u) {
        // a power of two (wave-uniform): x / 2^k and x * 2^-k are the same correctly rounded number, and the product
        // costs one instruction where the division costs eleven
        const float inv = M::rcp(fspp);
        result.x += energy.x * emitted.x * inv;
        result.y += energy.y * emitted.y * inv;
        result.z += energy.z * emitted.z * inv;
    } else {
        result.x += energy.x * emitted.x / fspp;
        result.y += energy.y * emitted.y / fspp;
        result.z += energy.z * emitted.z / fspp;
    }
    if (!hit) {
        energy = pt_make3(0, 0, 0);
        return;
    }
    const f3 normal = M::normalize(nraw);
    albedo = pt_make3(pt_min(1.0f - specular.x, albedo.x), pt_min(1.0f - specular.y, albedo.y),
                      pt_min(1.0f - specular.z, albedo.z));
    const float third = 1.0f / 3.0f;
    const float specChance = pt_dot3(specular, pt_make3(third, third, third));
    const float diffChance = pt_dot3(albedo, pt_make3(third, third, third));
    const float roulette = rng_next(rs);
    // SH:345-365.  The specular and the diffuse branch both end in SampleHemisphere(); they are
    // merged here so that a wave runs that code once, each lane with its own axis and alpha.
    const bool spec = roulette < specChance;
    const bool diff = !spec && diffChance > 0 && roulette < specChance + diffChance;
    if (MAT && spec && lobe == PT_LOBE_GGX) {
        const float k = M::rcp(specChance);
        o = pt_along3(position, 0.001f, normal);
        const f3 w = sample_ggx<FAST>(rs, normal, specular, smoothness, d);
        energy.x *= k * w.x;
        energy.y *= k * w.y;
        energy.z *= k * w.z;
    } else if (spec || diff) {
        // (1/chance) * colour, the left factor of SH:352 / SH:359
        const float k = M::rcp(spec ? specChance : diffChance);
        const f3 weight = spec ? pt_make3(k * specular.x, k * specular.y, k * specular.z)
                               : pt_make3(k * albedo.x, k * albedo.y, k * albedo.z);
        o = pt_along3(position, 0.001f, normal);
        float alpha = 1.0f;
        f3 axis = normal;
        if (spec) {
            alpha = M::pow(1000.0f, smoothness * smoothness);
            axis = pt_reflect3(d, normal);
        }
        d = sample_hemisphere<FAST>(rs, axis, alpha);
        if (spec) {
            float f = M::div(alpha + 2, alpha + 1);
            float sd = pt_saturate(pt_dot3(normal, d) * f);
            energy.x *= weight.x * sd;
            energy.y *= weight.y * sd;
            energy.z *= weight.z * sd;
        } else {
            energy.x *= weight.x;
            energy.y *= weight.y;
            energy.z *= weight.z;
        }
    } else {
        energy = pt_make3(0, 0, 0);
    }
}

// SH:108-120 with uv = id.xy / (W,H) * 2 - 1 (SH:842)
// EXT: honour RenderArgs::cam_old - CreateCameraRayOld (SH:374-387) puts clip-space z = 1 where CreateCameraRay has 0,
// taken as ((M0*u + M1*v) + M2) + M3.
template <bool EXT = false>
__device__ __forceinline__ void camera_ray(const RenderArgs &a, int px, int py, f3 &o, f3 &d)
{
    float u = (float)px / (float)a.W * 2.0f - 1.0f;
    float v = (float)py / (float)a.H * 2.0f - 1.0f;
    o = pt_make3(a.c2w[12], a.c2w[13], a.c2w[14]);
    f3 dir;
    if (EXT && a.cam_old) {
        dir.x = (pt_fma(a.ip[4], v, a.ip[0] * u) + a.ip[8]) + a.ip[12];
        dir.y = (pt_fma(a.ip[5], v, a.ip[1] * u) + a.ip[9]) + a.ip[13];
        dir.z = (pt_fma(a.ip[6], v, a.ip[2] * u) + a.ip[10]) + a.ip[14];
    } else {
        dir.x = pt_fma(a.ip[4], v, a.ip[0] * u) + a.ip[12];
        dir.y = pt_fma(a.ip[5], v, a.ip[1] * u) + a.ip[13];
        dir.z = pt_fma(a.ip[6], v, a.ip[2] * u) + a.ip[14];
    }
    dir = pt_mul_dir(a.c2w, dir);
    d = pt_normalize3(dir);
}

// Work item i -> 8x8 pixel block i/64 of the tile rectangle (row-major blocks), pixel i%64 inside
// it, so 64 consecutive items are one block.  Returns false for items outside the rectangle.
// With a block order the item's block is looked up instead: the frame's work counters each hand out a contiguous range
// of items, and counter s takes the blocks of rank s, s + n, s + 2n, ... (n counters), so that every counter - and with
// them the frame as a whole - goes from the most expensive blocks to the cheapest: the pixels still in flight when the
// counters run dry are then cheap ones, which shortens the tail of a persistent dispatch.  The table holds a block as
// column | row << 16 (order_blocks_kernel), so the lookup needs no division.
__device__ __forceinline__ bool block_pixel(const RenderArgs &a, uint32_t bx, uint32_t by, uint32_t in, int &x, int &y)
{
    x = a.tx0 + (int)bx * 8 + (int)(in & 7u);
    y = a.ty0 + (int)by * 8 + (int)(in >> 3);
    return x < a.tx0 + a.tw && y < a.ty0 + a.th;
}
// Row-major block number -> column and row.  The host sets the multiplier when it is exact for every block of the
// rectangle (blocks * tiles_x < 2^32); a 32-bit division costs some thirty instructions, and at one sample per pixel
// every shade phase claims pixels.
__device__ __forceinline__ bool numbered_pixel(const RenderArgs &a, uint32_t blk, uint32_t in, int &x, int &y)
{
    uint32_t bx, by;
    if (a.tiles_x_magic) { by = __umulhi(blk, a.tiles_x_magic); bx = blk - by * (uint32_t)a.tiles_x; }
    else { by = blk / (uint32_t)a.tiles_x; bx = blk % (uint32_t)a.tiles_x; }
    return block_pixel(a, bx, by, in, x, y);
}
__device__ __forceinline__ bool item_pixel(const RenderArgs &a, uint32_t item, int &x, int &y)
{
    uint32_t blk = item >> 6, in = item & 63u;
    if (a.block_order) {
        const uint32_t shard_blocks = a.shard_items >> 6;
        const uint32_t rank = (blk % shard_blocks) * (uint32_t)a.n_shards + blk / shard_blocks;
        if (rank >= a.n_blocks) { x = y = 0; return false; }
        const uint32_t packed = a.block_order[rank];
        return block_pixel(a, packed & 0xffffu, packed >> 16, in, x, y);
    }
    return numbered_pixel(a, blk, in, x, y);
}
// The same for an item a wave has just claimed from work counter `shard`: `pos` is the item's place in that counter's
// range, so the counter and the place need not be divided out of the item number again.
__device__ __forceinline__ bool claimed_pixel(const RenderArgs &a, uint32_t shard, uint32_t pos, int &x, int &y)
{
    if (a.block_order) {
        const uint32_t rank = (pos >> 6) * (uint32_t)a.n_shards + shard;
        if (rank >= a.n_blocks) { x = y = 0; return false; }
        const uint32_t packed = a.block_order[rank];
        return block_pixel(a, packed & 0xffffu, packed >> 16, pos & 63u, x, y);
    }
    return numbered_pixel(a, shard * (a.shard_items >> 6) + (pos >> 6), pos & 63u, x, y);
}

// What a pixel's first path cost (node visits + primitive tests, or 16 per ray where those are not at hand) goes to its
// block's cost counter, which keeps the largest.
constexpr int kFirstSample = 1 << 19;
__device__ __forceinline__ void note_cost(const RenderArgs &a, int pix_x, int pix_y, int work)
{
    // the block's most expensive pixel, not the sum of its pixels: what has to start early is whatever holds a long pixel
    // (measured against the sum: +1 % one dispatch at a time and three in flight)
    if (a.block_cost)
        atomicMax(a.block_cost + (uint32_t)((pix_y - a.ty0) >> 3) * (uint32_t)a.tiles_x + (uint32_t)((pix_x - a.tx0) >> 3), (uint32_t)work);
}

// dot(_Pixel, float2(12.9898, 78.233)) with _Pixel = id.xy * 2 (SH:831, SH:52)
__device__ __forceinline__ float pixel_dot(int x, int y)
{
    return pt_fma((float)y * 2.0f, 78.233f, ((float)x * 2.0f) * 12.9898f);
}

// ------------------------------------------------- renderMode 2: reprojection
//
// HLSL semantics this code pins down (the shader leaves them to the compiler / the API):
//   - mul(float4x4, float3) (SH:404) uses the upper-left 3x3 block;
//   - `pwidth / pheight` (SH:402) is an unsigned integer division;
//   - a float2 used as a texture index (SH:450-453, SH:968) is converted like HLSL's uint cast:
//     truncation, negative values and NaN to 0; a load outside the texture returns 0;
//   - `float s = 1.0 / textureSize` (SH:445) keeps the x component;
//   - a * b + c is a multiplication followed by an addition.

struct f2 { float x, y; };

// FindCameraRay(input, 0), SH:394-411: where the direction `dir` (from the old camera's origin) lands in
// the old camera's image, in pixels (+0.5).
__device__ __forceinline__ f2 find_camera_ray(const RenderArgs &a, f3 dir)
{
    const float m11 = a.old_proj[5], m22 = a.old_proj[10], m23 = a.old_proj[14];   // _mRC = [C * 4 + R]
    const float near = -m23 / m22;
    const float far = -(near * m22) / (1.0f - m22);
    const f3 pos = pt_scale3(pt_normalize3(dir), near + (far - near));
    const float ratio = (float)((uint32_t)a.W / (uint32_t)a.H);
    const f3 toCam = pt_mul_dir(a.old_w2c, pos);
    const float height = toCam.z / m11;
    const float width = ratio * height;
    const float dimx = (float)a.W, dimy = (float)a.H;
    const float uvx = (toCam.x - width * (dimx / dimy)) / width;
    const float uvy = (toCam.y - height) / height;
    f2 r;
    r.x = 0.5f - uvx * (dimy * ratio / 2.0f);
    r.y = (0.5f - uvy * (dimy / 2.0f)) - 0.005f;
    return r;
}

__device__ __forceinline__ uint32_t hlsl_ftou(float v) { return v > 0.0f ? (v >= 4294967296.0f ? 0xffffffffu : (uint32_t)v) : 0u; }

__device__ __forceinline__ float4 history_load(const RenderArgs &a, float x, float y)
{
    const uint32_t ix = hlsl_ftou(x), iy = hlsl_ftou(y);
    if (ix >= (uint32_t)a.W || iy >= (uint32_t)a.H) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return a.history[(size_t)iy * a.W + ix];
}

__device__ __forceinline__ float lerp1(float x, float y, float t) { return x + t * (y - x); }

// BilinearTextureSample(P), SH:440-463, on Result1.
__device__ __forceinline__ float4 bilinear_history(const RenderArgs &a, f2 P)
{
    const float tsx = (float)a.W, tsy = (float)a.H;
    const float s = 1.0f / tsx;
    P.x -= 0.5f; P.y -= 0.5f;
    float px = P.x * tsx + 0.5f, py = P.y * tsy + 0.5f;
    const float fx = pt_frac(P.x), fy = pt_frac(P.y);
    px = pt_floor(px) / tsx - s / 2.0f;
    py = pt_floor(py) / tsy - s / 2.0f;
    const float4 C11 = history_load(a, px + 0.0f, py + 0.0f), C21 = history_load(a, px + 1.0f, py + 0.0f);
    const float4 C12 = history_load(a, px + 0.0f, py + 1.0f), C22 = history_load(a, px + 1.0f, py + 1.0f);
    const float4 x1 = make_float4(lerp1(C11.x, C21.x, fx), lerp1(C11.y, C21.y, fx), lerp1(C11.z, C21.z, fx), lerp1(C11.w, C21.w, fx));
    const float4 x2 = make_float4(lerp1(C12.x, C22.x, fx), lerp1(C12.y, C22.y, fx), lerp1(C12.z, C22.z, fx), lerp1(C12.w, C22.w, fx));
    return make_float4(lerp1(x1.x, x2.x, fy), lerp1(x1.y, x2.y, fy), lerp1(x1.z, x2.z, fy), lerp1(x1.w, x2.w, fy));
}

// SH:953-982 / SH:888-925: the value renderMode 2 writes to Result, renderMode 0 to Result2, for a finished pixel.
__device__ __forceinline__ float4 reproject_pixel(const RenderArgs &a, f3 result, f3 hit_pos, bool sky, float alpha_dist, int pix_y)
{
    const f3 origin_old = pt_make3(a.old_c2w[12], a.old_c2w[13], a.old_c2w[14]);
    const f3 to_hit = pt_sub3(hit_pos, origin_old);
    const f2 uv = find_camera_ray(a, to_hit);
    const float divisor = a.accumulation_frames;
    if (uv.x < (float)a.W && uv.x > 0.0f && uv.y < (float)a.H && uv.y > 0.0f) {
        const float4 bl = bilinear_history(a, uv);
        const float dist_old = pt_length3(to_hit);
        const float direct_w = history_load(a, uv.x, uv.y).w;
        bool take = (pt_abs(dist_old - bl.w) < dist_old * 0.05f || direct_w > 80.0f) && !sky;
        // SH:908-915 (renderMode 0): history only from rows above the end of the pixel's 32-row thread group
        if (a.render_mode == 0 && !(uv.y < pt_floor((float)pix_y / 32.0f) * 32.0f + 32.0f)) take = false;
        if (take)
            return make_float4(bl.x * (divisor - 1.0f) / divisor + result.x / divisor,
                               bl.y * (divisor - 1.0f) / divisor + result.y / divisor,
                               bl.z * (divisor - 1.0f) / divisor + result.z / divisor, dist_old);
    }
    return make_float4(result.x, result.y, result.z, alpha_dist);
}

// FindCameraRayVR(input), SH:418-437: FindCameraRay with the headset's aspect ratio and per-eye offsets hard-coded.
__device__ __forceinline__ f2 find_camera_ray_vr(const RenderArgs &a, f3 dir)
{
    const float m11 = a.old_proj[5], m22 = a.old_proj[10], m23 = a.old_proj[14];
    const float near = -m23 / m22;
    const float far = -(near * m22) / (1.0f - m22);
    const f3 pos = pt_scale3(pt_normalize3(dir), near + (far - near));
    const float ratio = 2240.0f / 2016.0f;
    const f3 toCam = pt_mul_dir(a.old_w2c, pos);
    const float height = toCam.z / m11;
    const float width = ratio * height;
    const float uvx = (toCam.x - width * (2240.0f / 2016.0f)) / width;
    const float uvy = (toCam.y - height) / height;
    const float ph = (float)(uint32_t)a.H;
    f2 r;
    r.x = (1.0f - uvx * (ph * ratio / 2.043f)) - (a.left_eye ? 156.0f : 538.0f);
    r.y = (1.0f - uvy * (ph / 2.0f)) - (a.left_eye ? -1.3f : -0.6f);
    return r;
}

// What renderMode 3 (SH:986-1019), 4 (SH:1021-1046) and 10 (SH:1048-1069) look up for a pixel's primary hit.
struct HistoryLookup { bool on_screen; float4 bilinear; float4 texel; float dist_old; f2 uv; };
__device__ __forceinline__ HistoryLookup history_lookup(const RenderArgs &a, f3 hit_pos)
{
    HistoryLookup h;
    const f3 origin_old = pt_make3(a.old_c2w[12], a.old_c2w[13], a.old_c2w[14]);
    const f3 to_hit = pt_sub3(hit_pos, origin_old);
    h.dist_old = pt_length3(to_hit);
    h.uv = a.render_mode == 4 ? find_camera_ray_vr(a, to_hit) : find_camera_ray(a, to_hit);
    h.on_screen = h.uv.x < (float)a.W && h.uv.x > 0.0f && h.uv.y < (float)a.H && h.uv.y > 0.0f;
    h.bilinear = h.texel = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (h.on_screen) { h.bilinear = bilinear_history(a, h.uv); h.texel = history_load(a, h.uv.x, h.uv.y); }
    return h;
}

// renderMode 10: does the pixel show its history instead of a traced path (SH:1053-1055)?
__device__ __forceinline__ bool history_covers(const HistoryLookup &h, bool sky)
{
    return h.on_screen && (pt_abs(h.dist_old - h.bilinear.w) < h.dist_old * 0.05f || h.bilinear.w > 80.0f) && !sky;
}

// The value renderMode 3 or 4 writes to Result for a finished pixel.
__device__ __forceinline__ float4 reproject_pixel_ext(const RenderArgs &a, f3 result, f3 hit_pos, bool sky, float alpha_dist)
{
    const HistoryLookup h = history_lookup(a, hit_pos);
    if (a.render_mode == 3) {
        const bool take = h.on_screen && (pt_abs(h.dist_old - h.texel.w) < h.dist_old * 0.05f || h.texel.w > 80.0f) && !sky;   // SH:995-997
        if (!take) return make_float4(result.x, result.y, result.z, alpha_dist);                                          // SH:1011
        const float divisor = 10.0f;
        const float4 n0 = history_load(a, h.uv.x + 1.0f, h.uv.y + 1.0f), n1 = history_load(a, h.uv.x - 1.0f, h.uv.y - 1.0f);   // SH:1000-1001
        const float4 n2 = history_load(a, h.uv.x + 1.0f, h.uv.y + -1.0f), n3 = history_load(a, h.uv.x - 1.0f, h.uv.y - -1.0f); // SH:1002-1003
        float4 out;
        out.x = result.x / 2.5f; out.x += h.bilinear.x / 5.0f; out.x += n0.x / divisor; out.x += n1.x / divisor; out.x += n2.x / divisor; out.x += n3.x / divisor;
        out.y = result.y / 2.5f; out.y += h.bilinear.y / 5.0f; out.y += n0.y / divisor; out.y += n1.y / divisor; out.y += n2.y / divisor; out.y += n3.y / divisor;
        out.z = result.z / 2.5f; out.z += h.bilinear.z / 5.0f; out.z += n0.z / divisor; out.z += n1.z / divisor; out.z += n2.z / divisor; out.z += n3.z / divisor;
        out.w = alpha_dist;                                                                                               // SH:1005
        return out;
    }
    // renderMode 4: the distance test reads the unfiltered texel, the far-away test the filtered one (SH:1033-1035)
    const bool take = h.on_screen && (pt_abs(h.dist_old - h.texel.w) < h.dist_old * 0.05f || h.bilinear.w > 80.0f) && !sky;
    if (!take) return make_float4(result.x, result.y, result.z, alpha_dist);                                              // SH:1043
    const float divisor = a.accumulation_frames;
    return make_float4(h.bilinear.x * (divisor - 1.0f) / divisor + result.x / divisor,                                    // SH:1037
                       h.bilinear.y * (divisor - 1.0f) / divisor + result.y / divisor,
                       h.bilinear.z * (divisor - 1.0f) / divisor + result.z / divisor, h.dist_old);
}

// Converged-image update fused into the final write: AddMat.shader:44-63 with
// its SrcAlpha/OneMinusSrcAlpha blend (:11), or a running mean, or a sum.
__device__ __forceinline__ void accumulate_pixel(const RenderArgs &a, size_t p, float4 fresh)
{
    if (a.accumulate == PT_ACCUM_NONE || !a.converged) return;
    float S = (float)a.sample_index;
    float4 h = a.converged[p];
    float4 out;
    if (a.accumulate == PT_ACCUM_MEAN) {
        float w = 1.0f / (S + 1.0f);
        out.x = fresh.x * w + h.x * (1.0f - w);
        out.y = fresh.y * w + h.y * (1.0f - w);
        out.z = fresh.z * w + h.z * (1.0f - w);
        out.w = fresh.w * w + h.w * (1.0f - w);
    } else if (a.accumulate == PT_ACCUM_SUM) {
        if (a.sample_index == 0) out = fresh;
        else { out.x = h.x + fresh.x; out.y = h.y + fresh.y; out.z = h.z + fresh.z; out.w = h.w + fresh.w; }
    } else {
        float alpha = h.w;
        float firefly = 0.8f;
        if (h.x > firefly || h.y > firefly || h.z > firefly) alpha = 0.1f;
        float t = 1.0f + S * alpha;
        float A = 1.0f - 1.0f / t;
        out.x = h.x * A + fresh.x * (1.0f - A);
        out.y = h.y * A + fresh.y * (1.0f - A);
        out.z = h.z * A + fresh.z * (1.0f - A);
        out.w = A * A + fresh.w * (1.0f - A);
    }
    a.converged[p] = out;
}

// Counter totals go to one of kStatsShards copies (picked by block id) so that thousands of waves
// finishing together do not queue on one address; the host adds the copies up.
constexpr int kStatsShards = 64;
constexpr int kStatsStride = 16;  // u64 words per shard: rays, nodes, prims, paths; with PT_PROFILE_PHASES also
                                  // [4..9] = rounds and active lanes of the shade, node and leaf phases, [10..12] = their wave clocks
__device__ __forceinline__ void stats_add(unsigned long long *stats, int which, uint32_t v, int lane)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0 && v && stats) atomicAdd(&stats[(blockIdx.x % kStatsShards) * kStatsStride + which], (unsigned long long)v);
}

// --------------------------------------------------------------- path kernel
//
// Persistent kernel: the grid is sized to fill the chip once; every lane owns one pixel at a time
// and pulls the next one from a global counter as soon as its pixel is finished.  A pixel's samples
// are taken in order by one lane because the shader's RNG counter runs on across samples (SH:839:
// _Seed is never reset).  Work item i -> 8x8 pixel block i/64 of the tile rectangle (row-major
// blocks), pixel i%64 inside it, so the 64 items a fresh wave draws are one block.
//
// After a primary miss the shader traces the unchanged ray once more with zero energy (SH:867-870);
// that re-trace draws no random number and adds 0 * sky, so it is skipped.  Every other zero-energy
// case is traced, because its Shade() advances the RNG counter later samples of the pixel start from.
//
// The wave does not walk "trace everyone, then shade everyone".  Every lane is in one of four states -
// holding an inner BVH node, holding a leaf, holding a finished ray that waits for Shade(),
// or idle - and each round the wave runs the ONE phase that most lanes can take part in:
//
//   shade phase   when at least `vote_shade` lanes wait (or nothing else is runnable):
//                 Shade() for finished rays, new pixels for idle lanes, ray set-up
//   node phase    when at least `vote_node` lanes hold an inner node (or no lane holds a leaf)
//   leaf phase    otherwise
//
// A lane whose ray ends early therefore waits only until enough other lanes have ended too,
// not until the longest ray of the wave is done, and the expensive shading code runs with
// many lanes active.  Per-pixel arithmetic and its order are unchanged.
// Lane states that are not a BVH node reference (all negative, none a valid leaf code: a leaf's
// first-primitive field never reaches 2^28 - 1).
constexpr int kIdleNode = (int)0x80000001;   // no pixel yet / pixel finished: wants a new one
constexpr int kDeadNode = (int)0x80000002;   // no pixel and none left
__device__ __forceinline__ bool is_leaf_ref(int32_t node) { return node < 0 && node > kDeadNode; }

// Per-lane path state lives in LDS between shade phases ([slot][thread] float4, conflict-free); only what
// traversal needs stays in registers.  That keeps the register count of the traversal phases, and with it the
// kernel's occupancy, down, and costs no memory traffic.
//   ls0 = (energy.xyz, _Seed)   ls1 = (result.xyz, t of the primary hit)
//   ls2 = (pixel x | y << 16, sample | bounce << 20, primary record, primary id)   ls3 = (rays, nodes, prims, paths) counters
//   (sample: below 2^19; in a pixel's first sample kFirstSample | cost of the pixel so far)
template <int WIDE, bool MAT, bool INST = false, bool FAST = false>
__device__ __forceinline__ void path_sched_body(const RenderArgs &a)
{
#if PT_STACK_T
    // measurement build: half as many stack entries in LDS, each with its entry distance (same LDS footprint)
    constexpr bool kCull = WIDE == 4 && !INST;
    constexpr int kLdsEntries = kCull ? PT_STACK_T : kStackLds;
    typedef StackT<kLdsEntries, kBlock, kCull> PathStack;
    __shared__ uint32_t lds_stack[kLdsEntries * kBlock];
    __shared__ uint32_t lds_stack_t[kCull ? kLdsEntries * kBlock : 1];
#else
    typedef typename StackFor<WIDE>::type PathStack;
    __shared__ __align__(8) uint32_t lds_stack[StackFor<WIDE>::kWords * kBlock];
#endif
    __shared__ float lds_save[3 * kBlock];   // node, sp, ray_work parked here during a shade phase (WIDE == 8: sp in the spare bits 16..23 of the group mask)
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x, n_threads = gridDim.x * kBlock;
#if PT_STACK_T
    const PathStack st = {lds_stack + threadIdx.x, a.stack_overflow + gtid, n_threads, lds_stack_t + threadIdx.x};
#else
    const PathStack st = StackFor<WIDE>::make(lds_stack, a.stack_overflow + gtid, n_threads);
#endif
    const int lane = threadIdx.x & 63;
    const SceneDev &sc = a.sc;
    __shared__ float4 lds_state[4 * kBlock];   // ls0 .. ls3 of every thread, [slot][thread]
    float4 *const lsb = lds_state + threadIdx.x;
#define ls0 (lsb)
#define ls1 (lsb + kBlock)
#define ls2 (reinterpret_cast<int4 *>(lsb + 2 * kBlock))
#define ls3 (reinterpret_cast<int4 *>(lsb + 3 * kBlock))

    f3 o = pt_make3(0, 0, 0), d = pt_make3(0, 0, 1);
    RaySlabs rsl = {0, 0, 0, 0, 0, 0};
    float tbest = PT_INF; int32_t idbest = -1; uint32_t recbest = 0;
    int32_t node = kIdleNode; int sp = 0;
    // WIDE == 8: the lane works on a group of children (node_step8): `node` is its base (>= 0) and `gmask` its mask - an
    // inner group while bit 31 is set, else leaf records; the lane states (all negative) stay in `node`
    uint32_t gmask = 0;
    auto holds_inner = [&]() { if constexpr (WIDE == 8) return node >= 0 && (int32_t)gmask < 0; else return node >= 0; };
    auto holds_leaf = [&]() { if constexpr (WIDE == 8) return node >= 0 && (int32_t)gmask >= 0; else return is_leaf_ref(node); };
    int cur_inst = -1;                    // INST: the instance the ray is walking through, -1 at the top level
    uint32_t ray_work = 0;                // nodes | prims << 16 of the current ray
    int shard = (int)(blockIdx.x % (uint32_t)a.n_shards);   // wave-uniform: which part of the frame this wave draws pixels from
    bool frame_drained = false;
    *ls3 = make_int4(0, 0, 0, 0);
#ifdef PT_PROFILE_PHASES
    uint32_t ph_rounds[3] = {0, 0, 0}, ph_lanes[3] = {0, 0, 0};   // wave-uniform
    unsigned long long ph_cyc[3] = {0, 0, 0}, ph_t0 = clock64(); int ph_cur = -1;
    const unsigned long long ph_wall0 = wall_clock64();
#define PH_SWITCH(k) { unsigned long long t_ = clock64(); if (ph_cur >= 0) ph_cyc[ph_cur] += t_ - ph_t0; ph_t0 = t_; ph_cur = (k); }
#endif

    for (;;) {
        const bool inner = holds_inner();
        const bool leaf = holds_leaf();
        const bool waiting = node == kDoneNode || node == kIdleNode;
        const int n_inner = __popcll(__ballot(inner)), n_leaf = __popcll(__ballot(leaf)), n_wait = __popcll(__ballot(waiting));
        if (n_inner + n_leaf + n_wait == 0) break;
        // The votes are fractions of the lanes that still have work: once the frame is drained and lanes retire, a wave of ten
        // live lanes must not wait for 44 of them to finish a ray - it would only shade when all ten have, and every bounce
        // would take as long as the slowest of ten rays.
        const int n_live = n_inner + n_leaf + n_wait;
        const int vote_shade = n_live == 64 ? a.vote_shade : max(1, (a.vote_shade * n_live) >> 6);
        const int vote_node = n_live == 64 ? a.vote_node : max(1, (a.vote_node * n_live) >> 6);

        if (n_wait >= vote_shade || n_inner + n_leaf == 0) {
            // ================= shade phase =================
#ifdef PT_PROFILE_PHASES
            ph_rounds[0]++; ph_lanes[0] += n_wait; PH_SWITCH(0)
#endif
            // Lanes that are still traversing sit this phase out; what only traversal needs is parked in
            // LDS so that the shading code has those registers (the compiler cannot know they are idle).
            {
                float *sv = lds_save + threadIdx.x;
                sv[0 * kBlock] = as_f(node); sv[2 * kBlock] = as_f((int32_t)ray_work);
                if constexpr (WIDE == 8) sv[1 * kBlock] = as_f((int32_t)(gmask | ((uint32_t)sp << 16)));
                else sv[1 * kBlock] = as_f(INST ? sp | ((cur_inst + 1) << 8) : sp);
            }
            const bool was_done = node == kDoneNode, was_idle = node == kIdleNode;
            bool new_ray = false, finished = false, no_more = false;
            int cam_x = -1, cam_y = 0;             // >= 0: the new ray is this pixel's camera ray (next sample of a pixel, or a new pixel)
            bool cached = false;                   // new_ray whose hit is already known (the pixel's primary hit)
            float cached_t = PT_INF; int32_t cached_id = -1; uint32_t cached_rec = 0;
            int4 cnt = make_int4(0, 0, 0, 0);
            if (waiting) cnt = *ls3;
            if (was_done) {
                const int4 px = *ls2;
                const int pix_x = px.x & 0xffff, pix_y = px.x >> 16;
                // during a pixel's first sample the sample field holds kFirstSample | the node visits and primitive tests of the
                // pixel so far: what the pixel costs, for the next dispatch's block order
                const bool first_sample = (px.y & kFirstSample) != 0;
                int sample = first_sample ? 0 : (px.y & (kFirstSample - 1)), bounce = (int)((uint32_t)px.y >> 20), primary_id = px.w;
                uint32_t pix_work = first_sample ? (uint32_t)(px.y & (kFirstSample - 1)) : 0u;
                uint32_t primary_rec = (uint32_t)px.z;
                {
                    const uint32_t rw = (uint32_t)as_i(lds_save[threadIdx.x + 2 * kBlock]);
                    cnt.y += (int)(rw & 0xffffu); cnt.z += (int)(rw >> 16);
                    pix_work = min(pix_work + (rw & 0xffffu) + (rw >> 16), (uint32_t)kFirstSample - 1u);
                }
                if (a.depth == 1) {                                   // SH:799-809
                    size_t p = (size_t)pix_y * a.W + pix_x;
                    float dis = tbest / 16.0f;
                    a.result[p] = make_float4(dis, dis, dis, dis);
                    if (a.hit_ids) a.hit_ids[p] = idbest;
                    finished = true;
                } else if (a.depth == 2) {                            // primary hit only, for stripe_reproject_kernel (SH:757-759)
                    a.seed_out[(size_t)pix_y * a.W + pix_x] = tbest;
                    finished = true;
                } else {
                    const float4 s0 = *ls0, s1 = *ls1;
                    f3 energy = xyz(s0), result = xyz(s1);
                    float t_primary = s1.w;
                    Rng rs = {s0.w, pixel_dot(pix_x, pix_y)};
                    f3 hit_pos;
                    // plain path tracing, also inside a stripe (renderMode 1, 5, 6, 10, 11, 12): the pixel is (result, distance of the primary hit)
                    const bool alpha_early = !(a.render_mode == 0 || a.render_mode == 2 ||
                                               (MAT && (a.render_mode == 3 || a.render_mode == 4 || a.render_mode == 7 || a.render_mode == 8)));
                    bool from_history = false;
                    if (MAT && a.render_mode == 10 && bounce == 0 && sample == 0) {
                        // SH:1048-1057: a primary hit the history already shows is copied from there and no path is traced.
                        // The test depends on the camera ray and Result1 only, so it is the same for every sample of the
                        // pixel and is made once.
                        const bool sky = !(tbest < PT_INF);
                        const HistoryLookup h = history_lookup(a, sky ? pt_make3(0, 0, 0) : pt_along3(o, tbest, d));
                        if (history_covers(h, sky)) {
                            const size_t p = (size_t)pix_y * a.W + pix_x;
                            a.result[p] = h.bilinear;
                            if (a.hit_ids) a.hit_ids[p] = idbest;
                            from_history = true;
                            finished = true;
                        }
                    }
                    if (!from_history) {
                    shade<MAT, INST, FAST>(sc, rs, o, d, energy, result, (float)a.spp, tbest, idbest, recbest, hit_pos);
                    bool path_done;
                    if (bounce == 0) {
                        primary_id = idbest;
                        primary_rec = recbest;
                        t_primary = tbest;
                        // Where the pixel's final write needs nothing of the primary hit but its distance from the camera
                        // (SH:860, SH:935), that distance is taken here, when the LAST sample's primary hit comes back - hit_pos
                        // is the very bestHit.position the write would rebuild from a second camera ray - and kept in place
                        // of t, which no later sample will ask for.  At one sample per pixel this spares every shade phase a
                        // camera ray.
                        if (alpha_early && sample + 1 >= a.spp)
                            t_primary = pt_length3(pt_sub3(hit_pos, pt_make3(a.c2w[12], a.c2w[13], a.c2w[14])));
                        path_done = !(tbest < PT_INF) || a.maxb <= 1;   // primary miss: see the note above
                    } else {
                        path_done = !(energy.x != 0.0f || energy.y != 0.0f || energy.z != 0.0f) || bounce + 1 >= a.maxb;
                    }
                    bounce++;
                    new_ray = true;
                    if (path_done) {
                        if (sample == 0) note_cost(a, pix_x, pix_y, (int)pix_work);
                        sample++;
                        if (sample < a.spp) {                          // next sample of the same pixel, SH:839
                            bounce = 0;
                            energy = pt_make3(1, 1, 1);
                            cam_x = pix_x; cam_y = pix_y;              // its camera ray is made below, together with the new pixels'
                            cnt.w++;
                            if (a.reuse_primary) {
                                // SH:842 has no jitter: every sample of a pixel starts with the same camera ray, so
                                // its Trace() (SH:859) returns the hit found for the first sample
                                cached = true; cached_t = t_primary; cached_id = primary_id; cached_rec = primary_rec;
                            }
                        } else {                                       // SH:933-936
                            // SH:860, SH:935: hitPos of the (last) sample's primary ray, (0,0,0) after a miss, and
                            // its distance from the camera; every sample of a pixel has the same primary ray
                            size_t p = (size_t)pix_y * a.W + pix_x;
                            const bool full_reproj = MAT && (a.render_mode == 7 || a.render_mode == 8);
                            float4 fresh;
                            if (alpha_early) {
                                // the slot has held the distance itself since the last sample's primary hit came back (above)
                                fresh = make_float4(result.x, result.y, result.z, t_primary);
                            } else {
                                f3 co, cd;
                                camera_ray<MAT>(a, pix_x, pix_y, co, cd);
                                const bool sky = !(t_primary < PT_INF);
                                const f3 hit_pos0 = sky ? pt_make3(0, 0, 0) : pt_along3(co, t_primary, cd);
                                const f3 alpha_from = full_reproj ? pt_make3(a.alpha_origin[0], a.alpha_origin[1], a.alpha_origin[2]) : co;
                                const float alpha_dist = pt_length3(pt_sub3(hit_pos0, alpha_from));
                                fresh = make_float4(result.x, result.y, result.z, alpha_dist);
                                if (MAT && (a.render_mode == 3 || a.render_mode == 4)) fresh = reproject_pixel_ext(a, result, hit_pos0, sky, alpha_dist);
                                else if (a.render_mode == 0 || a.render_mode == 2) fresh = reproject_pixel(a, result, hit_pos0, sky, alpha_dist, pix_y);
                            }
                            if (a.render_mode == 0) { a.result2[p] = fresh; a.seed_out[p] = rs.seed; }
                            else if (full_reproj) {                    // SH:938-948
                                a.result2[p] = fresh;
                                if (a.copy_all) { a.converged[p] = fresh; a.result[p] = fresh; }
                            }
                            else a.result[p] = fresh;
                            if (a.hit_ids) a.hit_ids[p] = primary_id;
                            accumulate_pixel(a, p, fresh);
                            finished = true;
                            new_ray = false;
                        }
                    }
                    if (new_ray) {
                        *ls0 = make_float4(energy.x, energy.y, energy.z, rs.seed);
                        *ls1 = make_float4(result.x, result.y, result.z, t_primary);
                        *ls2 = make_int4(px.x, (sample == 0 ? (int)(kFirstSample | pix_work) : sample) | (bounce << 20), (int)primary_rec, primary_id);
                    }
                    }   // !from_history
                }
            }
            // ---- new pixels for lanes without one.  The wave claims them from its current shard of the frame with one
            // atomic; a shard that runs dry sends the wave on to the next one, for good.
            const bool need = was_idle || finished;
            unsigned long long m = __ballot(need);
            if (m) {
                uint32_t item_pos = 0, item_shard = 0; bool got = false;     // the claimed item: its counter and its place there
                int tries = 0;
                while (m && !frame_drained) {
                    const int leader = __ffsll((long long)m) - 1;
                    uint32_t base = 0;
                    if (lane == leader) base = atomicAdd(a.work_counter + shard * kWorkStride, (uint32_t)__popcll(m));
                    base = __shfl(base, leader);
                    const uint32_t sbeg = (uint32_t)shard * a.shard_items;
                    const uint32_t send = min(sbeg + a.shard_items, a.total_items);
                    const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (need && !got && sbeg + pos < send && base < a.shard_items) { item_pos = pos; item_shard = (uint32_t)shard; got = true; }
                    m = __ballot(need && !got);
                    if (m) {
                        shard = shard + 1 == a.n_shards ? 0 : shard + 1;
                        if (++tries == a.n_shards) frame_drained = true;   // every shard is empty: stop asking
                    }
                }
                if (need) {
                    if (!got) {
                        no_more = true;
                    } else {
                        int x, y;
                        if (claimed_pixel(a, item_shard, item_pos, x, y) && !(MAT && a.skip_t && !(a.skip_t[(size_t)y * a.W + x] < PT_INF))) {
                            *ls0 = make_float4(1.0f, 1.0f, 1.0f, a.seed);
                            *ls1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            *ls2 = make_int4(x | (y << 16), kFirstSample, 0, -1);   // first sample, no work yet, bounce 0, no primary hit yet
                            cam_x = x; cam_y = y;
                            cnt.w++;
                            new_ray = true;
                        }
                    }
                }
            }
            // ---- set up traversal of the new rays (SH:262-269: plane first)
            {
                const float *sv = lds_save + threadIdx.x;
                node = as_i(sv[0 * kBlock]); sp = as_i(sv[1 * kBlock]); ray_work = (uint32_t)as_i(sv[2 * kBlock]);
                if constexpr (WIDE == 8) { gmask = (uint32_t)sp & 0xff00ffffu; sp = (sp >> 16) & 0xff; }
                if (INST) { cur_inst = (sp >> 8) - 1; sp &= 0xff; }
            }
            if (no_more) node = kDeadNode;
            else if (finished && !new_ray) node = kIdleNode;   // pixel done, its replacement lay outside the tile
            // one camera ray for both kinds of regeneration: the wave runs this code once instead of twice
            if (new_ray && cam_x >= 0) camera_ray<MAT>(a, cam_x, cam_y, o, d);
            if (new_ray) {
                tbest = PT_INF; idbest = -1; recbest = 0;
                intersect_ground_plane(o, d, tbest, idbest);
                intersect_far_spheres(sc, o, d, tbest, idbest);
                sp = 0;
                cur_inst = -1;
                node = sc.n_nodes ? 0 : kDoneNode;
                ray_work = 0;
                cnt.x++;
                if (cached) {                       // nothing to traverse: straight back to the shade queue
                    tbest = cached_t; idbest = cached_id; recbest = cached_rec;
                    node = kDoneNode;
                    cnt.x--;
                }
            }
            // The slab constants are not kept across the shade phase: every lane computes them here, the lanes that kept
            // their ray getting the values they had (the wave runs this code for the new rays anyway).
            rsl = INST && cur_inst >= 0 ? instance_slabs(sc, cur_inst, o, d) : ray_slabs(o, d);
            if constexpr (WIDE == 8) { if (new_ray && node == 0) gmask = root_group(ray_octant(rsl)).mask; }   // a new ray starts at the root
            if (waiting) *ls3 = cnt;
            continue;
        }

        if (n_inner >= vote_node || n_leaf == 0) {
            // ================= node phase =================
#ifdef PT_PROFILE_PHASES
            ph_rounds[1]++; ph_lanes[1] += n_inner; PH_SWITCH(1)
#endif
            // keep stepping while enough lanes hold an inner node; the other lanes' states cannot change
            // meanwhile, so only that one count is re-evaluated
            for (;;) {
                if constexpr (WIDE == 8) {
                    if (holds_inner()) {
                        Group g = {(uint32_t)node, gmask};
                        node_step8(sc, g, rsl, tbest, st, sp);
                        node = g.mask ? (int32_t)g.base : kDoneNode; gmask = g.mask;
                        ray_work++;
                    }
                } else
                if (node >= 0) {
#if defined(PT_COUNT_EMPTY) || defined(PT_COUNT_OVERFLOW)   // measurement builds: the node statistic counts only the visits that entered no child / the stack entries that overflowed
                    if constexpr (WIDE == 4) { uint32_t e_ = 0; node = node_step4(sc, node, rsl, tbest, st, sp, &e_); ray_work += e_; } else
#endif
                    { node = node_step<WIDE>(sc, node, rsl, tbest, st, sp); ray_work++; }
                    if (INST) resolve_instance_links(sc, node, o, d, cur_inst, rsl, st, sp);
                }
                const int still = __popcll(__ballot(holds_inner()));
#ifdef PT_PROFILE_PHASES
                if (still >= vote_node) { ph_rounds[1]++; ph_lanes[1] += still; }
#endif
                if (still < vote_node) break;
            }
            continue;
        }

        // ================= leaf phase =================
#ifdef PT_PROFILE_PHASES
        ph_rounds[2]++; ph_lanes[2] += n_leaf; PH_SWITCH(2)
#endif
        if (leaf) {
            uint32_t np = 0;
            if constexpr (WIDE == 8) {
                Group g = {(uint32_t)node, gmask};
                leaf_step8(sc, g, o, d, ray_octant(rsl), tbest, idbest, recbest, st, sp, np);
                node = g.mask ? (int32_t)g.base : kDoneNode; gmask = g.mask;
            } else
            if (INST) node = leaf_step_inst(sc, node, o, d, cur_inst, rsl, tbest, idbest, recbest, st, sp, np);
            else node = leaf_step(sc, node, o, d, tbest, idbest, recbest, st, sp, np);
            ray_work += np << 16;
        }
    }

    const int4 cnt = *ls3;
    stats_add(a.stats, 0, (uint32_t)cnt.x, lane);
    stats_add(a.stats, 1, (uint32_t)cnt.y, lane);
    stats_add(a.stats, 2, (uint32_t)cnt.z, lane);
    stats_add(a.stats, 3, (uint32_t)cnt.w, lane);
#ifdef PT_PROFILE_PHASES
    if (lane == 0 && a.stats)
        for (int k = 0; k < 3; k++) {
            atomicAdd(&a.stats[(blockIdx.x % kStatsShards) * kStatsStride + 4 + 2 * k], (unsigned long long)ph_rounds[k]);
            atomicAdd(&a.stats[(blockIdx.x % kStatsShards) * kStatsStride + 5 + 2 * k], (unsigned long long)ph_lanes[k]);
            atomicAdd(&a.stats[(blockIdx.x % kStatsShards) * kStatsStride + 10 + k], ph_cyc[k]);
        }
    if (lane == 0 && a.stats) {      // how long this wave lived, in ticks of the constant 100 MHz counter: sum and maximum
        const unsigned long long life = wall_clock64() - ph_wall0;
        atomicAdd(&a.stats[(blockIdx.x % kStatsShards) * kStatsStride + 13], life);
        atomicMax(&a.stats[(blockIdx.x % kStatsShards) * kStatsStride + 14], life);
    }
#endif
}

#undef ls0
#undef ls1
#undef ls2
#undef ls3

// The extended kernel for two-level scenes (instances; leaf_step_inst).
#ifndef PT_INST_WAVES
#define PT_INST_WAVES 4   // 128 VGPRs: at 5 waves (96) the kernel spills 81 registers and runs 10 % slower
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_INST_WAVES))) void path_sched_inst_kernel(const RenderArgs a)
{
    path_sched_body<4, true, true>(a);
}

// Five waves per SIMD (96 VGPRs): the few values that do not fit are kernel constants spilled once at entry and read back
// in the shade phase; the node and leaf loops hold no scratch access.  Measured against four waves (128 VGPRs, nothing
// spilled): +2.5 % rays/s one dispatch at a time, +6 % with three in flight; six waves (80 VGPRs) lose.
#ifndef PT_SCHED_WAVES
#define PT_SCHED_WAVES 5   // measurement builds: 4 = 128 VGPRs, nothing spilled inside the loop
#endif
template <int WIDE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_SCHED_WAVES))) void path_sched_kernel(const RenderArgs a)
{
    path_sched_body<WIDE, false>(a);
}

// Option "fast_intrinsics": the default kernel with Shade() on the hardware's approximate transcendental instructions
// (Fm<true>); not bit-identical to the oracle - same primary hits, same random numbers, an image within a stated RMSE.
template <int WIDE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5))) void path_sched_fast_kernel(const RenderArgs a)
{
    path_sched_body<WIDE, false, false, true>(a);
}

// The same kernel for scenes whose mesh objects carry materials (pt_set_mesh_materials); kept apart so that
// the extra shading code cannot cost the default kernel registers.
template <int WIDE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5))) void path_sched_mat_kernel(const RenderArgs a)
{
    path_sched_body<WIDE, true>(a);
}

// order[] = the blocks (column | row << 16) by falling cost (a counting sort on the clamped cost; ties in no particular order: any
// permutation renders the same image), and the counters back to zero for the frame that is about to start.
// One small workgroup - four waves, 2 KB of LDS, 16 registers - so that it fits beside the resident workgroups of other
// contexts' path kernels: five of those per CU leave three wave slots per SIMD, 32 registers per lane and, after the LDS
// allocator's rounding, under 4 KB of LDS.  A kernel trace of three dispatches in flight showed the earlier forms (1024
// threads and 8 KB; 256 threads and 4.1 KB) pending for 6-7 ms on average, up to 22 ms, until a CU had drained, and the
// context's next frame behind them; this one starts at once (0.11 ms) and the three-in-flight rate is 0.8 % higher.
constexpr int kOrderThreads = 256, kOrderBins = 512, kOrderShift = 0;    // a pixel's first path costs tens to some hundreds of steps: a bin per step
__global__ __launch_bounds__(kOrderThreads) void order_blocks_kernel(uint32_t *cost, uint32_t *order, uint32_t n_blocks, uint32_t tiles_x)
{
    __shared__ uint32_t bin[kOrderBins];
    __shared__ uint32_t wave_sum[kOrderThreads / 64];
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < kOrderBins; i += kOrderThreads) bin[i] = 0;
    __syncthreads();
    // four blocks per thread and round, so that four loads are in flight
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 4 * kOrderThreads) {
        uint32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t b = b0 + j * kOrderThreads + t; c[j] = b < n_blocks ? cost[b] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (b0 + j * kOrderThreads + t < n_blocks) atomicAdd(&bin[min(c[j] >> kOrderShift, (uint32_t)kOrderBins - 1u)], 1u);
    }
    __syncthreads();
    // first rank of every bin, most expensive bin first: thread t owns the four bins of rank 4t .. 4t+3 from the top
    {
        constexpr int kPer = kOrderBins / kOrderThreads;
        uint32_t c[kPer], sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; j++) { c[j] = bin[kOrderBins - 1 - (t * kPer + j)]; sum += c[j]; }
        uint32_t incl = sum;                                   // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off); if ((t & 63u) >= (uint32_t)off) incl += v; }
        if ((t & 63u) == 63u) wave_sum[t >> 6] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < (t >> 6); w++) run += wave_sum[w];
#pragma unroll
        for (int j = 0; j < kPer; j++) { bin[kOrderBins - 1 - (t * kPer + j)] = run; run += c[j]; }
    }
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 4 * kOrderThreads) {
        uint32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t b = b0 + j * kOrderThreads + t; c[j] = b < n_blocks ? cost[b] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t b = b0 + j * kOrderThreads + t;
            if (b < n_blocks) {
                order[atomicAdd(&bin[min(c[j] >> kOrderShift, (uint32_t)kOrderBins - 1u)], 1u)] = (b % tiles_x) | ((b / tiles_x) << 16);
                cost[b] = 0;
            }
        }
    }
}

template <int WIDE>
__global__ __launch_bounds__(kBlock) void trace_rays_bvh_kernel(SceneDev sc, uint32_t *stack_overflow, const float *od, int n,
                                                                 int32_t *ids, float *ts)
{
    __shared__ __align__(8) uint32_t lds_stack[StackFor<WIDE>::kWords * kBlock];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const typename StackFor<WIDE>::type st = StackFor<WIDE>::make(lds_stack, stack_overflow + i, gridDim.x * kBlock);
    if (i >= n) return;
    f3 o = pt_make3(od[i * 6 + 0], od[i * 6 + 1], od[i * 6 + 2]);
    f3 d = pt_make3(od[i * 6 + 3], od[i * 6 + 4], od[i * 6 + 5]);
    float t = PT_INF; int32_t id = -1; uint32_t rec = 0, cn = 0, cp = 0;
    intersect_ground_plane(o, d, t, id);
    intersect_far_spheres(sc, o, d, t, id);
    traverse<WIDE>(sc, o, d, t, id, rec, st, cn, cp);
    ids[i] = id; ts[i] = t;
}

// One thread per ray against every primitive, no tree: the exhaustive loop of the shader (SH:262-285), which keeps the
// first of equally near hits in id order - what intersect_record's (t, id) rule keeps in any order.  The records of
// the top level hold every flattened primitive (a triangle cut into several references several times: harmless);
// the triangles of instances come from the reference's own buffers.
__global__ __launch_bounds__(kBlock) void trace_rays_brute_kernel(SceneDev sc, const float *od, int n, int32_t *ids, float *ts)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 o = pt_make3(od[i * 6 + 0], od[i * 6 + 1], od[i * 6 + 2]);
    f3 d = pt_make3(od[i * 6 + 3], od[i * 6 + 4], od[i * 6 + 5]);
    float t = PT_INF; int32_t id = -1;
    intersect_ground_plane(o, d, t, id);
    uint32_t rec = 0;
    for (int k = 0; k < sc.n_top_records; k++) {
        const float4 *p = sc.prims + (size_t)k * 3;
        const float4 a = p[0], b = p[1], c = p[2];
        if (as_i(b.w) != 2) intersect_record(a, b, c, (uint32_t)k, o, d, t, id, rec);      // kind 2 stands for an instance, below
    }
    for (int ii = 0; ii < sc.n_inst; ii++) {
        const float4 meta = sc.inst[(size_t)ii * 8 + 6];
        const int obj = as_i(meta.z), id0 = as_i(meta.y);
        const int4 ob = sc.objects[obj];
        const int n_tris = sc.objects[obj + 1].x - ob.x;
        const float *m = sc.matrices + 16 * (size_t)obj;
        for (int k = 0; k < n_tris; k++) {
            const int32_t *ix = sc.indices + ob.y + 3 * k;
            const float *pa = sc.vertices + 3 * (size_t)ix[0], *pb = sc.vertices + 3 * (size_t)ix[1], *pc = sc.vertices + 3 * (size_t)ix[2];
            const f3 v0 = pt_mul_point(m, pt_make3(pa[0], pa[1], pa[2]));      // SH:238-240
            const f3 v1 = pt_mul_point(m, pt_make3(pb[0], pb[1], pb[2]));
            const f3 v2 = pt_mul_point(m, pt_make3(pc[0], pc[1], pc[2]));
            const f3 e1 = pt_sub3(v1, v0), e2 = pt_sub3(v2, v0);
            intersect_record(make_float4(v0.x, v0.y, v0.z, as_f(id0 + k)), make_float4(e1.x, e1.y, e1.z, 0.0f), make_float4(e2.x, e2.y, e2.z, 0.0f),
                             0, o, d, t, id, rec);
        }
    }
    ids[i] = id; ts[i] = t;
}

}  // namespace ptrace
