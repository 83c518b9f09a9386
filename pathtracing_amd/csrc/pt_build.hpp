// Build of the four-wide BVH on the device, for scenes whose geometry changes at a size the host builder (bvh.cpp: binned
// SAH on sixteen threads, seconds for ten million triangles) cannot follow.
//
// The reference has no tree: RebuildMeshObjectBuffers (RayTracingMaster.cs:371-413) re-uploads every vertex and index when
// an object is added or removed and the shader visits every triangle per ray (RayTracingShader.compute:232-257).  Here
// the same upload is followed by:
//   1. one pass over the triangles for the scene's and the objects' bounds (build_bounds_kernel);
//   2. a record and a padded box per primitive (build_prims_kernel, then refit_prims_kernel - the same kernel and therefore
//      the same boxes as a transform update);
//   3. 63-bit Morton codes of the box centres and a radix sort (rocPRIM);
//   4. the two-child tree: top-down by binned SAH (pt_build_sah.hpp; the default), or - option tree_build 3 - by parallel
//      locally-ordered clustering (Meister and Bittner, "Parallel locally-ordered clustering for bounding volume hierarchy
//      construction", IEEE TVCG 24(3), 2018; the ploc_* kernels below): every cluster looks kPlocRadius places to either side in
//      the sorted order for the neighbour whose union with it has the smallest area, pairs that chose each other merge, the
//      survivors are compacted - bottom-up, a few dozen rounds - and the host's SAH builder makes the top of the tree over the
//      last few thousand clusters;
//   5. the collapse to four children per node level by level from the root (open the inner child of largest area, as
//      bvh.cpp does), then a depth-first renumbering of nodes and records;
//   6. the node boxes and their quantisation from the leaves up (refit_nodes_kernel), level by level.
// Everything is a function of the input alone: neighbours tie-break by position, places are given out by prefix sums, never
// by atomic counters, so two builds of one scene give the same tree.  The closest hit does not depend on the tree at all.
#pragma once

#include <rocprim/rocprim.hpp>

#include "pt_refit.hpp"

namespace ptrace {

constexpr int kBuildBlock = 256;
constexpr int kPlocRadius = 16;            // places to either side a cluster looks for its partner
constexpr int kPlocRadiusMax = 64;
constexpr int32_t kNoKid = INT32_MIN;          // an empty slot in collapse_pick_kernel's output (a leaf is ~primitive, never this)

// float <-> unsigned with the same order, for atomicMin / atomicMax on coordinates
__device__ __forceinline__ uint32_t ordered_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
inline float ordered_float(uint32_t u)
{
    u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
    float f; std::memcpy(&f, &u, 4);
    return f;
}

struct BuildScene {
    const float *vertices;        // object space, 3 floats per vertex
    const int32_t *indices;
    const float *matrices;        // 16 floats per object
    const int4 *objects;          // (first triangle, first index, 0, 0) per object, then (n_tris, 0, 0, 0)
    const float4 *spheres;        // four float4 per sphere: (position, radius) first
    int32_t n_objects, n_spheres, n_tris;
};

// The mesh object of triangle `gid` (triangles are numbered object by object, RayTracingShader.compute:232-257): the last
// object whose first triangle is <= gid - objects without triangles share their first triangle with the next one.
__device__ __forceinline__ int object_of(const BuildScene &s, int gid)
{
    int lo = 0, hi = s.n_objects - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s.objects[mid].x <= gid) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

// How the reductions over the whole primitive array are laid out: every wave takes kWaveChunk consecutive places, 64 at a
// time, and sends its atomics once per chunk (or whenever the thing it reduces for changes) - one atomic per wave and per 64
// primitives, all on the same few words, is ten milliseconds of serialised traffic for ten million primitives.
constexpr int kWaveChunk = 64 * 32;
__host__ __device__ inline size_t wave_chunk_blocks(size_t n) { return (n + (size_t)kWaveChunk * (kBuildBlock / 64) - 1) / ((size_t)kWaveChunk * (kBuildBlock / 64)); }

// scene_box: 6 ordered words (lo.xyz from +inf, hi.xyz from -inf) over the world-space vertices; object_box: the same per
// object over its object-space vertices.  Coordinates that are NaN or beyond kHugeCoord take no part (bvh.hpp).
__global__ __launch_bounds__(kBuildBlock) void build_bounds_kernel(const BuildScene s, uint32_t *scene_box, uint32_t *object_box)
{
    const int lane = threadIdx.x & 63;
    const size_t chunk = ((size_t)blockIdx.x * (kBuildBlock / 64) + (threadIdx.x >> 6)) * kWaveChunk;
    if (chunk >= (size_t)s.n_tris) return;
    float wlo[3] = {PT_INF, PT_INF, PT_INF}, whi[3] = {-PT_INF, -PT_INF, -PT_INF};      // this lane's share of the scene box
    float olo[3] = {PT_INF, PT_INF, PT_INF}, ohi[3] = {-PT_INF, -PT_INF, -PT_INF};      // and of the object the wave is in
    int held = -1;                                                                       // that object (the same in every lane)
    const auto flush_object = [&]() {
        if (held < 0) return;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float a = wave_min(olo[k]), b = wave_max(ohi[k]);
            if (lane == 0 && a <= b) { atomicMin(object_box + 6 * (size_t)held + k, ordered_bits(a)); atomicMax(object_box + 6 * (size_t)held + 3 + k, ordered_bits(b)); }
            olo[k] = PT_INF; ohi[k] = -PT_INF;
        }
    };
    for (int step = 0; step < kWaveChunk / 64; step++) {
        const size_t first = chunk + (size_t)step * 64;
        if (first >= (size_t)s.n_tris) break;
        const int gid = (int)min(first + (size_t)lane, (size_t)s.n_tris - 1);           // a duplicate changes no minimum
        const int o = object_of(s, gid);
        const int4 ob = s.objects[o];
        const int32_t *ix = s.indices + (size_t)ob.y + 3 * (size_t)(gid - ob.x);
        float mm[16];
#pragma unroll
        for (int k = 0; k < 16; k++) mm[k] = s.matrices[16 * (size_t)o + k];
        const bool one_object = __all(o == __builtin_amdgcn_readfirstlane(o));
        if (!one_object || o != held) { flush_object(); held = one_object ? o : -1; }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float *p = s.vertices + 3 * (size_t)ix[c];
            const float v[3] = {p[0], p[1], p[2]};
            const f3 w = pt_mul_point(mm, pt_make3(v[0], v[1], v[2]));      // RayTracingShader.compute:238-240
            const float wv[3] = {w.x, w.y, w.z};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (fabsf(wv[k]) < kHugeCoord) { wlo[k] = fminf(wlo[k], wv[k]); whi[k] = fmaxf(whi[k], wv[k]); }
                if (!(fabsf(v[k]) < kHugeCoord)) continue;
                if (one_object) { olo[k] = fminf(olo[k], v[k]); ohi[k] = fmaxf(ohi[k], v[k]); }
                else { atomicMin(object_box + 6 * (size_t)o + k, ordered_bits(v[k])); atomicMax(object_box + 6 * (size_t)o + 3 + k, ordered_bits(v[k])); }
            }
        }
    }
    flush_object();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float l = wave_min(wlo[k]), h = wave_max(whi[k]);
        if (lane == 0 && l <= h) { atomicMin(scene_box + k, ordered_bits(l)); atomicMax(scene_box + 3 + k, ordered_bits(h)); }
    }
}

// Per primitive p (spheres first, then the triangles in the reference's order): where refit_prims_kernel finds it, and the
// parts of its record that kernel leaves alone - the id, and all of a sphere's record.
__global__ __launch_bounds__(kBuildBlock) void build_prims_kernel(const BuildScene s, int4 *src, float4 *prims)
{
    const int p = blockIdx.x * kBuildBlock + threadIdx.x;
    if (p >= s.n_spheres + s.n_tris) return;
    if (p < s.n_spheres) {
        const float4 sp = s.spheres[(size_t)p * 4];
        src[p] = make_int4(-1, 0, 0, 0);
        prims[(size_t)p * 3 + 0] = make_float4(sp.x, sp.y, sp.z, __int_as_float(1 + p));
        prims[(size_t)p * 3 + 1] = make_float4(sp.w, 0.0f, 0.0f, __int_as_float(1));
        prims[(size_t)p * 3 + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const int gid = p - s.n_spheres;
    const int o = object_of(s, gid);
    const int4 ob = s.objects[o];
    const int32_t *ix = s.indices + (size_t)ob.y + 3 * (size_t)(gid - ob.x);
    src[p] = make_int4(o, ix[0], ix[1], ix[2]);
    prims[(size_t)p * 3 + 0] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(1 + p));
    prims[(size_t)p * 3 + 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    prims[(size_t)p * 3 + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---- reference splitting (bvh.hpp, split_references) in a device build.  Which triangles are cut, how often and where is
// the host's code and the host's arithmetic - they are few: only a triangle whose box spans a sixteenth of the scene is a
// candidate.  The device names the candidates (list_large_kernel, with a margin: the host applies its own rule again), the
// host cuts them and sends back one (reference, primitive, cut box) entry per piece; the first piece takes the primitive's
// own place, the others are appended behind the primitives.  From there on the build works on references: refit_prims_kernel
// clips a reference's box to its cut, several records name one triangle, and a later refit keeps the cuts of objects that
// did not move - all as after a host build.
__global__ __launch_bounds__(kBuildBlock) void list_large_kernel(const float *box, int first, int n, float threshold, uint32_t *list, uint32_t *count)
{
    const int p = first + blockIdx.x * kBuildBlock + threadIdx.x;
    if (p >= n) return;
    const float *b = box + (size_t)p * 6;
    const float e = fmaxf(fmaxf(b[3] - b[0], b[4] - b[1]), b[5] - b[2]);
    if (e >= threshold && e < kHugeCoord) list[atomicAdd(count, 1u)] = (uint32_t)p;      // the host sorts the list: its order is nobody's business
}

// clip[slot]: the part of its triangle's box a reference stands for (lo.xyz, hi.xyz); everything for a reference that is the
// whole triangle.
__global__ __launch_bounds__(kBuildBlock) void fill_clip_kernel(float *clip, size_t n)
{
    const size_t s = (size_t)blockIdx.x * kBuildBlock + threadIdx.x;
    if (s >= n) return;
    for (int k = 0; k < 3; k++) { clip[s * 6 + k] = -PT_INF; clip[s * 6 + 3 + k] = PT_INF; }
}

__global__ __launch_bounds__(kBuildBlock) void apply_split_kernel(const int2 *entry, const float *cut, int n_entries, int4 *src, float4 *prims, float *clip)
{
    const int j = blockIdx.x * kBuildBlock + threadIdx.x;
    if (j >= n_entries) return;
    const int r = entry[j].x, p = entry[j].y;
    if (r != p) {
        src[r] = src[p];
        for (int k = 0; k < 3; k++) prims[(size_t)r * 3 + k] = prims[(size_t)p * 3 + k];      // the id; the rest is refit_prims_kernel's
    }
    for (int k = 0; k < 6; k++) clip[(size_t)r * 6 + k] = cut[(size_t)j * 6 + k];
}

__device__ __forceinline__ uint64_t spread21(uint32_t v)      // bit i of v to bit 3 i
{
    uint64_t x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

struct MortonGrid { float lo[3], scale[3]; };      // scale = 2^21 / extent, or 0 for an axis without extent

__global__ __launch_bounds__(kBuildBlock) void build_morton_kernel(const float *box, int n, const MortonGrid g, uint64_t *keys, uint32_t *vals)
{
    const int p = blockIdx.x * kBuildBlock + threadIdx.x;
    if (p >= n) return;
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float c = 0.5f * box[(size_t)p * 6 + k] + 0.5f * box[(size_t)p * 6 + 3 + k];
        const float x = (c - g.lo[k]) * g.scale[k];
        q[k] = x >= 0.0f ? (uint32_t)fminf(x, 2097151.0f) : 0u;        // a box of +-kHugeCoord has its centre at 0; NaN cannot occur
    }
    keys[p] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
    vals[p] = (uint32_t)p;
}

// Clusters: lo = (box.lo, bits(reference)), hi = (box.hi, bits(primitives below)).  A reference is the number of a
// two-child node (>= 0) or ~primitive.
__global__ __launch_bounds__(kBuildBlock) void ploc_init_kernel(const uint32_t *sorted, const float *box, int n, float4 *lo, float4 *hi)
{
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = sorted[i];
    const float *b = box + (size_t)p * 6;
    lo[i] = make_float4(b[0], b[1], b[2], __int_as_float(~(int32_t)p));
    hi[i] = make_float4(b[3], b[4], b[5], __int_as_float(1));
}

// nn[i]: the cluster within kPlocRadius places of i whose union with i has the smallest area; of equals, the first.
__global__ __launch_bounds__(kBuildBlock) void ploc_nn_kernel(const float4 *lo, const float4 *hi, int c, int32_t *nn, const int R, const float lambda)
{
    constexpr int TMax = kBuildBlock + 2 * kPlocRadiusMax;
    const int T = kBuildBlock + 2 * R;
    __shared__ float sl[3][TMax], sh[3][TMax];
    const int base = blockIdx.x * kBuildBlock - R;
    for (int t = threadIdx.x; t < T; t += kBuildBlock) {
        const int i = base + t;
        if (i >= 0 && i < c) {
            const float4 a = lo[i], b = hi[i];
            sl[0][t] = a.x; sl[1][t] = a.y; sl[2][t] = a.z; sh[0][t] = b.x; sh[1][t] = b.y; sh[2][t] = b.z;
        }
    }
    __syncthreads();
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= c) return;
    const int me = threadIdx.x + R;
    const float l0 = sl[0][me], l1 = sl[1][me], l2 = sl[2][me], h0 = sh[0][me], h1 = sh[1][me], h2 = sh[2][me];
    const float own = (h0 - l0) * (h1 - l1) + (h1 - l1) * (h2 - l2) + (h2 - l2) * (h0 - l0);
    float best = PT_INF;
    int best_j = i > 0 ? i - 1 : i + 1;
#pragma unroll 4
    for (int d = -R; d <= R; d++) {
        const int j = i + d;
        if (d == 0 || j < 0 || j >= c) continue;
        const int t = me + d;
        const float ex = fmaxf(h0, sh[0][t]) - fminf(l0, sl[0][t]);
        const float ey = fmaxf(h1, sh[1][t]) - fminf(l1, sl[1][t]);
        const float ez = fmaxf(h2, sh[2][t]) - fminf(l2, sl[2][t]);
        const float ox = sh[0][t] - sl[0][t], oy = sh[1][t] - sl[1][t], oz = sh[2][t] - sl[2][t];
        const float a = (ex * ey + ey * ez + ez * ex) - lambda * (own + (ox * oy + oy * oz + oz * ox));
        if (a < best) { best = a; best_j = j; }
    }
    nn[i] = best_j;
}

// flags[i]: low word 1 if cluster i survives the round (it is not the later half of a pair), high word 1 if it is the
// earlier half of a pair and so becomes a new node.
__global__ __launch_bounds__(kBuildBlock) void ploc_flag_kernel(const int32_t *nn, int c, uint64_t *flags)
{
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= c) return;
    const int j = nn[i];
    const bool pair = nn[j] == i;
    flags[i] = (uint64_t)(pair && i > j ? 0u : 1u) | ((uint64_t)(pair && i < j ? 1u : 0u) << 32);
}

struct TwoChildTree {
    int2 *child;       // two references per node
    float4 *lo, *hi;   // box; hi.w = bits(primitives below)
};

// offs = exclusive prefix sums of flags.  totals[0] = survivors | new nodes << 32.
__global__ __launch_bounds__(kBuildBlock) void ploc_merge_kernel(const float4 *lo, const float4 *hi, const int32_t *nn, const uint64_t *flags, const uint64_t *offs, int c,
                                                                 int node_base, TwoChildTree tree, float4 *lo_out, float4 *hi_out, uint64_t *totals)
{
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= c) return;
    const uint64_t f = flags[i], o = offs[i];
    if (i == c - 1) totals[0] = o + f;
    if (!(uint32_t)f) return;
    const int dst = (int)(uint32_t)o;
    float4 a = lo[i], b = hi[i];
    if (f >> 32) {
        const int j = nn[i];
        const float4 a2 = lo[j], b2 = hi[j];
        const int node = node_base + (int)(o >> 32);
        tree.child[node] = make_int2(__float_as_int(a.w), __float_as_int(a2.w));
        a = make_float4(fminf(a.x, a2.x), fminf(a.y, a2.y), fminf(a.z, a2.z), __int_as_float(node));
        b = make_float4(fmaxf(b.x, b2.x), fmaxf(b.y, b2.y), fmaxf(b.z, b2.z), __int_as_float(__float_as_int(b.w) + __float_as_int(b2.w)));
        tree.lo[node] = a; tree.hi[node] = b;
    }
    lo_out[dst] = a; hi_out[dst] = b;
}

// ---- the bottom of the tree, again.  Clustering pairs primitives off in its first rounds by what happens to be mutual, and
// what is left over joins as singles: subtrees of three, five, seven primitives whose inner topology is whatever order the
// merges came in.  Every subtree of at most kBottomPrims primitives (whose parent has more) is rebuilt from its primitives'
// boxes by one thread, top-down: each set goes into the two parts of smallest area-weighted size, found by trying ALL
// two-part splits (127 for eight primitives).  The subtree keeps its node numbers and its root, so nothing above it changes.
// (Worth 2-5 % of the rays per second on the 70 k-triangle mesh, nothing on the 262 k-triangle hall.  The same with a sweep
// along three axes for subtrees of up to 64 primitives changed the hall's node visits per ray from 16.73 to 16.69 and took
// twenty times as long as the rest of the build: the clusters' insides are not where clustering loses to the host's
// top-down SAH.)
constexpr int kBottomPrims = 8;

__global__ __launch_bounds__(64) void bottom_rebuild_kernel(const TwoChildTree tree, int n_nodes, const float *prim_box)
{
    constexpr int M = kBottomPrims;
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= 2 * n_nodes) return;
    const int parent = t >> 1;
    if (__float_as_int(tree.hi[parent].w) <= M) return;
    const int2 pc = tree.child[parent];
    const int32_t top = (t & 1) ? pc.y : pc.x;
    if (top < 0) return;
    const int m = __float_as_int(tree.hi[top].w);
    if (m > M || m < 3) return;
    // the subtree's primitives and node numbers (m and m - 1 of them)
    int32_t prim[M], ids[M];
    {
        int32_t stack[M];
        int np = 0, ni = 0, sp = 0;
        stack[sp++] = top;
        while (sp) {
            const int32_t r = stack[--sp];
            if (r < 0) { if (np < M) prim[np++] = ~r; continue; }
            if (ni < M) ids[ni++] = r;
            const int2 ch = tree.child[r];
            if (sp + 2 <= M) { stack[sp++] = ch.y; stack[sp++] = ch.x; }
        }
        if (np != m || ni != m - 1) return;        // not a tree of m leaves (cannot happen): leave it as it is
    }
    float lo[M][3], hi[M][3];
    for (int k = 0; k < m; k++)
        for (int ax = 0; ax < 3; ax++) { lo[k][ax] = prim_box[(size_t)prim[k] * 6 + ax]; hi[k][ax] = prim_box[(size_t)prim[k] * 6 + 3 + ax]; }
    // half the surface area of every subset's box, and - for the nodes written below - the box itself on demand
    float area[1 << M];
    const int full = (1 << m) - 1;
    for (int s = 1; s <= full; s++) {
        float l[3] = {PT_INF, PT_INF, PT_INF}, h[3] = {-PT_INF, -PT_INF, -PT_INF};
        for (int k = 0; k < m; k++)
            if ((s >> k) & 1)
                for (int ax = 0; ax < 3; ax++) { l[ax] = fminf(l[ax], lo[k][ax]); h[ax] = fmaxf(h[ax], hi[k][ax]); }
        const float ex = h[0] - l[0], ey = h[1] - l[1], ez = h[2] - l[2];
        area[s] = ex * ey + ey * ez + ez * ex;
    }
    // top-down: sets[] holds the subsets still to be split, each with the node number it was given
    int sets[M]; int32_t set_node[M];
    int next_id = 0, n_sets = 0;
    sets[n_sets] = full; set_node[n_sets++] = ids[next_id++];
    while (n_sets) {
        const int s = sets[--n_sets];
        const int32_t node = set_node[n_sets];
        const int low = s & -s;
        int best = low; float best_cost = PT_INF;
        for (int sub = (s - 1) & s; sub; sub = (sub - 1) & s) {      // every split of s; the part with s's first member is `sub`
            if (!(sub & low)) continue;
            const int rest = s ^ sub;
            const float c = area[sub] * (float)__popc(sub) + area[rest] * (float)__popc(rest);
            if (c < best_cost || (c == best_cost && sub < best)) { best_cost = c; best = sub; }
        }
        const int part[2] = {best, s ^ best};
        int32_t ref[2];
        for (int k = 0; k < 2; k++) {
            if (__popc(part[k]) == 1) ref[k] = ~prim[__ffs(part[k]) - 1];
            else { ref[k] = ids[next_id]; sets[n_sets] = part[k]; set_node[n_sets++] = ids[next_id++]; }
        }
        float l[3] = {PT_INF, PT_INF, PT_INF}, h[3] = {-PT_INF, -PT_INF, -PT_INF};
        for (int k = 0; k < m; k++)
            if ((s >> k) & 1)
                for (int ax = 0; ax < 3; ax++) { l[ax] = fminf(l[ax], lo[k][ax]); h[ax] = fmaxf(h[ax], hi[k][ax]); }
        tree.child[node] = make_int2(ref[0], ref[1]);
        tree.lo[node] = make_float4(l[0], l[1], l[2], __int_as_float(node));
        tree.hi[node] = make_float4(h[0], h[1], h[2], __int_as_float(__popc(s)));
    }
}

// ---- collapse, one level of the four-wide tree per pair of launches
struct CollapseArgs {
    TwoChildTree tree;
    const int32_t *items;         // this level's nodes, as two-child node numbers; four-wide node level_base + t comes from items[t]
    const int32_t *stacked;       // stack entries a ray holds when it enters each
    int32_t n_items, level_base, next_base, rec_base, zero_rec;
    int32_t passes;               // 2: pairs of primitives may be opened into free slots; 1: they stay leaves of two records
    const float *prim_box;        // 6 floats per primitive
    int4 *kids;                   // per item, the up to four references chosen
    uint64_t *counts;             // per item: inner children | records << 32
    const uint64_t *offs;         // exclusive prefix sums of counts
    int32_t *items_next, *stacked_next;
    uint4 *nodes4;
    uint32_t *slot_prim;          // record -> primitive
    int32_t *stack_need;          // running maximum
    uint64_t *totals;
};

__device__ __forceinline__ int prims_below(const TwoChildTree &t, int32_t ref) { return ref < 0 ? 1 : __float_as_int(t.hi[ref].w); }
__device__ __forceinline__ float half_area(const TwoChildTree &t, int32_t ref)
{
    const float4 a = t.lo[ref], b = t.hi[ref];
    const float ex = b.x - a.x, ey = b.y - a.y, ez = b.z - a.z;
    return ex * ey + ey * ez + ez * ex;
}

// The children of one four-wide node: the two-child node's two, then - while a slot is free - the inner child of largest
// area opened into its two (bvh.cpp's rule).  A subtree of two primitives becomes one leaf of two records unless a slot is
// free AND the two boxes' areas add up to less than the pair's (the host builder's rule for ending at a leaf, bvh.cpp:133:
// testing both, 2 A, against one more visit and each on its own, A + A_left + A_right): a leaf visit costs a wave a round
// of its leaf phase whether the leaf holds one record or two, and most pairs are the two halves of a quad.
__global__ __launch_bounds__(kBuildBlock) void collapse_pick_kernel(const CollapseArgs a)
{
    const int t = blockIdx.x * kBuildBlock + threadIdx.x;
    if (t >= a.n_items) return;
    const int2 ch = a.tree.child[a.items[t]];
    int32_t kids[4] = {ch.x, ch.y, kNoKid, kNoKid};
    int nk = 2;
#pragma unroll 1
    for (int pass = 0; pass < a.passes; pass++)
        while (nk < 4) {
            int best = -1; float best_area = -1.0f;
            for (int k = 0; k < nk; k++) {
                if (kids[k] < 0) continue;
                const int cnt = prims_below(a.tree, kids[k]);
                if (pass == 0 ? cnt > 2 : cnt == 2) {
                    const float ar = half_area(a.tree, kids[k]);
                    if (pass == 1) {
                        const int2 two = a.tree.child[kids[k]];
                        const float *p = a.prim_box + (size_t)~two.x * 6, *q = a.prim_box + (size_t)~two.y * 6;
                        const float pa = (p[3] - p[0]) * (p[4] - p[1]) + (p[4] - p[1]) * (p[5] - p[2]) + (p[5] - p[2]) * (p[3] - p[0]);
                        const float qa = (q[3] - q[0]) * (q[4] - q[1]) + (q[4] - q[1]) * (q[5] - q[2]) + (q[5] - q[2]) * (q[3] - q[0]);
                        if (!(pa + qa < ar)) continue;
                    }
                    if (ar > best_area) { best_area = ar; best = k; }
                }
            }
            if (best < 0) break;
            const int2 open = a.tree.child[kids[best]];
            kids[best] = open.x; kids[nk++] = open.y;
        }
    uint32_t inner = 0, recs = 0;
    for (int k = 0; k < nk; k++) {
        const int cnt = prims_below(a.tree, kids[k]);
        if (kids[k] >= 0 && cnt > 2) inner++; else recs += (uint32_t)cnt;
    }
    a.kids[t] = make_int4(kids[0], kids[1], kids[2], kids[3]);
    a.counts[t] = (uint64_t)inner | ((uint64_t)recs << 32);
}

__global__ __launch_bounds__(kBuildBlock) void collapse_emit_kernel(const CollapseArgs a)
{
    const int t = blockIdx.x * kBuildBlock + threadIdx.x;
    if (t >= a.n_items) return;
    const uint64_t off = a.offs[t];
    if (t == a.n_items - 1) a.totals[0] = off + a.counts[t];
    const int4 kv = a.kids[t];
    const int32_t kids[4] = {kv.x, kv.y, kv.z, kv.w};
    int nk = 0;
    for (int k = 0; k < 4; k++) nk += kids[k] != kNoKid;
    const int held = a.stacked[t] + nk - 1;
    Bvh4Node nd;
    nd.org[0] = nd.org[1] = nd.org[2] = 0.0f; nd.ex = nd.ey = nd.ez = 27; nd.valid = 0; nd.pad[0] = nd.pad[1] = 0;      // boxes: refit_nodes_kernel
    uint32_t inner = (uint32_t)off, rec = (uint32_t)a.rec_base + (uint32_t)(off >> 32);
    for (int k = 0; k < 4; k++) {
        for (int ax = 0; ax < 3; ax++) { nd.qlo[ax][k] = 255; nd.qhi[ax][k] = 0; }
        const int32_t r = kids[k];
        if (r == kNoKid) { nd.child[k] = ~(int32_t)((uint32_t)a.zero_rec << 3); continue; }
        nd.valid |= (uint8_t)(1u << k);
        if (r < 0) {
            a.slot_prim[rec] = (uint32_t)~r;
            nd.child[k] = ~(int32_t)(rec << 3);
            rec++;
        } else if (prims_below(a.tree, r) == 2) {
            const int2 two = a.tree.child[r];
            a.slot_prim[rec] = (uint32_t)~two.x; a.slot_prim[rec + 1] = (uint32_t)~two.y;
            nd.child[k] = ~(int32_t)((rec << 3) | 1u);
            rec += 2;
        } else {
            a.items_next[inner] = r; a.stacked_next[inner] = held;
            nd.child[k] = a.next_base + (int32_t)inner;
            inner++;
        }
    }
    *reinterpret_cast<Bvh4Node *>(a.nodes4 + (size_t)(a.level_base + t) * 4) = nd;
    atomicMax(a.stack_need, held);
}

// ---- depth-first numbering.  The collapse numbers nodes and records level by level; a ray, though, goes down: with a
// node's first child right behind it and every subtree in one piece of the arrays, the nodes and records a wave touches
// next are the ones a neighbouring lane has just pulled into the cache (the host builder stores its tree the same way).
// Bottom-up: the nodes and records below each node.  Top-down: each node's place, its links rewritten on the way out.
struct ReorderArgs {
    const uint4 *nodes_in;        // level-order tree
    const uint32_t *slot_prim_in;
    int2 *below;                  // per level-order node: (nodes, records) of its subtree, itself included
    int2 *place;                  // per level-order node: (node number, first record) in depth-first order
    uint4 *nodes_out;
    uint32_t *slot_prim_out;
    int32_t *level_nodes_out;     // this level's nodes under their new numbers (for refit_nodes_kernel)
    int32_t first, count;         // this level's nodes, level-order numbers
};

__global__ __launch_bounds__(kBuildBlock) void reorder_count_kernel(const ReorderArgs a)
{
    const int t = blockIdx.x * kBuildBlock + threadIdx.x;
    if (t >= a.count) return;
    const Bvh4Node nd = *reinterpret_cast<const Bvh4Node *>(a.nodes_in + (size_t)(a.first + t) * 4);
    int nodes = 1, recs = 0;
    for (int k = 0; k < 4; k++) {
        if (!((nd.valid >> k) & 1)) continue;
        if (nd.child[k] >= 0) { const int2 b = a.below[nd.child[k]]; nodes += b.x; recs += b.y; }
        else recs += (int)(((uint32_t)~nd.child[k]) & 7u) + 1;
    }
    a.below[a.first + t] = make_int2(nodes, recs);
}

__global__ __launch_bounds__(kBuildBlock) void reorder_place_kernel(const ReorderArgs a)
{
    const int t = blockIdx.x * kBuildBlock + threadIdx.x;
    if (t >= a.count) return;
    Bvh4Node nd = *reinterpret_cast<const Bvh4Node *>(a.nodes_in + (size_t)(a.first + t) * 4);
    const int2 me = a.place[a.first + t];
    int node = me.x + 1, rec = me.y;
    for (int k = 0; k < 4; k++) {
        if (!((nd.valid >> k) & 1)) continue;
        if (nd.child[k] >= 0) {
            const int c = nd.child[k];
            const int2 b = a.below[c];
            a.place[c] = make_int2(node, rec);
            nd.child[k] = node;
            node += b.x; rec += b.y;
        } else {
            const uint32_t enc = (uint32_t)~nd.child[k], old = enc >> 3, cnt = (enc & 7u) + 1u;
            for (uint32_t q = 0; q < cnt; q++) a.slot_prim_out[(uint32_t)rec + q] = a.slot_prim_in[old + q];
            nd.child[k] = ~(int32_t)(((uint32_t)rec << 3) | (cnt - 1u));
            rec += (int)cnt;
        }
    }
    *reinterpret_cast<Bvh4Node *>(a.nodes_out + (size_t)me.x * 4) = nd;
    a.level_nodes_out[t] = me.x;
}

// The surface-area cost of the four-wide tree as it lies on the device: the sum over nodes of box area - divided by the
// root's (`relative`), the expected number of node visits of a ray that crosses the root's box at random (printed under
// PTRACE_BUILD_TIMING); as it is, the figure a refitted tree is compared with the freshly built one by (option
// "rebuild_percent").  One partial sum per block.
__global__ __launch_bounds__(kBuildBlock) void tree_cost_kernel(const float *node_box, int n_nodes, int relative, float *partial)
{
    __shared__ float sums[kBuildBlock];
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    float a = 0.0f;
    if (i < n_nodes) {
        const float *b = node_box + (size_t)i * 6, *r = node_box;
        const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2], rx = r[3] - r[0], ry = r[4] - r[1], rz = r[5] - r[2];
        a = ex * ey + ey * ez + ez * ex;
        if (relative) a /= rx * ry + ry * rz + rz * rx;
    }
    sums[threadIdx.x] = a;
    __syncthreads();
    for (int d = kBuildBlock / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) sums[threadIdx.x] += sums[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sums[0];
}

// The records in tree order: what build_prims_kernel and refit_prims_kernel made per primitive, moved to its slot.
__global__ __launch_bounds__(kBuildBlock) void build_gather_kernel(const uint32_t *slot_prim, int n, const int4 *src_in, const float4 *prims_in, const float *box_in,
                                                                   const float *clip_in, int4 *src, float4 *prims, float *box, float *clip)
{
    const int s = blockIdx.x * kBuildBlock + threadIdx.x;
    if (s > n) return;
    if (s == n) {             // the record after the last: a triangle without extent, where the links of unused slots point
        src[s] = make_int4(-1, 0, 0, 0);
        for (int k = 0; k < 3; k++) prims[(size_t)s * 3 + k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const uint32_t p = slot_prim[s];
    src[s] = src_in[p];
    for (int k = 0; k < 3; k++) prims[(size_t)s * 3 + k] = prims_in[(size_t)p * 3 + k];
    for (int k = 0; k < 6; k++) box[(size_t)s * 6 + k] = box_in[(size_t)p * 6 + k];
    if (clip_in)
        for (int k = 0; k < 6; k++) clip[(size_t)s * 6 + k] = clip_in[(size_t)p * 6 + k];
}

}  // namespace ptrace
