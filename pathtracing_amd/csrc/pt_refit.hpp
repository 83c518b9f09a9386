// Refit of the four-wide BVH on the device after a change of localToWorld matrices.
//
// The reference re-uploads every vertex, index and matrix whenever a watched transform changes
// (RayTracingMaster.cs:254-262, 371-413) and its shader transforms every triangle per ray
// (RayTracingShader.compute:238-240).  Here the object-space geometry stays on the device: an
// update sends the matrices (64 bytes per object), refit_prims_kernel redoes the world-space
// primitive records and their boxes, and refit_nodes_kernel recomputes the node boxes level by
// level from the leaves up.  The tree's topology is the one of the last full build, so its quality
// decays with large motion (a full build restores it); its answers do not: every box still
// contains its primitives, and the closest hit does not depend on the tree.
#pragma once

#include "bvh.hpp"
#include "pt_device.hpp"

namespace ptrace {

struct RefitArgs {
    const int4 *slot_src;     // per primitive slot: (object index or -1 for a sphere, vertex index 0, 1, 2)
    const float *vertices;    // object-space positions, 3 floats per vertex
    const float *matrices;    // 16 floats per object (UnityEngine.Matrix4x4 memory order)
    float4 *prims;            // the primitive records (pt_device.hpp)
    float *prim_box;          // 6 floats per slot: padded world-space box, lo then hi
    const float *slot_clip;   // 6 floats per slot or null: the part of the triangle's box this reference stands for,
    const float *build_matrices;   // valid while the object's matrix is bit for bit the one of the last full build
    uint4 *nodes4;            // Bvh4Node array
    uint4 *nodes8;            // Bvh8Node array (refit_nodes8_kernel)
    float *node_box;          // 6 floats per node: exact union of its children's boxes
    const int32_t *level_nodes;   // node indices, deepest level first
    float pad;                // absolute padding of primitive boxes (from the last full build)
    float reach;              // largest distance between a ray's origin and a primitive the boxes are padded for
    float sphere_slack2;      // a sphere's box has radius sqrt(r * r + sphere_slack2): the reach of IntersectSphere's rounding
    int32_t n_prims;
};

constexpr int kRefitBlock = 256;

__global__ __launch_bounds__(kRefitBlock) void refit_prims_kernel(const RefitArgs a)
{
    const int slot = blockIdx.x * kRefitBlock + threadIdx.x;
    if (slot >= a.n_prims) return;
    const int4 src = a.slot_src[slot];
    float lo[3], hi[3];
    float own = 0.0f;         // padding this primitive's own intersection arithmetic needs (triangle_fuzz)
    if (src.x >= 0) {
        const float *m = a.matrices + 16 * (size_t)src.x;
        float mm[16];
#pragma unroll
        for (int k = 0; k < 16; k++) mm[k] = m[k];
        const float *pa = a.vertices + 3 * (size_t)src.y, *pb = a.vertices + 3 * (size_t)src.z, *pc = a.vertices + 3 * (size_t)src.w;
        // RayTracingShader.compute:238-240
        const f3 v0 = pt_mul_point(mm, pt_make3(pa[0], pa[1], pa[2]));
        const f3 v1 = pt_mul_point(mm, pt_make3(pb[0], pb[1], pb[2]));
        const f3 v2 = pt_mul_point(mm, pt_make3(pc[0], pc[1], pc[2]));
        const f3 e1 = pt_sub3(v1, v0), e2 = pt_sub3(v2, v0);
        own = triangle_fuzz(e1, e2, a.reach);
        if (!(own < kHugeCoord)) own = kHugeCoord;
        const float idf = a.prims[(size_t)slot * 3].w;
        a.prims[(size_t)slot * 3 + 0] = make_float4(v0.x, v0.y, v0.z, idf);
        a.prims[(size_t)slot * 3 + 1] = make_float4(e1.x, e1.y, e1.z, 0.0f);
        a.prims[(size_t)slot * 3 + 2] = make_float4(e2.x, e2.y, e2.z, __int_as_float(src.x));   // the mesh object's index
        lo[0] = fminf(v0.x, fminf(v1.x, v2.x)); hi[0] = fmaxf(v0.x, fmaxf(v1.x, v2.x));
        lo[1] = fminf(v0.y, fminf(v1.y, v2.y)); hi[1] = fmaxf(v0.y, fmaxf(v1.y, v2.y));
        lo[2] = fminf(v0.z, fminf(v1.z, v2.z)); hi[2] = fmaxf(v0.z, fmaxf(v1.z, v2.z));
        // a NaN coordinate must widen the box, not vanish in fmin/fmax
        if (!(v0.x == v0.x && v1.x == v1.x && v2.x == v2.x)) { lo[0] = -PT_INF; hi[0] = PT_INF; }
        if (!(v0.y == v0.y && v1.y == v1.y && v2.y == v2.y)) { lo[1] = -PT_INF; hi[1] = PT_INF; }
        if (!(v0.z == v0.z && v1.z == v1.z && v2.z == v2.z)) { lo[2] = -PT_INF; hi[2] = PT_INF; }   // made finite below
        if (a.slot_clip) {
            // a split triangle's reference keeps its cut box while its object stays where it was built; a moved object's
            // references all take the whole triangle's box (still correct, and a full build cuts afresh)
            const uint32_t *bm = reinterpret_cast<const uint32_t *>(a.build_matrices + 16 * (size_t)src.x);
            bool same = true;
#pragma unroll
            for (int k = 0; k < 16; k++) same = same && bm[k] == __float_as_uint(mm[k]);
            if (same)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    lo[k] = fmaxf(lo[k], a.slot_clip[(size_t)slot * 6 + k]); hi[k] = fminf(hi[k], a.slot_clip[(size_t)slot * 6 + 3 + k]);
                }
        }
    } else {
        const float4 c = a.prims[(size_t)slot * 3 + 0];
        const float r0 = a.prims[(size_t)slot * 3 + 1].x;
        const float r = sqrtf(r0 * r0 + a.sphere_slack2);
        lo[0] = c.x - r; hi[0] = c.x + r; lo[1] = c.y - r; hi[1] = c.y + r; lo[2] = c.z - r; hi[2] = c.z + r;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float l = lo[k], h = hi[k];
        const float p = fmaxf(a.pad, own);
        const bool usable = fabsf(l) + p < kHugeCoord && fabsf(h) + p < kHugeCoord;   // false for NaN too: the same rule as the host build
        const float rel = 4e-7f * fmaxf(fabsf(l), fabsf(h));
        a.prim_box[(size_t)slot * 6 + k] = usable ? l - p - rel : -kHugeCoord;
        a.prim_box[(size_t)slot * 6 + 3 + k] = usable ? h + p + rel : kHugeCoord;
    }
}

// One four-wide node from its children's boxes (which must be ready: children before parents).
__device__ __forceinline__ void refit_node4(const RefitArgs &a, int node)
{
    Bvh4Node nd = *reinterpret_cast<const Bvh4Node *>(a.nodes4 + (size_t)node * 4);
    float clo[4][3], chi[4][3];
    float plo[3] = {PT_INF, PT_INF, PT_INF}, phi[3] = {-PT_INF, -PT_INF, -PT_INF};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!((nd.valid >> k) & 1)) continue;
        const int32_t link = nd.child[k];
        if (link < 0) {
            const uint32_t enc = (uint32_t)~link, firstp = enc >> 3, cnt = (enc & 7u) + 1u;
            for (int ax = 0; ax < 3; ax++) { clo[k][ax] = PT_INF; chi[k][ax] = -PT_INF; }
            for (uint32_t q = 0; q < cnt; q++)
                for (int ax = 0; ax < 3; ax++) {
                    clo[k][ax] = fminf(clo[k][ax], a.prim_box[(size_t)(firstp + q) * 6 + ax]);
                    chi[k][ax] = fmaxf(chi[k][ax], a.prim_box[(size_t)(firstp + q) * 6 + 3 + ax]);
                }
        } else {
            for (int ax = 0; ax < 3; ax++) { clo[k][ax] = a.node_box[(size_t)link * 6 + ax]; chi[k][ax] = a.node_box[(size_t)link * 6 + 3 + ax]; }
        }
        for (int ax = 0; ax < 3; ax++) { plo[ax] = fminf(plo[ax], clo[k][ax]); phi[ax] = fmaxf(phi[ax], chi[k][ax]); }
    }
    for (int ax = 0; ax < 3; ax++) { a.node_box[(size_t)node * 6 + ax] = plo[ax]; a.node_box[(size_t)node * 6 + 3 + ax] = phi[ax]; }
    // the node's grid: origin = low corner, per axis the smallest power-of-two cell with 255 cells >= extent
    float cell[3];
    uint8_t ebits[3];
    for (int ax = 0; ax < 3; ax++) {
        nd.org[ax] = plo[ax];
        const float ext = phi[ax] - plo[ax];
        int e = -100;
        if (ext > 0.0f && ext < PT_INF) {
            (void)frexpf(ext / 255.0f, &e);
            if (e < -100) e = -100;
            while (plo[ax] + 255.0f * ldexpf(1.0f, e) < phi[ax] && e < 100) e++;
        }
        cell[ax] = ldexpf(1.0f, e);
        ebits[ax] = (uint8_t)(e + 127);
    }
    nd.ex = ebits[0]; nd.ey = ebits[1]; nd.ez = ebits[2];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!((nd.valid >> k) & 1)) {
            for (int ax = 0; ax < 3; ax++) { nd.qlo[ax][k] = 255; nd.qhi[ax][k] = 0; }
            continue;
        }
        for (int ax = 0; ax < 3; ax++) {
            const float fl = floorf((clo[k][ax] - plo[ax]) / cell[ax]), fh = ceilf((chi[k][ax] - plo[ax]) / cell[ax]);
            int ql = (int)fminf(255.0f, fmaxf(0.0f, fl)), qh = (int)fminf(255.0f, fmaxf(0.0f, fh));
            if (!(fl == fl)) ql = 0;                       // an infinite or NaN extent: the whole grid
            if (!(fh == fh)) qh = 255;
            while (ql > 0 && plo[ax] + (float)ql * cell[ax] > clo[k][ax]) ql--;     // outward rounding survives
            while (qh < 255 && plo[ax] + (float)qh * cell[ax] < chi[k][ax]) qh++;   // the float arithmetic
            nd.qlo[ax][k] = (uint8_t)ql; nd.qhi[ax][k] = (uint8_t)qh;
        }
    }
    *reinterpret_cast<Bvh4Node *>(a.nodes4 + (size_t)node * 4) = nd;
}

// One thread per node of one tree level (children first, so every child's box is ready).
__global__ __launch_bounds__(kRefitBlock) void refit_nodes_kernel(const RefitArgs a, int first, int count)
{
    const int t = blockIdx.x * kRefitBlock + threadIdx.x;
    if (t >= count) return;
    refit_node4(a, a.level_nodes[first + t]);
}

// ---- shared meshes expanded on the device (option "instancing" 2) ----------------------------------------------------
// A mesh that several objects share is built once, on the host, as a tree over its object-space triangles; the device
// then makes one copy of that tree per object - node links and record numbers shifted to the copy's place - and refits
// every copy under its object's matrix, exactly as a transform update refits a flattened scene.  The result is an ordinary
// flattened tree in world space (no instance transform during traversal) that never existed on the host: ten million
// triangles in 38 copies cost the host one 262 k-triangle build and the device a few milliseconds.
struct ExpandArgs {
    const uint4 *mesh_nodes;      // the mesh tree's Bvh4Node array, links relative to the mesh
    const int4 *mesh_slots;       // per record slot of the mesh tree: (i0, i1, i2 minus the smallest index of the mesh's range, triangle number)
    const int4 *inst;             // per copy: (first node, first record, mesh object, smallest vertex index of the object's range)
    const int32_t *inst_id0;      // per copy: id of the object's first triangle
    int32_t n_mesh_nodes, n_mesh_slots, n_inst;
    uint4 *nodes4;
    float4 *prims;
    int4 *slot_src;
};

__global__ __launch_bounds__(kRefitBlock) void expand_nodes_kernel(const ExpandArgs a)
{
    const size_t t = (size_t)blockIdx.x * kRefitBlock + threadIdx.x;
    if (t >= (size_t)a.n_inst * (size_t)a.n_mesh_nodes) return;
    const int inst = (int)(t / (size_t)a.n_mesh_nodes), k = (int)(t - (size_t)inst * (size_t)a.n_mesh_nodes);
    const int4 in = a.inst[inst];
    Bvh4Node nd = *reinterpret_cast<const Bvh4Node *>(a.mesh_nodes + (size_t)k * 4);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (nd.child[c] >= 0) nd.child[c] += in.x;
        else { const uint32_t enc = (uint32_t)~nd.child[c]; nd.child[c] = ~(int32_t)((((enc >> 3) + (uint32_t)in.y) << 3) | (enc & 7u)); }
    }
    *reinterpret_cast<Bvh4Node *>(a.nodes4 + ((size_t)in.x + (size_t)k) * 4) = nd;
}

// The copies' records: where each comes from (for refit_prims_kernel, which fills in the world-space triangle) and its id;
// one more record per copy stays zero - the triangle nothing hits that the links of unused node slots point at.
__global__ __launch_bounds__(kRefitBlock) void expand_slots_kernel(const ExpandArgs a)
{
    const size_t per = (size_t)a.n_mesh_slots + 1;
    const size_t t = (size_t)blockIdx.x * kRefitBlock + threadIdx.x;
    if (t >= (size_t)a.n_inst * per) return;
    const int inst = (int)(t / per), s = (int)(t - (size_t)inst * per);
    const int4 in = a.inst[inst];
    const size_t rec = (size_t)in.y + (size_t)s;
    float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int4 src = make_int4(-1, 0, 0, 0);
    if (s < a.n_mesh_slots) {
        const int4 ms = a.mesh_slots[s];
        src = make_int4(in.z, ms.x + in.w, ms.y + in.w, ms.z + in.w);
        r0.w = __int_as_float(a.inst_id0[inst] + ms.w);
    }
    a.slot_src[rec] = src;
    a.prims[rec * 3 + 0] = r0;
    a.prims[rec * 3 + 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    a.prims[rec * 3 + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// One level of the mesh tree in every copy at once: thread t refits node first_node(copy) + level_nodes[first + k].
__global__ __launch_bounds__(kRefitBlock) void refit_nodes_copies_kernel(const RefitArgs a, const int4 *inst, int n_inst, int first, int count)
{
    const size_t t = (size_t)blockIdx.x * kRefitBlock + threadIdx.x;
    if (t >= (size_t)n_inst * (size_t)count) return;
    const int i = (int)(t / (size_t)count), k = (int)(t - (size_t)i * (size_t)count);
    refit_node4(a, inst[i].x + a.level_nodes[first + k]);
}

// The same for the eight-wide tree: the children of a node are its inner nodes from child_base on and the records of
// its leaves (one each) from leaf_base on, both in slot order (bvh.hpp, Bvh8Node).
__global__ __launch_bounds__(kRefitBlock) void refit_nodes8_kernel(const RefitArgs a, int first, int count)
{
    const int t = blockIdx.x * kRefitBlock + threadIdx.x;
    if (t >= count) return;
    const int node = a.level_nodes[first + t];
    Bvh8Node nd = *reinterpret_cast<const Bvh8Node *>(a.nodes8 + (size_t)node * 5);
    float clo[8][3], chi[8][3];
    float plo[3] = {PT_INF, PT_INF, PT_INF}, phi[3] = {-PT_INF, -PT_INF, -PT_INF};
    uint32_t n_inner = 0, n_leaf = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float *src;
        if ((nd.imask >> k) & 1) src = a.node_box + (size_t)(nd.child_base + n_inner++) * 6;
        else if ((nd.lmask >> k) & 1) src = a.prim_box + (size_t)(nd.leaf_base + n_leaf++) * 6;
        else continue;
        for (int ax = 0; ax < 3; ax++) {
            clo[k][ax] = src[ax]; chi[k][ax] = src[3 + ax];
            plo[ax] = fminf(plo[ax], clo[k][ax]); phi[ax] = fmaxf(phi[ax], chi[k][ax]);
        }
    }
    for (int ax = 0; ax < 3; ax++) { a.node_box[(size_t)node * 6 + ax] = plo[ax]; a.node_box[(size_t)node * 6 + 3 + ax] = phi[ax]; }
    float cell[3];
    uint8_t ebits[3];
    for (int ax = 0; ax < 3; ax++) {
        nd.org[ax] = plo[ax];
        const float ext = phi[ax] - plo[ax];
        int e = -100;
        if (ext > 0.0f && ext < PT_INF) {
            (void)frexpf(ext / 255.0f, &e);
            if (e < -100) e = -100;
            while (plo[ax] + 255.0f * ldexpf(1.0f, e) < phi[ax] && e < 100) e++;
        }
        cell[ax] = ldexpf(1.0f, e);
        ebits[ax] = (uint8_t)(e + 127);
    }
    nd.ex = ebits[0]; nd.ey = ebits[1]; nd.ez = ebits[2];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint8_t ql[3] = {255, 255, 255}, qh[3] = {0, 0, 0};
        if (((nd.imask | nd.lmask) >> k) & 1)
            for (int ax = 0; ax < 3; ax++) {
                const float fl = floorf((clo[k][ax] - plo[ax]) / cell[ax]), fh = ceilf((chi[k][ax] - plo[ax]) / cell[ax]);
                int l = (int)fminf(255.0f, fmaxf(0.0f, fl)), h = (int)fminf(255.0f, fmaxf(0.0f, fh));
                if (!(fl == fl)) l = 0;                        // an infinite or NaN extent: the whole grid
                if (!(fh == fh)) h = 255;
                while (l > 0 && plo[ax] + (float)l * cell[ax] > clo[k][ax]) l--;
                while (h < 255 && plo[ax] + (float)h * cell[ax] < chi[k][ax]) h++;
                ql[ax] = (uint8_t)l; qh[ax] = (uint8_t)h;
            }
        nd.qlo_x[k] = ql[0]; nd.qlo_y[k] = ql[1]; nd.qlo_z[k] = ql[2];
        nd.qhi_x[k] = qh[0]; nd.qhi_y[k] = qh[1]; nd.qhi_z[k] = qh[2];
    }
    *reinterpret_cast<Bvh8Node *>(a.nodes8 + (size_t)node * 5) = nd;
}

}  // namespace ptrace
