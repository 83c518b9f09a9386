// Host-side BVH construction for libptrace.
//
// The reference has no acceleration structure: RayTracingShader.compute:232-257
// loops over every triangle for every ray.  This builder produces a binary BVH
// over the scene's spheres and world-space triangles whose closest hit is, by
// construction of the traversal (padded boxes, (distance, id) ordering), the hit
// that loop returns.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace ptrace {

struct Aabb {
    float lo[3], hi[3];
};

// Device node, 64 bytes = four float4 loads:
//   q0 = (left.lo.xyz,  bits(left))   q1 = (left.hi.xyz,  bits(right))
//   q2 = (right.lo.xyz, 0)            q3 = (right.hi.xyz, 0)
// child >= 0: index of an inner node.  child < 0: leaf, ~child = (first << 3) | (count - 1)
// over the reordered primitive records.  An absent child has an inverted box
// (lo = +inf, hi = -inf) that no ray enters.
struct alignas(64) BvhNode {
    float llo[3]; int32_t left;
    float lhi[3]; int32_t right;
    float rlo[3]; int32_t pad0;
    float rhi[3]; int32_t pad1;
};

// A primitive with a NaN or infinite coordinate (or one beyond this magnitude) gets the box [-kHugeCoord, kHugeCoord] on that
// axis: finite, so that centroids, extents and the quantised node grids stay well defined, and so large that every ray of a
// real scene passes through it.  With the traversal's reciprocal direction clamped at 1e18 the products stay below 3.4e38.
constexpr float kHugeCoord = 1e18f;

constexpr int kMaxLeafPrims = 8;   // what the 3-bit count field can hold
constexpr int kMaxDepth = 60;      // traversal stack capacity bounds the depth

// Wide device node, 64 bytes (56 used): four children whose boxes are quantised to 8 bits per plane
// on a per-node grid (origin + power-of-two cell per axis), rounded outwards so a quantised box
// always contains the exact one.  A slot without a child (valid bit clear) holds the inside-out box
// qlo = 255, qhi = 0, which the slab test rejects by itself, and a link to a leaf of one record: record number
// boxes.size(), one past the last primitive, which the caller provides filled with zeros (nothing hits it).
//   q0 = (org.x, org.y, org.z, ex | ey << 8 | ez << 16 | valid << 24)    cell = 2^(e - 127)
//   q1 = child[0..3]            same encoding as BvhNode's links
//   q2 = (qlo.x[4], qlo.y[4], qlo.z[4], qhi.x[4])   one byte per child
//   q3 = (qhi.y[4], qhi.z[4], -, -)
struct alignas(64) Bvh4Node {
    float org[3];
    uint8_t ex, ey, ez, valid;
    int32_t child[4];
    uint8_t qlo[3][4];
    uint8_t qhi[3][4];
    uint32_t pad[2];
};
static_assert(sizeof(Bvh4Node) == 64, "Bvh4Node layout");

// Eight-wide device node, 80 bytes = five 16-byte loads.  Children whose boxes are quantised like Bvh4Node's; a child is an
// inner node (its bit in `imask`), a leaf of exactly ONE primitive record (its bit in `lmask`) or absent (neither bit; the
// slot holds the inside-out box qlo = 255, qhi = 0).  A node's inner children are consecutive in the node array from
// `child_base`, its leaf children's records consecutive in the record array from `leaf_base`, both in slot order: the
// child in slot s is node child_base + popcount(imask & ((1 << s) - 1)), or record leaf_base + popcount(lmask & ...).
// Slots are assigned by position: bit a of the slot number says on which side of the node's centre the child lies along
// axis a, so that for a ray whose direction has the sign bits `oct` (bit a set: negative along a) the order
// slot ^ oct = 0, 1, ... 7 runs roughly front to back.  The traversal keeps (base, masks) pairs instead of links.
//   q0 = (org.x, org.y, org.z, ex | ey << 8 | ez << 16 | imask << 24)
//   q1 = (child_base, leaf_base, lmask, 0)
//   q2 = (qlo.x[0..3], qlo.x[4..7], qlo.y[0..3], qlo.y[4..7])
//   q3 = (qlo.z[0..3], qlo.z[4..7], qhi.x[0..3], qhi.x[4..7])
//   q4 = (qhi.y[0..3], qhi.y[4..7], qhi.z[0..3], qhi.z[4..7])
struct alignas(16) Bvh8Node {
    float org[3];
    uint8_t ex, ey, ez, imask;
    uint32_t child_base, leaf_base;
    uint32_t lmask;                   // low byte
    uint32_t pad;
    uint8_t qlo_x[8], qlo_y[8], qlo_z[8], qhi_x[8], qhi_y[8], qhi_z[8];
};
static_assert(sizeof(Bvh8Node) == 80, "Bvh8Node layout");

struct BvhResult {
    std::vector<BvhNode> nodes;       // nodes[0] is the root (always an inner node) unless empty
    std::vector<Bvh4Node> nodes4;     // the same tree collapsed to four children per node; nodes4[0] is the root
    int depth4 = 0;
    std::vector<uint32_t> order;      // order[k] = input primitive index of reordered slot k
    int depth = 0;
    // Most traversal-stack entries a ray can hold at once: on its way down a ray that enters every child of every node
    // stacks all siblings of the child it follows (children - 1 per node).  stack_need2 <= depth <= kMaxDepth + 1; the
    // four-wide tree's figure depends on how the collapse fell and can approach 3 * kMaxDepth for lopsided trees.
    int stack_need2 = 0, stack_need4 = 0;
    // The eight-wide tree (build_bvh with wide8 set; leaves hold one primitive each).  Its records follow order8, not
    // `order`: the records of a node's leaf children are consecutive.  A ray holds at most one stack entry per level
    // (what is left of the group of children it descended from) plus one: stack_need8 = depth8 + 1.
    std::vector<Bvh8Node> nodes8;
    std::vector<uint32_t> order8;
    int depth8 = 0, stack_need8 = 0;
};

// boxes: one (already padded) box per primitive.  leaf_size <= kMaxLeafPrims.
// threads <= 0: pick from hardware_concurrency (capped at 16).
// traversal_cost: SAH cost of visiting a node, in primitive tests; ranges of <= leaf_size primitives become
// leaves when that is estimated cheaper than splitting them.
// wide8: also collapse the tree to eight children per node (nodes8, order8); the two-child tree is then built with one
// primitive per leaf whatever leaf_size says, and `nodes4` / `order` describe that same tree.
BvhResult build_bvh(const std::vector<Aabb> &boxes, int leaf_size, int threads, float traversal_cost = 0.5f, bool wide8 = false);

// Measurement layout (-DPT_NODES_3LOAD): the four-wide tree renumbered so that a node's inner children are consecutive
// nodes and its leaf children's records consecutive records, both in slot order; a node then needs no links - a first
// child, a first record and two bits per slot (0 absent, 1 inner node, 2 / 3 leaf of one / two records) - and fits 48
// bytes, three 16-byte loads instead of four.  Rewrites nodes4 (same structs, links renumbered) and `order`; fills `kinds`
// (one byte per node) and `first_child` / `first_record`.  Leaves of more than two records are not representable.
struct Consecutive4 { std::vector<uint8_t> kinds; std::vector<uint32_t> first_child, first_record; bool ok = false; };
Consecutive4 consecutive_layout(BvhResult &bvh);

// Reference splitting before the build: a large triangle's box is cut at spatial-median planes of the scene box into
// several smaller boxes ("references"), each holding the part of the triangle on its side of the cuts, so that a long or
// large triangle no longer inflates every node on the way down to it.  The tree is then built over references; a leaf
// names the whole triangle, so a ray that enters any reference's box tests the triangle itself and the closest hit is
// unchanged.  Which triangles are cut, how often and where follows Karras & Aila, "Fast parallel construction of
// high-quality bounding volume hierarchies" (HPG 2013), section 4: priority cbrt(2^-level * (A_box - A_ideal)), cuts at the
// most important median plane crossing the box, a budget of `budget` * n_tris additional references in total.
//   tri9: world-space vertices, 9 floats per triangle; boxes: each triangle's exact (unpadded) box;
//   ref_tri / ref_box: appended, one entry per reference (triangles that cannot be cut - a NaN, a coordinate beyond
//   kHugeCoord - get their own box back).  Boxes are rounded outwards; the union of a triangle's boxes contains it.
void split_references(const float *tri9, const Aabb *boxes, size_t n_tris, const float scene_lo[3], const float scene_hi[3],
                      float budget, std::vector<uint32_t> &ref_tri, std::vector<Aabb> &ref_box);

}  // namespace ptrace
