// The two-child tree of the device build, top-down by the surface-area heuristic - the host builder's algorithm (bvh.cpp:
// binned SAH) laid out for the GPU.
//
// Clustering from the leaves up (pt_build.hpp, ploc_*) is quick, but greedy: on the 262 k-triangle hall its tree costs
// 12 % more node visits per ray than the host's, and neither rebuilding the insides of its clusters nor handing their tops
// to the host's builder recovers more than a quarter of that - which primitives end up together is decided badly, at every
// scale.  Splitting from the root down decides exactly that, with the whole range in view.
//
//   upper levels, one pass per level over all primitives (they lie in Morton order, and a stable partition keeps every
//   node's primitives one contiguous, still Morton-ordered range):
//     sah_centres_kernel    the bounds of the box centres of the root's primitives: the range its bins are laid over (for the
//                           levels below, sah_side_kernel collects them while it sorts the primitives into children)
//     sah_bin_kernel        every primitive of a node with more than kSahSmall adds its box to one of 32 bins per axis of
//                           its node (by its centre's place among the node's centres); a wave collects in LDS bins of its
//                           own what goes to the node it is in, and sends them on when it comes to the next node
//     sah_split_kernel      per node: the cheapest of the 93 bin boundaries, area(left) n_left + area(right) n_right; the
//                           two children get numbers, boxes (exact: unions of bins) and ranges
//     sah_next_kernel       children with more than kSahSmall primitives form the next level, the others go on a list
//     sah_side_kernel, scan, sah_partition_kernel   left primitives to the front of the node's range, order kept
//   lower levels: sah_small_kernel, one wave per listed node, builds the whole subtree of its <= kSahSmall primitives in LDS
//   by the same rule (ranges of kSahExhaustive or fewer: by trying every split) - no global traffic but the primitives in
//   and the nodes out.
// Min, max and integer add commute exactly, so the bins - and the tree - do not depend on the order the atomics land in.
#pragma once

#include "pt_build.hpp"

namespace ptrace {

constexpr int kSahBins = 32;           // the host builder's count (bvh.cpp, kBins)
constexpr int kSahBinWords = 3 * kSahBins * 7;       // per node: [axis][bin] (lo.xyz as ordered words, hi.xyz, count)
constexpr int kSahSmall = 128;
constexpr int kSahExhaustive = 8;       // ranges this small are split by trying every two-part split

struct SahRange {
    float lo[3]; int32_t begin;        // box of the range's primitives; first place in the primitive array
    float hi[3]; int32_t count;
    int32_t node, parent, side, pad;   // its number in the two-child tree; where the link to it sits
};
static_assert(sizeof(SahRange) == 48, "SahRange layout");

__device__ __forceinline__ int sah_bin(float lo, float hi, float range_lo, float scale)
{
    const float x = (0.5f * lo + 0.5f * hi - range_lo) * scale;
    return x >= 0.0f ? min(kSahBins - 1, (int)x) : 0;
}
__device__ __forceinline__ float sah_scale(float lo, float hi)
{
    const float ext = hi - lo;
    return ext > 0.0f && ext < PT_INF ? (float)kSahBins / ext : 0.0f;
}

// The box of all primitives: 6 ordered words, lo from 0xffffffff, hi from 0.  (Chunks per wave: pt_build.hpp, kWaveChunk.)
__global__ __launch_bounds__(kBuildBlock) void sah_bounds_kernel(const float4 *p_lo, const float4 *p_hi, int n, uint32_t *bounds)
{
    const int lane = threadIdx.x & 63;
    const size_t chunk = ((size_t)blockIdx.x * (kBuildBlock / 64) + (threadIdx.x >> 6)) * kWaveChunk;
    if (chunk >= (size_t)n) return;
    float l[3] = {PT_INF, PT_INF, PT_INF}, h[3] = {-PT_INF, -PT_INF, -PT_INF};
    for (int step = 0; step < kWaveChunk / 64; step++) {
        const size_t i = chunk + (size_t)step * 64 + lane;
        if (i >= (size_t)n) break;
        const float4 a = p_lo[i], b = p_hi[i];
        l[0] = fminf(l[0], a.x); l[1] = fminf(l[1], a.y); l[2] = fminf(l[2], a.z);
        h[0] = fmaxf(h[0], b.x); h[1] = fmaxf(h[1], b.y); h[2] = fmaxf(h[2], b.z);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float wl = wave_min(l[k]), wh = wave_max(h[k]);
        if (lane == 0) { atomicMin(bounds + k, ordered_bits(wl)); atomicMax(bounds + 3 + k, ordered_bits(wh)); }
    }
}

__global__ __launch_bounds__(kBuildBlock) void sah_fill_kernel(uint32_t *bins, size_t words)
{
    const size_t w = (size_t)blockIdx.x * kBuildBlock + threadIdx.x;
    if (w >= words) return;
    const uint32_t k = (uint32_t)(w % 7);
    bins[w] = k < 3 ? 0xffffffffu : 0u;
}

// cb[a]: the bounds of the box CENTRES of node a's primitives, 6 ordered words (lo from 0xffffffff, hi from 0) - the range the
// bins are laid over.  (Over the node's box instead, the host builder's tree of the 262 k-triangle hall costs 15.80 node visits
// per ray instead of 14.97: half a primitive of slack at either end is a bin or two of sixteen once nodes get small.)  A node's
// primitives are contiguous, so nearly every wave holds one node and sends six atomics.
__global__ __launch_bounds__(kBuildBlock) void sah_fill_centres_kernel(uint32_t *cb, int words)
{
    const int w = blockIdx.x * kBuildBlock + threadIdx.x;
    if (w < words) cb[w] = (w % 6) < 3 ? 0xffffffffu : 0u;
}

__global__ __launch_bounds__(kBuildBlock) void sah_centres_kernel(const float4 *p_lo, const float4 *p_hi, const int32_t *seg, int n, uint32_t *cb)
{
    const int lane = threadIdx.x & 63;
    const size_t chunk = ((size_t)blockIdx.x * (kBuildBlock / 64) + (threadIdx.x >> 6)) * kWaveChunk;
    if (chunk >= (size_t)n) return;
    float l[3] = {PT_INF, PT_INF, PT_INF}, h[3] = {-PT_INF, -PT_INF, -PT_INF};      // this lane's share of node `held`
    int held = -1;                                                                   // the same in every lane
    const auto flush = [&]() {
        if (held < 0) return;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float a = wave_min(l[k]), b = wave_max(h[k]);
            if (lane == 0 && a <= b) { atomicMin(cb + (size_t)held * 6 + k, ordered_bits(a)); atomicMax(cb + (size_t)held * 6 + 3 + k, ordered_bits(b)); }
            l[k] = PT_INF; h[k] = -PT_INF;
        }
    };
    for (int step = 0; step < kWaveChunk / 64; step++) {
        const size_t first = chunk + (size_t)step * 64;
        if (first >= (size_t)n) break;
        const size_t i = min(first + (size_t)lane, (size_t)n - 1);                   // a duplicate changes no minimum
        const int a = seg[i];
        const bool one_node = __all(a == __builtin_amdgcn_readfirstlane(a));
        if (!one_node || a != held) { flush(); held = one_node ? a : -1; }
        if (a < 0) continue;
        const float4 pl = p_lo[i], ph = p_hi[i];
        const float c[3] = {0.5f * pl.x + 0.5f * ph.x, 0.5f * pl.y + 0.5f * ph.y, 0.5f * pl.z + 0.5f * ph.z};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (one_node) { l[k] = fminf(l[k], c[k]); h[k] = fmaxf(h[k], c[k]); }
            else { atomicMin(cb + (size_t)a * 6 + k, ordered_bits(c[k])); atomicMax(cb + (size_t)a * 6 + 3 + k, ordered_bits(c[k])); }
        }
    }
    flush();
}

__device__ __forceinline__ float ordered_to_float(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu)); }

// seg[i]: this level's node of the primitive at place i (an index into `act`), or -1 once it lies in a listed node.
// Every wave takes kBinChunk consecutive places, 64 at a time, and collects in LDS bins of its own what goes to its `home`
// node - the node all 64 places of a step belong to; when that changes, the bins are sent to the old home's global bins and
// start again.  Nodes are contiguous and hold more than kSahSmall primitives, so nearly everything goes through LDS at every
// level: a few dozen global atomics per node and wave instead of 21 per primitive.  A step that straddles nodes is taken node
// by node.  (Only the wave itself touches its bins: no block barrier.)
// The smallest / largest value among the 16 lanes of a row (lanes 16 r .. 16 r + 15), in all of them: four exchanges through the
// data-parallel-primitive lane selects (neighbour, other pair of the quad, mirror within 8, mirror within 16).  Every lane
// of the row must be active.
__device__ __forceinline__ uint32_t row_umin(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ uint32_t row_umax(uint32_t v)
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));
    return v;
}

constexpr int kBinChunk = 64 * 8;
__global__ __launch_bounds__(kBuildBlock) void sah_bin_kernel(const float4 *p_lo, const float4 *p_hi, const int32_t *seg, int n, const uint32_t *cb, uint32_t *bins)
{
    __shared__ uint32_t local_bins[kBuildBlock / 64][kSahBinWords];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *local = local_bins[wave];
    const size_t first = ((size_t)blockIdx.x * (kBuildBlock / 64) + wave) * kBinChunk;
    if (first >= (size_t)n) return;
    for (int w = lane; w < kSahBinWords; w += 64) local[w] = (w % 7) < 3 ? 0xffffffffu : 0u;
    int home = -1;
    const auto flush = [&]() {          // the wave's bins into home's, and empty again
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (home >= 0) {
            uint32_t *dst = bins + (size_t)home * kSahBinWords;
            for (int w = lane; w < kSahBinWords; w += 64) {
                const uint32_t v = local[w], k = (uint32_t)(w % 7);
                if (k < 3) { if (v != 0xffffffffu) { atomicMin(dst + w, v); local[w] = 0xffffffffu; } }
                else if (v) { if (k < 6) atomicMax(dst + w, v); else atomicAdd(dst + w, v); local[w] = 0u; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    };
    for (int step = 0; step < kBinChunk / 64; step++) {
        const size_t at = first + (size_t)step * 64;
        if (at >= (size_t)n) break;
        const size_t i = at + lane;
        const int a = i < (size_t)n ? seg[i] : -1;
        float l[3] = {0, 0, 0}, h[3] = {0, 0, 0};
        uint32_t ol[3] = {0, 0, 0}, oh[3] = {0, 0, 0};
        if (a >= 0) {
            const float4 pl = p_lo[i], ph = p_hi[i];
            l[0] = pl.x; l[1] = pl.y; l[2] = pl.z; h[0] = ph.x; h[1] = ph.y; h[2] = ph.z;
#pragma unroll
            for (int k = 0; k < 3; k++) { ol[k] = ordered_bits(l[k]); oh[k] = ordered_bits(h[k]); }
        }
        // the step's places node by node (a node's places are consecutive: one node nearly always, a few where small nodes meet)
        unsigned long long todo = __ballot(a >= 0);
        while (todo) {
            const int node = __shfl(a, __ffsll((long long)todo) - 1, 64);
            if (node != home) { flush(); home = node; }
            // Neighbours in Morton order mostly fall into one bin, and 64 atomics on one LDS word take their turns: where the 16
            // lanes of a row all belong to the node and agree on an axis' bin, the row is reduced in registers first and one
            // lane sends it.
            const bool mine = a == node;
            const uint32_t *r = cb + (size_t)node * 6;
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                const float rl = ordered_to_float(r[ax]), rh = ordered_to_float(r[3 + ax]);
                const uint32_t bin = mine ? (uint32_t)sah_bin(l[ax], h[ax], rl, sah_scale(rl, rh)) : 0xffffffffu;
                const uint32_t bmax = row_umax(bin);
                if (bmax != 0xffffffffu && row_umin(bin) == bmax) {
                    uint32_t rlo[3], rhi[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) { rlo[k] = row_umin(ol[k]); rhi[k] = row_umax(oh[k]); }
                    if ((lane & 15) == 0) {
                        uint32_t *w = local + (ax * kSahBins + (int)bin) * 7;
#pragma unroll
                        for (int k = 0; k < 3; k++) { atomicMin(w + k, rlo[k]); atomicMax(w + 3 + k, rhi[k]); }
                        atomicAdd(w + 6, 16u);
                    }
                } else if (mine) {
                    uint32_t *w = local + (ax * kSahBins + (int)bin) * 7;
#pragma unroll
                    for (int k = 0; k < 3; k++) { atomicMin(w + k, ol[k]); atomicMax(w + 3 + k, oh[k]); }
                    atomicAdd(w + 6, 1u);
                }
            }
            todo &= ~__ballot(a == node);
        }
    }
    flush();
}

struct SahBox { float l[3], h[3]; uint32_t n; };
__device__ __forceinline__ void sah_box_reset(SahBox &b) { for (int k = 0; k < 3; k++) { b.l[k] = PT_INF; b.h[k] = -PT_INF; } b.n = 0; }
__device__ __forceinline__ void sah_box_add(SahBox &b, const uint32_t *w)
{
    if (!w[6]) return;
    for (int k = 0; k < 3; k++) { b.l[k] = fminf(b.l[k], ordered_to_float(w[k])); b.h[k] = fmaxf(b.h[k], ordered_to_float(w[3 + k])); }
    b.n += w[6];
}
__device__ __forceinline__ float sah_box_cost(const SahBox &b)
{
    const float ex = b.h[0] - b.l[0], ey = b.h[1] - b.l[1], ez = b.h[2] - b.l[2];
    return (ex * ey + ey * ez + ez * ex) * (float)b.n;
}

// The cheapest of the 93 splits "bins 0 .. s of an axis against bins s + 1 .. 31", found by a whole wave: two rounds of 32-lane
// halves, a half per axis; every lane takes a bin, prefix and suffix unions run over the half (five shuffle steps each), and
// the lane of bin s prices its split.  Candidate c = axis * 31 + s; of equal costs the smallest c (the order a loop over the
// axes and bins would meet them in).  The result is the same in every lane; cost is PT_INF when no split has primitives on
// both sides and a finite price.
struct SahPick { float cost; int cand; SahBox left, right; };
__device__ __forceinline__ SahPick sah_pick_split(const uint32_t *bins, int lane)
{
    static_assert(kSahBins == 32, "one bin per lane of a half-wave");
    SahPick pick;
    pick.cost = PT_INF; pick.cand = 3 * (kSahBins - 1);
    sah_box_reset(pick.left); sah_box_reset(pick.right);
    for (int round = 0; round < 2; round++) {
        const int ax = 2 * round + (lane >> 5), bin = lane & 31;
        SahBox pre; sah_box_reset(pre);
        if (ax < 3) sah_box_add(pre, bins + (ax * kSahBins + bin) * 7);
        SahBox suf = pre;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            SahBox up, down;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                up.l[q] = __shfl_up(pre.l[q], d, 32); up.h[q] = __shfl_up(pre.h[q], d, 32);
                down.l[q] = __shfl_down(suf.l[q], d, 32); down.h[q] = __shfl_down(suf.h[q], d, 32);
            }
            up.n = __shfl_up(pre.n, d, 32); down.n = __shfl_down(suf.n, d, 32);
            if (bin >= d) { for (int q = 0; q < 3; q++) { pre.l[q] = fminf(pre.l[q], up.l[q]); pre.h[q] = fmaxf(pre.h[q], up.h[q]); } pre.n += up.n; }
            if (bin + d < 32) { for (int q = 0; q < 3; q++) { suf.l[q] = fminf(suf.l[q], down.l[q]); suf.h[q] = fmaxf(suf.h[q], down.h[q]); } suf.n += down.n; }
        }
        SahBox right;            // bins bin + 1 .. 31: the neighbour's suffix
#pragma unroll
        for (int q = 0; q < 3; q++) { right.l[q] = __shfl_down(suf.l[q], 1, 32); right.h[q] = __shfl_down(suf.h[q], 1, 32); }
        right.n = __shfl_down(suf.n, 1, 32);
        if (ax < 3 && bin < kSahBins - 1 && pre.n && right.n) {
            const float cost = sah_box_cost(pre) + sah_box_cost(right);
            if (cost < pick.cost) { pick.cost = cost; pick.cand = ax * (kSahBins - 1) + bin; pick.left = pre; pick.right = right; }
        }
    }
    float best = pick.cost; int best_cand = pick.cand;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float oc = __shfl_xor(best, d, 64); const int ocand = __shfl_xor(best_cand, d, 64);
        if (oc < best || (oc == best && ocand < best_cand)) { best = oc; best_cand = ocand; }
    }
    // the boxes of the winning split, from the lane that priced it
    const int winner = ((best_cand / (kSahBins - 1)) & 1) * 32 + best_cand % (kSahBins - 1);
    pick.cost = best; pick.cand = best_cand;
#pragma unroll
    for (int q = 0; q < 3; q++) {
        pick.left.l[q] = __shfl(pick.left.l[q], winner, 64); pick.left.h[q] = __shfl(pick.left.h[q], winner, 64);
        pick.right.l[q] = __shfl(pick.right.l[q], winner, 64); pick.right.h[q] = __shfl(pick.right.h[q], winner, 64);
    }
    pick.left.n = __shfl(pick.left.n, winner, 64); pick.right.n = __shfl(pick.right.n, winner, 64);
    return pick;
}

// split[a] = (axis or -1, last bin of the left part, primitives on the left, 0).  Candidates 2 a and 2 a + 1 are the
// children; flags says which of them form the next level (low word) and which go on the list (high word).  One wave per node.
__global__ __launch_bounds__(kBuildBlock) void sah_split_kernel(const SahRange *act, int n_act, const uint32_t *bins, int node_base, TwoChildTree tree, int4 *split,
                                                                SahRange *cand, uint64_t *flags)
{
    const int lane = threadIdx.x & 63;
    const int a = blockIdx.x * (kBuildBlock / 64) + (threadIdx.x >> 6);
    if (a >= n_act) return;
    const SahRange r = act[a];
    const SahPick pick = sah_pick_split(bins + (size_t)a * kSahBinWords, lane);
    if (lane != 0) return;
    int best_axis = -1, best_bin = 0;
    SahBox left = pick.left, right = pick.right;
    if (pick.cost < PT_INF) { best_axis = pick.cand / (kSahBins - 1); best_bin = pick.cand % (kSahBins - 1); }
    else {
        // every centre in one bin on every axis: halve the range as it lies; the children's boxes are the parent's (they
        // hold what is in them, which is all a box has to do)
        for (int k = 0; k < 3; k++) { left.l[k] = right.l[k] = r.lo[k]; left.h[k] = right.h[k] = r.hi[k]; }
        left.n = (uint32_t)r.count / 2; right.n = (uint32_t)r.count - left.n;
    }
    split[a] = make_int4(best_axis, best_bin, (int)left.n, 0);
    for (int s = 0; s < 2; s++) {
        const SahBox &b = s ? right : left;
        SahRange c;
        for (int k = 0; k < 3; k++) { c.lo[k] = b.l[k]; c.hi[k] = b.h[k]; }
        c.begin = r.begin + (s ? (int)left.n : 0); c.count = (int)b.n;
        c.node = node_base + 2 * a + s; c.parent = r.node; c.side = s; c.pad = 0;
        cand[2 * a + s] = c;
        flags[2 * a + s] = c.count > kSahSmall ? (uint64_t)1 : (uint64_t)1 << 32;
    }
    tree.child[r.node] = make_int2(node_base + 2 * a, node_base + 2 * a + 1);
    tree.lo[r.node] = make_float4(r.lo[0], r.lo[1], r.lo[2], __int_as_float(r.node));
    tree.hi[r.node] = make_float4(r.hi[0], r.hi[1], r.hi[2], __int_as_float(r.count));
}

// offs = exclusive prefix sums of flags.  slot[c]: candidate c's index in the next level, or -1.
__global__ __launch_bounds__(kBuildBlock) void sah_next_kernel(const SahRange *cand, int n_cand, const uint64_t *flags, const uint64_t *offs, SahRange *act_next, SahRange *small,
                                                               int small_base, int32_t *slot, uint64_t *totals)
{
    const int c = blockIdx.x * kBuildBlock + threadIdx.x;
    if (c >= n_cand) return;
    const uint64_t f = flags[c], o = offs[c];
    if (c == n_cand - 1) totals[0] = o + f;
    if ((uint32_t)f) { act_next[(uint32_t)o] = cand[c]; slot[c] = (int32_t)(uint32_t)o; }
    else { small[small_base + (int)(o >> 32)] = cand[c]; slot[c] = -1; }
}

// left[i]: whether the primitive at place i goes to its node's left child.  On the way, the bounds of the box centres of
// every child that forms the next level (slot[c] >= 0: its index there) are collected in cb_next, which the caller has filled
// with sah_fill_centres_kernel - what sah_centres_kernel would find after the partition, without another pass.  A wave takes
// `places` consecutive places (a multiple of 64: many while the level has few nodes and every wave's atomics meet on the same
// words, few once they spread) and sends its share of a node's two children when it comes to the next node.
__global__ __launch_bounds__(kBuildBlock) void sah_side_kernel(const float4 *p_lo, const float4 *p_hi, const int32_t *seg, int n, const SahRange *act, const uint32_t *cb, const int4 *split,
                                                               const int32_t *slot, uint32_t *left, uint32_t *cb_next, int places)
{
    const int lane = threadIdx.x & 63;
    const size_t chunk = ((size_t)blockIdx.x * (kBuildBlock / 64) + (threadIdx.x >> 6)) * (size_t)places;
    if (chunk >= (size_t)n) return;
    float cl[2][3], ch[2][3];                // this lane's share of the two children of node `held`
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int k = 0; k < 3; k++) { cl[s][k] = PT_INF; ch[s][k] = -PT_INF; }
    int held = -1;                           // the same in every lane
    const auto flush = [&]() {
        if (held < 0) return;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int to = slot[2 * held + s];
            if (to < 0) {                    // a child that leaves the levels needs no bins
#pragma unroll
                for (int k = 0; k < 3; k++) { cl[s][k] = PT_INF; ch[s][k] = -PT_INF; }
                continue;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float a = wave_min(cl[s][k]), b = wave_max(ch[s][k]);
                if (lane == 0 && to >= 0 && a <= b) { atomicMin(cb_next + (size_t)to * 6 + k, ordered_bits(a)); atomicMax(cb_next + (size_t)to * 6 + 3 + k, ordered_bits(b)); }
                cl[s][k] = PT_INF; ch[s][k] = -PT_INF;
            }
        }
    };
    for (int step = 0; step < places / 64; step++) {
        const size_t first = chunk + (size_t)step * 64;
        if (first >= (size_t)n) break;
        const size_t i = first + (size_t)lane;
        const bool valid = i < (size_t)n;
        const int a = valid ? seg[i] : -1;
        uint32_t l = 0;
        float c[3] = {0, 0, 0};
        if (a >= 0) {
            const int4 sp = split[a];
            const float4 pl = p_lo[i], ph = p_hi[i];
            if (sp.x < 0) l = (int)i - act[a].begin < sp.z;
            else {
                const float rl = ordered_to_float(cb[(size_t)a * 6 + sp.x]), rh = ordered_to_float(cb[(size_t)a * 6 + 3 + sp.x]);
                const float lo = sp.x == 0 ? pl.x : sp.x == 1 ? pl.y : pl.z, hi = sp.x == 0 ? ph.x : sp.x == 1 ? ph.y : ph.z;
                l = sah_bin(lo, hi, rl, sah_scale(rl, rh)) <= sp.y;
            }
            c[0] = 0.5f * pl.x + 0.5f * ph.x; c[1] = 0.5f * pl.y + 0.5f * ph.y; c[2] = 0.5f * pl.z + 0.5f * ph.z;
        }
        if (valid) left[i] = l;
        // the step's places node by node (one node nearly always)
        unsigned long long todo = __ballot(a >= 0);
        while (todo) {
            const int node = __shfl(a, __ffsll((long long)todo) - 1, 64);
            if (node != held) { flush(); held = node; }
            if (a == node) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    cl[0][k] = fminf(cl[0][k], l ? c[k] : PT_INF); ch[0][k] = fmaxf(ch[0][k], l ? c[k] : -PT_INF);
                    cl[1][k] = fminf(cl[1][k], l ? PT_INF : c[k]); ch[1][k] = fmaxf(ch[1][k], l ? -PT_INF : c[k]);
                }
            }
            todo &= ~__ballot(a == node);
        }
    }
    flush();
}

// rank = exclusive prefix sums of left.  Within each node's range: the left primitives first, the right ones after, both in
// the order they had; everything else stays where it is.
__global__ __launch_bounds__(kBuildBlock) void sah_partition_kernel(const float4 *p_lo, const float4 *p_hi, const int32_t *seg, int n, const SahRange *act, const int4 *split,
                                                                    const uint32_t *left, const uint32_t *rank, const int32_t *slot, float4 *q_lo, float4 *q_hi, int32_t *seg_out)
{
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= n) return;
    const int a = seg[i];
    int pos = i, s = -1;
    if (a >= 0) {
        const int begin = act[a].begin, n_left = split[a].z;
        const int before = (int)(rank[i] - rank[begin]);          // left primitives of the node before this one
        const bool l = left[i] != 0;
        pos = l ? begin + before : begin + n_left + (i - begin - before);
        s = slot[2 * a + (l ? 0 : 1)];
    }
    q_lo[pos] = p_lo[i]; q_hi[pos] = p_hi[i]; seg_out[pos] = s;
}

// extra[j]: node numbers listed node j needs beyond its own (a subtree of k primitives has k - 1 nodes).
__global__ __launch_bounds__(kBuildBlock) void sah_extra_kernel(const SahRange *small, int n_small, uint64_t *extra)
{
    const int j = blockIdx.x * kBuildBlock + threadIdx.x;
    if (j < n_small) extra[j] = (uint64_t)max(0, small[j].count - 2);
}

// One wave per listed node: its subtree, built in LDS.  order[] holds the node's primitives (as indices into the arrays
// loaded below) so that every range still to be split is a stretch of it.  Ranges of more than kSahExhaustive primitives are
// split by the whole wave, one after the other, with bins (the stack holds them); the ranges of kSahExhaustive or fewer that
// fall out of that - seven of every eight nodes of the subtree lie in those - go on a list and are then finished by groups of
// eight lanes, eight ranges side by side, each by trying every split into two parts.
__global__ __launch_bounds__(64) void sah_small_kernel(const SahRange *small, const uint64_t *extra_offs, int extra_base, const float4 *p_lo, const float4 *p_hi, TwoChildTree tree)
{
    constexpr int S = kSahSmall, E = kSahExhaustive;
    static_assert(E == 8, "a group of eight lanes per listed range, one lane per place");
    __shared__ float blo[3][S], bhi[3][S];
    __shared__ int32_t ref[S];
    __shared__ uint8_t order[2][S];
    __shared__ uint32_t bins[kSahBinWords];
    __shared__ uint16_t stack_begin[S], stack_end[S];
    __shared__ int32_t stack_node[S];
    __shared__ uint8_t tiny_begin[S / 2], tiny_count[S / 2];       // ranges of 2 .. E primitives, for the groups
    __shared__ int32_t tiny_node[S / 2], tiny_first[S / 2];        // the range's node number; the first number for what lies below it
    __shared__ uint8_t group_begin[8][E], group_count[8][E];       // a group's own stack while it finishes one listed range
    __shared__ int32_t group_node[8][E];
    static_assert(S <= 256, "order[] holds bytes");
    const int lane = threadIdx.x;
    const SahRange r = small[blockIdx.x];
    const int count = r.count;
    if (count <= 0 || count > S) return;
    if (count == 1) {                        // a leaf: the link that names this node names the primitive instead
        if (lane == 0) reinterpret_cast<int32_t *>(tree.child + r.parent)[r.side] = __float_as_int(p_lo[r.begin].w);
        return;
    }
    for (int i = lane; i < count; i += 64) {
        const float4 a = p_lo[r.begin + i], b = p_hi[r.begin + i];
        blo[0][i] = a.x; blo[1][i] = a.y; blo[2][i] = a.z; bhi[0][i] = b.x; bhi[1][i] = b.y; bhi[2][i] = b.z;
        ref[i] = __float_as_int(a.w);
        order[0][i] = (uint8_t)i;
    }
    int next_node = extra_base + (int)extra_offs[blockIdx.x];
    int sp = 0, n_tiny = 0;
    if (count <= E) {
        if (lane == 0) { tiny_begin[0] = 0; tiny_count[0] = (uint8_t)count; tiny_node[0] = r.node; tiny_first[0] = next_node; }
        n_tiny = 1;
    } else {
        if (lane == 0) { stack_begin[0] = 0; stack_end[0] = (uint16_t)count; stack_node[0] = r.node; }
        sp = 1;
    }
    __syncthreads();
    while (sp > 0) {
        sp--;
        const int b = stack_begin[sp], e = stack_end[sp], node = stack_node[sp], k = e - b;      // k > E
        __syncthreads();                     // everyone has read the entry before it is overwritten
        float l[3] = {PT_INF, PT_INF, PT_INF}, h[3] = {-PT_INF, -PT_INF, -PT_INF};          // the range's box
        float cl[3] = {PT_INF, PT_INF, PT_INF}, ch[3] = {-PT_INF, -PT_INF, -PT_INF};        // and the bounds of its box centres
        for (int i = b + lane; i < e; i += 64) {
            const int p = order[0][i];
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                l[ax] = fminf(l[ax], blo[ax][p]); h[ax] = fmaxf(h[ax], bhi[ax][p]);
                const float c = 0.5f * blo[ax][p] + 0.5f * bhi[ax][p];
                cl[ax] = fminf(cl[ax], c); ch[ax] = fmaxf(ch[ax], c);
            }
        }
#pragma unroll
        for (int ax = 0; ax < 3; ax++) { l[ax] = wave_min(l[ax]); h[ax] = wave_max(h[ax]); cl[ax] = wave_min(cl[ax]); ch[ax] = wave_max(ch[ax]); }
        int n_left;
        {
            for (int w = lane; w < kSahBinWords; w += 64) bins[w] = (w % 7) < 3 ? 0xffffffffu : 0u;
            __syncthreads();
            float scale[3];
#pragma unroll
            for (int ax = 0; ax < 3; ax++) scale[ax] = sah_scale(cl[ax], ch[ax]);
            for (int i = b + lane; i < e; i += 64) {
                const int p = order[0][i];
                uint32_t ol[3], oh[3];
#pragma unroll
                for (int ax = 0; ax < 3; ax++) { ol[ax] = ordered_bits(blo[ax][p]); oh[ax] = ordered_bits(bhi[ax][p]); }
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    uint32_t *w = bins + (ax * kSahBins + sah_bin(blo[ax][p], bhi[ax][p], cl[ax], scale[ax])) * 7;
#pragma unroll
                    for (int q = 0; q < 3; q++) { atomicMin(w + q, ol[q]); atomicMax(w + 3 + q, oh[q]); }
                    atomicAdd(w + 6, 1u);
                }
            }
            __syncthreads();
            const SahPick pick = sah_pick_split(bins, lane);
            const float best = pick.cost; const int best_cand = pick.cand; const uint32_t my_left = pick.left.n;
            if (best < PT_INF) {
                n_left = (int)my_left;
                const int ax = best_cand / (kSahBins - 1), s = best_cand % (kSahBins - 1);
                int done_left = 0, done_right = 0;
                for (int base = b; base < e; base += 64) {
                    const int i = base + lane;
                    const bool valid = i < e;
                    const int p = valid ? order[0][i] : 0;
                    const bool left = valid && sah_bin(blo[ax][p], bhi[ax][p], cl[ax], scale[ax]) <= s;
                    const unsigned long long ml = __ballot(left), mr = __ballot(valid && !left);
                    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
                    if (left) order[1][b + done_left + __popcll(ml & below)] = (uint8_t)p;
                    else if (valid) order[1][b + n_left + done_right + __popcll(mr & below)] = (uint8_t)p;
                    done_left += __popcll(ml); done_right += __popcll(mr);
                }
                __syncthreads();
                for (int i = b + lane; i < e; i += 64) order[0][i] = order[1][i];
            } else {
                n_left = k / 2;              // all centres in one bin on every axis: halve the range as it lies
            }
            __syncthreads();
        }
        // the node, and what is still to be split
        if (lane == 0) {
            int32_t link[2];
            const int part_begin[2] = {b, b + n_left}, part_end[2] = {b + n_left, e};
            for (int s = 0; s < 2; s++) {
                const int pk = part_end[s] - part_begin[s];
                if (pk == 1) link[s] = ref[order[0][part_begin[s]]];
                else if (pk <= E) {          // for the groups: its own number now, and the pk - 2 numbers below it set aside
                    link[s] = next_node;
                    tiny_begin[n_tiny] = (uint8_t)part_begin[s]; tiny_count[n_tiny] = (uint8_t)pk; tiny_node[n_tiny] = next_node; tiny_first[n_tiny] = next_node + 1;
                    n_tiny++; next_node += pk - 1;
                } else {
                    link[s] = next_node;
                    stack_begin[sp] = (uint16_t)part_begin[s]; stack_end[sp] = (uint16_t)part_end[s]; stack_node[sp] = next_node;
                    sp++; next_node++;
                }
            }
            tree.child[node] = make_int2(link[0], link[1]);
            tree.lo[node] = make_float4(l[0], l[1], l[2], __int_as_float(node));
            tree.hi[node] = make_float4(h[0], h[1], h[2], __int_as_float(k));
        }
        // the other lanes follow lane 0's bookkeeping
        sp = __shfl(sp, 0, 64); next_node = __shfl(next_node, 0, 64); n_tiny = __shfl(n_tiny, 0, 64);
        __syncthreads();
    }
    // ---- the listed ranges: group g takes ranges g, g + 8, ...; within a group every lane follows the same steps with the
    // same numbers (what is pushed and which node numbers are given out depends on the split alone), lane 0 of the group writes.
    // Nothing here is shared between groups, and a group's lanes run in step, so fences for the compiler are all it takes.
    const int grp = lane >> 3, gl = lane & 7;
    for (int t = grp; t < n_tiny; t += 8) {
        int first = tiny_first[t], gsp = 1;
        if (gl == 0) { group_begin[grp][0] = tiny_begin[t]; group_count[grp][0] = tiny_count[t]; group_node[grp][0] = tiny_node[t]; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        while (gsp > 0) {
            gsp--;
            const int b = group_begin[grp][gsp], k = group_count[grp][gsp], node = group_node[grp][gsp];
            float pl[E][3], ph[E][3];
            float l[3] = {PT_INF, PT_INF, PT_INF}, h[3] = {-PT_INF, -PT_INF, -PT_INF};
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int p = order[0][b + min(j, k - 1)];
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    pl[j][ax] = j < k ? blo[ax][p] : PT_INF; ph[j][ax] = j < k ? bhi[ax][p] : -PT_INF;
                    l[ax] = fminf(l[ax], pl[j][ax]); h[ax] = fmaxf(h[ax], ph[j][ax]);
                }
            }
            const int mine = order[0][b + min(gl, k - 1)];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // (the stack entry and the places are read; they may be written now)
            // every split into two parts, shared out over the group's lanes.  Part `sub` (a mask over the range's places, the
            // first place always in it) against the rest; of equal costs the smaller mask.
            const uint32_t full = (1u << k) - 1u;
            float best = PT_INF; uint32_t best_sub = 1u;
            for (uint32_t c = (uint32_t)gl; c + 1u < (1u << (k - 1)); c += 8u) {
                const uint32_t sub = (c << 1) | 1u;
                SahBox lb, rb; sah_box_reset(lb); sah_box_reset(rb);
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const bool in = (sub >> j) & 1u;       // (places from k on hold empty boxes and are in no mask: they change nothing but rb.n)
#pragma unroll
                    for (int ax = 0; ax < 3; ax++) {
                        lb.l[ax] = fminf(lb.l[ax], in ? pl[j][ax] : PT_INF); lb.h[ax] = fmaxf(lb.h[ax], in ? ph[j][ax] : -PT_INF);
                        rb.l[ax] = fminf(rb.l[ax], in ? PT_INF : pl[j][ax]); rb.h[ax] = fmaxf(rb.h[ax], in ? -PT_INF : ph[j][ax]);
                    }
                }
                lb.n = (uint32_t)__popc(sub); rb.n = (uint32_t)k - lb.n;
                const float cost = sah_box_cost(lb) + sah_box_cost(rb);
                if (cost < best || (cost == best && sub < best_sub)) { best = cost; best_sub = sub; }
            }
#pragma unroll
            for (int d = 4; d >= 1; d >>= 1) {
                const float oc = __shfl_xor(best, d, 8); const uint32_t os = __shfl_xor(best_sub, d, 8);
                if (oc < best || (oc == best && os < best_sub)) { best = oc; best_sub = os; }
            }
            const int n_left = __popc(best_sub);
            if (gl < k) {                // the members of best_sub to the front, order kept on both sides
                const uint32_t below = (1u << gl) - 1u;
                const int pos = ((best_sub >> gl) & 1u) ? __popc(best_sub & below) : n_left + __popc(~best_sub & full & below);
                order[0][b + pos] = (uint8_t)mine;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            int32_t link[2];
            const int part_begin[2] = {b, b + n_left}, part_count[2] = {n_left, k - n_left};
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (part_count[s] == 1) link[s] = ref[order[0][part_begin[s]]];
                else {
                    link[s] = first;
                    if (gl == 0) { group_begin[grp][gsp] = (uint8_t)part_begin[s]; group_count[grp][gsp] = (uint8_t)part_count[s]; group_node[grp][gsp] = first; }
                    gsp++; first++;
                }
            }
            if (gl == 0) {
                tree.child[node] = make_int2(link[0], link[1]);
                tree.lo[node] = make_float4(l[0], l[1], l[2], __int_as_float(node));
                tree.hi[node] = make_float4(h[0], h[1], h[2], __int_as_float(k));
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
    }
}

}  // namespace ptrace
