// Binned-SAH BVH builder (host, multi-threaded).  See bvh.hpp.
#include "bvh.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <limits>
#include <thread>

namespace ptrace {
namespace {

constexpr int kBins = 32;   // 16 / 32 / 64 / 128 bins: 16.2 / 14.9 / 15.2 / 15.5 node visits per ray on the 262 k-triangle bench scene, equal within 2 % on the 70 k one
constexpr float kInf = std::numeric_limits<float>::infinity();

struct Tmp {
    Aabb box;
    int32_t left = -1, right = -1;
    uint32_t first = 0, count = 0;  // count > 0: leaf
};

inline void box_reset(Aabb &b)
{
    for (int k = 0; k < 3; k++) { b.lo[k] = kInf; b.hi[k] = -kInf; }
}
inline void box_grow(Aabb &b, const Aabb &o)
{
    for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); }
}
inline float box_area(const Aabb &b)
{
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}
inline float centroid(const Aabb &b, int axis) { return 0.5f * (b.lo[axis] + b.hi[axis]); }

// A primitive as the builder moves it around: its box travels with its index, so that every pass over a
// range reads memory in order (an index array would make each pass a gather).
struct Ref { Aabb box; uint32_t id; };

struct Builder {
    std::vector<Ref> refs;
    std::vector<Tmp> tmp;
    std::atomic<int32_t> next{0};
    std::atomic<int> live_threads{1};
    std::atomic<int> max_depth{0};
    int leaf_size;          // largest leaf
    int min_leaf = 1;       // ranges this small always become a leaf
    float traversal_cost = 0.5f;   // cost of a node visit in units of one primitive test
    int max_threads;

    Builder(const std::vector<Aabb> &b, int leaf, int thr)
        : refs(b.size()), tmp(2 * b.size() + 1), leaf_size(leaf), max_threads(thr)
    {
        for (size_t i = 0; i < b.size(); i++) { refs[i].box = b[i]; refs[i].id = (uint32_t)i; }
    }

    static int levels_needed(uint32_t n, int leaf)
    {
        int l = 0;
        uint64_t cap = (uint64_t)leaf;
        while (cap < n) { cap <<= 1; l++; }
        return l;
    }

    int32_t build(uint32_t first, uint32_t n, int depth)
    {
        int32_t me = next.fetch_add(1);
        Tmp &node = tmp[me];
        Aabb box, cbox;
        box_reset(box); box_reset(cbox);
        for (uint32_t i = first; i < first + n; i++) {
            const Aabb &b = refs[i].box;
            box_grow(box, b);
            for (int k = 0; k < 3; k++) {
                float c = centroid(b, k);
                cbox.lo[k] = std::min(cbox.lo[k], c); cbox.hi[k] = std::max(cbox.hi[k], c);
            }
        }
        node.box = box; node.first = first; node.count = 0;
        static const bool bin_by_box = std::getenv("PTRACE_HOST_BIN_BOX") != nullptr;      // measurement knob: bins over the box, as the device builder's
        if (bin_by_box) cbox = box;
        int d = max_depth.load();
        while (depth > d && !max_depth.compare_exchange_weak(d, depth)) {}

        if (n <= (uint32_t)min_leaf || n == 1) { node.count = n; return me; }

        // Once the remaining depth budget only just fits a balanced tree over
        // n primitives, stop using SAH and halve the range.
        bool force_median = depth + levels_needed(n, 1) + 2 >= kMaxDepth;
        uint32_t mid = first + n / 2;
        bool split_done = false;

        if (!force_median) {
            float best_cost = kInf; int best_axis = -1, best_bin = -1;
            // one pass over the range fills the bins of all three axes
            Aabb bins[3][kBins]; uint32_t cnts[3][kBins];
            float blo[3], bscale[3]; bool usable[3];
            for (int axis = 0; axis < 3; axis++) {
                blo[axis] = cbox.lo[axis];
                float ext = cbox.hi[axis] - blo[axis];
                usable[axis] = ext > 0.0f;
                bscale[axis] = usable[axis] ? (float)kBins / ext : 0.0f;
                for (int b = 0; b < kBins; b++) { box_reset(bins[axis][b]); cnts[axis][b] = 0; }
            }
            for (uint32_t i = first; i < first + n; i++) {
                const Aabb &pb = refs[i].box;
                for (int axis = 0; axis < 3; axis++) {
                    if (!usable[axis]) continue;
                    int b = (int)((centroid(pb, axis) - blo[axis]) * bscale[axis]);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    box_grow(bins[axis][b], pb); cnts[axis][b]++;
                }
            }
            for (int axis = 0; axis < 3; axis++) {
                if (!usable[axis]) continue;
                const Aabb *bb = bins[axis]; const uint32_t *cnt = cnts[axis];
                float right_area[kBins]; uint32_t right_cnt[kBins];
                Aabb acc; box_reset(acc); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    box_grow(acc, bb[b]); c += cnt[b];
                    right_area[b] = box_area(acc); right_cnt[b] = c;
                }
                box_reset(acc); c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    box_grow(acc, bb[b]); c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = box_area(acc) * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
            // SAH termination: a range that may be a leaf becomes one when testing all its primitives is
            // estimated cheaper than one more node visit plus the children's expected primitive tests
            if (n <= (uint32_t)leaf_size) {
                float parent_area = box_area(box);
                float split_cost = best_axis >= 0 && parent_area > 0.0f ? traversal_cost + best_cost / parent_area
                                                                         : std::numeric_limits<float>::infinity();
                if ((float)n <= split_cost) { node.count = n; return me; }
            }
            if (best_axis >= 0) {
                float lo = cbox.lo[best_axis], ext = cbox.hi[best_axis] - lo;
                float scale = (float)kBins / ext;
                auto it = std::partition(refs.begin() + first, refs.begin() + first + n, [&](const Ref &rf) {
                    int b = (int)((centroid(rf.box, best_axis) - lo) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    return b <= best_bin;
                });
                mid = (uint32_t)(it - refs.begin());
                split_done = mid > first && mid < first + n;
            }
        }
        if (!split_done && n <= (uint32_t)leaf_size) { node.count = n; return me; }   // nothing separates them
        if (!split_done) {
            // median split along the widest centroid axis (or plain halving when all centroids coincide)
            int axis = 0;
            for (int k = 1; k < 3; k++)
                if (cbox.hi[k] - cbox.lo[k] > cbox.hi[axis] - cbox.lo[axis]) axis = k;
            mid = first + n / 2;
            if (cbox.hi[axis] > cbox.lo[axis])
                std::nth_element(refs.begin() + first, refs.begin() + mid, refs.begin() + first + n,
                                 [&](const Ref &a, const Ref &b) { return centroid(a.box, axis) < centroid(b.box, axis); });
        }

        uint32_t nl = mid - first, nr = n - nl;
        int32_t l, r;
        bool spawn = nl > 2048 && nr > 2048 && live_threads.load() < max_threads;
        if (spawn) {
            live_threads.fetch_add(1);
            std::thread t([&] { l = build(first, nl, depth + 1); live_threads.fetch_sub(1); });
            r = build(mid, nr, depth + 1);
            t.join();
        } else {
            l = build(first, nl, depth + 1);
            r = build(mid, nr, depth + 1);
        }
        tmp[me].left = l; tmp[me].right = r;
        return me;
    }
};

inline int32_t encode_leaf(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

// The grid of a quantised node over the box pb: per axis the smallest power-of-two cell with 255 cells >= the extent.
void node_grid(const Aabb &pb, float cell[3], uint8_t ebits[3])
{
    for (int a = 0; a < 3; a++) {
        float ext = pb.hi[a] - pb.lo[a];
        int e = -100;
        if (ext > 0.0f && std::isfinite(ext)) {
            (void)std::frexp(ext / 255.0f, &e);            // ext/255 = m * 2^e, m in [0.5, 1)
            if (e < -100) e = -100;
            while (pb.lo[a] + 255.0f * std::ldexp(1.0f, e) < pb.hi[a] && e < 100) e++;
        }
        cell[a] = std::ldexp(1.0f, e);
        ebits[a] = (uint8_t)(e + 127);
    }
}
// A child's planes on that grid, rounded outwards.
void quantise(const Aabb &pb, const float cell[3], const Aabb &cb, uint8_t ql[3], uint8_t qh[3])
{
    for (int a = 0; a < 3; a++) {
        float fl = std::floor((cb.lo[a] - pb.lo[a]) / cell[a]);
        float fh = std::ceil((cb.hi[a] - pb.lo[a]) / cell[a]);
        int l = (int)std::min(255.0f, std::max(0.0f, fl)), h = (int)std::min(255.0f, std::max(0.0f, fh));
        while (l > 0 && pb.lo[a] + (float)l * cell[a] > cb.lo[a]) l--;      // outward rounding survives
        while (h < 255 && pb.lo[a] + (float)h * cell[a] < cb.hi[a]) h++;    // the float arithmetic
        ql[a] = (uint8_t)l; qh[a] = (uint8_t)h;
    }
}

// Collapse to eight children per node (bvh.hpp, Bvh8Node): open the inner child with the largest area until there are
// eight or none is left to open; give every child the slot whose corner direction (+-1, +-1, +-1) it lies farthest
// along, seen from the node's centre (greedily, best pair first: Ylitie, Karras, Laine, "Efficient incoherent ray
// traversal on GPUs through compressed wide BVHs", HPG 2017, section 4.2); number the inner children consecutively
// and append the leaf children's primitives to the record order, both in slot order.
// Which subtrees become the children of an eight-wide node, chosen to minimise the tree's SAH cost instead of by "open the
// largest": C(n, i) = the cheapest way to stand for subtree n with at most i slots of a parent, a slot being an eight-wide
// node (cost: its area times kWideNodeCost, plus its own eight slots' worth of subtree) or a single primitive (its area
// times kWidePrimCost) - the dynamic program of Ylitie, Karras, Laine (HPG 2017, section 3), with one primitive per leaf:
//   C(n, 1) = area(n) * kWideNodeCost + D(n, 8),    C(n, i) = min(C(n, 1), D(n, i)),    D(n, j) = min_k C(left, k) + C(right, j - k).
// Filled bottom-up in one pass (a child's index is larger than its parent's); `take[j]` is the k of D(n, j), `node_at[i]`
// whether C(n, i) is the node.  The greedy rule fills the bottom of the tree badly - sixteen primitives become one node of
// eight two-primitive nodes instead of two nodes of eight - and every node costs a full visit however few children it has.
constexpr float kWideNodeCost = 1.0f, kWidePrimCost = 0.35f;
struct WideDp { float c[8]; uint8_t take[8]; uint8_t node_at; };

void wide_dp(const std::vector<Tmp> &tmp, int32_t n_tmp, int32_t root, std::vector<WideDp> &dp, float prim_cost, int width = 8, float leaf_cost = 0.0f)
{
    dp.resize((size_t)n_tmp);
    // areas relative to the root's, so that the sums stay finite whatever the scene's scale; a box without area (or with an
    // infinite one) still costs a little per node: among equal costs the tree with fewer nodes wins
    const float root_area = box_area(tmp[root].box);
    const float scale = root_area > 0.0f && std::isfinite(root_area) ? 1.0f / root_area : 0.0f;
    for (int32_t n = n_tmp - 1; n >= 0; n--) {
        const Tmp &t = tmp[n];
        WideDp &d = dp[(size_t)n];
        float area = box_area(t.box) * scale;
        if (!(area <= 1.0f)) area = 1.0f;
        area += 1e-7f;
        if (t.count) {                       // a primitive
            for (int i = 0; i < 8; i++) { d.c[i] = area * (leaf_cost + prim_cost * (float)t.count); d.take[i] = 0; }
            d.node_at = 0;
            continue;
        }
        const WideDp &l = dp[(size_t)t.left], &r = dp[(size_t)t.right];
        float dist[8];                       // dist[j - 1] = D(n, j), j >= 2
        dist[0] = kInf;
        for (int j = 2; j <= width; j++) {
            float best = kInf; int bk = 1;
            for (int k = 1; k < j; k++) {
                const float c = l.c[k - 1] + r.c[j - k - 1];
                if (c < best) { best = c; bk = k; }
            }
            dist[j - 1] = best; d.take[j - 1] = (uint8_t)bk;
        }
        d.c[0] = area * kWideNodeCost + dist[width - 1];
        d.node_at = 1;
        for (int i = 2; i <= width; i++) {
            if (d.c[0] <= dist[i - 1]) { d.c[i - 1] = d.c[0]; d.node_at |= (uint8_t)(1u << (i - 1)); }
            else d.c[i - 1] = dist[i - 1];
        }
    }
}
// The slots subtree n takes when it may use at most i of them.
void wide_slots(const std::vector<Tmp> &tmp, const std::vector<WideDp> &dp, int32_t n, int i, int32_t *kids, int &nk)
{
    const Tmp &t = tmp[n];
    if (t.count || ((dp[(size_t)n].node_at >> (i - 1)) & 1)) { kids[nk++] = n; return; }
    const int k = dp[(size_t)n].take[i - 1];
    wide_slots(tmp, dp, t.left, k, kids, nk);
    wide_slots(tmp, dp, t.right, i - k, kids, nk);
}

void collapse8(const std::vector<Tmp> &tmp, int32_t n_tmp, const std::vector<Ref> &refs, int32_t root, BvhResult &out)
{
    struct Item { int32_t tmp_index; uint32_t me; int depth; };
    std::vector<Item> work;
    bool greedy = false;
    float prim_cost = kWidePrimCost;
    if (const char *env = std::getenv("PTRACE_COLLAPSE")) { if (std::strcmp(env, "greedy") == 0) greedy = true; else if (std::atof(env) > 0.0) prim_cost = (float)std::atof(env); }
    std::vector<WideDp> dp;
    if (!greedy && !tmp[root].count) wide_dp(tmp, n_tmp, root, dp, prim_cost);
    out.nodes8.clear(); out.order8.clear();
    out.order8.reserve(refs.size());
    out.nodes8.emplace_back();
    int32_t root_kids[1] = {root};
    work.push_back({tmp[root].count ? -1 : root, 0u, 1});
    while (!work.empty()) {
        const Item it = work.back(); work.pop_back();
        int32_t kids[8]; int nk;
        const Aabb *pbp;
        if (it.tmp_index < 0) { kids[0] = root_kids[0]; nk = 1; pbp = &tmp[root].box; }     // the whole tree is one leaf
        else {
            const Tmp &t = tmp[it.tmp_index];
            pbp = &t.box;
            if (!greedy) {
                nk = 0;
                const int k = dp[(size_t)it.tmp_index].take[7];
                wide_slots(tmp, dp, t.left, k, kids, nk);
                wide_slots(tmp, dp, t.right, 8 - k, kids, nk);
            } else { kids[0] = t.left; kids[1] = t.right; nk = 2; }
            while (greedy && nk < 8) {
                int best = -1; float best_area = -1.0f;
                for (int k = 0; k < nk; k++)
                    if (tmp[kids[k]].count == 0) {
                        float ar = box_area(tmp[kids[k]].box);
                        if (ar > best_area) { best_area = ar; best = k; }
                    }
                if (best < 0) break;
                const int32_t open = kids[best];
                kids[best] = tmp[open].left;
                kids[nk++] = tmp[open].right;
            }
        }
        const Aabb &pb = *pbp;
        // ---- slots
        int slot_of[8]; bool kid_done[8] = {false, false, false, false, false, false, false, false}, slot_used[8] = {false, false, false, false, false, false, false, false};
        double cost[8][8];
        for (int k = 0; k < nk; k++)
            for (int sl = 0; sl < 8; sl++) {
                double c = 0.0;
                for (int a = 0; a < 3; a++) {
                    double pc = 0.5 * ((double)pb.lo[a] + pb.hi[a]), kc = 0.5 * ((double)tmp[kids[k]].box.lo[a] + tmp[kids[k]].box.hi[a]);
                    double d = kc - pc;
                    if (!std::isfinite(d)) d = 0.0;
                    c += ((sl >> a) & 1) ? d : -d;
                }
                cost[k][sl] = c;
            }
        for (int round = 0; round < nk; round++) {
            int bk = -1, bs = -1;
            for (int k = 0; k < nk; k++)
                if (!kid_done[k])
                    for (int sl = 0; sl < 8; sl++)
                        if (!slot_used[sl] && (bk < 0 || cost[k][sl] > cost[bk][bs])) { bk = k; bs = sl; }
            slot_of[bk] = bs; kid_done[bk] = true; slot_used[bs] = true;
        }
        int kid_in[8];
        for (int sl = 0; sl < 8; sl++) kid_in[sl] = -1;
        for (int k = 0; k < nk; k++) kid_in[slot_of[k]] = k;
        // ---- the node
        Bvh8Node nd; std::memset(&nd, 0, sizeof nd);
        float cell[3]; uint8_t ebits[3];
        node_grid(pb, cell, ebits);
        for (int a = 0; a < 3; a++) nd.org[a] = pb.lo[a];
        nd.ex = ebits[0]; nd.ey = ebits[1]; nd.ez = ebits[2];
        nd.child_base = (uint32_t)out.nodes8.size();
        nd.leaf_base = (uint32_t)out.order8.size();
        uint32_t n_inner = 0;
        for (int sl = 0; sl < 8; sl++) {
            if (kid_in[sl] < 0) {
                nd.qlo_x[sl] = nd.qlo_y[sl] = nd.qlo_z[sl] = 255; nd.qhi_x[sl] = nd.qhi_y[sl] = nd.qhi_z[sl] = 0;
                continue;
            }
            const Tmp &c = tmp[kids[kid_in[sl]]];
            uint8_t ql[3], qh[3];
            quantise(pb, cell, c.box, ql, qh);
            nd.qlo_x[sl] = ql[0]; nd.qlo_y[sl] = ql[1]; nd.qlo_z[sl] = ql[2];
            nd.qhi_x[sl] = qh[0]; nd.qhi_y[sl] = qh[1]; nd.qhi_z[sl] = qh[2];
            if (c.count) { nd.lmask |= 1u << sl; out.order8.push_back(refs[c.first].id); }     // one primitive per leaf
            else { nd.imask |= (uint8_t)(1u << sl); n_inner++; }
        }
        out.nodes8[it.me] = nd;
        out.depth8 = std::max(out.depth8, it.depth);
        for (uint32_t k = 0; k < n_inner; k++) out.nodes8.emplace_back();
        uint32_t next = nd.child_base;
        for (int sl = 0; sl < 8; sl++)
            if ((nd.imask >> sl) & 1) work.push_back({kids[kid_in[sl]], next++, it.depth + 1});
    }
    out.stack_need8 = out.depth8 + 1;
}

}  // namespace

BvhResult build_bvh(const std::vector<Aabb> &boxes, int leaf_size, int threads, float traversal_cost, bool wide8)
{
    BvhResult out;
    const uint32_t n = (uint32_t)boxes.size();
    if (n == 0) return out;
    leaf_size = wide8 ? 1 : std::max(1, std::min(leaf_size, kMaxLeafPrims));
    if (threads <= 0) threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    Builder b(boxes, leaf_size, threads);
    b.traversal_cost = traversal_cost;
    int32_t root = b.build(0, n, 0);
    out.depth = b.max_depth.load();
    out.order.resize(n);
    for (uint32_t i = 0; i < n; i++) out.order[i] = b.refs[i].id;

    // Emit the device layout in depth-first order: each inner node carries both children's boxes.
    const std::vector<Tmp> &tmp = b.tmp;
    auto set_child = [](float *lo, float *hi, const Aabb *box) {
        for (int k = 0; k < 3; k++) { lo[k] = box ? box->lo[k] : kInf; hi[k] = box ? box->hi[k] : -kInf; }
    };
    if (tmp[root].count > 0) {
        BvhNode nd; std::memset(&nd, 0, sizeof nd);
        set_child(nd.llo, nd.lhi, &tmp[root].box);
        nd.left = encode_leaf(tmp[root].first, tmp[root].count);
        set_child(nd.rlo, nd.rhi, nullptr);
        nd.right = encode_leaf(0, 1);
        out.nodes.push_back(nd);
        out.depth = 1;
        Bvh4Node w; std::memset(&w, 0, sizeof w);
        for (int a = 0; a < 3; a++) w.org[a] = tmp[root].box.lo[a];
        for (int a = 0; a < 3; a++) {
            float ext = tmp[root].box.hi[a] - tmp[root].box.lo[a];
            int e = -100;
            if (ext > 0.0f && std::isfinite(ext)) {
                (void)std::frexp(ext / 255.0f, &e);
                if (e < -100) e = -100;
                while (w.org[a] + 255.0f * std::ldexp(1.0f, e) < tmp[root].box.hi[a] && e < 100) e++;
            }
            (a == 0 ? w.ex : a == 1 ? w.ey : w.ez) = (uint8_t)(e + 127);
            w.qlo[a][0] = 0; w.qhi[a][0] = 255;
        }
        w.valid = 1;
        w.child[0] = encode_leaf(tmp[root].first, tmp[root].count);
        for (int k = 1; k < 4; k++) {
            for (int a = 0; a < 3; a++) { w.qlo[a][k] = 255; w.qhi[a][k] = 0; }
            w.child[k] = encode_leaf((uint32_t)boxes.size(), 1);
        }
        out.nodes4.push_back(w);
        out.depth4 = 1;
        if (wide8) collapse8(tmp, b.next.load(), b.refs, root, out);
        return out;                      // a single leaf: nothing is ever stacked
    }
    out.nodes.reserve((size_t)b.next.load());
    struct Item { int32_t tmp_index; int32_t parent; int side; };
    std::vector<Item> stack;
    stack.push_back({root, -1, 0});
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        const Tmp &t = tmp[it.tmp_index];
        int32_t me = (int32_t)out.nodes.size();
        BvhNode nd; std::memset(&nd, 0, sizeof nd);
        const Tmp &l = tmp[t.left], &r = tmp[t.right];
        set_child(nd.llo, nd.lhi, &l.box);
        set_child(nd.rlo, nd.rhi, &r.box);
        nd.left = l.count ? encode_leaf(l.first, l.count) : 0;
        nd.right = r.count ? encode_leaf(r.first, r.count) : 0;
        out.nodes.push_back(nd);
        if (it.parent >= 0) {
            if (it.side == 0) out.nodes[it.parent].left = me; else out.nodes[it.parent].right = me;
        }
        // push right first so the left subtree is emitted right after its parent
        if (!r.count) stack.push_back({t.right, me, 1});
        if (!l.count) stack.push_back({t.left, me, 0});
    }
    out.depth += 1;
    out.stack_need2 = out.depth;         // one sibling per inner node on the way down

    // ---- collapse to four children per node: repeatedly open the inner child with the largest area
    struct Item4 { int32_t tmp_index; int32_t parent; int slot; int depth; int stacked; };   // stacked: entries held when the node is entered
    std::vector<Item4> st4;
    st4.push_back({root, -1, 0, 1, 0});
    out.nodes4.reserve(out.nodes.size() / 2 + 1);
    std::vector<WideDp> dp4;
    bool use_dp4 = false;
    if (const char *env = std::getenv("PTRACE_COLLAPSE4")) {         // measurement knob: "prim cost,leaf cost"
        float pc = 0.35f, lc = 0.0f;
        std::sscanf(env, "%f,%f", &pc, &lc);
        wide_dp(tmp, b.next.load(), root, dp4, pc, 4, lc);
        use_dp4 = true;
    }
    while (!st4.empty()) {
        Item4 it = st4.back(); st4.pop_back();
        const Tmp &t = tmp[it.tmp_index];
        int32_t kids[4] = {t.left, t.right, -1, -1};
        int nk = 2;
        if (use_dp4) {
            nk = 0;
            const int k = dp4[(size_t)it.tmp_index].take[3];
            wide_slots(tmp, dp4, t.left, k, kids, nk);
            wide_slots(tmp, dp4, t.right, 4 - k, kids, nk);
        }
        while (!use_dp4 && nk < 4) {
            int best = -1; float best_area = -1.0f;
            for (int k = 0; k < nk; k++)
                if (tmp[kids[k]].count == 0) {
                    float ar = box_area(tmp[kids[k]].box);
                    if (ar > best_area) { best_area = ar; best = k; }
                }
            if (best < 0) break;
            int32_t open = kids[best];
            kids[best] = tmp[open].left;
            kids[nk++] = tmp[open].right;
        }
        int32_t me = (int32_t)out.nodes4.size();
        Bvh4Node nd; std::memset(&nd, 0, sizeof nd);
        const Aabb &pb = t.box;
        float cell[3];
        uint8_t ebits[3];
        for (int a = 0; a < 3; a++) {
            nd.org[a] = pb.lo[a];
            float ext = pb.hi[a] - pb.lo[a];
            int e = -100;
            if (ext > 0.0f && std::isfinite(ext)) {
                (void)std::frexp(ext / 255.0f, &e);            // ext/255 = m * 2^e, m in [0.5, 1)
                if (e < -100) e = -100;
                while (pb.lo[a] + 255.0f * std::ldexp(1.0f, e) < pb.hi[a] && e < 100) e++;
            }
            cell[a] = std::ldexp(1.0f, e);
            ebits[a] = (uint8_t)(e + 127);
        }
        nd.ex = ebits[0]; nd.ey = ebits[1]; nd.ez = ebits[2];
        for (int k = 0; k < nk; k++) {
            const Tmp &c = tmp[kids[k]];
            nd.valid |= (uint8_t)(1u << k);
            for (int a = 0; a < 3; a++) {
                float fl = std::floor((c.box.lo[a] - pb.lo[a]) / cell[a]);
                float fh = std::ceil((c.box.hi[a] - pb.lo[a]) / cell[a]);
                int ql = (int)std::min(255.0f, std::max(0.0f, fl)), qh = (int)std::min(255.0f, std::max(0.0f, fh));
                while (ql > 0 && pb.lo[a] + (float)ql * cell[a] > c.box.lo[a]) ql--;      // outward rounding survives
                while (qh < 255 && pb.lo[a] + (float)qh * cell[a] < c.box.hi[a]) qh++;    // the float arithmetic
                nd.qlo[a][k] = (uint8_t)ql; nd.qhi[a][k] = (uint8_t)qh;
            }
            nd.child[k] = c.count ? encode_leaf(c.first, c.count) : 0;
        }
        // An unused slot holds an inside-out box (low plane 255, high plane 0 on every axis), which no ray
        // enters, so traversal needs no validity test.  A ray with a wide margin on its box tests (far_slack in
        // pt_device.hpp) can pass it all the same: the slot's link (set below) is a leaf of one record, the one past the
        // last primitive, which the caller fills with zeros - a triangle without extent that nothing hits.
        for (int k = nk; k < 4; k++)
            for (int a = 0; a < 3; a++) { nd.qlo[a][k] = 255; nd.qhi[a][k] = 0; }
        out.nodes4.push_back(nd);
        if (it.parent >= 0) out.nodes4[it.parent].child[it.slot] = me;
        out.depth4 = std::max(out.depth4, it.depth);
        out.stack_need4 = std::max(out.stack_need4, it.stacked + nk - 1);
        for (int k = nk - 1; k >= 0; k--)
            if (!tmp[kids[k]].count) st4.push_back({kids[k], me, k, it.depth + 1, it.stacked + nk - 1});
    }
    for (Bvh4Node &nd : out.nodes4)
        for (int k = 0; k < 4; k++)
            if (!((nd.valid >> k) & 1)) nd.child[k] = encode_leaf((uint32_t)boxes.size(), 1);
    if (wide8) collapse8(tmp, b.next.load(), b.refs, root, out);
    return out;
}

Consecutive4 consecutive_layout(BvhResult &bvh)
{
    Consecutive4 out;
    const size_t n = bvh.nodes4.size();
    if (n == 0) return out;
    // breadth-first numbering: a node's inner children get consecutive new numbers, in slot order
    std::vector<int32_t> old_of; old_of.reserve(n);
    std::vector<int32_t> new_of(n, -1);
    old_of.push_back(0); new_of[0] = 0;
    for (size_t head = 0; head < old_of.size(); head++) {
        const Bvh4Node &nd = bvh.nodes4[(size_t)old_of[head]];
        for (int k = 0; k < 4; k++)
            if (((nd.valid >> k) & 1) && nd.child[k] >= 0) { new_of[(size_t)nd.child[k]] = (int32_t)old_of.size(); old_of.push_back(nd.child[k]); }
    }
    if (old_of.size() != n) return out;
    std::vector<Bvh4Node> nodes(n);
    std::vector<uint32_t> order; order.reserve(bvh.order.size());
    out.kinds.assign(n, 0); out.first_child.assign(n, 0); out.first_record.assign(n, 0);
    for (size_t i = 0; i < n; i++) {
        Bvh4Node nd = bvh.nodes4[(size_t)old_of[i]];
        out.first_record[i] = (uint32_t)order.size();
        bool first_inner = true;
        for (int k = 0; k < 4; k++) {
            if (!((nd.valid >> k) & 1)) continue;
            if (nd.child[k] >= 0) {
                nd.child[k] = new_of[(size_t)nd.child[k]];
                if (first_inner) { out.first_child[i] = (uint32_t)nd.child[k]; first_inner = false; }
                out.kinds[i] |= (uint8_t)(1u << (2 * k));
            } else {
                const uint32_t enc = (uint32_t)~nd.child[k], first = enc >> 3, count = (enc & 7u) + 1u;
                if (count > 2) return out;
                nd.child[k] = ~(int32_t)(((uint32_t)order.size() << 3) | (count - 1));
                for (uint32_t q = 0; q < count; q++) order.push_back(bvh.order[first + q]);
                out.kinds[i] |= (uint8_t)((count + 1) << (2 * k));
            }
        }
        nodes[i] = nd;
    }
    if (order.size() != bvh.order.size()) return out;
    for (Bvh4Node &nd : nodes)
        for (int k = 0; k < 4; k++)
            if (!((nd.valid >> k) & 1)) nd.child[k] = ~(int32_t)(((uint32_t)order.size() << 3) | 0u);
    bvh.nodes4.swap(nodes); bvh.order.swap(order);
    out.ok = true;
    return out;
}

// ---------------------------------------------------------------- reference splitting
namespace {

struct D3 { double v[3]; };

struct Piece {
    D3 poly[12]; int n;       // the part of the triangle inside the cuts made so far (a convex polygon)
    Aabb box;
};

// The most important spatial-median plane of [lo, hi] (level 1 = its middle, level 2 = the quarters ...) strictly inside
// (a, b); returns its level (1 .. 30) and position, or 0 when none exists at that resolution.
int median_plane(double lo, double hi, double a, double b, double &pos)
{
    if (!(hi > lo) || !(b > a)) return 0;
    // boxes are rounded outwards, so a plane within rounding distance of a face may only touch what is inside: keep clear of the faces
    const double margin = 1e-4 * (b - a) + 4e-7 * std::max(std::fabs(a), std::fabs(b));
    a += margin; b -= margin;
    if (!(b > a)) return 0;
    const double scale = 1073741824.0 / (hi - lo);     // 2^30 cells
    double fa = (a - lo) * scale, fb = (b - lo) * scale;
    if (!(fa >= 0.0)) fa = 0.0;
    if (!(fb <= 1073741824.0)) fb = 1073741824.0;
    const int64_t j0 = (int64_t)std::floor(fa) + 1, j1 = (int64_t)std::ceil(fb) - 1;     // plane indices strictly inside
    if (j0 > j1) return 0;
    const uint64_t x = (uint64_t)(j0 - 1) ^ (uint64_t)j1;
    int h = 0;
    while ((x >> (h + 1)) != 0) h++;
    const int64_t j = (j1 >> h) << h;
    pos = lo + (double)j / scale;
    if (!(pos > a && pos < b)) return 0;
    int level = 30 - h;
    return level < 1 ? 1 : level;
}

Aabb poly_box(const Piece &p, const Aabb &within)
{
    Aabb b;
    for (int k = 0; k < 3; k++) {
        double lo = p.poly[0].v[k], hi = lo;
        for (int i = 1; i < p.n; i++) { lo = std::min(lo, p.poly[i].v[k]); hi = std::max(hi, p.poly[i].v[k]); }
        float fl = (float)lo, fh = (float)hi;
        if ((double)fl > lo) fl = std::nextafterf(fl, -kInf);
        if ((double)fh < hi) fh = std::nextafterf(fh, kInf);
        b.lo[k] = std::max(fl, within.lo[k]); b.hi[k] = std::min(fh, within.hi[k]);
    }
    return b;
}

// Sutherland-Hodgman against the plane v[axis] = pos: `side` -1 keeps v[axis] <= pos, +1 keeps v[axis] >= pos.
void clip(const Piece &in, int axis, double pos, int side, Piece &out)
{
    out.n = 0;
    for (int i = 0; i < in.n; i++) {
        const D3 &p = in.poly[i], &q = in.poly[(i + 1) % in.n];
        const double dp = (p.v[axis] - pos) * side, dq = (q.v[axis] - pos) * side;
        if (dp >= 0.0 && out.n < 12) out.poly[out.n++] = p;
        if ((dp > 0.0 && dq < 0.0) || (dp < 0.0 && dq > 0.0)) {
            const double t = dp / (dp - dq);
            D3 r;
            for (int k = 0; k < 3; k++) r.v[k] = p.v[k] + t * (q.v[k] - p.v[k]);
            r.v[axis] = pos;
            if (out.n < 12) out.poly[out.n++] = r;
        }
    }
}

void split_piece(const Piece &pc, int splits, const float slo[3], const float shi[3], uint32_t tri,
                 std::vector<uint32_t> &ref_tri, std::vector<Aabb> &ref_box)
{
    if (splits > 0) {
        int best_level = 0, best_axis = -1; double best_pos = 0.0, best_ext = -1.0;
        for (int k = 0; k < 3; k++) {
            double pos;
            const int level = median_plane(slo[k], shi[k], pc.box.lo[k], pc.box.hi[k], pos);
            const double ext = (double)pc.box.hi[k] - pc.box.lo[k];
            if (level > 0 && (best_axis < 0 || level < best_level || (level == best_level && ext > best_ext))) {
                best_level = level; best_axis = k; best_pos = pos; best_ext = ext;
            }
        }
        if (best_axis >= 0) {
            Piece l, r;
            clip(pc, best_axis, best_pos, -1, l);
            clip(pc, best_axis, best_pos, +1, r);
            if (l.n >= 3 && r.n >= 3) {
                l.box = poly_box(l, pc.box); r.box = poly_box(r, pc.box);
                double wl = 0.0, wr = 0.0;
                for (int k = 0; k < 3; k++) { wl = std::max(wl, (double)l.box.hi[k] - l.box.lo[k]); wr = std::max(wr, (double)r.box.hi[k] - r.box.lo[k]); }
                int sl = wl + wr > 0.0 ? (int)std::floor((splits - 1) * wl / (wl + wr) + 0.5) : 0;
                sl = std::max(0, std::min(splits - 1, sl));
                split_piece(l, sl, slo, shi, tri, ref_tri, ref_box);
                split_piece(r, splits - 1 - sl, slo, shi, tri, ref_tri, ref_box);
                return;
            }
        }
    }
    ref_tri.push_back(tri); ref_box.push_back(pc.box);
}

}  // namespace

void split_references(const float *tri9, const Aabb *boxes, size_t n_tris, const float scene_lo[3], const float scene_hi[3],
                      float budget, std::vector<uint32_t> &ref_tri, std::vector<Aabb> &ref_box)
{
    const double kMaxSplits = 512.0;   // per triangle
    double ext_max = 0.0;
    for (int k = 0; k < 3; k++) ext_max = std::max(ext_max, (double)scene_hi[k] - scene_lo[k]);
    const bool scene_ok = budget > 0.0f && ext_max > 0.0 && std::isfinite(ext_max);
    // ---- the triangles worth cutting and their priorities
    std::vector<uint32_t> cand; std::vector<float> prio;
    for (size_t t = 0; scene_ok && t < n_tris; t++) {
        const Aabb &b = boxes[t];
        // only triangles that span a sixteenth of the scene: cutting the small ones of an evenly tessellated mesh adds
        // references and nodes and removes no overlap
        if (!(std::max(std::max((double)b.hi[0] - b.lo[0], (double)b.hi[1] - b.lo[1]), (double)b.hi[2] - b.lo[2]) >= ext_max / 16.0)) continue;
        const float *v = tri9 + 9 * t;
        bool ok = true;
        for (int k = 0; k < 9; k++) ok = ok && std::fabs(v[k]) < kHugeCoord;     // false for NaN as well
        if (!ok) continue;
        int level = 0;
        for (int k = 0; k < 3; k++) {
            double pos;
            const int l = median_plane(scene_lo[k], scene_hi[k], b.lo[k], b.hi[k], pos);
            if (l > 0 && (level == 0 || l < level)) level = l;
        }
        if (level == 0) continue;
        const double ex = ((double)b.hi[0] - b.lo[0]) / ext_max, ey = ((double)b.hi[1] - b.lo[1]) / ext_max, ez = ((double)b.hi[2] - b.lo[2]) / ext_max;
        const double e1[3] = {((double)v[3] - v[0]) / ext_max, ((double)v[4] - v[1]) / ext_max, ((double)v[5] - v[2]) / ext_max};
        const double e2[3] = {((double)v[6] - v[0]) / ext_max, ((double)v[7] - v[1]) / ext_max, ((double)v[8] - v[2]) / ext_max};
        const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
        const double a_box = 2.0 * (ex * ey + ey * ez + ez * ex), a_ideal = std::fabs(nx) + std::fabs(ny) + std::fabs(nz);
        const double d = a_box - a_ideal;
        if (d > 0.0) { cand.push_back((uint32_t)t); prio.push_back((float)std::cbrt(std::ldexp(d, -level))); }
    }
    // ---- the scale D with sum(min(kMaxSplits, floor(D * priority))) <= budget * n_tris, by bisection
    const double s_max = std::floor((double)budget * (double)n_tris);
    double d_lo = 0.0;
    if (!cand.empty() && s_max >= 1.0) {
        double psum = 0.0;
        for (float p : prio) psum += p;
        auto total = [&](double D) { double s = 0.0; for (float p : prio) s += std::min(kMaxSplits, std::floor(D * p)); return s; };
        double d_hi = 2.0 * s_max / psum + 1.0;
        while (total(d_hi) <= s_max && d_hi < 1e30) d_hi *= 2.0;      // ends at 1e30 when every triangle is at its cap
        for (int it = 0; it < 48; it++) {
            const double mid = 0.5 * (d_lo + d_hi);
            if (total(mid) <= s_max) d_lo = mid; else d_hi = mid;
        }
    }
    ref_tri.reserve(ref_tri.size() + n_tris + (size_t)(cand.empty() ? 0.0 : s_max));
    ref_box.reserve(ref_box.size() + n_tris + (size_t)(cand.empty() ? 0.0 : s_max));
    size_t c = 0;
    for (size_t t = 0; t < n_tris; t++) {
        int splits = 0;
        if (c < cand.size() && cand[c] == t) { splits = (int)std::min(kMaxSplits, std::floor(d_lo * prio[c])); c++; }
        if (splits <= 0) { ref_tri.push_back((uint32_t)t); ref_box.push_back(boxes[t]); continue; }
        Piece pc; pc.n = 3; pc.box = boxes[t];
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 3; k++) pc.poly[i].v[k] = tri9[9 * t + 3 * i + k];
        split_piece(pc, splits, scene_lo, scene_hi, (uint32_t)t, ref_tri, ref_box);
    }
}

}  // namespace ptrace
