// Host-only check of the BVH builder (csrc/bvh.cpp), meant to be built with -fsanitize=address,undefined:
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -I include -I pathtracing_amd/csrc -I pathtracing_amd/host \
//       tools/bvh_check.cpp pathtracing_amd/csrc/bvh.cpp pathtracing_amd/host/scene_gen.cpp -lpthread -o /tmp/bvh_check && /tmp/bvh_check
// For named scenes and for degenerate inputs it verifies, on both device layouts, that every primitive sits in exactly one
// leaf, that every (dequantised) child box contains the boxes below it, and that the depth stays within kMaxDepth.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "bvh.hpp"
#include "scene.hpp"

using ptrace::Aabb;

static bool contains(const float *lo, const float *hi, const Aabb &b)
{
    for (int k = 0; k < 3; k++)
        if (std::isfinite(b.lo[k]) && std::isfinite(b.hi[k]) && !(lo[k] <= b.lo[k] && hi[k] >= b.hi[k])) return false;
    return true;
}

static int check(const char *what, const std::vector<Aabb> &boxes, int leaf)
{
    ptrace::BvhResult r = ptrace::build_bvh(boxes, leaf, 0, 1.0f);
    const size_t n = boxes.size();
    if (n == 0) { std::printf("%-28s empty: %zu nodes\n", what, r.nodes.size()); return r.nodes.empty() && r.nodes4.empty() ? 0 : 1; }
    int bad = 0;
    std::vector<int> seen(n, 0);
    for (uint32_t id : r.order) { if (id >= n) bad++; else seen[id]++; }
    for (int c : seen) if (c != 1) bad++;
    // four-wide layout: walk it, dequantise, test containment of every primitive of a leaf and count leaf coverage
    std::vector<int> cover(n, 0);
    struct It { int32_t node; int depth; int stacked; };
    std::vector<It> st{{0, 1, 0}};
    int depth = 0, need4 = 0;
    while (!st.empty()) {
        It it = st.back(); st.pop_back();
        depth = std::max(depth, it.depth);
        int kids = 0;
        for (int k = 0; k < 4; k++) kids += (r.nodes4[(size_t)it.node].valid >> k) & 1;
        const int held = it.stacked + kids - 1;      // a ray that enters every child stacks all siblings of the one it follows
        if (n > 1) need4 = std::max(need4, held);
        const ptrace::Bvh4Node &nd = r.nodes4[(size_t)it.node];
        const float cell[3] = {std::ldexp(1.0f, nd.ex - 127), std::ldexp(1.0f, nd.ey - 127), std::ldexp(1.0f, nd.ez - 127)};
        for (int k = 0; k < 4; k++) {
            if (!((nd.valid >> k) & 1)) continue;
            float lo[3], hi[3];
            for (int a = 0; a < 3; a++) { lo[a] = nd.org[a] + nd.qlo[a][k] * cell[a]; hi[a] = nd.org[a] + nd.qhi[a][k] * cell[a]; }
            if (nd.child[k] >= 0) { st.push_back({nd.child[k], it.depth + 1, held}); continue; }
            uint32_t enc = (uint32_t)~nd.child[k], first = enc >> 3, count = (enc & 7u) + 1u;
            for (uint32_t s = first; s < first + count; s++) {
                if (s >= n) { bad++; continue; }
                cover[r.order[s]]++;
                if (!contains(lo, hi, boxes[r.order[s]])) bad++;
            }
        }
    }
    for (int c : cover) if (c != 1) bad++;
    if (depth > ptrace::kMaxDepth || r.depth > ptrace::kMaxDepth + 1) bad++;
    if (need4 != r.stack_need4 || r.stack_need2 > ptrace::kMaxDepth + 1 || r.stack_need4 > 3 * depth) bad++;
    std::printf("%-28s %8zu prims: %7zu nodes, %7zu wide nodes, depth %2d / %2d, stack need %2d / %3d, %s\n", what, n, r.nodes.size(),
                r.nodes4.size(), r.depth, depth, r.stack_need2, r.stack_need4, bad ? "INVALID" : "ok");
    return bad;
}

// split_references: every point of a triangle lies in one of the triangle's reference boxes, the number of additional
// references stays within the budget, triangles it must leave alone come back as they were.
static int check_split(const char *what, const std::vector<float> &tri9, float budget, bool expect_splits)
{
    const size_t n = tri9.size() / 9;
    std::vector<Aabb> boxes(n);
    float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t t = 0; t < n; t++)
        for (int k = 0; k < 3; k++) {
            const float *v = &tri9[9 * t];
            boxes[t].lo[k] = std::min(v[k], std::min(v[3 + k], v[6 + k])); boxes[t].hi[k] = std::max(v[k], std::max(v[3 + k], v[6 + k]));
            if (std::fabs(boxes[t].lo[k]) < ptrace::kHugeCoord) slo[k] = std::min(slo[k], boxes[t].lo[k]);
            if (std::fabs(boxes[t].hi[k]) < ptrace::kHugeCoord) shi[k] = std::max(shi[k], boxes[t].hi[k]);
        }
    std::vector<uint32_t> ref_tri; std::vector<Aabb> ref_box;
    split_references(tri9.data(), boxes.data(), n, slo, shi, budget, ref_tri, ref_box);
    int bad = 0;
    if (ref_tri.size() != ref_box.size() || ref_tri.size() < n || (double)ref_tri.size() > (double)n + std::floor((double)budget * n)) { std::printf("%s: %zu references for %zu triangles\n", what, ref_tri.size(), n); bad++; }
    if (expect_splits != (ref_tri.size() > n)) { std::printf("%s: %s\n", what, expect_splits ? "nothing was split" : "something was split"); bad++; }
    std::vector<std::vector<Aabb>> per(n);
    for (size_t r = 0; r < ref_tri.size(); r++) {
        if (ref_tri[r] >= n) { bad++; continue; }
        if (r > 0 && ref_tri[r] < ref_tri[r - 1]) { std::printf("%s: references out of triangle order\n", what); bad++; }
        per[ref_tri[r]].push_back(ref_box[r]);
        for (int k = 0; k < 3; k++)
            if (ref_box[r].lo[k] < boxes[ref_tri[r]].lo[k] || ref_box[r].hi[k] > boxes[ref_tri[r]].hi[k]) { std::printf("%s: a reference box leaves its triangle's box\n", what); bad++; }
    }
    std::mt19937 rng(11);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (size_t t = 0; t < n; t++) {
        if (per[t].empty()) { std::printf("%s: triangle %zu lost\n", what, t); bad++; continue; }
        if (per[t].size() == 1) { if (std::memcmp(&per[t][0], &boxes[t], sizeof(Aabb)) != 0) { std::printf("%s: unsplit triangle %zu changed its box\n", what, t); bad++; } continue; }
        const float *v = &tri9[9 * t];
        for (int s = 0; s < 200; s++) {
            double a = u(rng), b = u(rng);
            if (s < 3) { a = s == 1; b = s == 2; }        // the corners too
            else if (a + b > 1.0) { a = 1.0 - a; b = 1.0 - b; }
            double p[3];
            for (int k = 0; k < 3; k++) p[k] = (double)v[k] + a * ((double)v[3 + k] - v[k]) + b * ((double)v[6 + k] - v[k]);
            bool in = false;
            for (const Aabb &bx : per[t]) {
                bool here = true;
                for (int k = 0; k < 3; k++) { const double tol = 1e-6 * (std::fabs(p[k]) + 1.0); here = here && p[k] >= bx.lo[k] - tol && p[k] <= bx.hi[k] + tol; }
                in = in || here;
            }
            if (!in) { std::printf("%s: a point of triangle %zu is in none of its %zu boxes\n", what, t, per[t].size()); bad++; break; }
        }
    }
    std::printf("%-28s %zu triangles -> %zu references\n", what, n, ref_tri.size());
    return bad;
}

int main()
{
    int bad = 0;
    {
        std::mt19937 rng(9);
        std::uniform_real_distribution<float> u(-1.0f, 1.0f);
        std::vector<float> small, mixed, odd;
        for (int i = 0; i < 3000; i++) { float c[3] = {10 * u(rng), 10 * u(rng), 10 * u(rng)}; for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) small.push_back(c[k] + 0.1f * u(rng)); }
        mixed = small;
        for (int i = 0; i < 60; i++) for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) mixed.push_back(12 * u(rng));      // large, any orientation
        for (float y : {-10.0f, 10.0f}) { const float q[9] = {-10, y, -10, 10, y, -10, 10, y, 10}; mixed.insert(mixed.end(), q, q + 9); }   // axis-aligned walls
        { const float sliver[9] = {-9, 0, 0, 9, 0.001f, 0, 9, 0, 0.001f}; mixed.insert(mixed.end(), sliver, sliver + 9); }
        odd = mixed;
        { const float nan_tri[9] = {-9, 0, 0, 9, NAN, 0, 0, 9, 0}, inf_tri[9] = {-9, 0, 0, INFINITY, 1, 0, 0, 9, 0}, far_tri[9] = {-9, 0, 0, 3e18f, 1, 0, 0, 9, 0}, point[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
          odd.insert(odd.end(), nan_tri, nan_tri + 9); odd.insert(odd.end(), inf_tri, inf_tri + 9); odd.insert(odd.end(), far_tri, far_tri + 9); odd.insert(odd.end(), point, point + 9); }
        bad += check_split("split: small triangles only", small, 0.3f, false);
        bad += check_split("split: large among small", mixed, 0.3f, true);
        bad += check_split("split: budget 0", mixed, 0.0f, false);
        bad += check_split("split: tiny budget", mixed, 0.001f, true);
        bad += check_split("split: budget 4", mixed, 4.0f, true);
        bad += check_split("split: NaN, inf, huge, point", odd, 0.3f, true);
        bad += check_split("split: nothing", {}, 0.3f, false);
    }
    for (const char *name : {"fixture", "blob70k", "hall260k"}) {
        std::unique_ptr<ptm::Scene> s(ptm::Scene::create(name));
        std::vector<Aabb> boxes;
        for (const PtSphere &sp : s->spheres) {
            Aabb b;
            for (int k = 0; k < 3; k++) { b.lo[k] = sp.position[k] - sp.radius; b.hi[k] = sp.position[k] + sp.radius; }
            boxes.push_back(b);
        }
        for (size_t i = 0; i + 2 < s->indices.size(); i += 3) {
            Aabb b;
            for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
            for (int c = 0; c < 3; c++)
                for (int k = 0; k < 3; k++) {
                    float v = s->vertices[(size_t)s->indices[i + c] * 3 + k];
                    b.lo[k] = std::fmin(b.lo[k], v); b.hi[k] = std::fmax(b.hi[k], v);
                }
            boxes.push_back(b);
        }
        for (int leaf : {1, 2, 4}) bad += check((std::string(name) + " leaf " + std::to_string(leaf)).c_str(), boxes, leaf);
    }
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    auto point_box = [](float x, float y, float z, float h) { Aabb b = {{x - h, y - h, z - h}, {x + h, y + h, z + h}}; return b; };
    bad += check("no primitives", {}, 2);
    bad += check("one primitive", {point_box(0, 0, 0, 1)}, 2);
    bad += check("two identical", {point_box(1, 2, 3, 0.5f), point_box(1, 2, 3, 0.5f)}, 2);
    { std::vector<Aabb> b(5000, point_box(0.25f, -3, 7, 0.125f)); bad += check("5000 identical boxes", b, 2); }
    { std::vector<Aabb> b; for (int i = 0; i < 4000; i++) b.push_back(point_box((float)i * 1e-3f, 0, 0, 0.0f)); bad += check("4000 points on a line", b, 2); }
    { std::vector<Aabb> b; for (int i = 0; i < 3000; i++) b.push_back(point_box(u(rng), u(rng), u(rng), 0.01f));
      // what the library hands the builder for a primitive with a NaN or infinite coordinate
      b[7].lo[1] = -ptrace::kHugeCoord; b[7].hi[1] = ptrace::kHugeCoord;
      for (int k = 0; k < 3; k++) { b[99].lo[k] = -ptrace::kHugeCoord; b[99].hi[k] = ptrace::kHugeCoord; }
      bad += check("random + two huge boxes", b, 2); }
    { std::vector<Aabb> b; for (int i = 0; i < 20000; i++) { float s = std::ldexp(1.0f, -(i % 40)); b.push_back(point_box(s, s, s, s * 0.25f)); }
      bad += check("geometric sizes (deep tree)", b, 1); }
    { // a comb: at every scale a cluster of four wide boxes beside everything smaller; the collapse opens the cluster
      // (largest area) and the rest of the scene advances one level per wide node, three stacked siblings each
      std::vector<Aabb> b;
      for (int k = 0; k < 56; k++) {
          const float s = std::ldexp(1.0f, -k);
          for (int j = 0; j < 4; j++) { Aabb bx = {{s * (0.55f + 0.1f * j), -3 * s, -3 * s}, {s * (0.6f + 0.1f * j), 3 * s, 3 * s}}; b.push_back(bx); }
      }
      bad += check("comb of clusters (lopsided)", b, 1); }
    { std::vector<Aabb> b; for (int i = 0; i < 2000; i++) b.push_back(point_box(u(rng) * 1e30f, u(rng) * 1e-30f, u(rng), 1e20f)); bad += check("huge and tiny extents", b, 2); }
    std::printf(bad ? "bvh_check FAILED (%d)\n" : "bvh_check ok\n", bad);
    return bad != 0;
}
