// Wavefront pipeline: the same path tracer as path_kernel (pt_device.hpp), cut into
// stages that each keep a wave's 64 lanes busy with the same kind of work.
//
//   wf_generate   one slot per pixel of the tile: path state + camera ray, slot -> queue 0
//   repeat (iteration i = 0, 1, ...):
//     wf_trace    persistent waves pull slots from queue i; a lane whose ray is finished
//                 stores the hit and immediately pulls the next slot, so short rays do not
//                 idle until the longest ray of the wave ends
//     wf_shade    one lane per slot of queue i: Shade(); paths that go on - and pixels whose
//                 next sample starts (regeneration) - are compacted into queue i+1; finished
//                 pixels write Result and the fused accumulation
//
// A pixel's samples stay in order in its slot (the shader's RNG counter runs on across
// samples, SH:839), so the image is bit-identical to path_kernel's.  Kernel boundaries
// order all memory traffic between stages; nothing is handed between workgroups inside a
// launch except through atomics on the queue counters.
#pragma once
#include "pt_device.hpp"

namespace ptrace {

__device__ __forceinline__ void wave_append(uint32_t *queue, uint32_t *count, bool want, uint32_t value, int lane)
{
    unsigned long long m = __ballot(want);
    if (!m) return;
    int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    if (want) queue[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

__global__ __launch_bounds__(kBlock) void wf_generate(const RenderArgs a)
{
    const int lane = threadIdx.x & 63;
    uint32_t item = blockIdx.x * kBlock + threadIdx.x;
    int x = 0, y = 0;
    bool valid = item < a.total_items && item_pixel(a, item, x, y);
    if (valid) {
        f3 o, d;
        camera_ray(a, x, y, o, d);
        a.ray_o[item] = make_float4(o.x, o.y, o.z, a.seed);
        a.ray_d[item] = make_float4(d.x, d.y, d.z, as_f(0));
        a.path_e[item] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        a.path_r[item] = make_float4(0.0f, 0.0f, 0.0f, as_f(-1));
    }
    wave_append(a.q_out, a.counts, valid, item, lane);
    stats_add(a.stats, 3, valid ? 1u : 0u, lane);
}

// Persistent traversal.  Each wave owns a range [q_next, q_end) of queue indices claimed from the
// global cursor in chunks; idle lanes are refilled from it at the top of every round.
template <int WIDE>
__global__ __launch_bounds__(kBlock) void wf_trace(const RenderArgs a)
{
    __shared__ uint32_t lds_stack[kStackLds * kBlock];
    const uint32_t n = a.counts[a.iter];
    if (n == 0) return;
    const Stack st = {lds_stack + threadIdx.x, a.stack_overflow + (blockIdx.x * kBlock + threadIdx.x), gridDim.x * kBlock};
    const int lane = threadIdx.x & 63;
    const SceneDev &sc = a.sc;
    constexpr uint32_t kChunk = 64;

    // The first chunk of a wave is fixed by its position in the grid (no atomic, and a wave beyond the
    // end of a short queue leaves at once); later chunks come from the cursor, which starts behind them.
    const uint32_t wave_id = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = (gridDim.x * kBlock) >> 6;
    if (wave_id * kChunk >= n) return;
    uint32_t q_next = wave_id * kChunk, q_end = min(q_next + kChunk, n);   // wave-uniform
    bool exhausted = false;             // wave-uniform: the global queue has no more chunks
    int32_t slot = -1;
    f3 o = pt_make3(0, 0, 0), d = pt_make3(0, 0, 1);
    RaySlabs rsl = {0, 0, 0, 0, 0, 0};
    float tbest = PT_INF; int32_t idbest = -1; uint32_t recbest = 0;
    int32_t node = kDoneNode; int sp = 0;
    uint32_t c_rays = 0, c_nodes = 0, c_prims = 0;

    for (;;) {
        // ---- refill idle lanes
        unsigned long long idle = __ballot(slot < 0);
        if (__popcll(idle) >= a.refill_threshold && !(exhausted && q_next == q_end)) {
            if (q_next == q_end && !exhausted) {
                uint32_t base = 0;
                if (lane == 0) base = n_waves * kChunk + atomicAdd(&a.cursors[a.iter], kChunk);
                base = __shfl(base, 0);
                if (base >= n) { exhausted = true; }
                else { q_next = base; q_end = min(base + kChunk, n); }
            }
            uint32_t avail = q_end - q_next;
            uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (slot < 0 && rank < avail) {
                slot = (int32_t)a.q_in[q_next + rank];
                float4 ro = a.ray_o[slot], rd = a.ray_d[slot];
                o = xyz(ro); d = xyz(rd);
                tbest = PT_INF; idbest = -1; recbest = 0;
                intersect_ground_plane(o, d, tbest, idbest);
                intersect_far_spheres(sc, o, d, tbest, idbest);
                rsl = ray_slabs(o, d);
                sp = 0;
                node = sc.n_nodes ? 0 : kDoneNode;
                c_rays++;
            }
            q_next += min(avail, (uint32_t)__popcll(idle));
        }
        if (__ballot(slot >= 0) == 0ull) break;

        // ---- inner nodes: keep stepping while at least node_vote lanes still hold an inner node
        for (;;) {
            const bool inner = node >= 0;
            const int n_inner = __popcll(__ballot(inner));
            // leave when no lane holds an inner node, or when only a few do while others wait with a leaf
            if (n_inner == 0 || (n_inner < a.node_vote && __ballot(node < 0 && node != kDoneNode) != 0ull)) break;
            if (!inner) continue;
            node = node_step<WIDE>(sc, node, rsl, tbest, st, sp);
            c_nodes++;
        }
        // ---- one leaf
        if (node < 0 && node != kDoneNode) node = leaf_step(sc, node, o, d, tbest, idbest, recbest, st, sp, c_prims);
        // ---- finished rays
        if (slot >= 0 && node == kDoneNode) {
            a.hit[slot] = make_float4(tbest, as_f(idbest), as_f((int32_t)recbest), 0.0f);
            slot = -1;
        }
    }

    stats_add(a.stats, 0, c_rays, lane);
    stats_add(a.stats, 1, c_nodes, lane);
    stats_add(a.stats, 2, c_prims, lane);
}

// Shade + compaction + regeneration + final write.  Grid-stride over the queue of iteration `iter`.
__global__ __launch_bounds__(kBlock) void wf_shade(const RenderArgs a)
{
    const uint32_t n = a.counts[a.iter];
    if (n == 0) return;
    const int lane = threadIdx.x & 63;
    const f3 cam_o = pt_make3(a.c2w[12], a.c2w[13], a.c2w[14]);
    const float fspp = (float)a.spp;
    uint32_t c_paths = 0;
    const uint32_t stride = gridDim.x * kBlock;
    // every lane of a wave runs the same number of rounds so that wave_append sees whole waves
    for (uint32_t base = blockIdx.x * kBlock + (threadIdx.x & ~63u); base < n; base += stride) {
        const uint32_t i = base + (uint32_t)lane;
        bool go_on = false;
        uint32_t slot = 0;
        if (i < n) {
            slot = a.q_in[i];
            float4 ro = a.ray_o[slot], rd = a.ray_d[slot], h = a.hit[slot], pe = a.path_e[slot], pr = a.path_r[slot];
            f3 o = xyz(ro), d = xyz(rd), energy = xyz(pe), result = xyz(pr);
            int x, y;
            item_pixel(a, slot, x, y);
            Rng rs = {ro.w, pixel_dot(x, y)};
            uint32_t state = (uint32_t)as_i(rd.w);
            uint32_t sample = state & 0xffffffu, bounce = state >> 24;
            float alpha_dist = pe.w;
            int32_t primary_id = as_i(pr.w);
            const float t = h.x; const int32_t id = as_i(h.y); const uint32_t rec = (uint32_t)as_i(h.z);

            f3 hit_pos;
            shade(a.sc, rs, o, d, energy, result, fspp, t, id, rec, hit_pos);
            bool path_done;
            if (bounce == 0) {
                primary_id = id;
                alpha_dist = pt_length3(pt_sub3(hit_pos, cam_o));
                path_done = !(t < PT_INF) || a.maxb <= 1;        // primary miss: see the path kernel's note
            } else {
                path_done = !(energy.x != 0.0f || energy.y != 0.0f || energy.z != 0.0f) || (int)bounce + 1 >= a.maxb;
            }
            bounce++;
            if (path_done) {
                sample++;
                if ((int)sample < a.spp) {                        // regeneration: next sample of this pixel
                    bounce = 0;
                    energy = pt_make3(1, 1, 1);
                    camera_ray(a, x, y, o, d);
                    c_paths++;
                    go_on = true;
                } else {                                          // SH:933-936
                    size_t p = (size_t)y * a.W + x;
                    float4 fresh = make_float4(result.x, result.y, result.z, alpha_dist);
                    a.result[p] = fresh;
                    if (a.hit_ids) a.hit_ids[p] = primary_id;
                    accumulate_pixel(a, p, fresh);
                }
            } else {
                go_on = true;
            }
            if (go_on) {
                a.ray_o[slot] = make_float4(o.x, o.y, o.z, rs.seed);
                a.ray_d[slot] = make_float4(d.x, d.y, d.z, as_f((int32_t)(sample | (bounce << 24))));
                a.path_e[slot] = make_float4(energy.x, energy.y, energy.z, alpha_dist);
                a.path_r[slot] = make_float4(result.x, result.y, result.z, as_f(primary_id));
            }
        }
        wave_append(a.q_out, &a.counts[a.iter + 1], go_on, slot, lane);
    }
    stats_add(a.stats, 3, c_paths, lane);
}

}  // namespace ptrace
