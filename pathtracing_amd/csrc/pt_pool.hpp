// The path kernel with a ray pool per wave (gfx950, wave64; one wave per workgroup).
//
// path_sched_kernel (pt_device.hpp) gives every lane one ray and lets the wave vote which phase - inner-node step, leaf
// step, Shade() - to run next; the lanes in another state idle through it, and the counters show that this idling, not
// memory, bounds the kernel: the SIMDs issue vector instructions back to back, with 28 of 64 lanes doing useful work.
// Here a wave owns P = 64 * K rays ("slots") whose whole state lives in LDS (the ray, its closest hit so far, its
// traversal stack) and, for what only Shade() touches (energy, radiance, pixel and sample counters), in a global array.
// Each round the wave
//   1. classifies its slots by state (every lane looks at K of them),
//   2. picks the phase whose state is best represented,
//   3. gathers up to 64 slots in that state into its lanes (ranks from the ballot masks, a 64-entry list in LDS),
//   4. loads what the phase needs, runs it with (nearly) all lanes active, stores what changed.
// Nothing is shared between waves, so no wave ever waits for another.  Per-pixel arithmetic and its order are those of
// path_sched_kernel: the image is the same bit for bit.
#pragma once
#include "pt_device.hpp"

#ifndef PT_POOL_K
#define PT_POOL_K 2
#endif
#ifndef PT_POOL_SE
#define PT_POOL_SE 8
#endif
#ifndef PT_POOL_HOME
#define PT_POOL_HOME 0
#endif
#ifndef PT_POOL_WAVES
#define PT_POOL_WAVES 3
#endif
#ifndef PT_POOL_TRAVERSE
#define PT_POOL_TRAVERSE 0   // 1: one traversal phase (inner nodes and leaves voted inside it) instead of a node and a leaf phase
#endif

namespace ptrace {

// K slots per lane, SE stack entries per slot in LDS (deeper ones in the global overflow buffer).
// LDS per wave: 64 * K * (56 + 4 * SE) + 256 bytes.
template <int WIDE, bool MAT, int K, int SE>
__device__ __forceinline__ void path_pool_body(const RenderArgs &a)
{
    constexpr int P = 64 * K;
    __shared__ float4 sA[P];            // (origin.xyz, distance of the closest hit so far)
    __shared__ float4 sB[P];            // (direction.xyz, bits(id of that hit))
    __shared__ float4 sC[P];            // (clamped reciprocal direction.xyz, bits(primitive record of that hit))
    __shared__ int32_t sNode[P];        // traversal state: inner node, leaf reference, kDoneNode, kIdleNode, kDeadNode
    __shared__ int32_t sSp[P];          // stack entries in use
    __shared__ uint32_t sStack[SE * P]; // [entry][slot]
    __shared__ uint32_t sList[64];      // the slots gathered for the current phase
    typedef StackT<SE, P> PStack;
    const int lane = threadIdx.x;
    const SceneDev &sc = a.sc;
    const uint32_t n_slots = gridDim.x * P, gbase = blockIdx.x * P;
    // what only Shade() needs, per slot: (energy.xyz, _Seed) (result.xyz, t of the primary hit)
    // (pixel x | y << 16, sample | bounce << 20, primary record, primary id)
    float4 *const g0 = a.pool_state, *const g1 = a.pool_state + n_slots;
    int4 *const g2 = reinterpret_cast<int4 *>(a.pool_state + 2 * (size_t)n_slots);

#pragma unroll
    for (int k = 0; k < K; k++) { sNode[lane + 64 * k] = kIdleNode; sSp[lane + 64 * k] = 0; }
    uint32_t c_rays = 0, c_nodes = 0, c_prims = 0, c_paths = 0;     // this lane's share of the counters
    int shard = (int)(blockIdx.x % (uint32_t)a.n_shards);
    bool frame_drained = false;
#ifdef PT_PROFILE_PHASES
    uint32_t ph_rounds[3] = {0, 0, 0}, ph_lanes[3] = {0, 0, 0};
    unsigned long long ph_cyc[3] = {0, 0, 0}, ph_t0 = clock64(); int ph_cur = -1;
#define PH_SWITCH(k) { unsigned long long t_ = clock64(); if (ph_cur >= 0) ph_cyc[ph_cur] += t_ - ph_t0; ph_t0 = t_; ph_cur = (k); }
#endif
    __syncthreads();

    for (;;) {
        // ---- 1. classify the wave's slots
        unsigned long long m_inner[K], m_leaf[K], m_wait[K];
        int n_inner = 0, n_leaf = 0, n_wait = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int32_t nd = sNode[lane + 64 * k];
            m_inner[k] = __ballot(nd >= 0);
            m_leaf[k] = __ballot(is_leaf_ref(nd));
            m_wait[k] = __ballot(nd == kDoneNode || nd == kIdleNode);
            n_inner += __popcll(m_inner[k]); n_leaf += __popcll(m_leaf[k]); n_wait += __popcll(m_wait[k]);
        }
        if (n_inner + n_leaf + n_wait == 0) break;

#if PT_POOL_TRAVERSE
        // ---- 2. two phases only: traversal (inner nodes and leaves, voted lane by lane inside the phase) or Shade().
        // A ray stays in its lane from the gather until its traversal ends, so gathers are several node steps apart.
        const int phase = n_wait == 0 ? 1 : (n_inner + n_leaf == 0 ? 0
                          : (min(n_inner + n_leaf, 64) * a.pool_vote_shade >= min(n_wait, 64) * a.pool_vote_node ? 1 : 0));
        int n_sel = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const unsigned long long m = phase == 1 ? (m_inner[k] | m_leaf[k]) : m_wait[k];
            const int rank = n_sel + (int)__popcll(m & ((1ull << lane) - 1ull));
            if (((m >> lane) & 1ull) && rank < 64) sList[rank] = (uint32_t)(lane + 64 * k);
            n_sel += __popcll(m);
        }
        n_sel = min(n_sel, 64);
        __syncthreads();
        const int slot = lane < n_sel ? (int)sList[lane] : -1;
        __syncthreads();

        if (phase == 1) {
#ifdef PT_PROFILE_PHASES
            PH_SWITCH(1)
#endif
            int32_t node = kDoneNode, idbest = 0; int sp = 0; float tbest = 0.0f; uint32_t recbest = 0;
            f3 o = pt_make3(0, 0, 0), d = o;
            RaySlabs r = {0, 0, 0, 0, 0, 0};
            PStack st = {sStack, a.stack_overflow + gbase, n_slots};
            if (slot >= 0) {
                const float4 A = sA[slot], B = sB[slot], C = sC[slot];
                node = sNode[slot]; sp = sSp[slot]; tbest = A.w; idbest = as_i(B.w); recbest = (uint32_t)as_i(C.w);
                o = xyz(A); d = xyz(B);
                r.ix = C.x; r.iy = C.y; r.iz = C.z; r.ox = -A.x * C.x; r.oy = -A.y * C.y; r.oz = -A.z * C.z;
                st.lds = sStack + slot; st.overflow = a.stack_overflow + gbase + slot;
            }
            const int keep = max(1, (n_sel * a.pool_cont) >> 6);
            for (;;) {
                const int ni = __popcll(__ballot(node >= 0)), nl = __popcll(__ballot(is_leaf_ref(node)));
                if (ni + nl < keep) break;
                if (nl == 0 || ni * a.pool_vote_leaf > nl * 64) {
#ifdef PT_PROFILE_PHASES
                    PH_SWITCH(1) ph_rounds[1]++; ph_lanes[1] += ni;
#endif
                    if (node >= 0) { node = node_step<WIDE>(sc, node, r, tbest, st, sp); c_nodes++; }
                } else {
#ifdef PT_PROFILE_PHASES
                    PH_SWITCH(2) ph_rounds[2]++; ph_lanes[2] += nl;
#endif
                    if (is_leaf_ref(node)) {
                        uint32_t np = 0;
                        node = leaf_step(sc, node, o, d, tbest, idbest, recbest, st, sp, np);
                        c_prims += np;
                    }
                }
            }
            if (slot >= 0) {
                sA[slot].w = tbest; sB[slot].w = as_f(idbest); sC[slot].w = as_f((int32_t)recbest);
                sNode[slot] = node; sSp[slot] = sp;
            }
            __syncthreads();
            continue;
        }
#elif PT_POOL_HOME
        // ---- 2. / 3. home variant: a lane runs one of its own K slots (slot = lane + 64 k), so nothing moves between lanes
        // and every LDS access of the phases is conflict-free; the phase is the one most lanes can take part in
        unsigned long long l_inner = 0, l_leaf = 0, l_wait = 0;
#pragma unroll
        for (int k = 0; k < K; k++) { l_inner |= m_inner[k]; l_leaf |= m_leaf[k]; l_wait |= m_wait[k]; }
        int phase;   // 0 shade, 1 node, 2 leaf
        {
            const int s_shade = __popcll(l_wait) * a.pool_vote_node * a.pool_vote_leaf;
            const int s_node = __popcll(l_inner) * a.pool_vote_shade * a.pool_vote_leaf;
            const int s_leaf = __popcll(l_leaf) * a.pool_vote_shade * a.pool_vote_node;
            phase = s_node >= s_shade && s_node >= s_leaf ? 1 : (s_shade >= s_leaf ? 0 : 2);
        }
        int slot = -1;
#pragma unroll
        for (int k = K - 1; k >= 0; k--) {
            const unsigned long long m = phase == 1 ? m_inner[k] : phase == 2 ? m_leaf[k] : m_wait[k];
            if ((m >> lane) & 1ull) slot = lane + 64 * k;
        }
        const int n_sel = __popcll(phase == 1 ? l_inner : phase == 2 ? l_leaf : l_wait);
#else
        // ---- 2. the phase: the state with the most slots relative to what that phase wants to see (its vote); a wave
        // with rays in one state only runs that state's phase whatever the votes say
        int phase;   // 0 shade, 1 node, 2 leaf
        {
            const int s_shade = min(n_wait, 64) * a.pool_vote_node * a.pool_vote_leaf;
            const int s_node = min(n_inner, 64) * a.pool_vote_shade * a.pool_vote_leaf;
            const int s_leaf = min(n_leaf, 64) * a.pool_vote_shade * a.pool_vote_node;
            phase = s_node >= s_shade && s_node >= s_leaf ? 1 : (s_shade >= s_leaf ? 0 : 2);
        }

        // ---- 3. gather up to 64 slots of that state
        int n_sel = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const unsigned long long m = phase == 1 ? m_inner[k] : phase == 2 ? m_leaf[k] : m_wait[k];
            const int rank = n_sel + (int)__popcll(m & ((1ull << lane) - 1ull));
            if (((m >> lane) & 1ull) && rank < 64) sList[rank] = (uint32_t)(lane + 64 * k);
            n_sel += __popcll(m);
        }
        n_sel = min(n_sel, 64);
        __syncthreads();
        const int slot = lane < n_sel ? (int)sList[lane] : -1;
        __syncthreads();
#endif

#if !PT_POOL_TRAVERSE
        if (phase == 1) {
            // ================= node phase =================
#ifdef PT_PROFILE_PHASES
            ph_rounds[1]++; ph_lanes[1] += n_sel; PH_SWITCH(1)
#endif
            int32_t node = kDoneNode; int sp = 0; float tbest = 0.0f;
            RaySlabs r = {0, 0, 0, 0, 0, 0};
            PStack st = {sStack, a.stack_overflow + gbase, n_slots};
            if (slot >= 0) {
                const float4 A = sA[slot], C = sC[slot];
                node = sNode[slot]; sp = sSp[slot]; tbest = A.w;
                r.ix = C.x; r.iy = C.y; r.iz = C.z; r.ox = -A.x * C.x; r.oy = -A.y * C.y; r.oz = -A.z * C.z;
                st.lds = sStack + slot; st.overflow = a.stack_overflow + gbase + slot;
            }
            // step while most of the gathered rays still hold an inner node; then go back and gather afresh
            const int keep = max(1, (n_sel * a.pool_cont) >> 6);
            for (;;) {
                if (node >= 0) { node = node_step<WIDE>(sc, node, r, tbest, st, sp); c_nodes++; }
                const int still = __popcll(__ballot(node >= 0));
#ifdef PT_PROFILE_PHASES
                if (still >= keep) { ph_rounds[1]++; ph_lanes[1] += still; }
#endif
                if (still < keep) break;
            }
            if (slot >= 0) { sNode[slot] = node; sSp[slot] = sp; }
            __syncthreads();
            continue;
        }

        if (phase == 2) {
            // ================= leaf phase =================
#ifdef PT_PROFILE_PHASES
            ph_rounds[2]++; ph_lanes[2] += n_sel; PH_SWITCH(2)
#endif
            if (slot >= 0) {
                const float4 A = sA[slot], B = sB[slot];
                float tbest = A.w; int32_t idbest = as_i(B.w); uint32_t recbest = (uint32_t)as_i(sC[slot].w);
                int sp = sSp[slot];
                const PStack st = {sStack + slot, a.stack_overflow + gbase + slot, n_slots};
                uint32_t np = 0;
                const int32_t node = leaf_step(sc, sNode[slot], xyz(A), xyz(B), tbest, idbest, recbest, st, sp, np);
                c_prims += np;
                sA[slot].w = tbest; sB[slot].w = as_f(idbest); sC[slot].w = as_f((int32_t)recbest);
                sNode[slot] = node; sSp[slot] = sp;
            }
            __syncthreads();
            continue;
        }

#endif
        // ================= shade phase =================
#ifdef PT_PROFILE_PHASES
        ph_rounds[0]++; ph_lanes[0] += n_sel; PH_SWITCH(0)
#endif
        {
            const uint32_t gslot = gbase + (uint32_t)(slot >= 0 ? slot : 0);
            const int32_t state = slot >= 0 ? sNode[slot] : kDeadNode;
            const bool was_done = state == kDoneNode, was_idle = state == kIdleNode;
            f3 o = pt_make3(0, 0, 0), d = pt_make3(0, 0, 1);
            float tbest = PT_INF; int32_t idbest = -1; uint32_t recbest = 0;
            bool new_ray = false, finished = false, no_more = false;
            bool cached = false;                   // new_ray whose hit is already known (the pixel's primary hit)
            float cached_t = PT_INF; int32_t cached_id = -1; uint32_t cached_rec = 0;
            if (was_done) {
                const float4 A = sA[slot], B = sB[slot];
                o = xyz(A); d = xyz(B); tbest = A.w; idbest = as_i(B.w); recbest = (uint32_t)as_i(sC[slot].w);
                const int4 px = g2[gslot];
                const float4 s0 = g0[gslot], s1 = g1[gslot];
                const int pix_x = px.x & 0xffff, pix_y = px.x >> 16;
                int sample = px.y & 0xfffff, bounce = (int)((uint32_t)px.y >> 20), primary_id = px.w;
                uint32_t primary_rec = (uint32_t)px.z;
                f3 energy = xyz(s0), result = xyz(s1);
                float t_primary = s1.w;
                Rng rs = {s0.w, pixel_dot(pix_x, pix_y)};
                f3 hit_pos;
                bool from_history = false;
                if (MAT && a.render_mode == 10 && bounce == 0 && sample == 0) {
                    // SH:1048-1057: a primary hit the history already shows is copied from there and no path is traced
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
                    shade<MAT>(sc, rs, o, d, energy, result, (float)a.spp, tbest, idbest, recbest, hit_pos);
                    bool path_done;
                    if (bounce == 0) {
                        primary_id = idbest;
                        primary_rec = recbest;
                        t_primary = tbest;
                        path_done = !(tbest < PT_INF) || a.maxb <= 1;   // a primary miss is not traced again (SH:867-870 adds 0 * sky)
                    } else {
                        path_done = !(energy.x != 0.0f || energy.y != 0.0f || energy.z != 0.0f) || bounce + 1 >= a.maxb;
                    }
                    bounce++;
                    new_ray = true;
                    if (path_done) {
                        if (sample == 0) note_cost(a, pix_x, pix_y, 16 * bounce);
                        sample++;
                        if (sample < a.spp) {                          // next sample of the same pixel, SH:839
                            bounce = 0;
                            energy = pt_make3(1, 1, 1);
                            camera_ray<MAT>(a, pix_x, pix_y, o, d);
                            c_paths++;
                            if (a.reuse_primary) {
                                // SH:842 has no jitter: every sample of a pixel starts with the same camera ray
                                cached = true; cached_t = t_primary; cached_id = primary_id; cached_rec = primary_rec;
                            }
                        } else {                                       // SH:933-936
                            const size_t p = (size_t)pix_y * a.W + pix_x;
                            f3 co, cd;
                            camera_ray<MAT>(a, pix_x, pix_y, co, cd);
                            const bool sky = !(t_primary < PT_INF);
                            const f3 hit_pos0 = sky ? pt_make3(0, 0, 0) : pt_along3(co, t_primary, cd);
                            const bool full_reproj = MAT && (a.render_mode == 7 || a.render_mode == 8);
                            const f3 alpha_from = full_reproj ? pt_make3(a.alpha_origin[0], a.alpha_origin[1], a.alpha_origin[2]) : co;
                            const float alpha_dist = pt_length3(pt_sub3(hit_pos0, alpha_from));
                            float4 fresh = make_float4(result.x, result.y, result.z, alpha_dist);
                            if (MAT && (a.render_mode == 3 || a.render_mode == 4)) fresh = reproject_pixel_ext(a, result, hit_pos0, sky, alpha_dist);
                            else if (a.render_mode == 0 || a.render_mode == 2) fresh = reproject_pixel(a, result, hit_pos0, sky, alpha_dist, pix_y);
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
                        g0[gslot] = make_float4(energy.x, energy.y, energy.z, rs.seed);
                        g1[gslot] = make_float4(result.x, result.y, result.z, t_primary);
                        g2[gslot] = make_int4(px.x, sample | (bounce << 20), (int)primary_rec, primary_id);
                    }
                }
            }
            // ---- new pixels for slots without one: one atomic per wave on its current shard of the frame; a shard
            // that runs dry sends the wave on to the next one, for good
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
                            g0[gslot] = make_float4(1.0f, 1.0f, 1.0f, a.seed);
                            g1[gslot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            g2[gslot] = make_int4(x | (y << 16), 0, 0, -1);   // sample 0, bounce 0, no primary hit yet
                            camera_ray<MAT>(a, x, y, o, d);
                            c_paths++;
                            new_ray = true;
                        }
                    }
                }
            }
            // ---- the slot's next state; a new ray starts with the ground plane (SH:262-269) and the tree's root
            if (new_ray) {
                tbest = PT_INF; idbest = -1; recbest = 0;
                intersect_ground_plane(o, d, tbest, idbest);
                intersect_far_spheres(sc, o, d, tbest, idbest);
                int32_t node = sc.n_nodes ? 0 : kDoneNode;
                c_rays++;
                if (cached) {                       // nothing to traverse: straight back to the shade queue
                    tbest = cached_t; idbest = cached_id; recbest = cached_rec;
                    node = kDoneNode;
                    c_rays--;
                }
                const RaySlabs r = ray_slabs(o, d);
                sA[slot] = make_float4(o.x, o.y, o.z, tbest);
                sB[slot] = make_float4(d.x, d.y, d.z, as_f(idbest));
                sC[slot] = make_float4(r.ix, r.iy, r.iz, as_f((int32_t)recbest));
                sNode[slot] = node; sSp[slot] = 0;
            } else if (no_more) {
                sNode[slot] = kDeadNode;
            } else if (finished) {
                sNode[slot] = kIdleNode;            // pixel done, its replacement lay outside the tile
            }
        }
        __syncthreads();
    }

    stats_add(a.stats, 0, c_rays, lane);
    stats_add(a.stats, 1, c_nodes, lane);
    stats_add(a.stats, 2, c_prims, lane);
    stats_add(a.stats, 3, c_paths, lane);
#ifdef PT_PROFILE_PHASES
    PH_SWITCH(-1)
    if (lane == 0 && a.stats)
        for (int k = 0; k < 3; k++) {
            atomicAdd(&a.stats[(blockIdx.x % kStatsShards) * kStatsStride + 4 + 2 * k], (unsigned long long)ph_rounds[k]);
            atomicAdd(&a.stats[(blockIdx.x % kStatsShards) * kStatsStride + 5 + 2 * k], (unsigned long long)ph_lanes[k]);
            atomicAdd(&a.stats[(blockIdx.x % kStatsShards) * kStatsStride + 10 + k], ph_cyc[k]);
        }
#undef PH_SWITCH
#endif
}

constexpr int kPoolK = PT_POOL_K;     // slots per lane
constexpr int kPoolSE = PT_POOL_SE;   // stack entries per slot in LDS

template <int WIDE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT_POOL_WAVES))) void path_pool_kernel(const RenderArgs a)
{
    path_pool_body<WIDE, false, kPoolK, kPoolSE>(a);
}

template <int WIDE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT_POOL_WAVES))) void path_pool_mat_kernel(const RenderArgs a)
{
    path_pool_body<WIDE, true, kPoolK, kPoolSE>(a);
}

}  // namespace ptrace
