// Micro-benchmark: how many dependent, divergent 64-byte record fetches per second a gfx950 CU sustains - the access
// pattern of BVH traversal (every lane chases its own chain of records through a table that fits the L2).
// Knobs: 16-byte loads per record (1..4), waves per SIMD, vector FMAs of filler work per step, table size,
// active lanes per wave (the others sit the loop out, as lanes in another traversal state do).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/gather_rate.hip -o tools/ubench/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LOADS, int FILL>
__global__ __launch_bounds__(256) void chase(const uint4 *table, uint32_t mask, int steps, int active, uint32_t *out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = (tid * 2654435761u) & mask;
    float acc = 0.0f;
    if ((int)(threadIdx.x & 63) < active) {
        for (int s = 0; s < steps; s++) {
            const uint4 *rec = table + (size_t)idx * 4;
            uint4 q0 = rec[0], q1 = make_uint4(0, 0, 0, 0), q2 = q1, q3 = q1;
            if (LOADS > 1) q1 = rec[1];
            if (LOADS > 2) q2 = rec[2];
            if (LOADS > 3) q3 = rec[3];
            float x = __uint_as_float((q0.y & 0x007fffffu) | 0x3f800000u);
#pragma unroll
            for (int f = 0; f < FILL; f++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
            acc += x;
            idx = (q0.x + q1.x + q2.x + q3.x + (uint32_t)(acc > 1e30f)) & mask;   // the next record depends on every load
        }
    }
    if (acc == 12345.678f) out[tid] = idx;
    if (tid == 0) out[0] = idx;
}

// The same chase with the wave fetching cooperatively: lanes 4r .. 4r+3 of instruction j load the four quarters of the
// record lane 16j + r is about to visit, so each load instruction touches 16 whole 64-byte records instead of 64 quarter
// records; the quarters go through an LDS staging area (per wave, 4 KiB, swizzled so that reads are conflict-free) back
// to their owners.  DMA: the loads write LDS directly (global_load_lds_dwordx4), no staging registers.
template <int LOADS, int FILL, bool DMA>
__global__ __launch_bounds__(256) void chase_coop(const uint4 *table, uint32_t mask, int steps, int active, uint32_t *out)
{
    __shared__ uint4 stage_all[4][256];
    uint4 *stage = stage_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = (tid * 2654435761u) & mask;
    float acc = 0.0f;
    const bool on = lane < active;
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int owner = 16 * j + (lane >> 2);
            const uint32_t r = (uint32_t)__shfl((int)idx, owner);
            const int quarter = ((lane & 3) - (owner >> 2)) & 3;          // what position lane & 3 of this record holds
            const uint4 *src = table + (size_t)r * 4 + quarter;
            if (owner < active) {
                if (DMA) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                          (__attribute__((address_space(3))) void *)(stage + 64 * j), 16, 0, 0);
                else stage[64 * j + lane] = *src;
            }
        }
        if (DMA) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
        __syncthreads();
        if (on) {
            uint4 q0 = stage[lane * 4 + ((0 + (lane >> 2)) & 3)], q1 = make_uint4(0, 0, 0, 0), q2 = q1, q3 = q1;
            if (LOADS > 1) q1 = stage[lane * 4 + ((1 + (lane >> 2)) & 3)];
            if (LOADS > 2) q2 = stage[lane * 4 + ((2 + (lane >> 2)) & 3)];
            if (LOADS > 3) q3 = stage[lane * 4 + ((3 + (lane >> 2)) & 3)];
            float x = __uint_as_float((q0.y & 0x007fffffu) | 0x3f800000u);
#pragma unroll
            for (int f = 0; f < FILL; f++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
            acc += x;
            idx = (q0.x + q1.x + q2.x + q3.x + (uint32_t)(acc > 1e30f)) & mask;
        }
        __syncthreads();
    }
    if (acc == 12345.678f) out[tid] = idx;
    if (tid == 0) out[0] = idx;
}

template <int LOADS, int FILL, bool DMA>
static double run_coop(const uint4 *table, uint32_t mask, int blocks_per_cu, int cus, int active, uint32_t *out)
{
    const int steps = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    chase_coop<LOADS, FILL, DMA><<<cus * blocks_per_cu, 256>>>(table, mask, 50, active, out);
    (void)hipEventRecord(e0);
    chase_coop<LOADS, FILL, DMA><<<cus * blocks_per_cu, 256>>>(table, mask, steps, active, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)cus * blocks_per_cu * 4 * active * steps / (ms * 1e-3) / 1e9;
}

template <int LOADS, int FILL>
static double run(const uint4 *table, uint32_t mask, int blocks_per_cu, int cus, int active, uint32_t *out)
{
    const int steps = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    chase<LOADS, FILL><<<cus * blocks_per_cu, 256>>>(table, mask, 50, active, out);
    (void)hipEventRecord(e0);
    chase<LOADS, FILL><<<cus * blocks_per_cu, 256>>>(table, mask, steps, active, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fetches = (double)cus * blocks_per_cu * 4 * active * steps;
    return fetches / (ms * 1e-3) / 1e9;      // G record fetches per second, whole chip
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : (1u << 16);   // records of 64 B: 65536 = 4 MiB
    std::vector<uint4> h((size_t)n * 4);
    srand(7);
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < 4; k++) h[(size_t)i * 4 + k] = make_uint4((uint32_t)rand() * 2654435761u + (uint32_t)rand(), (uint32_t)rand(), 0, 0);
    uint4 *d; uint32_t *out;
    (void)hipMalloc(&d, h.size() * sizeof(uint4)); (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(uint32_t));
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("table %u records (%.1f MiB), %d CUs; G record fetches / s, whole chip (x %d = per-lane 16-byte loads)\n", n, n * 64.0 / 1048576.0, cus, 1);
    printf("%-28s %8s %8s %8s %8s\n", "waves/SIMD, lanes, filler", "1 load", "2 loads", "3 loads", "4 loads");
    for (int bpc : {2, 3, 5, 8})
        for (int active : {64, 32}) {
            printf("%d w/SIMD %2d lanes   0 fma    %8.1f %8.1f %8.1f %8.1f\n", bpc, active, run<1, 0>(d, n - 1, bpc, cus, active, out), run<2, 0>(d, n - 1, bpc, cus, active, out),
                   run<3, 0>(d, n - 1, bpc, cus, active, out), run<4, 0>(d, n - 1, bpc, cus, active, out));
            printf("%d w/SIMD %2d lanes 150 fma    %8.1f %8.1f %8.1f %8.1f\n", bpc, active, run<1, 150>(d, n - 1, bpc, cus, active, out), run<2, 150>(d, n - 1, bpc, cus, active, out),
                   run<3, 150>(d, n - 1, bpc, cus, active, out), run<4, 150>(d, n - 1, bpc, cus, active, out));
        }
    printf("\ncooperative fetch (4 lanes per 64-byte record, quarters exchanged through LDS); columns: 16-byte LDS reads per record\n");
    for (int bpc : {3, 5})
        for (int active : {64, 32}) {
            printf("%d w/SIMD %2d lanes   0 fma regs %8.1f %8.1f %8.1f %8.1f\n", bpc, active, run_coop<1, 0, false>(d, n - 1, bpc, cus, active, out), run_coop<2, 0, false>(d, n - 1, bpc, cus, active, out),
                   run_coop<3, 0, false>(d, n - 1, bpc, cus, active, out), run_coop<4, 0, false>(d, n - 1, bpc, cus, active, out));
            printf("%d w/SIMD %2d lanes 150 fma regs %8.1f %8.1f %8.1f %8.1f\n", bpc, active, run_coop<1, 150, false>(d, n - 1, bpc, cus, active, out), run_coop<2, 150, false>(d, n - 1, bpc, cus, active, out),
                   run_coop<3, 150, false>(d, n - 1, bpc, cus, active, out), run_coop<4, 150, false>(d, n - 1, bpc, cus, active, out));
            printf("%d w/SIMD %2d lanes   0 fma dma  %8.1f %8.1f %8.1f %8.1f\n", bpc, active, run_coop<1, 0, true>(d, n - 1, bpc, cus, active, out), run_coop<2, 0, true>(d, n - 1, bpc, cus, active, out),
                   run_coop<3, 0, true>(d, n - 1, bpc, cus, active, out), run_coop<4, 0, true>(d, n - 1, bpc, cus, active, out));
            printf("%d w/SIMD %2d lanes 150 fma dma  %8.1f %8.1f %8.1f %8.1f\n", bpc, active, run_coop<1, 150, true>(d, n - 1, bpc, cus, active, out), run_coop<2, 150, true>(d, n - 1, bpc, cus, active, out),
                   run_coop<3, 150, true>(d, n - 1, bpc, cus, active, out), run_coop<4, 150, true>(d, n - 1, bpc, cus, active, out));
        }
    return 0;
}
