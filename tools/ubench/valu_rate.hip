// Micro-benchmark: issue rate of plain v_fma_f32 vs packed v_pk_fma_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ void k_plain(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_packed(float *out, int iters, float a, float b)
{
    float2v A = {a, a}, B = {b, b};
    float2v x0 = {(float)threadIdx.x, 1}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; i++) {
        x0 = __builtin_elementwise_fma(x0, A, B); x1 = __builtin_elementwise_fma(x1, A, B);
        x2 = __builtin_elementwise_fma(x2, A, B); x3 = __builtin_elementwise_fma(x3, A, B);
    }
    float2v s = x0 + x1 + x2 + x3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_minmax(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        x0 = fminf(fmaxf(x0, a), b + x1); x1 = fminf(fmaxf(x1, a), b + x2); x2 = fminf(fmaxf(x2, a), b + x3); x3 = fminf(fmaxf(x3, a), b + x0);
        x4 = fminf(fmaxf(x4, a), b + x5); x5 = fminf(fmaxf(x5, a), b + x6); x6 = fminf(fmaxf(x6, a), b + x7); x7 = fminf(fmaxf(x7, a), b + x4);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpe = 1; wpe <= 8; wpe *= 2) {
        int blocks = 256 * wpe;   // 256-thread blocks: wpe waves per SIMD
        for (int which = 0; which < 3; which++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                else if (which == 1) hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k_minmax, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    double instr_per_wave = (double)iters * (which == 0 ? 8 : which == 1 ? 4 : 24);
                    double waves_per_simd = wpe;
                    double cycles = ms * 1e-3 * 2.4e9;
                    printf("%s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (at 2.4 GHz), %.1f Gflop-lanes\n",
                           which == 0 ? "v_fma_f32   " : which == 1 ? "v_pk_fma_f32" : "min/max/add ", wpe, ms,
                           cycles / (instr_per_wave * waves_per_simd), 0.0);
                }
            }
        }
    }
    return 0;
}
