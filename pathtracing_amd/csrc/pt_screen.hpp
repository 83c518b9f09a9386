// Screen-space passes of renderMode 0 (RenderMode.Denoise): the edge-avoiding a-trous filter
// (SH:466-615) and the temporal anti-aliasing resolve (SH:617-714) of
// Assets/Scripts/RayTracingShader.compute, one thread per pixel.
//
// In the shader both run at the end of CSMain, per thread, with no barrier between them
// (SH:928-929): TAA() reads the `Result` texels of neighbouring threads that may or may not have
// written their denoiser() output yet.  Here the two are separate launches: the filter writes every
// pixel of a scratch image, then the resolve reads that image and writes `Result`, so every thread
// sees its neighbours' filtered value and nobody's resolved one.
//
// Both kernels stream: per pixel the filter reads 25 history texels that lie within +-8 pixels of it
// (served by L1/L2: algorithmic traffic is one 16-byte read and one 16-byte write per pixel) and the
// resolve reads 4 history texels, 4 filtered texels and writes one (48 bytes per pixel).  Neither is
// bound by those bytes: the filter is about 1 300 vector instructions per pixel (25 software exp, 25
// bounds-checked taps), the resolve 700.  Two rearrangements of the memory side were built, found bit-
// identical and measured at 1920x1080 against 58 and 41 microseconds for the kernels below: each
// workgroup's neighbourhood copied into LDS first and the taps read from there (7 and 4 sixteen-byte
// loads per lane instead of 26 and 13): 82 and 51; every tap's load made unconditional (texel 0
// stands in outside the image, selected away afterwards) so that none sits behind a branch: 66 and 41.
#pragma once

#include "pt_device.hpp"

namespace ptrace {

struct ScreenArgs {
    const float4 *history;    // Result1
    const float *seeds;       // _Seed of every pixel after its last sample (SH:564 reads the running counter)
    float4 *filtered;         // scratch: denoiser() of every pixel
    float4 *result;           // Result
    int32_t W, H;
};

// Texture2D.Load(int2): a float2 location is truncated toward zero; outside the texture reads 0.
__device__ __forceinline__ float4 load_trunc_int(const float4 *tex, int W, int H, float x, float y)
{
    const bool bad = !(x > -2147483648.0f && x < 2147483648.0f && y > -2147483648.0f && y < 2147483648.0f);
    const int ix = bad ? -1 : (int)x, iy = bad ? -1 : (int)y;
    if (ix < 0 || iy < 0 || ix >= W || iy >= H) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return tex[(size_t)iy * W + ix];
}

// tex[float2] (operator[] takes uint2): truncation, negative and NaN to 0; outside reads 0.
__device__ __forceinline__ float4 load_trunc_uint(const float4 *tex, int W, int H, float x, float y)
{
    const uint32_t ix = hlsl_ftou(x), iy = hlsl_ftou(y);
    if (ix >= (uint32_t)W || iy >= (uint32_t)H) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return tex[(size_t)iy * W + ix];
}

__device__ __forceinline__ float hash1(float seed) { return pt_frac(pt_sin(seed) * 43758.5453123f); }   // SH:475-477

constexpr int kScreenBlockX = 64, kScreenBlockY = 4;

// denoiser(fragCoord), SH:479-615.  The 25 weights of SH:513-542 are the outer product of
// (1/16, 1/4, 3/8, 1/4, 1/16) with itself; all products are exact in binary32.
__global__ __launch_bounds__(kScreenBlockX * kScreenBlockY) void denoise_kernel(const ScreenArgs a)
{
    const int x = blockIdx.x * kScreenBlockX + threadIdx.x, y = blockIdx.y * kScreenBlockY + threadIdx.y;
    if (x >= a.W || y >= a.H) return;
    const size_t p = (size_t)y * a.W + x;
    const float fx = (float)x, fy = (float)y;
    const float c_phi = 1.0f, n_phi = 0.5f, p_phi = 0.25f;
    const float4 c4 = load_trunc_int(a.history, a.W, a.H, fx, fy);                          // SH:552
    const f3 cval = pt_make3(c4.x, c4.y, c4.z);
    const float strength = 1.0f + (4.0f - 1.0f) * hash1((641.128752f * fx + 312.321374f * fy) + 1.92357812f * a.seeds[p]);   // SH:564
    const float n_w = pt_min(pt_exp(-(0.0f) / n_phi), 1.0f);                                // SH:592
    f3 sum = pt_make3(0, 0, 0), sum_f = pt_make3(0, 0, 0);
    float cum_w = 0.0f, cum_fw = 0.0f;
    const float kw[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
#pragma unroll
    for (int j = 0; j < 5; j++) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const float kern = kw[i] * kw[j];
            const float4 t4 = load_trunc_int(a.history, a.W, a.H, fx + (float)(i - 2) * strength, fy + (float)(j - 2) * strength);   // SH:572-575
            const f3 ctmp = pt_make3(t4.x, t4.y, t4.z);
            const f3 t = pt_sub3(cval, ctmp);
            const float dist2 = pt_dot3(t, t);
            const float c_w = pt_min(pt_exp(-(dist2) / c_phi), 1.0f);                        // SH:578
            const float r_w = c_w;                                                          // SH:593
            const float weight0 = c_w * n_w;
            sum.x += ctmp.x * weight0 * kern; sum.y += ctmp.y * weight0 * kern; sum.z += ctmp.z * weight0 * kern;   // SH:597-598
            cum_w += weight0 * kern;
            const float weight1 = r_w * n_w;
            sum_f.x += ctmp.x * weight1 * kern; sum_f.y += ctmp.y * weight1 * kern; sum_f.z += ctmp.z * weight1 * kern;
            cum_fw += weight1 * kern;
        }
    }
    const f3 u = pt_make3(sum.x / cum_w, sum.y / cum_w, sum.z / cum_w);
    const f3 v = pt_make3(sum_f.x / cum_fw, sum_f.y / cum_fw, sum_f.z / cum_fw);
    const float p_w = pt_min(pt_exp(-(pt_dot3(u, u)) / p_phi), 1.0f);                       // SH:608-612, ptmp = 0
    a.filtered[p] = make_float4(pt_min(pt_max(lerp1(u.x, v.x, p_w), 0.0f), 1.0f), pt_min(pt_max(lerp1(u.y, v.y, p_w), 0.0f), 1.0f),
                                pt_min(pt_max(lerp1(u.z, v.z, p_w), 0.0f), 1.0f), 0.0f);    // SH:614
}

__device__ __forceinline__ f3 encode_pal_yuv(f3 rgb)   // SH:624-632
{
    rgb = pt_make3(pt_pow_2(rgb.x), pt_pow_2(rgb.y), pt_pow_2(rgb.z));
    return pt_make3(pt_dot3(rgb, pt_make3(0.299f, 0.587f, 0.114f)), pt_dot3(rgb, pt_make3(-0.14713f, -0.28886f, 0.436f)),
                    pt_dot3(rgb, pt_make3(0.615f, -0.51499f, -0.10001f)));
}
__device__ __forceinline__ f3 decode_pal_yuv(f3 yuv)   // SH:634-642
{
    const f3 rgb = pt_make3(pt_dot3(yuv, pt_make3(1.0f, 0.0f, 1.13983f)), pt_dot3(yuv, pt_make3(1.0f, -0.39465f, -0.58060f)),
                            pt_dot3(yuv, pt_make3(1.0f, 2.03211f, 0.0f)));
    return pt_make3(pt_pow_half(rgb.x), pt_pow_half(rgb.y), pt_pow_half(rgb.z));
}
__device__ __forceinline__ f3 vmin3(f3 a, f3 b) { return pt_make3(pt_min(a.x, b.x), pt_min(a.y, b.y), pt_min(a.z, b.z)); }
__device__ __forceinline__ f3 vmax3(f3 a, f3 b) { return pt_make3(pt_max(a.x, b.x), pt_max(a.y, b.y), pt_max(a.z, b.z)); }
__device__ __forceinline__ f3 vlerp3(f3 a, f3 b, float t) { return pt_make3(lerp1(a.x, b.x, t), lerp1(a.y, b.y, t), lerp1(a.z, b.z, t)); }

// TAA(fragCoord), SH:644-714.  The eight neighbour reads of SH:673-680 offset the pixel position by
// 1/textureSize, so after index truncation they land on the pixel itself (+) or its left / upper
// neighbour (-); the arithmetic is kept as written so that sizes where 1/W vanishes behave alike.
__global__ __launch_bounds__(kScreenBlockX * kScreenBlockY) void taa_kernel(const ScreenArgs a, const RenderArgs ra)
{
    const int x = blockIdx.x * kScreenBlockX + threadIdx.x, y = blockIdx.y * kScreenBlockY + threadIdx.y;
    if (x >= a.W || y >= a.H) return;
    const float fx = (float)x, fy = (float)y;
    f2 uv; uv.x = fx; uv.y = fy;
    const float4 last = bilinear_history(ra, uv);                                            // SH:651
    f3 aa = pt_make3(last.x, last.y, last.z);
    float mixRate = 0.5f;
    const float offx = 1.0f / (float)a.W, offy = 1.0f / (float)a.H;                          // SH:657
    auto cur = [&](float px, float py) { const float4 q = load_trunc_uint(a.filtered, a.W, a.H, px, py); return pt_make3(q.x, q.y, q.z); };
    f3 in0 = cur(fx, fy);                                                                    // SH:659
    aa = vlerp3(pt_make3(aa.x * aa.x, aa.y * aa.y, aa.z * aa.z), pt_make3(in0.x * in0.x, in0.y * in0.y, in0.z * in0.z), mixRate);   // SH:661
    aa = pt_make3(pt_sqrt(aa.x), pt_sqrt(aa.y), pt_sqrt(aa.z));
    f3 in1 = cur(fx + offx, fy + 0.0f), in2 = cur(fx + -offx, fy + 0.0f);                    // SH:673-680
    f3 in3 = cur(fx + 0.0f, fy + offy), in4 = cur(fx + 0.0f, fy + -offy);
    f3 in5 = cur(fx + offx, fy + offy), in6 = cur(fx + -offx, fy + offy);
    f3 in7 = cur(fx + offx, fy + -offy), in8 = cur(fx + -offx, fy + -offy);
    aa = encode_pal_yuv(aa);
    in0 = encode_pal_yuv(in0); in1 = encode_pal_yuv(in1); in2 = encode_pal_yuv(in2); in3 = encode_pal_yuv(in3); in4 = encode_pal_yuv(in4);
    in5 = encode_pal_yuv(in5); in6 = encode_pal_yuv(in6); in7 = encode_pal_yuv(in7); in8 = encode_pal_yuv(in8);
    f3 minColor = vmin3(vmin3(vmin3(in0, in1), vmin3(in2, in3)), in4);                       // SH:693-698
    f3 maxColor = vmax3(vmax3(vmax3(in0, in1), vmax3(in2, in3)), in4);
    minColor = vlerp3(minColor, vmin3(vmin3(vmin3(in5, in6), vmin3(in7, in8)), minColor), 0.5f);
    maxColor = vlerp3(maxColor, vmax3(vmax3(vmax3(in5, in6), vmax3(in7, in8)), maxColor), 0.5f);
    const f3 preclamping = aa;
    aa = vmin3(vmax3(aa, minColor), maxColor);                                               // clamp(), SH:701
    mixRate = 1.0f / (1.0f / mixRate + 1.0f);                                                // SH:703
    const f3 diff = pt_sub3(aa, preclamping);
    mixRate += pt_dot3(diff, diff) * 4.0f;                                                   // SH:706-708
    mixRate = pt_min(pt_max(mixRate, 0.05f), 0.5f);
    aa = decode_pal_yuv(aa);
    a.result[(size_t)y * a.W + x] = make_float4(aa.x, aa.y, aa.z, mixRate);                  // SH:713, SH:929
}

// ---- primary rays only: the _Depth == 1 dispatch (SH:799-809) and the primary-hit pass of renderMode 6 ----
//
// One pixel per thread and pass over the rectangle (8x8 pixel blocks, so a wave's rays are neighbours and walk
// the tree almost in step); no path state, no regeneration, so none of the persistent kernel's machinery.
// what == 1: Result = distance / 16 in all channels (+ hit ids); what == 2: t of the hit to a.seed_out,
// rows skip_lo .. skip_hi (the stripe being path-traced) left out; what == 3: as 1, but only for pixels whose
// entry in a.skip_t is finite.
template <int WIDE>
__global__ __launch_bounds__(kBlock) void primary_kernel(const RenderArgs a, int what, int skip_lo, int skip_hi)
{
    __shared__ __align__(8) uint32_t lds_stack[StackFor<WIDE>::kWords * kBlock];
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x, n_threads = gridDim.x * kBlock;
    const typename StackFor<WIDE>::type st = StackFor<WIDE>::make(lds_stack, a.stack_overflow + tid, n_threads);
    const int lane = threadIdx.x & 63;
    uint32_t cn = 0, cp = 0, rays = 0;
    // the grid is capped (the stack overflow area is per resident thread); a thread takes every n_threads-th pixel
    for (uint32_t item = tid; item < a.total_items; item += n_threads) {
        int x = 0, y = 0;
        if (!item_pixel(a, item, x, y) || (what == 2 && y >= skip_lo && y <= skip_hi)) continue;
        const size_t p = (size_t)y * a.W + x;
        if (what == 3 && !(a.skip_t[p] < PT_INF)) continue;               // renderMode 8: the thread returned at SH:760-763
        f3 o, d;
        camera_ray(a, x, y, o, d);
        float t = PT_INF; int32_t id = -1; uint32_t rec = 0;
        intersect_ground_plane(o, d, t, id);
        intersect_far_spheres(a.sc, o, d, t, id);
        traverse<WIDE>(a.sc, o, d, t, id, rec, st, cn, cp);
        rays++;
        if (what == 1 || what == 3) {
            const float dis = t / 16.0f;                                  // SH:806-807
            a.result[p] = make_float4(dis, dis, dis, dis);
            if (a.hit_ids) a.hit_ids[p] = id;
        } else {
            a.seed_out[p] = t;
        }
    }
    stats_add(a.stats, 0, rays, lane);
    stats_add(a.stats, 1, cn, lane);
    stats_add(a.stats, 2, cp, lane);
    stats_add(a.stats, 3, rays, lane);
}

// ---- the display pass Assets/Shaders/Denoiser.shader (Graphics.Blit(_target, destination, denoiser),
// RayTracingMaster.cs:760-764, when imageBlur is on) ----
//
// A fragment shader: the same 25-tap filter as above, but on `Result` through tex2D with normalised
// coordinates, weights c_w * n_w with n_w from the same colour difference at n_phi = 0.5 (:153-156), and a
// stride hash without _Seed (:136).  What the shader leaves to the sampler is fixed here as: bilinear
// filtering with full float weights, clamp addressing, texel centres at (i + 0.5) / size; the fixed4
// return value is kept in float.

// tex2D(_MainTex, uv) under those rules.
__device__ __forceinline__ float4 tex2d_bilinear_clamp(const float4 *tex, int W, int H, float u, float v)
{
    const float px = u * (float)W - 0.5f, py = v * (float)H - 0.5f;
    const float x0f = pt_floor(px), y0f = pt_floor(py);
    const float fx = px - x0f, fy = py - y0f;
    const int x0 = (int)fminf(fmaxf(x0f, 0.0f), (float)(W - 1)), x1 = (int)fminf(fmaxf(x0f + 1.0f, 0.0f), (float)(W - 1));
    const int y0 = (int)fminf(fmaxf(y0f, 0.0f), (float)(H - 1)), y1 = (int)fminf(fmaxf(y0f + 1.0f, 0.0f), (float)(H - 1));
    const float4 c00 = tex[(size_t)y0 * W + x0], c10 = tex[(size_t)y0 * W + x1];
    const float4 c01 = tex[(size_t)y1 * W + x0], c11 = tex[(size_t)y1 * W + x1];
    const float4 a = make_float4(lerp1(c00.x, c10.x, fx), lerp1(c00.y, c10.y, fx), lerp1(c00.z, c10.z, fx), lerp1(c00.w, c10.w, fx));
    const float4 b = make_float4(lerp1(c01.x, c11.x, fx), lerp1(c01.y, c11.y, fx), lerp1(c01.z, c11.z, fx), lerp1(c01.w, c11.w, fx));
    return make_float4(lerp1(a.x, b.x, fy), lerp1(a.y, b.y, fy), lerp1(a.z, b.z, fy), lerp1(a.w, b.w, fy));
}

// Denoiser.shader:50-184 + frag (:186-215); src = Result, dst = the display image.
__global__ __launch_bounds__(kScreenBlockX * kScreenBlockY) void display_denoise_kernel(const float4 *src, float4 *dst, int W, int H)
{
    const int x = blockIdx.x * kScreenBlockX + threadIdx.x, y = blockIdx.y * kScreenBlockY + threadIdx.y;
    if (x >= W || y >= H) return;
    const float fw = (float)W, fh = (float)H;
    const float u = ((float)x + 0.5f) / fw, v = ((float)y + 0.5f) / fh;     // the interpolated i.uv at the pixel centre
    const float c_phi = 1.0f, r_phi = 1.0f, n_phi = 0.5f, p_phi = 0.25f;
    const float4 c4 = tex2d_bilinear_clamp(src, W, H, u, v);                // :124-126, cval = rval = nval
    const f3 cval = pt_make3(c4.x, c4.y, c4.z);
    const float strength = (1.0f + (4.0f - 1.0f) * hash1((641.128752f * u + 312.321374f * v) + 1.92357812f)) * 1.0f;   // :136
    f3 sum = pt_make3(0, 0, 0), sum_f = pt_make3(0, 0, 0);
    float cum_w = 0.0f, cum_fw = 0.0f;
    const float kw[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
#pragma unroll
    for (int j = 0; j < 5; j++) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const float kern = kw[i] * kw[j];
            const float4 t4 = tex2d_bilinear_clamp(src, W, H, u + (float)(i - 2) * strength / fw, v + (float)(j - 2) * strength / fh);   // :144-147
            const f3 ctmp = pt_make3(t4.x, t4.y, t4.z);
            const f3 t = pt_sub3(cval, ctmp);
            const float dist2 = pt_dot3(t, t);
            const float c_w = pt_min(pt_exp(-(dist2) / c_phi), 1.0f);                 // :150
            const float n_w = pt_min(pt_exp(-(pt_max(dist2, 0.0f)) / n_phi), 1.0f);   // :153-156, ntmp = ctmp
            const float r_w = pt_min(pt_exp(-(dist2) / r_phi), 1.0f);                 // :159-162, rtmp = ctmp
            const float weight0 = c_w * n_w;
            sum.x += ctmp.x * weight0 * kern; sum.y += ctmp.y * weight0 * kern; sum.z += ctmp.z * weight0 * kern;
            cum_w += weight0 * kern;
            const float weight1 = r_w * n_w;
            sum_f.x += ctmp.x * weight1 * kern; sum_f.y += ctmp.y * weight1 * kern; sum_f.z += ctmp.z * weight1 * kern;
            cum_fw += weight1 * kern;
        }
    }
    const f3 a = pt_make3(sum.x / cum_w, sum.y / cum_w, sum.z / cum_w);
    const f3 b = pt_make3(sum_f.x / cum_fw, sum_f.y / cum_fw, sum_f.z / cum_fw);
    const float p_w = pt_min(pt_exp(-(pt_dot3(a, a)) / p_phi), 1.0f);                 // :177-181
    dst[(size_t)y * W + x] = make_float4(pt_min(pt_max(lerp1(a.x, b.x, p_w), 0.0f), 1.0f), pt_min(pt_max(lerp1(a.y, b.y, p_w), 0.0f), 1.0f),
                                         pt_min(pt_max(lerp1(a.z, b.z, p_w), 0.0f), 1.0f), 0.0f);   // :183
}

// ---- pixels that show the history instead of a traced path: renderMode 5 / 6 outside the stripe being rendered
// (SH:742-789), every pixel of renderMode 7 / 8 (same lines) and of renderMode 11 / 12 (SH:1071-1104) ----

struct StripeArgs {
    const PtCameraHistory *cams;   // the four *Old arrays, PT_MAX_CAMERA_HISTORY elements
    const float *t_primary;        // even modes: t of every pixel's primary hit (+inf = miss)
    int32_t divisions, counter, mode;
};

// FindCameraRay(input, framesFrom) (SH:394-411) against history element `cam`.
__device__ __forceinline__ f2 find_camera_ray_in(const PtCameraHistory &cam, int W, int H, f3 dir)
{
    const float m11 = cam.camera_projection[5], m22 = cam.camera_projection[10], m23 = cam.camera_projection[14];
    const float near = -m23 / m22;
    const float far = -(near * m22) / (1.0f - m22);
    const f3 pos = pt_scale3(pt_normalize3(dir), near + (far - near));
    const float ratio = (float)((uint32_t)W / (uint32_t)H);
    const f3 toCam = pt_mul_dir(cam.world_to_camera, pos);
    const float height = toCam.z / m11;
    const float width = ratio * height;
    const float dimx = (float)W, dimy = (float)H;
    const float uvx = (toCam.x - width * (dimx / dimy)) / width;
    const float uvy = (toCam.y - height) / height;
    f2 r;
    r.x = 0.5f - uvx * (dimy * ratio / 2.0f);
    r.y = (0.5f - uvy * (dimy / 2.0f)) - 0.005f;
    return r;
}

// One thread per pixel; in renderMode 5 / 6 pixels of the stripe being path-traced are left alone.
__global__ __launch_bounds__(kScreenBlockX * kScreenBlockY) void stripe_reproject_kernel(const RenderArgs a, const StripeArgs sa)
{
    const int x = blockIdx.x * kScreenBlockX + threadIdx.x, y = blockIdx.y * kScreenBlockY + threadIdx.y;
    if (x >= a.W || y >= a.H) return;
    const float fh = (float)a.H, fy = (float)y;
    const bool partial = sa.mode == 5 || sa.mode == 6;
    const float stripe = fh / (float)sa.divisions;                                            // SH:742
    if (partial && !(fy > fh - stripe * (float)sa.counter || fy < fh - stripe * (float)(sa.counter + 1))) return;   // SH:751
    const size_t p = (size_t)y * a.W + x;
    f3 o, d;
    camera_ray(a, x, y, o, d);                                                                // SH:753-754, SH:848
    f3 hit_pos = pt_along3(o, 10.0f, d);                                                      // SH:755, SH:855
    if (sa.mode % 2 == 0) {                                                                   // SH:757-764, SH:858-863
        const float t = sa.t_primary[p];
        if (!(t < PT_INF)) {
            if (sa.mode == 12) a.result[p] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);             // SH:1078, SH:1102
            else { const f3 sky = sky_sample(a.sc, d); a.result[p] = make_float4(sky.x, sky.y, sky.z, 1.0f); }   // SH:760-762
            return;
        }
        hit_pos = pt_along3(o, t, d);
    }
    int frames_back = 0;
    if (partial) {
        const int one_mod = 1 % sa.divisions;
        frames_back = (int)(((float)sa.divisions - fy / stripe) - (float)one_mod);            // SH:766
        frames_back = frames_back < 0 ? 0 : frames_back > PT_MAX_CAMERA_HISTORY - 1 ? PT_MAX_CAMERA_HISTORY - 1 : frames_back;   // stay inside the arrays
    }
    const PtCameraHistory &cam = sa.cams[frames_back];
    const f3 origin_old = pt_make3(cam.camera_to_world[12], cam.camera_to_world[13], cam.camera_to_world[14]);
    const f2 uv = find_camera_ray_in(cam, a.W, a.H, pt_sub3(hit_pos, origin_old));            // SH:770, SH:1074
    float4 out = make_float4(0.0f, 0.0f, 0.0f, 1.0f);                                         // SH:780, SH:1102
    if (uv.x < (float)a.W && uv.x > 0.0f && uv.y < (float)a.H && uv.y > 0.0f) out = bilinear_history(a, uv);   // SH:772-776, SH:1076-1097
    a.result[p] = out;
}

}  // namespace ptrace
