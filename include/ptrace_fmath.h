/*
 * ptrace_fmath.h - the fp32 definition of every HLSL intrinsic the path-tracing
 * kernel uses.
 *
 * The reference kernel (Assets/Scripts/RayTracingShader.compute) is written on
 * top of HLSL intrinsics (sin cos acos atan2 pow sqrt normalize cross dot
 * reflect frac saturate lerp mul) whose precision is left to whichever shader
 * compiler and GPU run it.  Its random-number hash multiplies sin() of an
 * argument of up to ~1e6 by 43758.5453 (RayTracingShader.compute:50-55), so a
 * single ulp of difference in sin() changes which way a path bounces.  Two
 * implementations can therefore only agree bit for bit if they share ONE
 * explicit definition of these intrinsics.  This header is that definition:
 *
 *   - every function is built from IEEE-754 binary32 +,-,*,/ and sqrt (all
 *     correctly rounded), explicit fused multiply-adds, and exact bit
 *     operations; nothing calls a math library;
 *   - it must be compiled with floating-point contraction OFF
 *     (-ffp-contract=off) and without fast-math on every side, so the only
 *     fused operations are the ones spelled pt_fma();
 *   - it is plain C99 and also valid C++ / HIP device code.
 *
 * Users: the HIP kernels (device side) and the CPU oracle (host side).  The
 * header holds intrinsics only; the path-tracing algorithm itself is written
 * separately on each side.
 */
#ifndef PTRACE_FMATH_H
#define PTRACE_FMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PT_FN __host__ __device__ static inline __attribute__((always_inline))
#else
#define PT_FN static inline
#endif

typedef struct pt_f3 { float x, y, z; } pt_f3;

/* ---- exact primitives ------------------------------------------------- */
PT_FN float pt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_FN float pt_sqrt(float x) { return __builtin_sqrtf(x); }
PT_FN float pt_floor(float x) { return __builtin_floorf(x); }
PT_FN float pt_rint(float x) { return __builtin_rintf(x); } /* ties to even */
PT_FN float pt_abs(float x) { return __builtin_fabsf(x); }
PT_FN uint32_t pt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
PT_FN float pt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
/* min/max with a fixed NaN rule (second operand wins), as compare+select. */
PT_FN float pt_min(float a, float b) { return (a < b) ? a : b; }
PT_FN float pt_max(float a, float b) { return (a > b) ? a : b; }

#define PT_INF (__builtin_inff())
#define PT_PI 3.14159265f          /* RayTracingShader.compute:28 */
#define PT_EPSILON 1e-8f           /* RayTracingShader.compute:29 */

/* ---- scalar helpers --------------------------------------------------- */
PT_FN float pt_frac(float x) { return x - pt_floor(x); }
PT_FN float pt_saturate(float x) { return !(x > 0.0f) ? 0.0f : (x > 1.0f ? 1.0f : x); }

/* ---- sin / cos ---------------------------------------------------------
 * k = rint(x * 2/pi); r = x - k*pi/2 by a three-term FMA reduction (pi/2 split
 * into three floats), then a polynomial on |r| <= pi/4 (+ reduction slack),
 * selected by quadrant.  Arguments with |x| >= 1e9 (or NaN) return 0.      */
#define PT_TWO_OVER_PI 0x1.45f306p-1f
#define PT_PIO2_HI 0x1.921fb6p+0f
#define PT_PIO2_MID (-0x1.777a5cp-25f)
#define PT_PIO2_LO (-0x1.ee59dap-50f)

PT_FN float pt__sin_poly(float r)
{
    float w = r * r;
    float p = 0x1.6da91ep-19f;
    p = pt_fma(p, w, -0x1.a01366p-13f);
    p = pt_fma(p, w, 0x1.11110ep-7f);
    p = pt_fma(p, w, -0x1.555556p-3f);
    return pt_fma(r * w, p, r);
}
PT_FN float pt__cos_poly(float r)
{
    float w = r * r;
    float p = -0x1.25170cp-22f;
    p = pt_fma(p, w, 0x1.a0159cp-16f);
    p = pt_fma(p, w, -0x1.6c16c0p-10f);
    p = pt_fma(p, w, 0x1.555556p-5f);
    return pt_fma(w * w, p, pt_fma(-0.5f, w, 1.0f));
}
PT_FN float pt__reduce_pio2(float x, int *quadrant)
{
    float k = pt_rint(x * PT_TWO_OVER_PI);
    float r = pt_fma(k, -PT_PIO2_HI, x);
    r = pt_fma(k, -PT_PIO2_MID, r);
    r = pt_fma(k, -PT_PIO2_LO, r);
    *quadrant = ((int)k) & 3;
    return r;
}
PT_FN float pt_sin(float x)
{
    if (!(pt_abs(x) < 1.0e9f)) return 0.0f;
    int q;
    float r = pt__reduce_pio2(x, &q);
    float v = (q & 1) ? pt__cos_poly(r) : pt__sin_poly(r);
    return (q & 2) ? -v : v;
}
PT_FN float pt_cos(float x)
{
    if (!(pt_abs(x) < 1.0e9f)) return 0.0f;
    int q;
    float r = pt__reduce_pio2(x, &q);
    q += 1;
    float v = (q & 1) ? pt__cos_poly(r) : pt__sin_poly(r);
    return (q & 2) ? -v : v;
}

/* ---- log2 / exp2 / pow -------------------------------------------------
 * log2: x = m * 2^e with m in [sqrt(.5), sqrt(2)); s = (m-1)/(m+1);
 *       log2(m) = s * L(s^2).  Subnormal x is pre-scaled by 2^24.
 * exp2: n = rint(x), f = x - n in [-.5,.5]; 2^f by a degree-6 polynomial;
 *       scaled by 2^n in two normal-range steps.
 * pow(x,y) = exp2(y * log2(x)) for x > 0; for x <= 0 it is 1 if y == 0
 *       else 0 (the shader only raises non-negative bases, to positive
 *       powers: RayTracingShader.compute:307, :321).                        */
PT_FN float pt_log2(float x)
{
    int e = 0;
    if (x < 0x1p-126f) { x = x * 0x1p24f; e = -24; }
    uint32_t b = pt_f2u(x);
    e += (int)(b >> 23) - 127;
    float m = pt_u2f((b & 0x007fffffu) | 0x3f800000u);
    if (m > 0x1.6a09e6p+0f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float w = s * s;
    float p = 0x1.5d55b2p-2f;
    p = pt_fma(p, w, 0x1.a587dap-2f);
    p = pt_fma(p, w, 0x1.27778cp-1f);
    p = pt_fma(p, w, 0x1.ec709ep-1f);
    p = pt_fma(p, w, 0x1.715476p+1f);
    return pt_fma(s, p, (float)e);
}
PT_FN float pt_exp2(float x)
{
    if (!(x > -252.0f)) x = -252.0f;
    if (x > 254.0f) x = 254.0f;
    float nf = pt_rint(x);
    float f = x - nf;
    int n = (int)nf;
    float p = 0x1.446c7ep-13f;
    p = pt_fma(p, f, 0x1.5f48c8p-10f);
    p = pt_fma(p, f, 0x1.3b29d8p-7f);
    p = pt_fma(p, f, 0x1.c6aeccp-5f);
    p = pt_fma(p, f, 0x1.ebfbe0p-3f);
    p = pt_fma(p, f, 0x1.62e430p-1f);
    p = pt_fma(p, f, 1.0f);
    int n1 = n >> 1;
    int n2 = n - n1;
    float s1 = pt_u2f((uint32_t)(n1 + 127) << 23);
    float s2 = pt_u2f((uint32_t)(n2 + 127) << 23);
    return (p * s1) * s2;
}
/* exp(x) = 2^(x * log2(e)); used by the denoiser weights (RayTracingShader.compute:578,612). */
PT_FN float pt_exp(float x) { return pt_exp2(x * 0x1.715476p+0f); }
PT_FN float pt_pow(float x, float y)
{
    if (!(x > 0.0f)) return (y == 0.0f) ? 1.0f : 0.0f;
    return pt_exp2(y * pt_log2(x));
}

/* pow(x, 2.0) and pow(x, 0.5) with literal exponents (RayTracingShader.compute:626,641): the correctly
 * rounded results x*x and sqrt(x) - what a shader compiler emits for these literals - with pt_pow's
 * convention for x <= 0. */
PT_FN float pt_pow_2(float x) { return (x > 0.0f) ? x * x : 0.0f; }
PT_FN float pt_pow_half(float x) { return (x > 0.0f) ? pt_sqrt(x) : 0.0f; }

/* ---- acos / atan2 ------------------------------------------------------ */
PT_FN float pt__asin_poly(float z)
{
    float w = z * z;
    float p = 0x1.1504e2p-5f;
    p = pt_fma(p, w, 0x1.17b5fap-6f);
    p = pt_fma(p, w, 0x1.fdd33ep-6f);
    p = pt_fma(p, w, 0x1.6d58acp-5f);
    p = pt_fma(p, w, 0x1.33343cp-4f);
    p = pt_fma(p, w, 0x1.555554p-3f);
    return pt_fma(z * w, p, z);
}
/* acos with the argument clamped to [-1,1] (NaN -> acos(1) = 0). */
PT_FN float pt_acos(float x)
{
    if (!(x < 1.0f)) x = 1.0f;
    if (x < -1.0f) x = -1.0f;
    float ax = pt_abs(x);
    if (ax <= 0.5f) return 0x1.921fb6p+0f - pt__asin_poly(x);
    float z = pt_sqrt((1.0f - ax) * 0.5f);
    float a = 2.0f * pt__asin_poly(z);
    return (x > 0.0f) ? a : (0x1.921fb6p+1f - a);
}
PT_FN float pt__atan_poly(float z)
{
    float w = z * z;
    float p = -0x1.06de6cp-4f;
    p = pt_fma(p, w, 0x1.b7a68ep-4f);
    p = pt_fma(p, w, -0x1.241b82p-3f);
    p = pt_fma(p, w, 0x1.999710p-3f);
    p = pt_fma(p, w, -0x1.555554p-2f);
    return pt_fma(z * w, p, z);
}
/* atan2(y,x) in (-pi, pi]; atan2(0,0) = 0; -0 counts as +0. */
PT_FN float pt_atan2(float y, float x)
{
    float ax = pt_abs(x), ay = pt_abs(y);
    float mx = pt_max(ax, ay), mn = pt_min(ax, ay);
    if (!(mx > 0.0f)) return 0.0f;
    float a = mn / mx;
    float r;
    if (a > 0x1.a8279ap-2f)
        r = 0x1.921fb6p-1f + pt__atan_poly((a - 1.0f) / (a + 1.0f));
    else
        r = pt__atan_poly(a);
    if (ay > ax) r = 0x1.921fb6p+0f - r;
    if (x < 0.0f) r = 0x1.921fb6p+1f - r;
    return (y < 0.0f) ? -r : r;
}

/* ---- float3 ----------------------------------------------------------- */
PT_FN pt_f3 pt_make3(float x, float y, float z) { pt_f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_FN pt_f3 pt_add3(pt_f3 a, pt_f3 b) { return pt_make3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_FN pt_f3 pt_sub3(pt_f3 a, pt_f3 b) { return pt_make3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_FN pt_f3 pt_mul3(pt_f3 a, pt_f3 b) { return pt_make3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_FN pt_f3 pt_scale3(pt_f3 a, float s) { return pt_make3(a.x * s, a.y * s, a.z * s); }
/* dot(a,b): x product first, then y and z folded in by FMA. */
PT_FN float pt_dot3(pt_f3 a, pt_f3 b) { return pt_fma(a.z, b.z, pt_fma(a.y, b.y, a.x * b.x)); }
/* cross(a,b): each component is fma(a_i, b_j, -(a_j * b_i)). */
PT_FN pt_f3 pt_cross3(pt_f3 a, pt_f3 b)
{
    return pt_make3(pt_fma(a.y, b.z, -(a.z * b.y)),
                    pt_fma(a.z, b.x, -(a.x * b.z)),
                    pt_fma(a.x, b.y, -(a.y * b.x)));
}
PT_FN float pt_length3(pt_f3 a) { return pt_sqrt(pt_dot3(a, a)); }
PT_FN pt_f3 pt_normalize3(pt_f3 a) { return pt_scale3(a, 1.0f / pt_sqrt(pt_dot3(a, a))); }
/* reflect(i,n) = i - 2*dot(n,i)*n */
PT_FN pt_f3 pt_reflect3(pt_f3 i, pt_f3 n)
{
    float k = 2.0f * pt_dot3(n, i);
    return pt_make3(i.x - k * n.x, i.y - k * n.y, i.z - k * n.z);
}
/* o + t*d, component-wise, multiply and add rounded separately. */
PT_FN pt_f3 pt_along3(pt_f3 o, float t, pt_f3 d) { return pt_make3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z); }

/* ---- 4x4 matrices ------------------------------------------------------
 * m[16] is in UnityEngine.Matrix4x4 memory order (column-major): element
 * (row i, column j) is m[j*4+i].  mul(M, float4(v,1)).xyz and
 * mul(M, float4(v,0)).xyz as used at RayTracingShader.compute:111-116,:238. */
PT_FN pt_f3 pt_mul_point(const float *m, pt_f3 v)
{
    pt_f3 r;
    r.x = pt_fma(m[8], v.z, pt_fma(m[4], v.y, m[0] * v.x)) + m[12];
    r.y = pt_fma(m[9], v.z, pt_fma(m[5], v.y, m[1] * v.x)) + m[13];
    r.z = pt_fma(m[10], v.z, pt_fma(m[6], v.y, m[2] * v.x)) + m[14];
    return r;
}
PT_FN pt_f3 pt_mul_dir(const float *m, pt_f3 v)
{
    pt_f3 r;
    r.x = pt_fma(m[8], v.z, pt_fma(m[4], v.y, m[0] * v.x));
    r.y = pt_fma(m[9], v.z, pt_fma(m[5], v.y, m[1] * v.x));
    r.z = pt_fma(m[10], v.z, pt_fma(m[6], v.y, m[2] * v.x));
    return r;
}

#endif /* PTRACE_FMATH_H */
