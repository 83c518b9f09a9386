/*
 * pt_oracle.c - CPU oracle for libptrace.  TEST INFRASTRUCTURE ONLY; see
 * pt_oracle.h for what it is, who may load it, and why parity is unpinned.
 *
 * SH = Assets/Scripts/RayTracingShader.compute, ADD = Assets/Shaders/AddMat.shader
 * of the reference.  Every function names the lines it follows.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -mfma -fopenmp (see Makefile).
 */
#include "pt_oracle.h"
#include "../include/ptrace_fmath.h"

#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef pt_f3 f3;

/* ------------------------------------------------------------------ types */

typedef struct Ray { f3 origin, direction, energy; } Ray;        /* SH:92-97 */

typedef struct RayHit {                                          /* SH:126-135 */
    f3 position; float distance; f3 normal; f3 albedo; f3 specular; float smoothness; f3 emission;
    int32_t lobe; /* PtMaterial.lobe; not in the shader (always its Phong lobe there) */
    int32_t id; /* not in the shader: -1 miss, 0 plane, 1+i sphere i, 1+nS+k triangle k */
} RayHit;

typedef struct Tri { f3 v0, v1, v2; int32_t object; } Tri;

/* thin, len: the largest L^2 / |e1 x e2| and the longest edge L among the triangles below the node (see box_hit) */
typedef struct ONode { float lo[3], hi[3]; int32_t left, right; int32_t first, count; float thin, len; } ONode;

struct PtoPrepared {
    PtSphere *spheres; int32_t n_spheres;
    Tri *tris; int32_t n_tris;           /* world space, reference order */
    PtMaterial *materials; int32_t n_materials;   /* per mesh object; none = SH:250-253 for every triangle */
    float *sky; int32_t sky_w, sky_h;
    int accel;
    ONode *nodes; int32_t n_nodes; int32_t *order; /* BVH over the triangles' prim ids (n_spheres + 1 ..) */
};

typedef struct RngState { float seed; float px, py; } RngState;  /* _Seed, _Pixel  SH:47-48 */

typedef struct Counters { uint64_t trace_calls, rays, paths, prims; float *log; int log_cap, log_n; } Counters;

/* ------------------------------------------------------------- utilities */

static float sdot(f3 x, f3 y, float f) { return pt_saturate(pt_dot3(x, y) * f); }      /* SH:34-37 */
static float energy(f3 c) { float k = 1.0f / 3.0f; return pt_dot3(c, pt_make3(k, k, k)); } /* SH:39-42 */

/* SH:50-55 */
static float rng(RngState *s)
{
    float d = pt_fma(s->py, 78.233f, s->px * 12.9898f); /* dot(_Pixel, float2(12.9898, 78.233)) */
    float result = pt_frac(pt_sin(s->seed / 100.0f * d) * 43758.5453f);
    s->seed += 1.0f;
    return result;
}

static Ray create_ray(f3 o, f3 d) { Ray r; r.origin = o; r.direction = d; r.energy = pt_make3(1, 1, 1); return r; } /* SH:99-106 */

/* SH:108-120 */
static Ray create_camera_ray(const float *c2w, const float *ip, float u, float v)
{
    f3 origin = pt_make3(c2w[12], c2w[13], c2w[14]); /* mul(_CameraToWorld, (0,0,0,1)).xyz */
    /* mul(_CameraInverseProjection, float4(uv, 0, 1)).xyz */
    f3 d;
    d.x = pt_fma(ip[4], v, ip[0] * u) + ip[12];
    d.y = pt_fma(ip[5], v, ip[1] * u) + ip[13];
    d.z = pt_fma(ip[6], v, ip[2] * u) + ip[14];
    d = pt_mul_dir(c2w, d);
    d = pt_normalize3(d);
    return create_ray(origin, d);
}

/* SH:137-148 */
static RayHit create_ray_hit(void)
{
    RayHit h; memset(&h, 0, sizeof h);
    h.distance = PT_INF; h.id = -1;
    return h;
}

/* ---------------------------------------------------------- intersection */

/* SH:154-168 */
static void intersect_ground_plane(const Ray *ray, RayHit *best)
{
    float t = -ray->origin.y / ray->direction.y;
    if (t > 0 && t < best->distance) {
        best->distance = t;
        best->position = pt_along3(ray->origin, t, ray->direction);
        best->normal = pt_make3(0, 1, 0);
        best->albedo = pt_make3(0.5f, 0.5f, 0.5f);
        best->specular = pt_make3(0.03f, 0.03f, 0.03f);
        best->smoothness = 0.2f;
        best->lobe = PT_LOBE_PHONG;
        best->emission = pt_make3(0, 0, 0);
        best->id = 0;
    }
}

/* SH:170-190.  `id` only labels the hit. */
static void intersect_sphere(const Ray *ray, RayHit *best, const PtSphere *s, int32_t id, int lexi)
{
    f3 c = pt_make3(s->position[0], s->position[1], s->position[2]);
    f3 d = pt_sub3(ray->origin, c);
    float p1 = -pt_dot3(ray->direction, d);
    float p2sqr = p1 * p1 - pt_dot3(d, d) + s->radius * s->radius;
    if (p2sqr < 0) return;
    float p2 = pt_sqrt(p2sqr);
    float t = p1 - p2 > 0 ? p1 - p2 : p1 + p2;
    if (t > 0 && (t < best->distance || (lexi && t == best->distance && id < best->id))) {
        best->distance = t;
        best->position = pt_along3(ray->origin, t, ray->direction);
        best->normal = pt_normalize3(pt_sub3(best->position, c));
        best->albedo = pt_make3(s->albedo[0], s->albedo[1], s->albedo[2]);
        best->specular = pt_make3(s->specular[0], s->specular[1], s->specular[2]);
        best->smoothness = s->smoothness;
        best->emission = pt_make3(s->emission[0], s->emission[1], s->emission[2]);
        best->lobe = PT_LOBE_PHONG;
        best->id = id;
    }
}

/* SH:192-230 */
static int intersect_triangle_mt97(const Ray *ray, f3 vert0, f3 vert1, f3 vert2, float *t, float *u, float *v)
{
    f3 edge1 = pt_sub3(vert1, vert0);
    f3 edge2 = pt_sub3(vert2, vert0);
    f3 pvec = pt_cross3(ray->direction, edge2);
    float det = pt_dot3(edge1, pvec);
    if (det < PT_EPSILON) return 0;          /* backface culling */
    float inv_det = 1.0f / det;
    f3 tvec = pt_sub3(ray->origin, vert0);
    *u = pt_dot3(tvec, pvec) * inv_det;
    if (*u < 0.0f || *u > 1.0f) return 0;
    f3 qvec = pt_cross3(tvec, edge1);
    *v = pt_dot3(ray->direction, qvec) * inv_det;
    if (*v < 0.0f || *u + *v > 1.0f) return 0;
    *t = pt_dot3(edge2, qvec) * inv_det;
    return 1;
}

/* Body of the loop at SH:236-256 for one triangle already in world space (the
 * shader re-applies localToWorldMatrix per ray, SH:238-240; pto_prepare applies
 * the same pt_mul_point once, which yields the same bits). */
static void intersect_tri(const Ray *ray, RayHit *best, const Tri *tr, int32_t id, int lexi, const PtMaterial *materials)
{
    float t, u, v;
    if (intersect_triangle_mt97(ray, tr->v0, tr->v1, tr->v2, &t, &u, &v)) {
        if (t > 0 && (t < best->distance || (lexi && t == best->distance && id < best->id))) {
            best->distance = t;
            best->position = pt_along3(ray->origin, t, ray->direction);
            best->normal = pt_normalize3(pt_cross3(pt_sub3(tr->v1, tr->v0), pt_sub3(tr->v2, tr->v0)));
            best->albedo = pt_make3(0, 0, 0);                       /* SH:250-253 */
            best->specular = pt_make3(0.65f, 0.65f, 0.65f);
            best->smoothness = 0.99f;
            best->emission = pt_make3(0, 0, 0);
            best->lobe = PT_LOBE_PHONG;
            if (materials) {                                        /* extension: the object's own surface */
                const PtMaterial *m = &materials[tr->object];
                best->albedo = pt_make3(m->albedo[0], m->albedo[1], m->albedo[2]);
                best->specular = pt_make3(m->specular[0], m->specular[1], m->specular[2]);
                best->smoothness = m->smoothness;
                best->emission = pt_make3(m->emission[0], m->emission[1], m->emission[2]);
                best->lobe = m->lobe;
            }
            best->id = id;
        }
    }
}

/* The oracle's own acceleration structure (accel != 0): a median-split BVH
 * over the triangles.  Not in the reference.  It prunes with padded
 * boxes and picks the winner by (distance, id) - the same winner the
 * exhaustive strict-less-than loop in id order picks.
 * What the boxes must not prune are the hits that only rounding produces: IntersectSphere (SH:170-190) decides by
 * the sign of a difference of squares of the origin's distance D, so it accepts rays passing sqrt(r^2 + k eps D^2) from
 * the centre - units beside a sphere for an origin some 10^4 units out on the ground plane - which is why the spheres
 * (at most a few hundred) are not in the tree at all: trace() tests every one of them, as the shader does.  Moeller-
 * Trumbore's barycentrics err by about 8 eps (L^2 / |e1 x e2|) max(L, D) in world units - a sliver ten units long and
 * a thousandth wide reports hits three units beyond its end from 2 10^4 units away - so every box is widened, for one
 * ray at distance `far` from the scene, by that expression for the thinnest and the longest triangle below it, and by
 * `margin` (sixteen times a well-shaped triangle's reach at that distance) at least. */
static int box_hit(const ONode *n, const Ray *ray, f3 inv, float best, float margin, float far)
{
    const float own = 8.0f * 5.9604645e-8f * n->thin * fmaxf(n->len, far);
    if (own > margin) margin = own;
    if (!(margin < 1e30f)) margin = 1e30f;        /* a triangle without area below: always entered */
    float tx0 = (n->lo[0] - margin - ray->origin.x) * inv.x, tx1 = (n->hi[0] + margin - ray->origin.x) * inv.x;
    float ty0 = (n->lo[1] - margin - ray->origin.y) * inv.y, ty1 = (n->hi[1] + margin - ray->origin.y) * inv.y;
    float tz0 = (n->lo[2] - margin - ray->origin.z) * inv.z, tz1 = (n->hi[2] + margin - ray->origin.z) * inv.z;
    float tmin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.0f));
    float tmax = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    return tmin <= tmax * 1.000002f + 1e-30f && tmin <= best;
}

static void trace_bvh(const PtoPrepared *p, const Ray *ray, RayHit *best, Counters *cn)
{
    if (p->n_nodes == 0) return;
    /* |d| clamped away from zero so that no slab distance is 0 * inf */
    f3 dd = ray->direction;
    if (fabsf(dd.x) < 1e-18f) dd.x = copysignf(1e-18f, dd.x);
    if (fabsf(dd.y) < 1e-18f) dd.y = copysignf(1e-18f, dd.y);
    if (fabsf(dd.z) < 1e-18f) dd.z = copysignf(1e-18f, dd.z);
    f3 inv = pt_make3(1.0f / dd.x, 1.0f / dd.y, 1.0f / dd.z);
    /* the ray's distance from the farthest corner of the scene: 6.4e-6 of it is sixteen times a well-shaped
     * triangle's reach at that distance; the boxes' own padding covers the near field */
    const ONode *root = &p->nodes[0];
    const float fx = fmaxf(fabsf(ray->origin.x - root->lo[0]), fabsf(ray->origin.x - root->hi[0]));
    const float fy = fmaxf(fabsf(ray->origin.y - root->lo[1]), fabsf(ray->origin.y - root->hi[1]));
    const float fz = fmaxf(fabsf(ray->origin.z - root->lo[2]), fabsf(ray->origin.z - root->hi[2]));
    const float far = sqrtf(fx * fx + fy * fy + fz * fz);
    float margin = 6.4e-6f * far;
    if (!(margin < 1e30f)) margin = 1e30f;        /* NaN or infinite origin: every box is entered */
    int32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const ONode *n = &p->nodes[stack[--sp]];
        if (!box_hit(n, ray, inv, best->distance, margin, far)) continue;
        if (n->count) {
            for (int32_t k = 0; k < n->count; k++) {
                int32_t id = p->order[n->first + k];
                cn->prims++;
                intersect_tri(ray, best, &p->tris[id - 1 - p->n_spheres], id, 1, p->materials);
            }
        } else {
            stack[sp++] = n->left; stack[sp++] = n->right;
        }
    }
}

/* SH:262-285 */
static RayHit trace(const PtoPrepared *p, const Ray *ray, Counters *cn)
{
    RayHit best = create_ray_hit();
    cn->trace_calls++;
    intersect_ground_plane(ray, &best);
    if (p->accel) {
        for (int32_t i = 0; i < p->n_spheres; i++)
            intersect_sphere(ray, &best, &p->spheres[i], 1 + i, 1);
        cn->prims += (uint64_t)p->n_spheres;
        trace_bvh(p, ray, &best, cn);
        return best;
    }
    for (int32_t i = 0; i < p->n_spheres; i++)
        intersect_sphere(ray, &best, &p->spheres[i], 1 + i, 0);
    /* mesh objects in order, triangles in order: p->tris is that flattened order */
    for (int32_t k = 0; k < p->n_tris; k++)
        intersect_tri(ray, &best, &p->tris[k], 1 + p->n_spheres + k, 0, p->materials);
    cn->prims += (uint64_t)p->n_spheres + (uint64_t)p->n_tris;
    return best;
}

/* Trace() plus an optional log of (origin, direction, id, distance) per call, for tests. */
static RayHit trace_logged(const PtoPrepared *p, const Ray *ray, Counters *cn)
{
    RayHit h = trace(p, ray, cn);
    if (cn->log && cn->log_n < cn->log_cap) {
        float *e = cn->log + 8 * cn->log_n++;
        e[0] = ray->origin.x; e[1] = ray->origin.y; e[2] = ray->origin.z;
        e[3] = ray->direction.x; e[4] = ray->direction.y; e[5] = ray->direction.z;
        e[6] = (float)h.id; e[7] = h.distance;
    }
    return h;
}

/* ---------------------------------------------------------------- sampling */

/* SH:291-302; rows tangent, binormal, normal */
static void get_tangent_space(f3 normal, f3 *tangent, f3 *binormal)
{
    f3 helper = pt_make3(1, 0, 0);
    if (pt_abs(normal.x) > 0.99f) helper = pt_make3(0, 0, 1);
    *tangent = pt_normalize3(pt_cross3(normal, helper));
    *binormal = pt_normalize3(pt_cross3(normal, *tangent));
}

/* SH:304-314 */
static f3 tangent_to_world(f3 normal, f3 ts);

static f3 sample_hemisphere(RngState *rs, f3 normal, float alpha)
{
    float cosTheta = pt_pow(rng(rs), 1.0f / (alpha + 1.0f));
    float sinTheta = pt_sqrt(1.0f - cosTheta * cosTheta);
    float phi = 2 * PT_PI * rng(rs);
    f3 ts = pt_make3(pt_cos(phi) * sinTheta, pt_sin(phi) * sinTheta, cosTheta);
    return tangent_to_world(normal, ts);
}

static f3 tangent_to_world(f3 normal, f3 ts)   /* SH:310-313 */
{
    f3 tangent, binormal;
    get_tangent_space(normal, &tangent, &binormal);
    /* mul(tangentSpaceDir, float3x3(tangent, binormal, normal)): row vector times matrix */
    f3 r;
    r.x = pt_fma(ts.z, normal.x, pt_fma(ts.y, binormal.x, ts.x * tangent.x));
    r.y = pt_fma(ts.z, normal.y, pt_fma(ts.y, binormal.y, ts.x * tangent.y));
    r.z = pt_fma(ts.z, normal.z, pt_fma(ts.y, binormal.z, ts.x * tangent.z));
    return r;
}

static float smoothness_to_phong_alpha(float s) { return pt_pow(1000.0f, s * s); } /* SH:319-322 */

/* PT_LOBE_GGX (extension, no counterpart in the shader): sample the GGX normal distribution of roughness
 * a = max((1-smoothness)^2, 1e-3) - cos(theta_h) first, phi second, like SampleHemisphere -, mirror the ray
 * about the half-vector, return f*cos/pdf = F*G*(v.h)/((n.v)(n.h)) per channel (Schlick F with F0 = specular,
 * Smith separable G) or zero when the sample leaves the upper hemisphere. */
static f3 sample_ggx(RngState *rs, f3 normal, f3 specular, float smoothness, f3 *dir)
{
    float rough = 1.0f - smoothness;
    float a = pt_max(rough * rough, 1e-3f), a2 = a * a;
    float u1 = rng(rs);
    float cos2 = (1.0f - u1) / (1.0f + (a2 - 1.0f) * u1);
    float cosT = pt_sqrt(cos2), sinT = pt_sqrt(1.0f - cos2);
    float phi = 2 * PT_PI * rng(rs);
    f3 h = tangent_to_world(normal, pt_make3(pt_cos(phi) * sinT, pt_sin(phi) * sinT, cosT));
    f3 v = pt_make3(-dir->x, -dir->y, -dir->z);
    float vh = pt_dot3(v, h);
    *dir = pt_reflect3(*dir, h);
    float nl = pt_dot3(normal, *dir), nv = pt_dot3(normal, v), nh = pt_dot3(normal, h);
    if (!(nl > 0.0f && nv > 0.0f && vh > 0.0f && nh > 0.0f)) return pt_make3(0, 0, 0);
    float m = 1.0f - vh, m2 = m * m, m5 = m2 * m2 * m;
    float g1v = 2.0f * nv / (nv + pt_sqrt(a2 + (1.0f - a2) * nv * nv));
    float g1l = 2.0f * nl / (nl + pt_sqrt(a2 + (1.0f - a2) * nl * nl));
    float w = g1v * g1l * vh / (nv * nh);
    return pt_make3((specular.x + (1.0f - specular.x) * m5) * w, (specular.y + (1.0f - specular.y) * m5) * w,
                    (specular.z + (1.0f - specular.z) * m5) * w);
}

/* _SkyboxTexture.SampleLevel(sampler, uv, 0): bilinear, Repeat wrap on both
 * axes, texel centres at (i+0.5)/N, weights in fp32; lerp(a,b,t) = a + t*(b-a). */
static int wrapi(int i, int n) { i %= n; return i < 0 ? i + n : i; }
static f3 sky_fetch(const PtoPrepared *p, float u, float v)
{
    if (!p->sky) return pt_make3(0, 0, 0);
    int W = p->sky_w, H = p->sky_h;
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float x0f = pt_floor(x), y0f = pt_floor(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapi((int)x0f, W), y0 = wrapi((int)y0f, H);
    int x1 = x0 + 1 == W ? 0 : x0 + 1, y1 = y0 + 1 == H ? 0 : y0 + 1;
    const float *c00 = p->sky + 4 * ((size_t)y0 * W + x0), *c10 = p->sky + 4 * ((size_t)y0 * W + x1);
    const float *c01 = p->sky + 4 * ((size_t)y1 * W + x0), *c11 = p->sky + 4 * ((size_t)y1 * W + x1);
    float out[3];
    for (int k = 0; k < 3; k++) {
        float a = c00[k] + fx * (c10[k] - c00[k]);
        float b = c01[k] + fx * (c11[k] - c01[k]);
        out[k] = a + fy * (b - a);
    }
    return pt_make3(out[0], out[1], out[2]);
}

/* SH:324-332 */
static f3 skybox_sample(const PtoPrepared *p, Ray *ray)
{
    ray->energy = pt_make3(0, 0, 0);
    float theta = pt_acos(ray->direction.y) / -PT_PI;
    float phi = pt_atan2(ray->direction.x, -ray->direction.z) / -PT_PI * 0.5f;
    return pt_scale3(sky_fetch(p, phi, theta), 2.0f);
}

/* SH:334-373 */
static f3 shade(const PtoPrepared *p, RngState *rs, Ray *ray, RayHit hit)
{
    if (hit.distance < PT_INF) {
        hit.albedo = pt_make3(pt_min(1.0f - hit.specular.x, hit.albedo.x),
                              pt_min(1.0f - hit.specular.y, hit.albedo.y),
                              pt_min(1.0f - hit.specular.z, hit.albedo.z));
        float specChance = energy(hit.specular);
        float diffChance = energy(hit.albedo);
        float roulette = rng(rs);
        if (roulette < specChance && hit.lobe == PT_LOBE_GGX) {
            ray->origin = pt_along3(hit.position, 0.001f, hit.normal);
            float k = (1.0f / specChance);
            f3 w = sample_ggx(rs, hit.normal, hit.specular, hit.smoothness, &ray->direction);
            ray->energy.x *= k * w.x;
            ray->energy.y *= k * w.y;
            ray->energy.z *= k * w.z;
        } else if (roulette < specChance) {
            ray->origin = pt_along3(hit.position, 0.001f, hit.normal); /* position + normal*0.001 */
            float alpha = smoothness_to_phong_alpha(hit.smoothness);
            ray->direction = sample_hemisphere(rs, pt_reflect3(ray->direction, hit.normal), alpha);
            float f = (alpha + 2) / (alpha + 1);
            float k = (1.0f / specChance);
            float sd = sdot(hit.normal, ray->direction, f);
            /* ray.energy *= (1/specChance) * hit.specular * sdot(...), left to right */
            ray->energy.x *= (k * hit.specular.x) * sd;
            ray->energy.y *= (k * hit.specular.y) * sd;
            ray->energy.z *= (k * hit.specular.z) * sd;
        } else if (diffChance > 0 && roulette < specChance + diffChance) {
            ray->origin = pt_along3(hit.position, 0.001f, hit.normal);
            ray->direction = sample_hemisphere(rs, hit.normal, 1.0f);
            float k = (1.0f / diffChance);
            ray->energy.x *= k * hit.albedo.x;
            ray->energy.y *= k * hit.albedo.y;
            ray->energy.z *= k * hit.albedo.z;
        } else {
            ray->energy = pt_make3(0, 0, 0);
        }
        return hit.emission;
    }
    return skybox_sample(p, ray);
}

/* ------------------------------------------------------------------ kernel */

static int any3(f3 v) { return v.x != 0.0f || v.y != 0.0f || v.z != 0.0f; }

/* ---- renderMode 2: SH:394-411, SH:440-463, SH:950-983 ----
 * HLSL semantics fixed here because the shader leaves them to its compiler and API:
 * mul(float4x4, float3) uses the upper-left 3x3 block (SH:404); pwidth / pheight is an
 * unsigned division (SH:402); a float2 texture index is truncated like a uint cast, negative
 * and NaN to 0, and a load outside the texture gives 0 (SH:450-453, SH:968); float s =
 * 1.0/textureSize keeps .x (SH:445); a*b+c is not fused. */
typedef struct History {
    const PtCameraHistory *cam;   /* the element of the four *Old arrays to use (entry 0 unless stated) */
    const float *result1;         /* Result1: W*H*4, the previous frame's Result */
    const PtCameraHistory *all;   /* the whole arrays, PT_MAX_CAMERA_HISTORY elements (renderMode 5/6) */
} History;

static void find_camera_ray(const History *hs, int W, int H, f3 direction, float *uv_out) /* SH:394-411, framesFrom 0 */
{
    const float *P = hs->cam->camera_projection;
    float m11 = P[1 * 4 + 1], m22 = P[2 * 4 + 2], m23 = P[3 * 4 + 2];   /* _mRC at [C*4+R] */
    float nearp = -m23 / m22;                                            /* SH:399 */
    float farp = -(nearp * m22) / (1.0f - m22);                          /* SH:400 */
    f3 pos = pt_scale3(pt_normalize3(direction), nearp + (farp - nearp)); /* SH:401 */
    float ratio = (float)((uint32_t)W / (uint32_t)H);                    /* SH:402 */
    f3 toCam = pt_mul_dir(hs->cam->world_to_camera, pos);                /* SH:404 */
    float camPosZ = toCam.z;
    float height = camPosZ / m11;                                        /* SH:406 */
    float width = ratio * height;                                        /* SH:407 */
    float dx = (float)W, dy = (float)H;
    float uvx = (toCam.x - width * (dx / dy)) / width;                   /* SH:408 */
    float uvy = (toCam.y - height) / height;                             /* SH:409 */
    uv_out[0] = (0.5f - uvx * ((float)H * ratio / 2.0f)) - 0.00f;        /* SH:410 */
    uv_out[1] = (0.5f - uvy * ((float)H / 2.0f)) - 0.005f;
}

static uint32_t to_uint(float v)
{
    if (!(v > 0.0f)) return 0u;               /* negative, zero, NaN */
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}

static void load_result1(const History *hs, int W, int H, float x, float y, float *out4)
{
    uint32_t ix = to_uint(x), iy = to_uint(y);
    if (ix >= (uint32_t)W || iy >= (uint32_t)H) { out4[0] = out4[1] = out4[2] = out4[3] = 0.0f; return; }
    const float *q = hs->result1 + 4 * ((size_t)iy * W + ix);
    for (int k = 0; k < 4; k++) out4[k] = q[k];
}

static void bilinear_texture_sample(const History *hs, int W, int H, float Px, float Py, float *out4) /* SH:440-463 */
{
    float tsx = (float)W, tsy = (float)H;
    float s = 1.0f / tsx;                                     /* SH:445 */
    Px -= 0.5f; Py -= 0.5f;                                   /* SH:446 */
    float pixx = Px * tsx + 0.5f, pixy = Py * tsy + 0.5f;     /* SH:447 */
    float fx = pt_frac(Px), fy = pt_frac(Py);                 /* SH:448 */
    pixx = pt_floor(pixx) / tsx - s / 2.0f;                   /* SH:449 */
    pixy = pt_floor(pixy) / tsy - s / 2.0f;
    float C11[4], C21[4], C12[4], C22[4];
    load_result1(hs, W, H, pixx + 0.0f, pixy + 0.0f, C11);    /* SH:450-453 */
    load_result1(hs, W, H, pixx + 1.0f, pixy + 0.0f, C21);
    load_result1(hs, W, H, pixx + 0.0f, pixy + 1.0f, C12);
    load_result1(hs, W, H, pixx + 1.0f, pixy + 1.0f, C22);
    for (int k = 0; k < 4; k++) {
        float x1 = C11[k] + fx * (C21[k] - C11[k]);           /* SH:458-461: lerp(a,b,t) = a + t*(b-a) */
        float x2 = C12[k] + fx * (C22[k] - C12[k]);
        out4[k] = x1 + fy * (x2 - x1);
    }
}

/* ---- renderMode 0: SH:466-615 (denoiser), SH:617-714 (TAA) ----
 * Texture2D.Load / RWTexture2D.Load take int2: a float2 argument is truncated toward zero;
 * operator[] takes uint2 (to_uint above).  Either way a location outside the texture reads 0. */
static void load_int(const float *tex, int W, int H, float x, float y, float *out4)
{
    int ix = (x >= 2147483648.0f || x <= -2147483648.0f || x != x) ? -1 : (int)x;
    int iy = (y >= 2147483648.0f || y <= -2147483648.0f || y != y) ? -1 : (int)y;
    if (ix < 0 || iy < 0 || ix >= W || iy >= H) { out4[0] = out4[1] = out4[2] = out4[3] = 0.0f; return; }
    const float *q = tex + 4 * ((size_t)iy * W + ix);
    for (int k = 0; k < 4; k++) out4[k] = q[k];
}

static float hash1(float seed) { return pt_frac(pt_sin(seed) * 43758.5453123f); }   /* SH:475-477 */

static const float k_atrous[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};   /* SH:513-542 = outer product */

/* denoiser(fragCoord), SH:479-615, on Result1; `seed` = _Seed at the time of the call. */
static void denoiser(const float *result1, int W, int H, float fx, float fy, float seed, float *out4)
{
    const float c_phi = 1.0f, n_phi = 0.5f, p_phi = 0.25f;
    float cv[4];
    load_int(result1, W, H, fx, fy, cv);                                        /* SH:552 */
    f3 cval = pt_make3(cv[0], cv[1], cv[2]);
    float strength = 1.0f + (4.0f - 1.0f) * hash1((641.128752f * fx + 312.321374f * fy) + 1.92357812f * seed);  /* SH:564 */
    f3 sum = pt_make3(0, 0, 0), sum_f = pt_make3(0, 0, 0);
    float cum_w = 0.0f, cum_fw = 0.0f;
    for (int i = 0; i < 25; i++) {
        float ox = (float)(i % 5 - 2), oy = (float)(i / 5 - 2);                 /* SH:481-510 */
        float kern = k_atrous[i % 5] * k_atrous[i / 5];
        float tv[4];
        load_int(result1, W, H, fx + ox * strength, fy + oy * strength, tv);    /* SH:572-575 */
        f3 ctmp = pt_make3(tv[0], tv[1], tv[2]);
        f3 t = pt_sub3(cval, ctmp);
        float dist2 = pt_dot3(t, t);
        float c_w = pt_min(pt_exp(-(dist2) / c_phi), 1.0f);                     /* SH:578 */
        float n_w = pt_min(pt_exp(-(0.0f) / n_phi), 1.0f);                      /* SH:592 */
        float r_w = c_w;                                                        /* SH:593 */
        float weight0 = c_w * n_w;
        sum.x += ctmp.x * weight0 * kern; sum.y += ctmp.y * weight0 * kern; sum.z += ctmp.z * weight0 * kern;   /* SH:597 */
        cum_w += weight0 * kern;
        float weight1 = r_w * n_w;
        sum_f.x += ctmp.x * weight1 * kern; sum_f.y += ctmp.y * weight1 * kern; sum_f.z += ctmp.z * weight1 * kern;
        cum_fw += weight1 * kern;
    }
    f3 a = pt_make3(sum.x / cum_w, sum.y / cum_w, sum.z / cum_w);
    f3 b = pt_make3(sum_f.x / cum_fw, sum_f.y / cum_fw, sum_f.z / cum_fw);
    float dist2 = pt_dot3(a, a);                                                /* SH:608-611 with ptmp = 0 */
    float p_w = pt_min(pt_exp(-(dist2) / p_phi), 1.0f);
    out4[0] = pt_min(pt_max(a.x + p_w * (b.x - a.x), 0.0f), 1.0f);              /* SH:614 */
    out4[1] = pt_min(pt_max(a.y + p_w * (b.y - a.y), 0.0f), 1.0f);
    out4[2] = pt_min(pt_max(a.z + p_w * (b.z - a.z), 0.0f), 1.0f);
    out4[3] = 0.0f;
}

/* ---- Assets/Shaders/Denoiser.shader (the display blit, RayTracingMaster.cs:760-764) ----
 * tex2D is taken as: bilinear, float weights, clamp addressing, texel centres at (i + 0.5) / size. */
static void tex2d(const float *tex, int W, int H, float u, float v, float *out4)
{
    float px = u * (float)W - 0.5f, py = v * (float)H - 0.5f;
    float x0f = pt_floor(px), y0f = pt_floor(py);
    float fx = px - x0f, fy = py - y0f;
    int x0 = (int)pt_min(pt_max(x0f, 0.0f), (float)(W - 1)), x1 = (int)pt_min(pt_max(x0f + 1.0f, 0.0f), (float)(W - 1));
    int y0 = (int)pt_min(pt_max(y0f, 0.0f), (float)(H - 1)), y1 = (int)pt_min(pt_max(y0f + 1.0f, 0.0f), (float)(H - 1));
    const float *c00 = tex + 4 * ((size_t)y0 * W + x0), *c10 = tex + 4 * ((size_t)y0 * W + x1);
    const float *c01 = tex + 4 * ((size_t)y1 * W + x0), *c11 = tex + 4 * ((size_t)y1 * W + x1);
    for (int k = 0; k < 4; k++) {
        float a = c00[k] + fx * (c10[k] - c00[k]);
        float b = c01[k] + fx * (c11[k] - c01[k]);
        out4[k] = a + fy * (b - a);
    }
}

/* Denoiser.shader:50-184 for the fragment at pixel (x, y); `src` is _MainTex (= Result). */
static void display_denoiser(const float *src, int W, int H, int x, int y, float *out4)
{
    const float c_phi = 1.0f, r_phi = 1.0f, n_phi = 0.5f, p_phi = 0.25f;
    float fw = (float)W, fh = (float)H;
    float u = ((float)x + 0.5f) / fw, v = ((float)y + 0.5f) / fh;                     /* i.uv at the pixel centre */
    float cv[4];
    tex2d(src, W, H, u, v, cv);                                                       /* :124-126 */
    f3 cval = pt_make3(cv[0], cv[1], cv[2]), rval = cval, nval = cval;
    float strength = (1.0f + (4.0f - 1.0f) * hash1((641.128752f * u + 312.321374f * v) + 1.92357812f)) * 1.0f;   /* :136 */
    f3 sum = pt_make3(0, 0, 0), sum_f = pt_make3(0, 0, 0);
    float cum_w = 0.0f, cum_fw = 0.0f;
    for (int i = 0; i < 25; i++) {
        float ox = (float)(i % 5 - 2), oy = (float)(i / 5 - 2);
        float kern = k_atrous[i % 5] * k_atrous[i / 5];
        float tv[4];
        tex2d(src, W, H, u + ox * strength / fw, v + oy * strength / fh, tv);         /* :144-147 */
        f3 ctmp = pt_make3(tv[0], tv[1], tv[2]), ntmp = ctmp, rtmp = ctmp;            /* :147,153,159: the same texture */
        f3 t = pt_sub3(cval, ctmp);
        float dist2 = pt_dot3(t, t);
        float c_w = pt_min(pt_exp(-(dist2) / c_phi), 1.0f);                           /* :150 */
        t = pt_sub3(nval, ntmp);
        dist2 = pt_max(pt_dot3(t, t), 0.0f);
        float n_w = pt_min(pt_exp(-(dist2) / n_phi), 1.0f);                           /* :156 */
        t = pt_sub3(rval, rtmp);
        dist2 = pt_dot3(t, t);
        float r_w = pt_min(pt_exp(-(dist2) / r_phi), 1.0f);                           /* :162 */
        float weight0 = c_w * n_w;
        sum.x += ctmp.x * weight0 * kern; sum.y += ctmp.y * weight0 * kern; sum.z += ctmp.z * weight0 * kern;
        cum_w += weight0 * kern;
        float weight1 = r_w * n_w;
        sum_f.x += rtmp.x * weight1 * kern; sum_f.y += rtmp.y * weight1 * kern; sum_f.z += rtmp.z * weight1 * kern;
        cum_fw += weight1 * kern;
    }
    f3 a = pt_make3(sum.x / cum_w, sum.y / cum_w, sum.z / cum_w);
    f3 b = pt_make3(sum_f.x / cum_fw, sum_f.y / cum_fw, sum_f.z / cum_fw);
    float p_w = pt_min(pt_exp(-(pt_dot3(a, a)) / p_phi), 1.0f);                       /* :177-181 */
    out4[0] = pt_min(pt_max(a.x + p_w * (b.x - a.x), 0.0f), 1.0f);                    /* :183 */
    out4[1] = pt_min(pt_max(a.y + p_w * (b.y - a.y), 0.0f), 1.0f);
    out4[2] = pt_min(pt_max(a.z + p_w * (b.z - a.z), 0.0f), 1.0f);
    out4[3] = 0.0f;
}

static f3 encode_pal_yuv(f3 rgb)                                                 /* SH:624-632 */
{
    rgb = pt_make3(pt_pow_2(rgb.x), pt_pow_2(rgb.y), pt_pow_2(rgb.z));
    return pt_make3(pt_dot3(rgb, pt_make3(0.299f, 0.587f, 0.114f)),
                    pt_dot3(rgb, pt_make3(-0.14713f, -0.28886f, 0.436f)),
                    pt_dot3(rgb, pt_make3(0.615f, -0.51499f, -0.10001f)));
}

static f3 decode_pal_yuv(f3 yuv)                                                 /* SH:634-642 */
{
    f3 rgb = pt_make3(pt_dot3(yuv, pt_make3(1.0f, 0.0f, 1.13983f)),
                      pt_dot3(yuv, pt_make3(1.0f, -0.39465f, -0.58060f)),
                      pt_dot3(yuv, pt_make3(1.0f, 2.03211f, 0.0f)));
    return pt_make3(pt_pow_half(rgb.x), pt_pow_half(rgb.y), pt_pow_half(rgb.z));
}

static f3 min3v(f3 a, f3 b) { return pt_make3(pt_min(a.x, b.x), pt_min(a.y, b.y), pt_min(a.z, b.z)); }
static f3 max3v(f3 a, f3 b) { return pt_make3(pt_max(a.x, b.x), pt_max(a.y, b.y), pt_max(a.z, b.z)); }
static f3 lerp3(f3 a, f3 b, float t) { return pt_make3(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)); }

static f3 load_xyz_uint(const float *tex, int W, int H, float x, float y)        /* tex[float2]: operator[] */
{
    History hs = {NULL, tex, NULL};
    float v[4];
    load_result1(&hs, W, H, x, y, v);
    return pt_make3(v[0], v[1], v[2]);
}

/* TAA(fragCoord), SH:644-714: history from Result1, current frame and its neighbours from `cur`
 * (the denoised image of every pixel - the shader reads Result while other threads are still
 * writing it; here all of them have finished the denoiser and none has written the TAA output). */
static void taa(const float *result1, const float *cur, int W, int H, float fx, float fy, float *out4)
{
    History hs = {NULL, result1, NULL};
    float last[4];
    bilinear_texture_sample(&hs, W, H, fx, fy, last);                           /* SH:651 */
    f3 antialiased = pt_make3(last[0], last[1], last[2]);
    float mixRate = 0.5f;
    float offx = 1.0f / (float)W, offy = 1.0f / (float)H;                        /* SH:657 */
    f3 in0 = load_xyz_uint(cur, W, H, fx, fy);                                   /* SH:659 */
    antialiased = lerp3(pt_make3(antialiased.x * antialiased.x, antialiased.y * antialiased.y, antialiased.z * antialiased.z),
                        pt_make3(in0.x * in0.x, in0.y * in0.y, in0.z * in0.z), mixRate);   /* SH:661 */
    antialiased = pt_make3(pt_sqrt(antialiased.x), pt_sqrt(antialiased.y), pt_sqrt(antialiased.z));
    f3 in1 = load_xyz_uint(cur, W, H, fx + offx, fy + 0.0f);                     /* SH:673-680 */
    f3 in2 = load_xyz_uint(cur, W, H, fx + -offx, fy + 0.0f);
    f3 in3 = load_xyz_uint(cur, W, H, fx + 0.0f, fy + offy);
    f3 in4 = load_xyz_uint(cur, W, H, fx + 0.0f, fy + -offy);
    f3 in5 = load_xyz_uint(cur, W, H, fx + offx, fy + offy);
    f3 in6 = load_xyz_uint(cur, W, H, fx + -offx, fy + offy);
    f3 in7 = load_xyz_uint(cur, W, H, fx + offx, fy + -offy);
    f3 in8 = load_xyz_uint(cur, W, H, fx + -offx, fy + -offy);
    antialiased = encode_pal_yuv(antialiased);
    in0 = encode_pal_yuv(in0); in1 = encode_pal_yuv(in1); in2 = encode_pal_yuv(in2); in3 = encode_pal_yuv(in3);
    in4 = encode_pal_yuv(in4); in5 = encode_pal_yuv(in5); in6 = encode_pal_yuv(in6); in7 = encode_pal_yuv(in7);
    in8 = encode_pal_yuv(in8);
    f3 minColor = min3v(min3v(min3v(in0, in1), min3v(in2, in3)), in4);           /* SH:693-698 */
    f3 maxColor = max3v(max3v(max3v(in0, in1), max3v(in2, in3)), in4);
    minColor = lerp3(minColor, min3v(min3v(min3v(in5, in6), min3v(in7, in8)), minColor), 0.5f);
    maxColor = lerp3(maxColor, max3v(max3v(max3v(in5, in6), max3v(in7, in8)), maxColor), 0.5f);
    f3 preclamping = antialiased;
    antialiased = min3v(max3v(antialiased, minColor), maxColor);                 /* clamp(x, lo, hi) = min(max(x, lo), hi) */
    mixRate = 1.0f / (1.0f / mixRate + 1.0f);                                    /* SH:703 */
    f3 diff = pt_sub3(antialiased, preclamping);
    float clampAmount = pt_dot3(diff, diff);
    mixRate += clampAmount * 4.0f;
    mixRate = pt_min(pt_max(mixRate, 0.05f), 0.5f);
    antialiased = decode_pal_yuv(antialiased);
    out4[0] = antialiased.x; out4[1] = antialiased.y; out4[2] = antialiased.z; out4[3] = mixRate;   /* SH:713 */
}

/* SH:374-387 CreateCameraRayOld(uv, framesBack): as CreateCameraRay but against element framesBack of the history
 * arrays and with clip-space z = 1 (SH:381) instead of 0 (SH:115).  mul(M, float4(u, v, 1, 1)) is taken in the order
 * ((M0*u + M1*v) + M2) + M3. */
static Ray create_camera_ray_old(const float *c2w, const float *ip, float u, float v)
{
    f3 origin = pt_make3(c2w[12], c2w[13], c2w[14]);
    f3 d;
    d.x = (pt_fma(ip[4], v, ip[0] * u) + ip[8]) + ip[12];
    d.y = (pt_fma(ip[5], v, ip[1] * u) + ip[9]) + ip[13];
    d.z = (pt_fma(ip[6], v, ip[2] * u) + ip[10]) + ip[14];
    d = pt_mul_dir(c2w, d);
    d = pt_normalize3(d);
    return create_ray(origin, d);
}

/* SH:418-437 FindCameraRayVR(input): the headset variant of FindCameraRay with the eye texture's aspect ratio and
 * per-eye pixel offsets hard-coded; history element 0. */
static void find_camera_ray_vr(const History *hs, int H, f3 direction, int left_eye, float *uv_out)
{
    const float *P = hs->cam->camera_projection;
    float m11 = P[1 * 4 + 1], m22 = P[2 * 4 + 2], m23 = P[3 * 4 + 2];
    float nearp = -m23 / m22;                                            /* SH:425 */
    float farp = -(nearp * m22) / (1.0f - m22);                          /* SH:426 */
    f3 pos = pt_scale3(pt_normalize3(direction), nearp + (farp - nearp)); /* SH:427 */
    float ratio = 2240.0f / 2016.0f;                                     /* SH:428 */
    f3 toCam = pt_mul_dir(hs->cam->world_to_camera, pos);                /* SH:430 */
    float camPosZ = toCam.z;
    float height = camPosZ / m11;                                        /* SH:432 */
    float width = ratio * height;                                        /* SH:433 */
    float uvx = (toCam.x - width * (2240.0f / 2016.0f)) / width;         /* SH:434 */
    float uvy = (toCam.y - height) / height;                             /* SH:435 */
    float ph = (float)(uint32_t)H;
    uv_out[0] = (1.0f - uvx * (ph * ratio / 2.043f)) - (left_eye ? 156.0f : 538.0f);   /* SH:436 */
    uv_out[1] = (1.0f - uvy * (ph / 2.0f)) - (left_eye ? -1.3f : -0.6f);
}

/* CSMain for one thread id (SH:725-1112), every renderMode of RayTracingMaster.cs:977-990.
 * The prologue copies Result1 = Result / Result2 (SH:728-736) are done for the whole rectangle by the caller
 * before any pixel runs, and so are the copies that end a dispatch (SH:944-947, SH:1108-1112): the return value
 * is 1 for a thread that reaches the end of the sample loop, which is the thread those copies belong to.
 * out4 = Result[id], out2 = Result2[id] (renderMode 7 / 8; NULL otherwise); `hs` carries the history arrays and
 * Result1 as it was before the dispatch (NULL in renderMode 1). */
static int cs_main(const PtoPrepared *p, const PtFrameParams *prm, int W, int H, int idx, int idy,
                   float *out4, float *out2, int32_t *hit_id, Counters *cn, const History *hs, float *seed_out)
{
    const int mode = prm->render_mode;
    float fw = (float)W, fh = (float)H;
    float u = (float)idx / fw * 2.0f - 1.0f;   /* SH:802, SH:842: pixel corner, no jitter */
    float v = (float)idy / fh * 2.0f - 1.0f;
    float fy = (float)idy;
    float stripe = 0.0f;
    int outside = 0, inside_or_edge = 1;
    if (mode >= 5 && mode <= 8) {
        stripe = fh / (float)prm->divisions;                                         /* SH:742 */
        outside = fy > fh - stripe * (float)prm->counter || fy < fh - stripe * (float)(prm->counter + 1);        /* SH:751, SH:786 */
        inside_or_edge = fy <= fh - stripe * (float)prm->counter || fy >= fh - stripe * (float)(prm->counter + 1); /* SH:853, SH:939 */
    }
    if (((mode == 5 || mode == 6) && outside) || mode == 7 || mode == 8) {           /* SH:751-789 */
        Ray rayr = create_camera_ray(prm->camera_to_world, prm->camera_inverse_projection, u, v);   /* SH:753-754 */
        f3 hitPosr = pt_along3(rayr.origin, 10.0f, rayr.direction);                  /* SH:755 */
        if (mode % 2 == 0) {                                                         /* SH:757-764 */
            RayHit hitr = trace(p, &rayr, cn);
            cn->rays++; cn->paths++;
            hitPosr = hitr.position;
            if (!(hitr.distance < PT_INF)) {
                f3 sky = skybox_sample(p, &rayr);
                out4[0] = sky.x; out4[1] = sky.y; out4[2] = sky.z; out4[3] = 1.0f;
                return 0;
            }
        }
        int framesBack = 0;
        if (mode == 5 || mode == 6) {
            /* SH:766: (_Divisions - (id.y / stripe)) - 1 % _Divisions, a float, truncated into the int */
            framesBack = (int)(((float)prm->divisions - fy / stripe) - (float)(1 % prm->divisions));
            if (framesBack < 0) framesBack = 0;                                      /* keep the index inside the arrays */
            if (framesBack > PT_MAX_CAMERA_HISTORY - 1) framesBack = PT_MAX_CAMERA_HISTORY - 1;
        }
        History h = {&hs->all[framesBack], hs->result1, hs->all};
        f3 originOld = pt_make3(h.cam->camera_to_world[12], h.cam->camera_to_world[13], h.cam->camera_to_world[14]);   /* SH:769 */
        float reproUV[2];
        find_camera_ray(&h, W, H, pt_sub3(hitPosr, originOld), reproUV);             /* SH:770 */
        if (reproUV[0] < fw && reproUV[0] > 0 && reproUV[1] < fh && reproUV[1] > 0)
            bilinear_texture_sample(&h, W, H, reproUV[0], reproUV[1], out4);         /* SH:776 */
        else { out4[0] = out4[1] = out4[2] = 0.0f; out4[3] = 1.0f; }                 /* SH:780 */
        if (mode <= 6) return 0;                                                     /* SH:783-784 */
        if (prm->counter % prm->divisions != 0 && outside) return 0;                 /* SH:786-788 */
    }
    if (prm->depth == 1) {                     /* SH:799-809 */
        Ray ray = create_camera_ray(prm->camera_to_world, prm->camera_inverse_projection, u, v);
        RayHit hit = trace(p, &ray, cn);
        cn->rays++; cn->paths++;
        float dis = hit.distance / 16.0f;
        out4[0] = out4[1] = out4[2] = out4[3] = dis;
        if (hit_id) *hit_id = hit.id;
        return 0;
    }
    f3 result = pt_make3(0, 0, 0);
    RngState rs; rs.seed = prm->seed; rs.px = (float)idx * 2.0f; rs.py = (float)idy * 2.0f; /* SH:831 */
    int spp = prm->samples_per_pixel;
    float fspp = (float)spp;
    int maxb = prm->max_bounces;
    f3 camo = pt_make3(prm->camera_to_world[12], prm->camera_to_world[13], prm->camera_to_world[14]);
    /* _SamplesPerPixel <= 0: the loop below does not run and the thread writes nothing */
    for (int j = 1; j < spp + 1; j++) {            /* SH:839 */
        Ray ray = create_camera_ray(prm->camera_to_world, prm->camera_inverse_projection, u, v);   /* SH:848 */
        if (mode == 7 || mode == 8) {              /* SH:850-852 */
            /* (_Counter - 1) % _Divisions is -1 when _Counter is 0.  Element -1 of a uniform array is the uniform
             * declared just before it - the constant buffer is contiguous and SH:8-11 declare _CameraToWorld before
             * _CameraToWorldOld[100] and _CameraInverseProjection before _CameraInverseProjectionOld[100] - so that
             * dispatch renders from the current camera (through CreateCameraRayOld's z = 1). */
            int fb = (prm->counter - 1) % prm->divisions;
            if (fb < 0) ray = create_camera_ray_old(prm->camera_to_world, prm->camera_inverse_projection, u, v);
            else ray = create_camera_ray_old(hs->all[fb].camera_to_world, hs->all[fb].camera_inverse_projection, u, v);
        }
        int fullFrameReprojRender = (mode != 7 && mode != 8) || inside_or_edge;      /* SH:853 */
        cn->paths++;
        f3 hitPos = pt_along3(ray.origin, 10.0f, ray.direction);                     /* SH:855 */
        int skyboxS = 0;                           /* SH:856 */
        RayHit hit = create_ray_hit();             /* SH:857 leaves it unset; it is not read before SH:859 sets it */
        if (mode != 11 && fullFrameReprojRender) { /* SH:858-863 */
            hit = trace_logged(p, &ray, cn);
            cn->rays++;
            hitPos = hit.position;                 /* (0,0,0) on a miss */
            skyboxS = !(hit.distance < PT_INF);
            if (hit_id) *hit_id = hit.id;
        }
        if (mode < 10 && fullFrameReprojRender) {  /* SH:865-874 */
            {   /* SH:866: energy is read before Shade() changes it */
                f3 e = ray.energy;
                f3 s = shade(p, &rs, &ray, hit);
                result.x += e.x * s.x / fspp; result.y += e.y * s.y / fspp; result.z += e.z * s.z / fspp;
            }
            for (int i = 1; i < maxb; i++) {       /* SH:867 (4 in the shader) */
                hit = trace_logged(p, &ray, cn);
                if (!(i == 1 && skyboxS)) cn->rays++;
                f3 e = ray.energy;
                f3 s = shade(p, &rs, &ray, hit);
                result.x += e.x * s.x / fspp; result.y += e.y * s.y / fspp; result.z += e.z * s.z / fspp;
                if (!any3(ray.energy)) break;      /* SH:871 */
            }
        }
        if (seed_out) *seed_out = rs.seed;         /* _Seed as the denoiser's hash will see it (SH:564) */
        f3 origin = camo;
        const float alphaNew = pt_length3(pt_sub3(hitPos, origin));
        if (mode == 1 || mode == 5 || mode == 6 || !hs) {   /* SH:933-936 */
            out4[0] = result.x; out4[1] = result.y; out4[2] = result.z; out4[3] = alphaNew;
            continue;
        }
        if (mode == 7 || mode == 8) {              /* SH:938-948; the copies of :944-947 are the caller's */
            if (inside_or_edge) { out2[0] = result.x; out2[1] = result.y; out2[2] = result.z; out2[3] = alphaNew; }
            continue;
        }
        /* every other mode looks the primary hit up in the previous frame: history element 0 */
        f3 originOld = pt_make3(hs->cam->camera_to_world[12], hs->cam->camera_to_world[13], hs->cam->camera_to_world[14]);
        f3 toHit = pt_sub3(hitPos, originOld);
        float distOld = pt_length3(toHit);
        float reproUV[2];
        if (mode == 4) find_camera_ray_vr(hs, H, toHit, prm->left_eye, reproUV);     /* SH:1026 */
        else find_camera_ray(hs, W, H, toHit, reproUV);                              /* SH:890, 955, 991, 1051, 1074 */
        const int on_screen = reproUV[0] < fw && reproUV[0] > 0 && reproUV[1] < fh && reproUV[1] > 0;
        float bl[4] = {0, 0, 0, 0}, direct[4] = {0, 0, 0, 0};
        if (on_screen) {
            bilinear_texture_sample(hs, W, H, reproUV[0], reproUV[1], bl);
            load_result1(hs, W, H, reproUV[0], reproUV[1], direct);
        }
        if (mode == 2 || mode == 0) {              /* SH:950-983; SH:885-925 writes Result2 the same way */
            float divisor = (float)prm->accumulation_frames;
            int take = on_screen && (pt_abs(distOld - bl[3]) < distOld * 0.05f || direct[3] > 80.0f) && !skyboxS;
            if (take && mode == 0) {               /* SH:908-915: history only from rows above the end of this thread group */
                float maximum = pt_floor((float)idy / 32.0f) * 32.0f + 32.0f;
                if (!(reproUV[1] < maximum)) take = 0;
            }
            if (take) {                            /* SH:973, SH:912 */
                out4[0] = (bl[0] * (divisor - 1.0f)) / divisor + result.x / divisor;
                out4[1] = (bl[1] * (divisor - 1.0f)) / divisor + result.y / divisor;
                out4[2] = (bl[2] * (divisor - 1.0f)) / divisor + result.z / divisor;
                out4[3] = distOld;
            } else {                               /* SH:981, SH:914, SH:924 */
                out4[0] = result.x; out4[1] = result.y; out4[2] = result.z; out4[3] = alphaNew;
            }
        } else if (mode == 3) {                    /* SH:986-1019: the history's texel and its four diagonal neighbours */
            int take = on_screen && (pt_abs(distOld - direct[3]) < distOld * 0.05f || direct[3] > 80.0f) && !skyboxS;   /* SH:995-997 */
            if (take) {
                float divisor = 10.0f;
                float n0[4], n1[4], n2[4], n3[4];
                load_result1(hs, W, H, reproUV[0] + 1.0f, reproUV[1] + 1.0f, n0);    /* SH:1000 */
                load_result1(hs, W, H, reproUV[0] - 1.0f, reproUV[1] - 1.0f, n1);    /* SH:1001 */
                load_result1(hs, W, H, reproUV[0] + 1.0f, reproUV[1] + -1.0f, n2);   /* SH:1002 */
                load_result1(hs, W, H, reproUV[0] - 1.0f, reproUV[1] - -1.0f, n3);   /* SH:1003 */
                const float r3[3] = {result.x, result.y, result.z};
                for (int k = 0; k < 3; k++) {
                    float avg = r3[k] / 2.5f;      /* SH:998 */
                    avg += bl[k] / 5.0f;           /* SH:999 */
                    avg += n0[k] / divisor; avg += n1[k] / divisor; avg += n2[k] / divisor; avg += n3[k] / divisor;
                    out4[k] = avg;
                }
                out4[3] = alphaNew;                /* SH:1005 */
            } else {                               /* SH:1011 */
                out4[0] = result.x; out4[1] = result.y; out4[2] = result.z; out4[3] = alphaNew;
            }
        } else if (mode == 4) {                    /* SH:1021-1046 */
            float divisor = (float)prm->accumulation_frames;
            int take = on_screen && (pt_abs(distOld - direct[3]) < distOld * 0.05f || bl[3] > 80.0f) && !skyboxS;   /* SH:1033-1035 */
            if (take) {                            /* SH:1037 */
                out4[0] = (bl[0] * (divisor - 1.0f)) / divisor + result.x / divisor;
                out4[1] = (bl[1] * (divisor - 1.0f)) / divisor + result.y / divisor;
                out4[2] = (bl[2] * (divisor - 1.0f)) / divisor + result.z / divisor;
                out4[3] = distOld;
            } else {                               /* SH:1043 */
                out4[0] = result.x; out4[1] = result.y; out4[2] = result.z; out4[3] = alphaNew;
            }
        } else if (mode == 10) {                   /* SH:1048-1069: trace a path only where the history has nothing to show */
            int take = on_screen && (pt_abs(distOld - bl[3]) < distOld * 0.05f || bl[3] > 80.0f) && !skyboxS;   /* SH:1053-1055 */
            if (take) {
                for (int k = 0; k < 4; k++) out4[k] = bl[k];                         /* SH:1056 */
            } else {                               /* SH:1059-1067 */
                {
                    f3 e = ray.energy;
                    f3 s = shade(p, &rs, &ray, hit);
                    result.x += e.x * s.x / fspp; result.y += e.y * s.y / fspp; result.z += e.z * s.z / fspp;
                }
                for (int i = 1; i < maxb; i++) {
                    hit = trace_logged(p, &ray, cn);
                    if (!(i == 1 && skyboxS)) cn->rays++;
                    f3 e = ray.energy;
                    f3 s = shade(p, &rs, &ray, hit);
                    result.x += e.x * s.x / fspp; result.y += e.y * s.y / fspp; result.z += e.z * s.z / fspp;
                    if (!any3(ray.energy)) break;
                }
                out4[0] = result.x; out4[1] = result.y; out4[2] = result.z; out4[3] = alphaNew;
            }
        } else {                                   /* renderMode >= 11, SH:1071-1104: the frozen frame seen from the moving camera */
            if (on_screen && !skyboxS) { for (int k = 0; k < 4; k++) out4[k] = bl[k]; }   /* SH:1097 */
            else { out4[0] = out4[1] = out4[2] = 0.0f; out4[3] = 1.0f; }             /* SH:1102 */
        }
    }
    return spp > 0;
}

int pto_render(const PtoPrepared *p, const PtFrameParams *prm, int W, int H,
               float *result, int32_t *hit_ids, PtoStats *stats, int threads)
{
    if (prm && prm->render_mode != 1) return 1;
    return pto_render_history(p, prm, NULL, 0, W, H, result, NULL, NULL, hit_ids, stats, threads);
}

int pto_render_history(const PtoPrepared *p, const PtFrameParams *prm, const PtCameraHistory *history, int n_history,
                       int W, int H, float *result, float *result1, float *result2, int32_t *hit_ids, PtoStats *stats,
                       int threads)
{
    if (!p || !prm || !result || W <= 0 || H <= 0 || prm->max_bounces < 1) return 1;
    const int mode = prm->render_mode;
    const int stripes = mode == 5 || mode == 6;
    const int full_reproj = mode == 7 || mode == 8;
    if (mode < 0 || mode == 9 || mode > 12) return 1;       /* RayTracingMaster.cs:977-990 */
    if (mode != 1 && !result1) return 1;
    if ((mode == 0 || full_reproj) && !result2) return 1;
    if ((stripes || full_reproj) && (prm->divisions < 1 || prm->divisions > PT_MAX_CAMERA_HISTORY || prm->counter < 0 || prm->counter >= prm->divisions)) return 1;
    /* the four uniform arrays: elements never uploaded hold zeros */
    PtCameraHistory *cams = (PtCameraHistory *)calloc(PT_MAX_CAMERA_HISTORY, sizeof(PtCameraHistory));
    if (!cams) return 1;
    if (history && n_history > 0) memcpy(cams, history, sizeof(PtCameraHistory) * (size_t)(n_history < PT_MAX_CAMERA_HISTORY ? n_history : PT_MAX_CAMERA_HISTORY));
    History hs = {&cams[0], result1, cams};
    int x0 = 0, y0 = 0, tw = W, th = H;
    if (prm->tile_w > 0 && prm->tile_h > 0) { x0 = prm->tile_x0; y0 = prm->tile_y0; tw = prm->tile_w; th = prm->tile_h; }
    if (x0 < 0 || y0 < 0 || x0 + tw > W || y0 + th > H) { free(cams); return 1; }
    if ((mode == 0 || stripes || full_reproj || mode >= 11) && (tw != W || th != H)) { free(cams); return 1; }   /* whole frames only */
    if (threads <= 0) threads = 1;
    const int screen_passes = mode == 0 && prm->depth != 1 && prm->samples_per_pixel > 0;
    /* SH:728-736 for every dispatched thread, before any of them goes on: Result1 = Result2 (renderMode 0) or
     * Result1 = Result (renderMode 1, 2, 3, 4, 10).  renderMode 1 itself never reads Result1; the copy matters to
     * the modes that show Result1 without copying first (5..8, 11, 12) when the host switches to them.  It is left
     * out when the caller has no Result1 (pto_render) or accumulates into it (an extension: the shader has no
     * accumulation, AddMat.shader runs after it on another target). */
    if (mode == 0 || mode == 2 || mode == 3 || mode == 4 || mode == 10 || (mode == 1 && result1 && prm->accumulate == PT_ACCUM_NONE)) {
        const float *src = mode == 0 ? result2 : result;
        for (int y = y0; y < y0 + th; y++)
            memcpy(result1 + 4 * ((size_t)y * W + x0), src + 4 * ((size_t)y * W + x0), (size_t)tw * 4 * sizeof(float));
    }
    float *seeds = NULL, *denoised = NULL;
    if (screen_passes) {
        seeds = (float *)malloc((size_t)W * H * sizeof(float));
        denoised = (float *)malloc((size_t)W * H * 4 * sizeof(float));
        if (!seeds || !denoised) { free(seeds); free(denoised); free(cams); return 1; }
    }
    /* renderMode 7 / 8, first dispatch of a cycle: which threads get as far as SH:944-947 */
    unsigned char *reached_end = NULL;
    if (full_reproj && prm->counter % prm->divisions == 0) {
        reached_end = (unsigned char *)calloc((size_t)W * H, 1);
        if (!reached_end) { free(seeds); free(denoised); free(cams); return 1; }
    }
    /* the path tracer; renderMode 0 writes its (reprojected) radiance to Result2, SH:912-924 */
    float *trace_out = (mode == 0 && prm->depth != 1) ? result2 : result;
    uint64_t a = 0, b = 0, c = 0, d = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : a, b, c, d)
    for (int y = y0; y < y0 + th; y++) {
        Counters cn = {0, 0, 0, 0, NULL, 0, 0};
        for (int x = x0; x < x0 + tw; x++) {
            size_t px = (size_t)y * W + x;
            int reached = cs_main(p, prm, W, H, x, y, trace_out + 4 * px, full_reproj ? result2 + 4 * px : NULL,
                                  hit_ids ? hit_ids + px : NULL, &cn, &hs, seeds ? seeds + px : NULL);
            if (reached_end) reached_end[px] = (unsigned char)reached;
        }
        a += cn.trace_calls; b += cn.rays; c += cn.paths; d += cn.prims;
    }
    if (screen_passes) {
        /* SH:928: Result = denoiser(id) for every pixel, then SH:929: Result = TAA(id) for every pixel */
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                size_t px = (size_t)y * W + x;
                denoiser(result1, W, H, (float)x, (float)y, seeds[px], denoised + 4 * px);
            }
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++)
                taa(result1, denoised, W, H, (float)x, (float)y, result + 4 * ((size_t)y * W + x));
        free(seeds); free(denoised);
    }
    if (stripes && prm->depth != 1) {
        /* SH:1108-1112: Result1 = Result for the pixels of the rendered stripe - the only threads that get that
         * far - once every pixel above has read the Result1 of before the dispatch */
        float fh = (float)H, stripe = fh / (float)prm->divisions;
        for (int y = 0; y < H; y++) {
            float fy = (float)y;
            if (fy > fh - stripe * (float)prm->counter || fy < fh - stripe * (float)(prm->counter + 1)) continue;
            memcpy(result1 + 4 * (size_t)y * W, result + 4 * (size_t)y * W, (size_t)W * 4 * sizeof(float));
        }
    }
    if (reached_end) {
        /* SH:944-947: Result1 = Result2 and Result = Result2, once no thread reads Result1 any more */
        for (size_t px = 0; px < (size_t)W * H; px++)
            if (reached_end[px]) {
                memcpy(result1 + 4 * px, result2 + 4 * px, 4 * sizeof(float));
                memcpy(result + 4 * px, result2 + 4 * px, 4 * sizeof(float));
            }
        free(reached_end);
    }
    free(cams);
    if (stats) { stats->trace_calls = a; stats->rays = b; stats->paths = c; stats->prims_tested = d; }
    return 0;
}

/* ------------------------------------------------------------ accumulation */

/* ADD:44-63 with Blend SrcAlpha OneMinusSrcAlpha (ADD:11).  Roles: the sampled
 * _MainTex is the history (converged image), the blend destination is the
 * fresh frame; the blended value is the new converged image. */
void pto_accumulate(int mode, uint32_t sample_index, const float *fresh, float *conv, int64_t n)
{
    float S = (float)sample_index;
    for (int64_t i = 0; i < n; i++) {
        const float *f = fresh + 4 * i; float *h = conv + 4 * i;
        if (mode == PT_ACCUM_MEAN) {
            float a = 1.0f / (S + 1.0f);
            for (int k = 0; k < 4; k++) h[k] = f[k] * a + h[k] * (1.0f - a);
        } else if (mode == PT_ACCUM_SUM) {
            for (int k = 0; k < 4; k++) h[k] = sample_index == 0 ? f[k] : h[k] + f[k];
        } else if (mode == PT_ACCUM_ADDMAT) {
            float alpha = h[3];                                    /* ADD:56 */
            float firefly = 0.8f;                                  /* ADD:57 */
            if (h[0] > firefly || h[1] > firefly || h[2] > firefly) alpha = 0.1f; /* ADD:58-60 */
            float t = 1.0f + S * alpha;                            /* ADD:62 */
            float A = 1.0f - 1.0f / t;                             /* ADD:63 */
            for (int k = 0; k < 3; k++) h[k] = h[k] * A + f[k] * (1.0f - A);
            h[3] = A * A + f[3] * (1.0f - A);
        }
    }
}

/* ----------------------------------------------------------------- prepare */

static void node_bounds(PtoPrepared *p, const int32_t *ids, int32_t n, float *lo, float *hi, float *clo, float *chi)
{
    for (int k = 0; k < 3; k++) { lo[k] = clo[k] = INFINITY; hi[k] = chi[k] = -INFINITY; }
    for (int32_t i = 0; i < n; i++) {
        int32_t id = ids[i];
        float blo[3], bhi[3];
        if (id <= p->n_spheres) {
            const PtSphere *s = &p->spheres[id - 1];
            for (int k = 0; k < 3; k++) { blo[k] = s->position[k] - s->radius; bhi[k] = s->position[k] + s->radius; }
        } else {
            const Tri *t = &p->tris[id - 1 - p->n_spheres];
            const float *a = &t->v0.x, *b = &t->v1.x, *c = &t->v2.x;
            for (int k = 0; k < 3; k++) { blo[k] = fminf(a[k], fminf(b[k], c[k])); bhi[k] = fmaxf(a[k], fmaxf(b[k], c[k])); }
        }
        for (int k = 0; k < 3; k++) {
            float ce = 0.5f * (blo[k] + bhi[k]);
            lo[k] = fminf(lo[k], blo[k]); hi[k] = fmaxf(hi[k], bhi[k]);
            clo[k] = fminf(clo[k], ce); chi[k] = fmaxf(chi[k], ce);
        }
    }
}

static float centroid(const PtoPrepared *p, int32_t id, int axis)
{
    if (id <= p->n_spheres) return p->spheres[id - 1].position[axis];
    const Tri *t = &p->tris[id - 1 - p->n_spheres];
    const float *a = &t->v0.x, *b = &t->v1.x, *c = &t->v2.x;
    return 0.5f * (fminf(a[axis], fminf(b[axis], c[axis])) + fmaxf(a[axis], fmaxf(b[axis], c[axis])));
}

/* Rearranges ids[0..n) so that ids[k] is the element a sort by key[id - 1] would put there, smaller keys before it and
 * larger ones after it (quickselect, median-of-three pivot).  Which primitives share a leaf does not matter to the
 * results: the closest hit is the (distance, id)-smallest one whatever the tree looks like. */
static void select_kth(int32_t *ids, int32_t n, int32_t k, const float *key)
{
    int32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const int32_t mid = lo + (hi - lo) / 2;
        const float x = key[ids[lo] - 1], y = key[ids[mid] - 1], z = key[ids[hi] - 1];
        const float pv = x < y ? (y < z ? y : (x < z ? z : x)) : (x < z ? x : (y < z ? z : y));
        int32_t i = lo, j = hi;
        while (i <= j) {
            while (key[ids[i] - 1] < pv) i++;
            while (key[ids[j] - 1] > pv) j--;
            if (i <= j) { const int32_t t = ids[i]; ids[i] = ids[j]; ids[j] = t; i++; j--; }
        }
        if (k <= j) hi = j; else if (k >= i) lo = i; else return;
    }
}

/* cent: the primitives' box centres, [axis][id - 1].  Large subtrees are built as OpenMP tasks. */
static int32_t build_node(PtoPrepared *p, int32_t first, int32_t n, float pad, const float *cent, int32_t np)
{
    int32_t me = __atomic_fetch_add(&p->n_nodes, 1, __ATOMIC_RELAXED);
    float lo[3], hi[3], clo[3], chi[3];
    node_bounds(p, p->order + first, n, lo, hi, clo, chi);
    ONode *nd = &p->nodes[me];
    for (int k = 0; k < 3; k++) { nd->lo[k] = lo[k] - pad; nd->hi[k] = hi[k] + pad; }
    nd->first = first; nd->count = 0; nd->left = nd->right = -1;
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    if (n <= 4 || !(chi[axis] > clo[axis])) { nd->count = n; return me; }
    int32_t half = n / 2;
    select_kth(p->order + first, n, half, cent + (size_t)axis * (size_t)np);
    int32_t l = -1, r = -1;
    if (n > 65536) {
#pragma omp task shared(l) firstprivate(p, first, half, pad, cent, np)
        l = build_node(p, first, half, pad, cent, np);
        r = build_node(p, first + half, n - half, pad, cent, np);
#pragma omp taskwait
    } else {
        l = build_node(p, first, half, pad, cent, np);
        r = build_node(p, first + half, n - half, pad, cent, np);
    }
    p->nodes[me].left = l; p->nodes[me].right = r;
    return me;
}

PtoPrepared *pto_prepare(const PtoScene *sc, int accel)
{
    PtoPrepared *p = (PtoPrepared *)calloc(1, sizeof *p);
    if (!p) return NULL;
    p->accel = accel;
    p->n_spheres = sc->n_spheres;
    if (sc->n_spheres > 0) {
        p->spheres = (PtSphere *)malloc(sizeof(PtSphere) * (size_t)sc->n_spheres);
        memcpy(p->spheres, sc->spheres, sizeof(PtSphere) * (size_t)sc->n_spheres);
    }
    /* flatten mesh objects in order, triangles in order (SH:278-282, SH:236) */
    int64_t nt = 0;
    for (int32_t o = 0; o < sc->n_objects; o++) nt += sc->objects[o].indices_count / 3;
    p->n_tris = (int32_t)nt;
    if (nt > 0) p->tris = (Tri *)malloc(sizeof(Tri) * (size_t)nt);
    int64_t k = 0;
    for (int32_t o = 0; o < sc->n_objects; o++) {
        const PtMeshObject *mo = &sc->objects[o];
        for (int32_t i = mo->indices_offset; i + 2 < mo->indices_offset + mo->indices_count; i += 3) {
            const float *a = sc->vertices + 3 * (size_t)sc->indices[i];
            const float *b = sc->vertices + 3 * (size_t)sc->indices[i + 1];
            const float *c = sc->vertices + 3 * (size_t)sc->indices[i + 2];
            p->tris[k].v0 = pt_mul_point(mo->local_to_world, pt_make3(a[0], a[1], a[2])); /* SH:238-240 */
            p->tris[k].v1 = pt_mul_point(mo->local_to_world, pt_make3(b[0], b[1], b[2]));
            p->tris[k].v2 = pt_mul_point(mo->local_to_world, pt_make3(c[0], c[1], c[2]));
            p->tris[k].object = o;
            k++;
        }
    }
    p->n_tris = (int32_t)k;
    if (sc->materials && sc->n_materials == sc->n_objects && sc->n_objects > 0) {
        p->materials = (PtMaterial *)malloc(sizeof(PtMaterial) * (size_t)sc->n_materials);
        memcpy(p->materials, sc->materials, sizeof(PtMaterial) * (size_t)sc->n_materials);
        p->n_materials = sc->n_materials;
    }
    if (sc->sky && sc->sky_w > 0 && sc->sky_h > 0) {
        size_t n = (size_t)sc->sky_w * sc->sky_h * 4;
        p->sky = (float *)malloc(n * sizeof(float));
        memcpy(p->sky, sc->sky, n * sizeof(float));
        p->sky_w = sc->sky_w; p->sky_h = sc->sky_h;
    }
    /* the tree holds the triangles only (see box_hit); cent is indexed by prim id - 1, so it keeps the spheres' slots */
    int32_t np = p->n_spheres + p->n_tris;
    if (accel && p->n_tris > 0) {
        const int32_t nt = p->n_tris;
        p->order = (int32_t *)malloc(sizeof(int32_t) * (size_t)nt);
        for (int32_t i = 0; i < nt; i++) p->order[i] = p->n_spheres + i + 1;
        p->nodes = (ONode *)malloc(sizeof(ONode) * (size_t)(2 * nt));
        float lo[3], hi[3], clo[3], chi[3];
        node_bounds(p, p->order, nt, lo, hi, clo, chi);
        float diag = sqrtf((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
        float *cent = (float *)malloc(sizeof(float) * 3 * (size_t)np);
        for (int axis = 0; axis < 3; axis++)
            for (int32_t i = p->n_spheres; i < np; i++) cent[(size_t)axis * (size_t)np + i] = centroid(p, i + 1, axis);
#pragma omp parallel
#pragma omp single
        build_node(p, 0, nt, 1e-5f * diag + 1e-6f, cent, np);
        free(cent);
        /* thinness and edge length per node, children first (a node's index is smaller than its children's) */
        for (int32_t n = p->n_nodes - 1; n >= 0; n--) {
            ONode *nd = &p->nodes[n];
            nd->thin = 0.0f; nd->len = 0.0f;
            if (nd->count) {
                for (int32_t q = 0; q < nd->count; q++) {
                    const Tri *tr = &p->tris[p->order[nd->first + q] - 1 - p->n_spheres];
                    const f3 e1 = pt_sub3(tr->v1, tr->v0), e2 = pt_sub3(tr->v2, tr->v0), e3 = pt_sub3(e2, e1);
                    const float l2 = fmaxf(pt_dot3(e1, e1), fmaxf(pt_dot3(e2, e2), pt_dot3(e3, e3)));
                    const f3 nn = pt_cross3(e1, e2);
                    float thin = l2 / sqrtf(pt_dot3(nn, nn));
                    if (!(thin < 1e30f)) thin = 1e30f;              /* no area, NaN */
                    float len = sqrtf(l2);
                    if (!(len < 1e30f)) len = 1e30f;
                    nd->thin = fmaxf(nd->thin, thin); nd->len = fmaxf(nd->len, len);
                }
            } else {
                nd->thin = fmaxf(p->nodes[nd->left].thin, p->nodes[nd->right].thin);
                nd->len = fmaxf(p->nodes[nd->left].len, p->nodes[nd->right].len);
            }
        }
    }
    return p;
}

void pto_release(PtoPrepared *p)
{
    if (!p) return;
    free(p->spheres); free(p->tris); free(p->sky); free(p->nodes); free(p->order); free(p->materials); free(p);
}

/* ------------------------------------------------------ pieces for KATs */

void pto_rand_sequence(float seed, int idx, int idy, int n, float *out)
{
    RngState rs; rs.seed = seed; rs.px = (float)idx * 2.0f; rs.py = (float)idy * 2.0f;
    for (int i = 0; i < n; i++) out[i] = rng(&rs);
}

void pto_camera_ray(const float *c2w, const float *ip, int idx, int idy, int W, int H, float *od)
{
    float u = (float)idx / (float)W * 2.0f - 1.0f, v = (float)idy / (float)H * 2.0f - 1.0f;
    Ray r = create_camera_ray(c2w, ip, u, v);
    od[0] = r.origin.x; od[1] = r.origin.y; od[2] = r.origin.z;
    od[3] = r.direction.x; od[4] = r.direction.y; od[5] = r.direction.z;
}

void pto_trace_ray(const PtoPrepared *p, const float *od, int32_t *id, float *t)
{
    Counters cn = {0, 0, 0, 0, NULL, 0, 0};
    Ray r = create_ray(pt_make3(od[0], od[1], od[2]), pt_make3(od[3], od[4], od[5]));
    RayHit h = trace(p, &r, &cn);
    *id = h.id; *t = h.distance;
}

void pto_sky_sample(const PtoPrepared *p, const float *d, float *rgb)
{
    Ray r = create_ray(pt_make3(0, 0, 0), pt_make3(d[0], d[1], d[2]));
    f3 c = skybox_sample(p, &r);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* Every Trace() of one pixel, in order: 8 floats per call (origin, direction, id, distance). */
int pto_path_dump(const PtoPrepared *p, const PtFrameParams *prm, int W, int H, int idx, int idy, float *log, int cap)
{
    Counters cn = {0, 0, 0, 0, log, cap, 0};
    float out4[4]; int32_t id;
    cs_main(p, prm, W, H, idx, idy, out4, NULL, &id, &cn, NULL, NULL);
    return cn.log_n;
}

void pto_find_camera_ray(const PtCameraHistory *old_camera, int W, int H, const float *dir3, float *uv2)
{
    History hs = {old_camera, NULL, NULL};
    find_camera_ray(&hs, W, H, pt_make3(dir3[0], dir3[1], dir3[2]), uv2);
}

void pto_bilinear_sample(const float *result1, int W, int H, float px, float py, float *out4)
{
    History hs = {NULL, result1, NULL};
    bilinear_texture_sample(&hs, W, H, px, py, out4);
}

void pto_denoiser(const float *result1, int W, int H, int x, int y, float seed, float *out4) { denoiser(result1, W, H, (float)x, (float)y, seed, out4); }
void pto_taa(const float *result1, const float *current, int W, int H, int x, int y, float *out4) { taa(result1, current, W, H, (float)x, (float)y, out4); }

void pto_display_denoise(const float *result, int W, int H, float *display, int threads)
{
    if (threads <= 0) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) display_denoiser(result, W, H, x, y, display + 4 * ((size_t)y * W + x));
}

float pto_fmath(int which, float a, float b)
{
    switch (which) {
    case 0: return pt_sin(a);
    case 1: return pt_cos(a);
    case 2: return pt_pow(a, b);
    case 3: return pt_acos(a);
    case 4: return pt_atan2(a, b);
    case 5: return pt_log2(a);
    case 6: return pt_exp2(a);
    case 7: return pt_exp(a);
    case 8: return pt_frac(a);
    default: return 0.0f;
    }
}
