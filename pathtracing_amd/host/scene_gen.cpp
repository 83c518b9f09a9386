// Synthetic scenes and scene pieces (CPU only).  See ptmaster.h.
#include "scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

namespace ptm {

namespace {

// PCG32 (O'Neill), the build's own stand-in for UnityEngine.Random.
struct Pcg32 {
    uint64_t state, inc;
    explicit Pcg32(uint64_t seed, uint64_t seq = 54u)
    {
        state = 0u; inc = (seq << 1u) | 1u;
        next(); state += seed; next();
    }
    uint32_t next()
    {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    // Random.value: uniform in [0,1]
    float value() { return (float)(next() >> 8) * (1.0f / 16777215.0f); }
    float range(float a, float b) { return a + value() * (b - a); }
};

void hsv_to_rgb(float h, float s, float v, float rgb[3])
{
    float i = std::floor(h * 6.0f);
    float f = h * 6.0f - i;
    float p = v * (1.0f - s), q = v * (1.0f - f * s), t = v * (1.0f - (1.0f - f) * s);
    switch (((int)i) % 6) {
    case 0: rgb[0] = v; rgb[1] = t; rgb[2] = p; break;
    case 1: rgb[0] = q; rgb[1] = v; rgb[2] = p; break;
    case 2: rgb[0] = p; rgb[1] = v; rgb[2] = t; break;
    case 3: rgb[0] = p; rgb[1] = q; rgb[2] = v; break;
    case 4: rgb[0] = t; rgb[1] = p; rgb[2] = v; break;
    default: rgb[0] = v; rgb[1] = p; rgb[2] = q; break;
    }
}

// Cheap smooth value noise in [-1,1] for displacing surfaces.
float hash3(int x, int y, int z, uint32_t seed)
{
    uint32_t h = (uint32_t)x * 0x8da6b343u ^ (uint32_t)y * 0xd8163841u ^ (uint32_t)z * 0xcb1ab31fu ^ seed * 0x9e3779b9u;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return (float)(h & 0xffffff) * (2.0f / 16777215.0f) - 1.0f;
}
float noise3(float x, float y, float z, uint32_t seed)
{
    int xi = (int)std::floor(x), yi = (int)std::floor(y), zi = (int)std::floor(z);
    float fx = x - xi, fy = y - yi, fz = z - zi;
    auto sm = [](float t) { return t * t * (3.0f - 2.0f * t); };
    float u = sm(fx), v = sm(fy), w = sm(fz);
    float c[2][2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int d = 0; d < 2; d++) c[a][b][d] = hash3(xi + a, yi + b, zi + d, seed);
    auto lerp = [](float a, float b, float t) { return a + t * (b - a); };
    return lerp(lerp(lerp(c[0][0][0], c[1][0][0], u), lerp(c[0][1][0], c[1][1][0], u), v),
                lerp(lerp(c[0][0][1], c[1][0][1], u), lerp(c[0][1][1], c[1][1][1], u), v), w);
}

const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

}  // namespace

// ------------------------------------------------------------------ meshes

void matrix_trs(float m[16], float tx, float ty, float tz, float yaw_deg, float scale)
{
    float a = yaw_deg * 3.14159265358979f / 180.0f, c = std::cos(a) * scale, s = std::sin(a) * scale;
    // column-major: column 0 = (c,0,-s), column 1 = (0,scale,0), column 2 = (s,0,c), column 3 = translation
    float r[16] = {c, 0, -s, 0, 0, scale, 0, 0, s, 0, c, 0, tx, ty, tz, 1};
    std::memcpy(m, r, sizeof r);
}

int MeshBuilder::add_vertex(float x, float y, float z)
{
    vertices.push_back(x); vertices.push_back(y); vertices.push_back(z);
    return (int)(vertices.size() / 3) - 1;
}

// Adds (a,b,c) wound so that (b-a)x(c-a) points away from `inside`; the shader
// culls back faces (RayTracingShader.compute:206), so winding decides visibility.
void MeshBuilder::add_tri_outward(int a, int b, int c, const float inside[3])
{
    const float *A = &vertices[3 * a], *B = &vertices[3 * b], *C = &vertices[3 * c];
    float e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
    float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    float len2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    if (!(len2 > 0.0f)) return;  // degenerate
    float ctr[3] = {(A[0] + B[0] + C[0]) / 3 - inside[0], (A[1] + B[1] + C[1]) / 3 - inside[1], (A[2] + B[2] + C[2]) / 3 - inside[2]};
    bool outward = n[0] * ctr[0] + n[1] * ctr[1] + n[2] * ctr[2] >= 0.0f;
    indices.push_back(a);
    indices.push_back(outward ? b : c);
    indices.push_back(outward ? c : b);
}

// Closed lat-long surface around `c` with radius R * (1 + amp * noise).
void MeshBuilder::add_blob(const float c[3], float R, int nu, int nv, float amp, float freq, uint32_t seed)
{
    int base = (int)(vertices.size() / 3);
    for (int j = 0; j <= nv; j++) {
        double th = M_PI * j / nv;
        for (int i = 0; i < nu; i++) {
            double ph = 2.0 * M_PI * i / nu;
            float dx = (float)(std::sin(th) * std::cos(ph)), dy = (float)std::cos(th), dz = (float)(std::sin(th) * std::sin(ph));
            float r = R * (1.0f + amp * noise3(dx * freq + 11.3f, dy * freq + 5.7f, dz * freq + 2.9f, seed));
            add_vertex(c[0] + r * dx, c[1] + r * dy, c[2] + r * dz);
        }
    }
    for (int j = 0; j < nv; j++)
        for (int i = 0; i < nu; i++) {
            int i1 = (i + 1) % nu;
            int a = base + j * nu + i, b = base + j * nu + i1, d = base + (j + 1) * nu + i, e = base + (j + 1) * nu + i1;
            if (j != 0) add_tri_outward(a, b, e, c);
            if (j != nv - 1) add_tri_outward(a, e, d, c);
        }
}

void MeshBuilder::add_box(const float lo[3], const float hi[3])
{
    float c[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
    int v[8];
    for (int k = 0; k < 8; k++) v[k] = add_vertex(k & 1 ? hi[0] : lo[0], k & 2 ? hi[1] : lo[1], k & 4 ? hi[2] : lo[2]);
    const int f[6][4] = {{0, 1, 3, 2}, {4, 5, 7, 6}, {0, 1, 5, 4}, {2, 3, 7, 6}, {0, 2, 6, 4}, {1, 3, 7, 5}};
    for (auto &q : f) { add_tri_outward(v[q[0]], v[q[1]], v[q[2]], c); add_tri_outward(v[q[0]], v[q[2]], v[q[3]], c); }
}

// Fluted column: axis along +y from y0 to y1.
void MeshBuilder::add_column(float cx, float cz, float y0, float y1, float radius, int sides, int rings)
{
    int base = (int)(vertices.size() / 3);
    for (int j = 0; j <= rings; j++) {
        float t = (float)j / rings, y = y0 + t * (y1 - y0);
        float taper = 1.0f - 0.15f * t;
        for (int i = 0; i < sides; i++) {
            double ph = 2.0 * M_PI * i / sides;
            float flute = 1.0f + 0.04f * (float)std::cos(ph * 12.0);
            float r = radius * taper * flute;
            add_vertex(cx + r * (float)std::cos(ph), y, cz + r * (float)std::sin(ph));
        }
    }
    for (int j = 0; j < rings; j++)
        for (int i = 0; i < sides; i++) {
            int i1 = (i + 1) % sides;
            int a = base + j * sides + i, b = base + j * sides + i1, d = base + (j + 1) * sides + i, e = base + (j + 1) * sides + i1;
            float y = 0.5f * (vertices[3 * a + 1] + vertices[3 * d + 1]);
            float in[3] = {cx, y, cz};
            add_tri_outward(a, b, e, in);
            add_tri_outward(a, e, d, in);
        }
}

// Half torus (an arch) in the plane z = cz spanning x0..x1, springing from height y.
void MeshBuilder::add_arch(float x0, float x1, float y, float cz, float tube, int segs, int sides)
{
    int base = (int)(vertices.size() / 3);
    float cx = 0.5f * (x0 + x1), R = 0.5f * (x1 - x0);
    for (int j = 0; j <= segs; j++) {
        double a = M_PI * j / segs;
        float ox = cx + R * (float)std::cos(a), oy = y + R * (float)std::sin(a);
        for (int i = 0; i < sides; i++) {
            double ph = 2.0 * M_PI * i / sides;
            float rr = tube * (float)std::cos(ph);
            add_vertex(ox + rr * (float)std::cos(a), oy + rr * (float)std::sin(a), cz + tube * (float)std::sin(ph));
        }
    }
    for (int j = 0; j < segs; j++) {
        double am = M_PI * (j + 0.5) / segs;
        float in[3] = {cx + R * (float)std::cos(am), y + R * (float)std::sin(am), cz};
        for (int i = 0; i < sides; i++) {
            int i1 = (i + 1) % sides;
            int a = base + j * sides + i, b = base + j * sides + i1, d = base + (j + 1) * sides + i, e = base + (j + 1) * sides + i1;
            add_tri_outward(a, b, e, in);
            add_tri_outward(a, e, d, in);
        }
    }
}

// Bumpy wall: a height field over the rectangle x0..x1, y0..y1 at z, facing the side `face` (+1 or -1 in z).
void MeshBuilder::add_wall(float x0, float x1, float y0, float y1, float z, float face, int nx, int ny, float bump, uint32_t seed)
{
    int base = (int)(vertices.size() / 3);
    for (int j = 0; j <= ny; j++)
        for (int i = 0; i <= nx; i++) {
            float x = x0 + (x1 - x0) * i / nx, y = y0 + (y1 - y0) * j / ny;
            float dz = bump * noise3(x * 0.9f, y * 0.9f, 0.5f, seed);
            add_vertex(x, y, z + face * dz);
        }
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            int a = base + j * (nx + 1) + i, b = a + 1, d = a + (nx + 1), e = d + 1;
            float in[3] = {vertices[3 * a], vertices[3 * a + 1], z - face * 10.0f};
            add_tri_outward(a, b, e, in);
            add_tri_outward(a, e, d, in);
        }
}

// --------------------------------------------------------------- scene pieces

int sphere_field(int sphere_seed, uint32_t spheres_max, float rmin, float rmax, float placement_radius,
                 std::vector<PtSphere> &out)
{
    Pcg32 rng((uint64_t)(uint32_t)sphere_seed);                       // Random.InitState(SphereSeed)  M:317
    out.clear();
    for (uint32_t i = 0; i < spheres_max; i++) {                      // M:321
        PtSphere s; std::memset(&s, 0, sizeof s);
        s.radius = rmin + rng.value() * (rmax - rmin);                // M:326
        float px, pz;                                                 // Random.insideUnitCircle  M:327
        do { px = rng.range(-1.0f, 1.0f); pz = rng.range(-1.0f, 1.0f); } while (px * px + pz * pz > 1.0f);
        s.position[0] = px * placement_radius; s.position[1] = s.radius; s.position[2] = pz * placement_radius;  // M:328
        bool skip = false;                                            // M:331-336
        for (const PtSphere &o : out) {
            float min_dist = s.radius + o.radius;
            float dx = s.position[0] - o.position[0], dy = s.position[1] - o.position[1], dz = s.position[2] - o.position[2];
            if (dx * dx + dy * dy + dz * dz < min_dist * min_dist) { skip = true; break; }
        }
        if (skip) continue;
        float color[3];                                               // Random.ColorHSV()  M:339
        { float h = rng.value(), sa = rng.value(), v = rng.value(); hsv_to_rgb(h, sa, v, color); }
        float chance = rng.value();                                   // M:340
        if (chance < 0.8f) {                                          // M:341-347
            bool metal = chance < 0.4f;
            for (int k = 0; k < 3; k++) {
                s.albedo[k] = metal ? 0.0f : color[k];
                s.specular[k] = metal ? color[k] : 0.04f;
            }
            s.smoothness = rng.value();
        } else {                                                      // ColorHSV(0,1,0,1,3,8)  M:350-351
            float h = rng.value(), sa = rng.value(), v = rng.range(3.0f, 8.0f);
            hsv_to_rgb(h, sa, v, s.emission);
        }
        out.push_back(s);                                             // M:355
    }
    return (int)out.size();
}

void sky_procedural(int w, int h, float *rgba)
{
    // sun direction (normalised)
    const double sx = 0.45, sy = 0.55, sz = -0.70;
    const double sl = std::sqrt(sx * sx + sy * sy + sz * sz);
    for (int iy = 0; iy < h; iy++) {
        double v = (iy + 0.5) / h;
        double dy = std::cos((1.0 - v) * M_PI);          // v = 1 - acos(d.y)/pi  (SH:329, Repeat wrap)
        double st = std::sqrt(std::max(0.0, 1.0 - dy * dy));
        for (int ix = 0; ix < w; ix++) {
            double u = (ix + 0.5) / w;
            double a = -2.0 * M_PI * u;                   // u = atan2(d.x, -d.z) / -pi * 0.5  (SH:330)
            double dx = std::sin(a) * st, dz = -std::cos(a) * st;
            double t = 0.5 * (dy + 1.0);
            double r, g, b;
            if (dy >= 0.0) {                              // horizon haze -> zenith blue
                double k = std::pow(1.0 - dy, 3.0);
                r = 0.25 + 0.55 * k; g = 0.40 + 0.45 * k; b = 0.75 + 0.20 * k;
            } else {                                      // below the horizon: dim ground bounce
                r = 0.18 + 0.1 * t; g = 0.16 + 0.1 * t; b = 0.14 + 0.1 * t;
            }
            double cs = (dx * sx + dy * sy + dz * sz) / sl;
            double ang = std::acos(std::max(-1.0, std::min(1.0, cs)));
            double sun = 20.0 * std::exp(-(ang * ang) / (2.0 * 0.035 * 0.035)) + 0.6 * std::exp(-(ang * ang) / (2.0 * 0.25 * 0.25));
            float *p = rgba + 4 * ((size_t)iy * w + ix);
            p[0] = (float)(r + sun); p[1] = (float)(g + 0.95 * sun); p[2] = (float)(b + 0.8 * sun); p[3] = 1.0f;
        }
    }
}

// Radiance picture (.hdr, "32-bit_rle_rgbe"): the format of the reference's SkyboxTexture asset (Assets/HDRI/cape_hill_4k.hdr,
// bound at RayTracingMaster.cs:457).  Texels become mantissa * 2^(exponent - 136), alpha 1.  The file's first scanline is the top
// of the picture (resolution "-Y h +X w") and the texture's v = 0 is its bottom, so scanlines are stored last first.  A
// picture larger than max_size in either direction is halved (2x2 box) until it fits - the asset's import setting is
// maxTextureSize 2048 (cape_hill_4k.hdr.meta:34); the importer's own resampling filter and its BC6H compression are not reproduced.
bool sky_load_hdr(const char *path, int max_size, int &w, int &h, std::vector<float> &rgba, std::string &err)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    struct Closer { FILE *f; ~Closer() { std::fclose(f); } } closer{f};
    char line[512];
    if (!std::fgets(line, sizeof line, f) || (std::strncmp(line, "#?RADIANCE", 10) != 0 && std::strncmp(line, "#?RGBE", 6) != 0)) { err = "not a Radiance picture"; return false; }
    bool format_ok = false;
    for (;;) {
        if (!std::fgets(line, sizeof line, f)) { err = "header ends early"; return false; }
        if (line[0] == '\n' || (line[0] == '\r' && line[1] == '\n')) break;
        if (std::strncmp(line, "FORMAT=", 7) == 0) {
            if (std::strncmp(line + 7, "32-bit_rle_rgbe", 15) != 0) { err = "only FORMAT=32-bit_rle_rgbe is read"; return false; }
            format_ok = true;
        }
    }
    if (!format_ok) { err = "no FORMAT line"; return false; }
    char sy = 0, sx = 0; int fh = 0, fw = 0;
    if (!std::fgets(line, sizeof line, f) || std::sscanf(line, "%cY %d %cX %d", &sy, &fh, &sx, &fw) != 4 || (sy != '-' && sy != '+') || (sx != '-' && sx != '+') ||
        fw <= 0 || fh <= 0 || fw > 32768 || fh > 32768) { err = "resolution line is not '-Y h +X w'"; return false; }
    // a header may claim up to 32768 x 32768 texels (16 GB of floats): the picture is refused rather than the process ended
    if ((size_t)fw * (size_t)fh > ((size_t)1 << 28)) { err = "picture larger than 2^28 texels"; return false; }
    std::vector<unsigned char> scan;
    std::vector<float> img;
    try { scan.resize((size_t)fw * 4); img.resize((size_t)fw * fh * 4); }
    catch (const std::bad_alloc &) { err = "out of memory for the picture"; return false; }
    for (int y = 0; y < fh; y++) {
        unsigned char head[4];
        if (std::fread(head, 1, 4, f) != 4) { err = "pixel data ends early"; return false; }
        if (head[0] == 2 && head[1] == 2 && !(head[2] & 0x80) && fw >= 8 && fw < 32768) {
            // run-length encoded scanline: the four channels one after the other
            if (((int)head[2] << 8 | head[3]) != fw) { err = "scanline width differs from the header"; return false; }
            for (int c = 0; c < 4; c++)
                for (int x = 0; x < fw;) {
                    int n = std::fgetc(f);
                    if (n == EOF) { err = "pixel data ends early"; return false; }
                    if (n > 128) {
                        n -= 128;
                        const int v = std::fgetc(f);
                        if (v == EOF || x + n > fw) { err = "bad run in a scanline"; return false; }
                        while (n--) scan[(size_t)(x++) * 4 + c] = (unsigned char)v;
                    } else {
                        if (n == 0 || x + n > fw) { err = "bad run in a scanline"; return false; }
                        while (n--) { const int v = std::fgetc(f); if (v == EOF) { err = "pixel data ends early"; return false; } scan[(size_t)(x++) * 4 + c] = (unsigned char)v; }
                    }
                }
        } else {
            // flat scanline; (1, 1, 1, n) repeats the previous texel n times (the old run-length form)
            std::memcpy(scan.data(), head, 4);
            int x = 1, shift = 0;
            if (head[0] == 1 && head[1] == 1 && head[2] == 1) { err = "a scanline starts with a repeat"; return false; }
            while (x < fw) {
                unsigned char px[4];
                if (std::fread(px, 1, 4, f) != 4) { err = "pixel data ends early"; return false; }
                if (px[0] == 1 && px[1] == 1 && px[2] == 1) {
                    // consecutive repeat texels give the count's bytes, low byte first: a width below 32768 needs two at most,
                    // and a count of zero would leave x where it is while the shift grows without bound
                    if (px[3] == 0 || shift > 24) { err = "bad run in a scanline"; return false; }
                    long n = (long)px[3] << shift;
                    if (x + n > fw) { err = "bad run in a scanline"; return false; }
                    while (n--) { std::memcpy(&scan[(size_t)x * 4], &scan[(size_t)(x - 1) * 4], 4); x++; }
                    shift += 8;
                } else { std::memcpy(&scan[(size_t)x * 4], px, 4); x++; shift = 0; }
            }
        }
        const int row = sy == '-' ? fh - 1 - y : y;
        for (int x = 0; x < fw; x++) {
            const unsigned char *q = &scan[(size_t)x * 4];
            const float scale = q[3] ? std::ldexp(1.0f, (int)q[3] - 136) : 0.0f;
            float *o = &img[((size_t)row * fw + (sx == '+' ? x : fw - 1 - x)) * 4];
            o[0] = q[0] * scale; o[1] = q[1] * scale; o[2] = q[2] * scale; o[3] = 1.0f;
        }
    }
    while (max_size > 0 && (fw > max_size || fh > max_size) && fw % 2 == 0 && fh % 2 == 0) {
        const int nw = fw / 2, nh = fh / 2;
        std::vector<float> half;
        try { half.resize((size_t)nw * nh * 4); } catch (const std::bad_alloc &) { err = "out of memory for the picture"; return false; }
        for (int y = 0; y < nh; y++)
            for (int x = 0; x < nw; x++)
                for (int c = 0; c < 4; c++)
                    half[((size_t)y * nw + x) * 4 + c] = 0.25f * (img[((size_t)(2 * y) * fw + 2 * x) * 4 + c] + img[((size_t)(2 * y) * fw + 2 * x + 1) * 4 + c] +
                                                                img[((size_t)(2 * y + 1) * fw + 2 * x) * 4 + c] + img[((size_t)(2 * y + 1) * fw + 2 * x + 1) * 4 + c]);
        img.swap(half); fw = nw; fh = nh;
    }
    w = fw; h = fh; rgba.swap(img);
    return true;
}

void camera_look_at(const float eye[3], const float target[3], const float up[3], float fov_y_deg, float aspect,
                    float z_near, float z_far, float c2w[16], float ip[16])
{
    auto norm = [](double *v) { double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; };
    double f[3] = {target[0] - eye[0], target[1] - eye[1], target[2] - eye[2]};
    norm(f);
    double u0[3] = {up[0], up[1], up[2]};
    double r[3] = {f[1] * u0[2] - f[2] * u0[1], f[2] * u0[0] - f[0] * u0[2], f[0] * u0[1] - f[1] * u0[0]};
    norm(r);
    double u[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};
    // camera space: +x right, +y up, -z forward; columns of cameraToWorld are those axes in world space
    float m[16] = {(float)r[0], (float)r[1], (float)r[2], 0, (float)u[0], (float)u[1], (float)u[2], 0,
                   (float)-f[0], (float)-f[1], (float)-f[2], 0, eye[0], eye[1], eye[2], 1};
    std::memcpy(c2w, m, sizeof m);
    // inverse of the OpenGL perspective matrix
    double tf = std::tan(fov_y_deg * M_PI / 360.0);
    float p[16];
    std::memset(p, 0, sizeof p);
    p[0] = (float)(aspect * tf);                                     // (0,0)
    p[5] = (float)tf;                                                // (1,1)
    p[11] = (float)((z_near - z_far) / (2.0 * z_far * z_near));      // (3,2)
    p[14] = -1.0f;                                                   // (2,3)
    p[15] = (float)((z_far + z_near) / (2.0 * z_far * z_near));      // (3,3)
    std::memcpy(ip, p, sizeof p);
}

void camera_history_entry(const float eye[3], const float target[3], const float up[3], float fov_y_deg, float aspect,
                          float z_near, float z_far, PtCameraHistory *out)
{
    camera_look_at(eye, target, up, fov_y_deg, aspect, z_near, z_far, out->camera_to_world, out->camera_inverse_projection);
    // worldToCamera: inverse of the rigid cameraToWorld (rotation transposed, translation -R^T * eye)
    const float *m = out->camera_to_world;
    float *w = out->world_to_camera;
    std::memset(w, 0, sizeof(float) * 16);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) w[c * 4 + r] = m[r * 4 + c];
    for (int r = 0; r < 3; r++)
        w[12 + r] = (float)-((double)m[r * 4 + 0] * eye[0] + (double)m[r * 4 + 1] * eye[1] + (double)m[r * 4 + 2] * eye[2]);
    w[15] = 1.0f;
    // OpenGL perspective matrix (what Camera.projectionMatrix holds)
    double tf = std::tan(fov_y_deg * M_PI / 360.0);
    float *q = out->camera_projection;
    std::memset(q, 0, sizeof(float) * 16);
    q[0] = (float)(1.0 / (aspect * tf));                               // (0,0)
    q[5] = (float)(1.0 / tf);                                          // (1,1)
    q[10] = (float)(-(z_far + z_near) / (double)(z_far - z_near));     // (2,2)
    q[11] = -1.0f;                                                     // (3,2)
    q[14] = (float)(-2.0 * z_far * z_near / (double)(z_far - z_near)); // (2,3)
}

bool matrix_inverse(const float m[16], float out[16])
{
    double a[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { a[r][c] = m[c * 4 + r]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (!(std::fabs(a[piv][col]) > 0.0)) {
            for (int i = 0; i < 16; i++) out[i] = (i % 5 == 0) ? 1.0f : 0.0f;
            return false;
        }
        if (piv != col)
            for (int c = 0; c < 8; c++) std::swap(a[piv][c], a[col][c]);
        const double d = a[col][col];
        for (int c = 0; c < 8; c++) a[col][c] /= d;
        for (int r = 0; r < 4; r++) {
            if (r == col) continue;
            const double f = a[r][col];
            if (f != 0.0)
                for (int c = 0; c < 8; c++) a[r][c] -= f * a[col][c];
        }
    }
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[c * 4 + r] = (float)a[r][4 + c];
    return true;
}

void camera_stereo_eye(const float eye[3], const float target[3], const float up[3], float fov_y_deg, float aspect,
                       float z_near, float z_far, float offset, float shear, float view[16], float proj[16])
{
    // the mono camera's axes; the eye sits `offset` along the right axis and looks the same way (parallel axes)
    float c2w[16], ip[16];
    camera_look_at(eye, target, up, fov_y_deg, aspect, z_near, z_far, c2w, ip);
    const float e[3] = {eye[0] + c2w[0] * offset, eye[1] + c2w[1] * offset, eye[2] + c2w[2] * offset};
    const float t[3] = {target[0] + c2w[0] * offset, target[1] + c2w[1] * offset, target[2] + c2w[2] * offset};
    PtCameraHistory h;
    camera_history_entry(e, t, up, fov_y_deg, aspect, z_near, z_far, &h);
    std::memcpy(view, h.world_to_camera, sizeof h.world_to_camera);
    std::memcpy(proj, h.camera_projection, sizeof h.camera_projection);
    proj[8] = shear;                                                   // (0,2): the frustum's sideways shift
}

float frame_seed(uint32_t k)
{
    double v = 0.5 + (double)k * 0.6180339;
    return (float)(v - std::floor(v));
}

// ------------------------------------------------------------------ scenes

int Scene::register_object(const MeshBuilder &mesh, const float local_to_world[16])
{
    // RebuildMeshObjectBuffers (RayTracingMaster.cs:387-408): append vertices, append indices offset by
    // the first vertex, record (matrix, first index, index count).
    int first_vertex = (int)(vertices.size() / 3);
    vertices.insert(vertices.end(), mesh.vertices.begin(), mesh.vertices.end());
    int first_index = (int)indices.size();
    for (int idx : mesh.indices) indices.push_back(idx + first_vertex);
    PtMeshObject mo;
    std::memcpy(mo.local_to_world, local_to_world, sizeof mo.local_to_world);
    mo.indices_offset = first_index; mo.indices_count = (int)mesh.indices.size();
    objects.push_back(mo);
    return (int)objects.size() - 1;
}

int Scene::add_instance(int object, const float local_to_world[16])
{
    PtMeshObject mo = objects[object];
    std::memcpy(mo.local_to_world, local_to_world, sizeof mo.local_to_world);
    objects.push_back(mo);
    return (int)objects.size() - 1;
}

static void build_hall(MeshBuilder &mb)
{
    const float len = 60.0f, half_w = 9.0f, col_h = 9.0f;
    const int n_cols = 10;
    for (int side = -1; side <= 1; side += 2) {
        float z = side * 5.0f;
        for (int i = 0; i < n_cols; i++) {
            float x = -len / 2 + (i + 0.5f) * len / n_cols;
            mb.add_column(x, z, 0.4f, col_h, 0.55f, 72, 32);
            float lo[3] = {x - 0.8f, 0.0f, z - 0.8f}, hi[3] = {x + 0.8f, 0.4f, z + 0.8f};
            mb.add_box(lo, hi);
            float lo2[3] = {x - 0.75f, col_h, z - 0.75f}, hi2[3] = {x + 0.75f, col_h + 0.35f, z + 0.75f};
            mb.add_box(lo2, hi2);
            if (i + 1 < n_cols) mb.add_arch(x, x + len / n_cols, col_h + 0.35f, z, 0.3f, 48, 16);
        }
        mb.add_wall(-len / 2, len / 2, 0.0f, 14.0f, side * half_w, (float)-side, 160, 40, 0.12f, 77u + (uint32_t)side);
    }
    // statues along the nave
    for (int i = 0; i < 4; i++) {
        float c[3] = {-len / 2 + (i + 0.5f) * len / 4, 2.2f, 0.0f};
        mb.add_blob(c, 1.6f, 128, 112, 0.25f, 2.5f, 1000u + (uint32_t)i);
        float lo[3] = {c[0] - 1.2f, 0.0f, -1.2f}, hi[3] = {c[0] + 1.2f, 0.7f, 1.2f};
        mb.add_box(lo, hi);
    }
    // floor tiles: thin raised slabs, large flat triangles
    for (int i = 0; i < 30; i++)
        for (int j = 0; j < 8; j++) {
            float x = -len / 2 + i * 2.0f, z = -8.0f + j * 2.0f;
            float lo[3] = {x + 0.03f, -0.02f, z + 0.03f}, hi[3] = {x + 1.97f, 0.02f + 0.01f * ((i + j) & 1), z + 1.97f};
            mb.add_box(lo, hi);
        }
}

Scene *Scene::create(const std::string &name)
{
    Scene *s = new Scene();
    s->name = name;
    if (name == "spheres") {
        sphere_field(207407, 100, 3.0f, 8.0f, 100.0f, s->spheres);
        float e[3] = {0, 22, -125}, t[3] = {0, 4, 0};
        std::memcpy(s->eye, e, 12); std::memcpy(s->target, t, 12); s->fov = 60.0f;
    } else if (name == "xr") {
        sphere_field(207407, 200, 0.5f, 3.0f, 20.0f, s->spheres);
        MeshBuilder cube;
        float lo[3] = {-0.5f, -0.5f, -0.5f}, hi[3] = {0.5f, 0.5f, 0.5f};
        cube.add_box(lo, hi);
        float m[16]; matrix_trs(m, 0.0f, 3.0f, 6.0f, 33.0f, 2.0f);
        s->register_object(cube, m);
        float e[3] = {0, 4.6f, -14}, t[3] = {0, 2.0f, 4};
        std::memcpy(s->eye, e, 12); std::memcpy(s->target, t, 12); s->fov = 90.0f;
    } else if (name == "fixture") {
        sphere_field(12345, 40, 0.6f, 1.6f, 9.0f, s->spheres);
        if (s->spheres.size() > 20) s->spheres.resize(20);
        MeshBuilder mb;
        float c[3] = {0, 0, 0};
        mb.add_blob(c, 1.5f, 16, 11, 0.3f, 1.7f, 5u);
        float m[16]; matrix_trs(m, 0.5f, 2.4f, 1.0f, 20.0f, 1.0f);
        s->register_object(mb, m);
        float e[3] = {1, 5, -13}, t[3] = {0, 1.5f, 0};
        std::memcpy(s->eye, e, 12); std::memcpy(s->target, t, 12); s->fov = 60.0f;
    } else if (name == "blob70k") {
        sphere_field(424242, 60, 0.8f, 2.5f, 22.0f, s->spheres);
        // keep the centre clear for the mesh
        s->spheres.erase(std::remove_if(s->spheres.begin(), s->spheres.end(), [](const PtSphere &sp) {
            return sp.position[0] * sp.position[0] + sp.position[2] * sp.position[2] < 8.0f * 8.0f; }), s->spheres.end());
        MeshBuilder mb;
        float c[3] = {0, 0, 0};
        mb.add_blob(c, 4.0f, 188, 187, 0.35f, 1.9f, 9u);
        float m[16]; matrix_trs(m, 0.0f, 4.4f, 0.0f, 15.0f, 1.0f);
        s->register_object(mb, m);
        float e[3] = {3, 8, -19}, t[3] = {0, 3.8f, 0};
        std::memcpy(s->eye, e, 12); std::memcpy(s->target, t, 12); s->fov = 60.0f;
    } else if (name == "hall260k" || name == "city10m") {
        MeshBuilder mb;
        build_hall(mb);
        if (name == "hall260k") {
            sphere_field(77, 30, 0.3f, 0.9f, 4.0f, s->spheres);
            s->register_object(mb, kIdentity);
            float e[3] = {-27, 4.2f, 1.5f}, t[3] = {10, 3.5f, 0};
            std::memcpy(s->eye, e, 12); std::memcpy(s->target, t, 12); s->fov = 60.0f;
        } else {
            sphere_field(77, 200, 0.5f, 3.0f, 150.0f, s->spheres);
            Pcg32 rng(99);
            int first = -1;
            for (int cell = 0; cell < 38; cell++) {      // 38 x ~262 k triangles
                int gx = cell % 7, gz = cell / 7;
                float m[16];
                matrix_trs(m, (gx - 3) * 75.0f + rng.range(-4, 4), 0.0f, (gz - 2.5f) * 40.0f + rng.range(-4, 4),
                           rng.range(-20, 20), rng.range(0.9f, 1.1f));
                if (first < 0) first = s->register_object(mb, m);
                else s->add_instance(first, m);
            }
            float e[3] = {-40, 30, -170}, t[3] = {0, 0, 0};
            std::memcpy(s->eye, e, 12); std::memcpy(s->target, t, 12); s->fov = 60.0f;
        }
    } else {
        delete s;
        return nullptr;
    }
    return s;
}

int Scene::add_obj(const std::string &path, const float local_to_world[16])
{
    // Wavefront OBJ: "v x y z" positions and "f ..." faces (index forms i, i/t, i/t/n, i//n; negative = relative to the
    // end; polygons fan-triangulated); every "o" / "g" group with faces becomes one mesh object - the unit Unity's importer
    // turns into a MeshFilter and RayTracingObject registers (RayTracingObject.cs) - with its own compacted vertex list,
    // registered like RebuildMeshObjectBuffers does (RayTracingMaster.cs:387-408).  Returns the index of the first object
    // added, or a negative number: -1 unreadable file, -2 face index out of range, -3 no faces.
    std::ifstream in(path);
    if (!in) return -1;
    std::vector<float> pos;                       // every "v" of the file
    struct Group { std::vector<int> faces; };     // triangles as global vertex indices
    std::vector<Group> groups(1);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        ls >> tag;
        if (tag == "v") {
            float x, y, z;
            if (ls >> x >> y >> z) { pos.push_back(x); pos.push_back(y); pos.push_back(z); }
        } else if (tag == "o" || tag == "g") {
            if (!groups.back().faces.empty()) groups.emplace_back();
        } else if (tag == "f") {
            std::vector<int> poly;
            std::string tok;
            const int nv = (int)(pos.size() / 3);
            while (ls >> tok) {
                int idx = std::atoi(tok.c_str());
                if (idx < 0) idx = nv + idx; else idx -= 1;
                if (idx < 0 || idx >= nv) return -2;
                poly.push_back(idx);
            }
            for (size_t k = 1; k + 1 < poly.size(); k++) {
                groups.back().faces.push_back(poly[0]); groups.back().faces.push_back(poly[k]); groups.back().faces.push_back(poly[k + 1]);
            }
        }
    }
    int first = -3;
    for (const Group &g : groups) {
        if (g.faces.empty()) continue;
        MeshBuilder mb;
        std::vector<int> local(pos.size() / 3, -1);           // the group's vertices, in file order
        for (int gi : g.faces) local[(size_t)gi] = 0;
        for (size_t v = 0; v < local.size(); v++)
            if (local[v] == 0) local[v] = mb.add_vertex(pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]);
        for (int gi : g.faces) mb.indices.push_back(local[(size_t)gi]);
        const int idx = register_object(mb, local_to_world);
        if (first < 0) first = idx;
    }
    return first;
}

}  // namespace ptm
