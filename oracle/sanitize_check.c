/* Memory-safety check of the oracle (test infrastructure): every renderMode over a short moving-camera sequence on a
 * small generated scene, built together with pt_oracle.c under AddressSanitizer and UBSan:  make -C oracle sanitize  */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pt_oracle.h"

static void look_at(const float e[3], const float t[3], float fov_deg, float aspect, PtCameraHistory *h)
{
    float f[3] = {t[0] - e[0], t[1] - e[1], t[2] - e[2]};
    float l = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (int k = 0; k < 3; k++) f[k] /= l;
    float r[3] = {f[1] * 0 - f[2] * 1, f[2] * 0 - f[0] * 0, f[0] * 1 - f[1] * 0};
    l = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int k = 0; k < 3; k++) r[k] /= l;
    float u[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};
    memset(h, 0, sizeof *h);
    float *m = h->camera_to_world, *w = h->world_to_camera;
    for (int k = 0; k < 3; k++) { m[k] = r[k]; m[4 + k] = u[k]; m[8 + k] = -f[k]; m[12 + k] = e[k]; }
    m[15] = 1;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) w[b * 4 + a] = m[a * 4 + b];
    for (int a = 0; a < 3; a++) w[12 + a] = -(m[a * 4] * e[0] + m[a * 4 + 1] * e[1] + m[a * 4 + 2] * e[2]);
    w[15] = 1;
    float tf = tanf(fov_deg * 3.14159265f / 360.0f), zn = 0.1f, zf = 100.0f;
    float *q = h->camera_projection, *ip = h->camera_inverse_projection;
    q[0] = 1 / (aspect * tf); q[5] = 1 / tf; q[10] = -(zf + zn) / (zf - zn); q[11] = -1; q[14] = -2 * zf * zn / (zf - zn);
    ip[0] = aspect * tf; ip[5] = tf; ip[11] = (zn - zf) / (2 * zf * zn); ip[14] = -1; ip[15] = (zf + zn) / (2 * zf * zn);
}

int main(void)
{
    enum { W = 45, H = 29, NS = 6, NT = 40 };
    PtSphere sph[NS];
    memset(sph, 0, sizeof sph);
    for (int i = 0; i < NS; i++) {
        sph[i].radius = 0.5f + 0.2f * i;
        sph[i].position[0] = -4.0f + 1.7f * i; sph[i].position[1] = sph[i].radius; sph[i].position[2] = (float)(i % 3);
        sph[i].albedo[0] = 0.8f; sph[i].albedo[1] = 0.3f; sph[i].albedo[2] = 0.2f;
        sph[i].specular[0] = sph[i].specular[1] = sph[i].specular[2] = i % 2 ? 0.6f : 0.04f;
        sph[i].smoothness = 0.2f * i;
        if (i == 3) sph[i].emission[1] = 4.0f;
    }
    float verts[NT * 9]; int32_t idx[NT * 3];
    for (int k = 0; k < NT; k++) {
        float x = -5.0f + 0.25f * k, z = 3.0f + 0.05f * k;
        float tri[9] = {x, 0.2f, z, x + 0.1f, 2.5f, z, x + 0.6f, 0.2f, z + 0.3f};
        memcpy(verts + 9 * k, tri, sizeof tri);
        idx[3 * k] = 3 * k; idx[3 * k + 1] = 3 * k + (k & 1 ? 2 : 1); idx[3 * k + 2] = 3 * k + (k & 1 ? 1 : 2);
    }
    PtMeshObject obj[2];
    memset(obj, 0, sizeof obj);
    for (int o = 0; o < 2; o++) {
        for (int d = 0; d < 4; d++) obj[o].local_to_world[5 * d] = 1;
        obj[o].local_to_world[13] = 0.5f * o;
        obj[o].indices_offset = o * (NT / 2) * 3; obj[o].indices_count = (NT / 2) * 3;
    }
    float sky[8 * 4 * 4];
    for (int i = 0; i < 8 * 4 * 4; i++) sky[i] = 0.2f + 0.05f * (i % 7);
    PtMaterial mats[2];
    memset(mats, 0, sizeof mats);
    mats[0].lobe = PT_LOBE_GGX; mats[0].smoothness = 0.7f; mats[0].specular[0] = mats[0].specular[1] = mats[0].specular[2] = 0.8f;
    mats[1].albedo[0] = 0.5f; mats[1].emission[2] = 0.3f;
    int failures = 0;
    for (int accel = 0; accel < 2; accel++)
        for (int with_mat = 0; with_mat < 2; with_mat++) {
            PtoScene sc = {sph, NS, obj, 2, verts, NT * 3, idx, NT * 3, sky, 8, 4, with_mat ? mats : NULL, with_mat ? 2 : 0};
            PtoPrepared *p = pto_prepare(&sc, accel);
            if (!p) { fprintf(stderr, "pto_prepare failed\n"); return 1; }
            const int modes[] = {1, 2, 0, 3, 4, 10, 5, 6, 7, 8, 11, 12};
            for (size_t mi = 0; mi < sizeof modes / sizeof modes[0]; mi++) {
                float *res = calloc((size_t)W * H * 4, 4), *res1 = calloc((size_t)W * H * 4, 4), *res2 = calloc((size_t)W * H * 4, 4);
                int32_t *ids = malloc((size_t)W * H * 4);
                PtCameraHistory *hist = calloc(PT_MAX_CAMERA_HISTORY, sizeof *hist);
                int n_hist = 0;
                for (int k = 0; k < 7; k++) {
                    float eye[3] = {0.3f * k, 2.0f + 0.1f * k, -9.0f + 0.2f * k}, tgt[3] = {0, 1, 2};
                    PtCameraHistory cam;
                    look_at(eye, tgt, 60.0f, (float)W / H, &cam);
                    PtFrameParams prm;
                    memset(&prm, 0, sizeof prm);
                    memcpy(prm.camera_to_world, cam.camera_to_world, sizeof prm.camera_to_world);
                    memcpy(prm.camera_inverse_projection, cam.camera_inverse_projection, sizeof prm.camera_inverse_projection);
                    prm.seed = 0.1f + 0.13f * k; prm.samples_per_pixel = modes[mi] >= 10 ? 1 : 2; prm.max_bounces = 4;
                    prm.render_mode = modes[mi]; prm.accumulation_frames = 5; prm.divisions = 3; prm.counter = k % 3; prm.left_eye = k & 1;
                    prm.depth = k == 5;
                    PtoStats st;
                    if (pto_render_history(p, &prm, hist, n_hist, W, H, res, modes[mi] == 1 ? NULL : res1, res2, ids, &st, 2)) {
                        fprintf(stderr, "mode %d frame %d: bad arguments\n", modes[mi], k); failures++;
                    }
                    if (modes[mi] >= 5 && modes[mi] <= 8) { if (k % 3 == 0) n_hist = 0; hist[n_hist++] = cam; }
                    else { hist[0] = cam; n_hist = 1; }
                }
                double sum = 0;
                for (size_t i = 0; i < (size_t)W * H * 4; i++) if (isfinite(res[i])) sum += res[i];
                printf("accel %d materials %d mode %2d: sum of the last frame %.4f\n", accel, with_mat, modes[mi], sum);
                free(res); free(res1); free(res2); free(ids); free(hist);
            }
            /* the screen-space pieces on their own */
            float *img = calloc((size_t)W * H * 4, 4), out4[4];
            for (size_t i = 0; i < (size_t)W * H * 4; i++) img[i] = 0.01f * (float)(i % 97);
            for (int y = -2; y < H + 2; y += 3)
                for (int x = -2; x < W + 2; x += 3) {
                    if (x >= 0 && y >= 0 && x < W && y < H) { pto_denoiser(img, W, H, x, y, 1.5f, out4); pto_taa(img, img, W, H, x, y, out4); }
                    pto_bilinear_sample(img, W, H, (float)x + 0.3f, (float)y - 0.4f, out4);
                }
            float *disp = calloc((size_t)W * H * 4, 4);
            pto_display_denoise(img, W, H, disp, 2);
            free(img); free(disp);
            pto_release(p);
        }
    printf(failures ? "FAILED\n" : "sanitize_check ok\n");
    return failures != 0;
}
