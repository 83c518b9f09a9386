// ptrace_demo - a compiled host for libptrace with no Python anywhere: the C++ RayTracingMaster mirror renders a named
// scene for a number of frames, as the reference's Unity host would (RayTracingMaster.cs:704-747), and reports the
// device-counted rays, the time and a checksum of the image; optionally writes the image as a binary PPM.
//
//   ptrace_demo [--scene hall260k] [--size 1920x1080] [--spp 16] [--bounces 8] [--frames 5] [--mode 1]
//               [--accumulate 0|1|2|3] [--device 0] [--ppm out.ppm] [--sky picture.hdr] [--option name=value ...]
//
// --sky: a Radiance .hdr picture as SkyboxTexture instead of the procedural sky; --option: pt_set_option before the first
// frame (e.g. --option instancing=1 for a scene of many copies of one mesh, --option backend=2).
//
// The checksum is the XOR of the image's 32-bit words: equal to what bench.py and the tests compute from the same frames.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "ray_tracing_master.hpp"

static int fail(const char *what, const std::string &why)
{
    std::fprintf(stderr, "ptrace_demo: %s: %s\n", what, why.c_str());
    return 1;
}

int main(int argc, char **argv)
{
    std::string scene_name = "hall260k", ppm, sky;
    std::vector<std::pair<std::string, int>> options;
    int W = 1920, H = 1080, spp = 16, bounces = 8, frames = 5, mode = 1, device = 0, accumulate = PT_ACCUM_NONE;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--scene") scene_name = next();
        else if (a == "--size") { if (std::sscanf(next(), "%dx%d", &W, &H) != 2) return fail("--size", "expected WxH"); }
        else if (a == "--spp") spp = std::atoi(next());
        else if (a == "--bounces") bounces = std::atoi(next());
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--mode") mode = std::atoi(next());
        else if (a == "--accumulate") accumulate = std::atoi(next());
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--ppm") ppm = next();
        else if (a == "--sky") sky = next();
        else if (a == "--option") {
            const std::string o = next();
            const size_t eq = o.find('=');
            if (eq == std::string::npos) return fail("--option", "expected name=value");
            options.push_back({o.substr(0, eq), std::atoi(o.c_str() + eq + 1)});
        }
        else return fail("unknown argument", a);
    }
    std::unique_ptr<ptm::Scene> scene(ptm::Scene::create(scene_name));
    if (!scene) return fail("unknown scene", scene_name);

    ptm::RayTracingMaster master(device);
    if (!master.ok()) return fail("pt_create", master.error());
    master.RenderWidth = W; master.RenderHight = H;
    master.samplesPerPixel = spp; master.maxBounces = bounces;
    master.renderMode = mode; master.accumulate = accumulate; master.sampleFrames = 10;
    std::memcpy(master.eye, scene->eye, sizeof master.eye);
    std::memcpy(master.target, scene->target, sizeof master.target);
    master.fieldOfView = scene->fov;
    if (master.OnEnable(*scene, 2048, 1024)) return fail("OnEnable", master.error());
    if (!sky.empty() && master.SetSkyboxTexture(sky.c_str())) return fail("--sky", master.error());
    for (const auto &o : options)
        if (pt_set_option(master.context(), o.first.c_str(), o.second)) return fail("--option", pt_last_error(master.context()));

    if (master.Render()) return fail("Render", master.error());      // first frame: builds the acceleration structure
    PtStats st;
    if (pt_get_stats(master.context(), &st)) return fail("pt_get_stats", pt_last_error(master.context()));
    const float build_ms = st.bvh_build_ms;
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < frames; k++)
        if (master.Render()) return fail("Render", master.error());
    if (pt_get_stats(master.context(), &st)) return fail("pt_get_stats", pt_last_error(master.context()));   // waits for the renders
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    std::vector<float> img((size_t)W * H * 4);
    if (pt_read_result(master.context(), img.data())) return fail("pt_read_result", pt_last_error(master.context()));
    uint32_t crc = 0;
    double mean[3] = {0, 0, 0};
    for (size_t i = 0; i < img.size(); i++) {
        uint32_t w; std::memcpy(&w, &img[i], 4);
        crc ^= w;
        if ((i & 3) < 3) mean[i & 3] += img[i];
    }
    std::printf("{\"scene\": \"%s\", \"width\": %d, \"height\": %d, \"spp\": %d, \"bounces\": %d, \"render_mode\": %d, \"frames\": %d, "
                "\"ms_per_frame\": %.4f, \"mrays_per_s\": %.1f, \"rays\": %llu, \"bvh_nodes\": %d, \"bvh_build_ms\": %.1f, "
                "\"image_crc\": %u, \"image_mean_rgb\": [%.6f, %.6f, %.6f]}\n",
                scene_name.c_str(), W, H, spp, bounces, mode, frames, ms / std::max(1, frames),
                frames > 0 ? (double)st.rays / (ms * 1e-3) / 1e6 : 0.0, (unsigned long long)st.rays, st.bvh_nodes, build_ms, crc,
                mean[0] / ((double)W * H), mean[1] / ((double)W * H), mean[2] / ((double)W * H));
    if (!ppm.empty()) {
        FILE *f = std::fopen(ppm.c_str(), "wb");
        if (!f) return fail("--ppm", "cannot open " + ppm);
        std::fprintf(f, "P6\n%d %d\n255\n", W, H);
        for (int y = H - 1; y >= 0; y--)                               // row 0 is the bottom of the view
            for (int x = 0; x < W; x++)
                for (int c = 0; c < 3; c++) {
                    float v = img[((size_t)y * W + x) * 4 + c];
                    v = v / (1.0f + v);                                // simple tone curve for the HDR sky
                    std::fputc((int)(255.0f * std::sqrt(std::fmin(std::fmax(v, 0.0f), 1.0f)) + 0.5f), f);
                }
        std::fclose(f);
    }
    return 0;
}
