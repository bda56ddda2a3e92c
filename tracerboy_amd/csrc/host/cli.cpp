/* cli.cpp -- headless replacement of the reference's Win32 shell (WinMain/WinMain.cpp, D3D12App.cpp):
 *   tracerboy-hip scene.pbrt [--width W] [--height H] [--spp N] [--depth D] [--seed-time T] [--device I]
 *                 [--builder lbvh|sah] [--blue-noise 0|1] [--out frame.pfm]
 * Uses only the C ABI (include/tracerboy_hip.h), the way an embedding application would.
 * Output: PFM (RGB float32, bottom row first) of sum(rgb*w)/sum(w), i.e. the value PostProcessCS divides
 * out before tonemapping (PostProcessCS.hlsl:23-47). */
#include "../../../include/tracerboy_hip.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int fail(tb_context* c, const char* what, int rc)
{
    fprintf(stderr, "tracerboy-hip: %s failed (%d): %s\n", what, rc, tb_last_error(c));
    if (c) tb_destroy(c);
    return 1;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: tracerboy-hip scene.pbrt [--width W --height H --spp N --depth D --seed-time T --device I --builder lbvh|sah --blue-noise 0|1 --out f.pfm]\n"); return 2; }
    std::string scene = argv[1], out = "frame.pfm";
    uint32_t W = 0, H = 0, spp = 64; int depth = -1, device = 0, builder = 0, blue = -1; float t = 0.0f;
    for (int i = 2; i + 1 < argc; i += 2) {
        std::string k = argv[i]; const char* v = argv[i + 1];
        if (k == "--width") W = (uint32_t)atoi(v); else if (k == "--height") H = (uint32_t)atoi(v); else if (k == "--spp") spp = (uint32_t)atoi(v);
        else if (k == "--depth") depth = atoi(v); else if (k == "--seed-time") t = (float)atof(v); else if (k == "--device") device = atoi(v);
        else if (k == "--builder") builder = !strcmp(v, "sah") ? 1 : 0; else if (k == "--blue-noise") blue = atoi(v); else if (k == "--out") out = v;
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    tb_context* ctx = nullptr;
    int rc = tb_create(&ctx, device);
    if (rc) return fail(nullptr, "tb_create", rc);
    tb_set_option(ctx, "bvh_builder", builder);
    auto t0 = std::chrono::steady_clock::now();
    if ((rc = tb_load_scene(ctx, scene.c_str()))) return fail(ctx, "tb_load_scene", rc);
    double loadS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    tb_scene_info info; tb_scene_info_get(ctx, &info);
    if (!W) W = info.filmWidth ? info.filmWidth : 1920;
    if (!H) H = info.filmHeight ? info.filmHeight : 1080;
    tb_output_settings s; tb_default_output_settings(&s);
    if (depth >= 0) s.MaxBounces = depth;
    if (blue >= 0) s.EnableBlueNoise = (uint32_t)blue;
    if ((rc = tb_render(ctx, W, H, spp, &s, t))) return fail(ctx, "tb_render", rc);
    float ms = tb_last_render_ms(ctx);
    std::vector<float> acc((size_t)W * H * 4);
    if ((rc = tb_read_accum(ctx, acc.data(), nullptr))) return fail(ctx, "tb_read_accum", rc);
    FILE* f = fopen(out.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", out.c_str()); tb_destroy(ctx); return 1; }
    fprintf(f, "PF\n%u %u\n-1.0\n", W, H);
    std::vector<float> row((size_t)W * 3);
    for (uint32_t y = 0; y < H; y++) {
        const float* src = &acc[(size_t)(H - 1 - y) * W * 4];
        for (uint32_t x = 0; x < W; x++) { float w = src[4 * x + 3]; float inv = w > 0 ? 1.0f / w : 0.0f; row[3 * x] = src[4 * x] * inv; row[3 * x + 1] = src[4 * x + 1] * inv; row[3 * x + 2] = src[4 * x + 2] * inv; }
        fwrite(row.data(), 4, row.size(), f);
    }
    fclose(f);
    printf("%s: %u triangles, %ux%u x %u spp, depth %d: %.2f ms on the GPU (%.1f Msamples/s), scene load + BVH %.2f s -> %s\n",
           scene.c_str(), info.numTriangles, W, H, spp, s.MaxBounces, ms, (double)W * H * spp / (ms * 1e3), loadS, out.c_str());
    tb_destroy(ctx);
    return 0;
}
