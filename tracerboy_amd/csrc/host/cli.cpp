/* cli.cpp -- headless replacement of the reference's Win32 shell (WinMain/WinMain.cpp, D3D12App.cpp):
 *   tracerboy-hip scene.pbrt [--width W] [--height H] [--spp N] [--depth D] [--seed-time T] [--device I]
 *                 [--builder lbvh|sah|lbvh-gpu|treelets|treelets-gpu] [--blue-noise 0|1] [--tonemap 0..7] [--exposure E|auto]
 *                 [--out frame.png|frame.pfm|frame.exr]
 * Uses only the C ABI (include/tracerboy_hip.h), the way an embedding application would.
 * Output by extension: .png = what the reference presents (auto exposure + PostProcessCS tonemap, 8-bit back buffer,
 * tb_post_process); .pfm = linear radiance sum(rgb*w)/sum(w), the value PostProcessCS divides out before tonemapping
 * (PostProcessCS.hlsl:23-47), RGB float32, bottom row first; .exr = the same radiance as OpenEXR (RGBA float32, A = 1). */
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
    if (argc < 2) { fprintf(stderr, "usage: tracerboy-hip scene.pbrt [--width W --height H --spp N --depth D --seed-time T --device I --builder lbvh|sah|lbvh-gpu|treelets|treelets-gpu --blue-noise 0|1 --tonemap 0..7 --exposure E|auto --out f.png|f.pfm|f.exr]\n"); return 2; }
    std::string scene = argv[1], out = "frame.png";
    tb_post_settings post; tb_default_post_settings(&post);
    uint32_t W = 0, H = 0, spp = 64; int depth = -1, device = 0, builder = 0, blue = -1; float t = 0.0f;
    for (int i = 2; i + 1 < argc; i += 2) {
        std::string k = argv[i]; const char* v = argv[i + 1];
        if (k == "--width") W = (uint32_t)atoi(v); else if (k == "--height") H = (uint32_t)atoi(v); else if (k == "--spp") spp = (uint32_t)atoi(v);
        else if (k == "--depth") depth = atoi(v); else if (k == "--seed-time") t = (float)atof(v); else if (k == "--device") device = atoi(v);
        else if (k == "--builder") builder = !strcmp(v, "sah") ? 1 : !strcmp(v, "lbvh-gpu") ? 2 : !strcmp(v, "treelets") ? 3 : !strcmp(v, "treelets-gpu") ? 4 : 0; /* tb_set_option "bvh_builder" */ else if (k == "--blue-noise") blue = atoi(v); else if (k == "--out") out = v;
        else if (k == "--tonemap") post.TonemapType = (uint32_t)atoi(v);
        else if (k == "--exposure") { if (!strcmp(v, "auto")) post.EnableAutoExposure = 1; else { post.EnableAutoExposure = 0; post.ExposureMultiplier = (float)atof(v); } }
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
    const bool png = out.size() >= 4 && out.compare(out.size() - 4, 4, ".png") == 0;
    if (png) {
        std::vector<uint8_t> img((size_t)W * H * 4);
        if ((rc = tb_post_process(ctx, &post, TB_OUTPUT_TYPE_LIT, nullptr, img.data()))) return fail(ctx, "tb_post_process", rc);
        if ((rc = tb_write_image_rgba8(out.c_str(), W, H, img.data()))) return fail(ctx, "tb_write_image_rgba8", rc);
    } else {
        std::vector<float> acc((size_t)W * H * 4);
        if ((rc = tb_read_accum(ctx, acc.data(), nullptr))) return fail(ctx, "tb_read_accum", rc);
        for (size_t i = 0; i < (size_t)W * H; i++) { float w = acc[4 * i + 3], inv = w > 0 ? 1.0f / w : 0.0f; acc[4 * i] *= inv; acc[4 * i + 1] *= inv; acc[4 * i + 2] *= inv; acc[4 * i + 3] = w > 0 ? 1.0f : 0.0f; }
        if ((rc = tb_write_image_f32(out.c_str(), W, H, acc.data()))) return fail(ctx, "tb_write_image_f32 (use .png, .pfm or .exr)", rc);
    }
    printf("%s: %u triangles, %ux%u x %u spp, depth %d: %.2f ms on the GPU (%.1f Msamples/s), scene load + BVH %.2f s -> %s\n",
           scene.c_str(), info.numTriangles, W, H, spp, s.MaxBounces, ms, (double)W * H * spp / (ms * 1e3), loadS, out.c_str());
    tb_destroy(ctx);
    return 0;
}
