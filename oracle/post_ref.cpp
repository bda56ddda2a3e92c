/* post_ref.cpp -- CPU restatement of the reference's output stage (TEST INFRASTRUCTURE, see tb_oracle.h: only tests/,
 * smoke() and bench.py's cpu_baseline leg may use anything under oracle/).
 *
 * Follows, in the reference's own order:
 *   TracerBoy/GenerateHistogramCS.hlsl:19-52           luminance -> one of 256 log2 bins
 *   TracerBoy/CalculateAveragedLuminanceCS.hlsl:15-34  integer weighted mean of the bin indices -> averaged luminance
 *   TracerBoy/PostProcessCS.hlsl:23-195                per output type: divide by the sample weight, exposure, tonemap, gamma
 *   TracerBoy/Tonemap.h:12-211                         the eight tonemappers
 *   TracerBoy/TracerBoy.cpp:2948-3030                  MinLogLuminance -10, LogLuminanceRange 16, PixelCount = W*H
 * fp32 throughout, one rounding per written operation (-ffp-contract=off), transcendental functions from tb_math.h --
 * the same arithmetic contract as the path tracer.  Parity unpinned by the reference (it has no tests and
 * its shaders cannot be compiled here); pinned by the closed-form known answers in tests/test_post_process.py. */
#include "tb_oracle.h"
#include "../include/tb_math.h"
#include "../include/tb_vec.h"
#include <cstring>
#include <vector>

namespace {

typedef float (*Curve)(float);

float ColorToLuma(const float c[3]) { return (c[0] * 0.212671f + c[1] * 0.715160f) + c[2] * 0.072169f; } /* Tonemap.h:12-15 */
float GammaCorrect1(float c) { return tb_pow(c, 1.0f / 2.2f); }                                          /* :153-156 */
void GammaCorrect(float c[3]) { for (int k = 0; k < 3; k++) c[k] = GammaCorrect1(c[k]); }
float lerp(float a, float b, float t) { return a + t * (b - a); }

void mulMatVec(const float m[3][3], float v[3]) /* mul(M, v) */
{
    float r[3];
    for (int i = 0; i < 3; i++) r[i] = (m[i][0] * v[0] + m[i][1] * v[1]) + m[i][2] * v[2];
    memcpy(v, r, sizeof r);
}
void mulVecMat(float v[3], const float m[3][3]) /* mul(v, M) */
{
    float r[3];
    for (int j = 0; j < 3; j++) r[j] = (v[0] * m[0][j] + v[1] * m[1][j]) + v[2] * m[2][j];
    memcpy(v, r, sizeof r);
}

float RRTAndODTFit(float v) /* :35-40 */
{
    float a = v * (v + 0.0245786f) - 0.000090537f;
    float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}
void ACESFitted(float c[3]) /* :42-55 */
{
    static const float ACESInputMat[3][3] = {{0.59719f, 0.35458f, 0.04823f}, {0.07600f, 0.90834f, 0.01566f}, {0.02840f, 0.13383f, 0.83777f}};
    static const float ACESOutputMat[3][3] = {{1.60475f, -0.53108f, -0.07367f}, {-0.10208f, 1.10813f, -0.00605f}, {-0.00327f, -0.07276f, 1.07602f}};
    mulMatVec(ACESInputMat, c);
    for (int k = 0; k < 3; k++) c[k] = RRTAndODTFit(c[k]);
    mulMatVec(ACESOutputMat, c);
    for (int k = 0; k < 3; k++) c[k] = tb_saturate(c[k]);
}

float uncharted2_tonemap_partial(float x) /* :64-73 */
{
    float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
float uncharted2_filmic(float v) /* :76-84 */
{
    float exposure_bias = 2.0f;
    float curr = uncharted2_tonemap_partial(v * exposure_bias);
    float white_scale = 1.0f / uncharted2_tonemap_partial(11.2f);
    return curr * white_scale;
}

void CommerceToneMapping(float c[3]) /* :87-104 */
{
    float startCompression = 0.8f - 0.04f;
    float desaturation = 0.15f;
    float x = tb_min(c[0], tb_min(c[1], c[2]));
    float offset = x < 0.08f ? x - (6.25f * x) * x : 0.04f;
    for (int k = 0; k < 3; k++) c[k] = c[k] - offset;
    float peak = tb_max(c[0], tb_max(c[1], c[2]));
    if (peak < startCompression) return;
    float d = 1.0f - startCompression;
    float newPeak = 1.0f - (d * d) / ((peak + d) - startCompression);
    float scale = newPeak / peak;
    for (int k = 0; k < 3; k++) c[k] = c[k] * scale;
    float g = 1.0f - 1.0f / (desaturation * (peak - newPeak) + 1.0f);
    for (int k = 0; k < 3; k++) c[k] = lerp(c[k], newPeak * 1.0f, g);
}

float agxDefaultContrastApproximation(float x) /* :106-110 */
{
    float x2 = x * x;
    float x4 = x2 * x2;
    return ((((((15.5f * x4) * x2 - (40.14f * x4) * x) + 31.96f * x4) - (6.868f * x2) * x) + 0.4298f * x2) + 0.1191f * x) - 0.00232f;
}
void agx(float c[3]) /* :112-126 */
{
    static const float agxTransform[3][3] = {{0.842479062253094f, 0.0423282422610123f, 0.0423756549057051f},
                                             {0.0784335999999992f, 0.878468636469772f, 0.0784336f},
                                             {0.0792237451477643f, 0.0791661274605434f, 0.879142973793104f}};
    const float minEv = -12.47393f, maxEv = 4.026069f;
    mulVecMat(c, agxTransform);
    for (int k = 0; k < 3; k++) {
        float v = tb_clamp(tb_log2(c[k]), minEv, maxEv);
        v = (v - minEv) / (maxEv - minEv);
        c[k] = agxDefaultContrastApproximation(v);
    }
}
void agxLook(float val[3], bool punchy) /* :129-151 */
{
    float luma = (val[0] * 0.2126f + val[1] * 0.7152f) + val[2] * 0.0722f;
    float offset = 0.0f, slope = 1.0f, power = 1.0f, sat = 1.0f;
    if (punchy) { slope = 1.0f; power = 1.35f; sat = 1.4f; }
    for (int k = 0; k < 3; k++) {
        float v = tb_pow(val[k] * slope + offset, power); /* ASC CDL */
        val[k] = luma + sat * (v - luma);
    }
}

float GTTonemap(float x) /* :158-176 */
{
    float m = 0.22f, a = 1.0f, c = 1.33f, P = 1.0f, l = 0.4f;
    float l0 = ((P - m) * l) / a;
    float S0 = m + l0;
    float S1 = m + a * l0;
    float C2 = (a * P) / (P - S1);
    float L = m + a * (x - m);
    float T = m * tb_pow(x / m, c);
    float S = P - (P - S1) * tb_exp((-C2 * (x - S0)) / P);
    float t = tb_saturate((x - 0.0f) / (m - 0.0f)); /* smoothstep(0, m, x) */
    float w0 = 1.0f - (t * t) * (3.0f - 2.0f * t);
    float w2 = (x < m + l) ? 0.0f : 1.0f;
    float w1 = (1.0f - w0) - w2;
    return (T * w0 + L * w1) + S * w2;
}

void Tonemap(uint32_t type, float c[3]) /* :178-204 */
{
    switch (type) {
    case TB_TONEMAP_REINHARD: for (int k = 0; k < 3; k++) c[k] = c[k] / (1.0f + c[k]); GammaCorrect(c); return;
    case TB_TONEMAP_GT: for (int k = 0; k < 3; k++) c[k] = GTTonemap(c[k]); GammaCorrect(c); return;
    case TB_TONEMAP_ACES: ACESFitted(c); GammaCorrect(c); return;
    case TB_TONEMAP_UNCHARTED: for (int k = 0; k < 3; k++) c[k] = uncharted2_filmic(c[k]); GammaCorrect(c); return;
    case TB_TONEMAP_KHRONOS_PBR_NEUTRAL: CommerceToneMapping(c); GammaCorrect(c); return;
    case TB_TONEMAP_AGX: agx(c); agxLook(c, false); return;
    case TB_TONEMAP_AGX_PUNCHY: agx(c); agxLook(c, true); return;
    default: for (int k = 0; k < 3; k++) c[k] = tb_saturate(c[k]); GammaCorrect(c); return;
    }
}

uint32_t f2u(float f) { return f >= 4294967296.0f ? 0xffffffffu : (f > 0.0f ? (uint32_t)f : 0u); } /* HLSL float -> uint: saturating, NaN -> 0 */

uint32_t LuminanceToHistogramIndex(float luminance, float minLogLuminance, float oneOverLogLuminanceRange) /* GenerateHistogramCS.hlsl:19-31 */
{
    const float epsilon = 0.00001f;
    if (luminance < epsilon) return 0;
    float logLuminance = tb_saturate((tb_log2(luminance) - minLogLuminance) * oneOverLogLuminanceRange);
    return f2u(logLuminance * 254.0f + 1.0f);
}

} // namespace

extern "C" void tbo_post_process(const TbPostConstants* pc, const float* in, int inIsR32, float* outRgba, uint8_t* outRgba8, float* averagedOut,
    uint32_t* histogramOut)
{
    const uint32_t W = pc->W, H = pc->H;
    const size_t n = (size_t)W * H;
    float averagedLuminance = 0.0f;
    if (pc->UseAutoExposure && pc->OutputType == TB_OUTPUT_TYPE_LIT) {
        const float MinLogLuminance = -10.0f, LogLuminanceRange = 16.0f; /* TracerBoy.cpp:2950-2951 */
        std::vector<uint32_t> hist(256, 0);
        for (size_t i = 0; i < n; i++) {
            const float* a = in + i * 4;
            float color[3] = {a[0] / a[3], a[1] / a[3], a[2] / a[3]};
            hist[LuminanceToHistogramIndex(ColorToLuma(color), MinLogLuminance, 1.0f / LogLuminanceRange)]++;
        }
        uint32_t AveragedHistogramCount = 0;
        for (uint32_t b = 0; b < 256; b++) AveragedHistogramCount += hist[b] * b;
        const uint32_t BinCount = hist[0]; /* the first thread's bin, CalculateAveragedLuminanceCS.hlsl:24-31 */
        const uint32_t den = W * H - BinCount;
        const uint32_t q = den ? AveragedHistogramCount / den : 0xffffffffu;
        float averagedLogLuminance = ((float)q - 1.0f) / 254.0f;
        averagedLuminance = tb_exp2(averagedLogLuminance * LogLuminanceRange + MinLogLuminance);
        if (histogramOut) memcpy(histogramOut, hist.data(), 256 * 4);
    }
    if (averagedOut) *averagedOut = averagedLuminance;
    for (size_t i = 0; i < n; i++) {
        float color[4];
        if (inIsR32) { color[0] = in[i]; color[1] = 0.0f; color[2] = 0.0f; color[3] = 1.0f; } else memcpy(color, in + i * 4, 16);
        float o[3];
        switch (pc->OutputType) {
        default: { /* ProcessLit :23-47 */
            float FrameCount = color[3];
            for (int k = 0; k < 3; k++) o[k] = color[k] / FrameCount;
            float Exposure;
            if (pc->UseAutoExposure) { float LinearGray = tb_pow(0.5f, 2.2f); Exposure = LinearGray / averagedLuminance; }
            else Exposure = pc->ExposureMultiplier;
            for (int k = 0; k < 3; k++) o[k] = o[k] * Exposure;
            Tonemap(pc->TonemapType, o);
            break; }
        case TB_OUTPUT_TYPE_ALBEDO: /* ProcessAlbedo :64-76 */
            for (int k = 0; k < 3; k++) o[k] = color[k] * pc->ExposureMultiplier;
            Tonemap(pc->TonemapType, o);
            if (pc->UseGammaCorrection) GammaCorrect(o);
            break;
        case TB_OUTPUT_TYPE_NORMAL: { /* ProcessNormal :78-83 */
            uint32_t frameCount = f2u(color[3]);
            if (frameCount > 0) {
                tb3 v = tb3_normalize(tb3_make(color[0] / (float)frameCount, color[1] / (float)frameCount, color[2] / (float)frameCount));
                o[0] = tb_abs(v.x); o[1] = tb_abs(v.y); o[2] = tb_abs(v.z);
            } else o[0] = o[1] = o[2] = 0.0f;
            break; }
        case TB_OUTPUT_TYPE_DEPTH: case TB_OUTPUT_TYPE_LIVE_PIXELS: /* PassThroughColor :107-114 */
            for (int k = 0; k < 3; k++) o[k] = color[k] * pc->ExposureMultiplier;
            Tonemap(pc->TonemapType, o);
            break;
        case TB_OUTPUT_TYPE_LUMINANCE: { /* ProcessLuminance :49-62 */
            float FrameCount = color[3];
            for (int k = 0; k < 3; k++) o[k] = (color[k] / FrameCount) * pc->ExposureMultiplier;
            Tonemap(pc->TonemapType, o);
            float l = ColorToLuma(o);
            o[0] = o[1] = o[2] = l;
            if (pc->UseGammaCorrection) GammaCorrect(o);
            break; }
        case TB_OUTPUT_TYPE_HEATMAP: { /* ProcessHeatmap :127-140 */
            uint32_t TotalTests = f2u(color[0]) + f2u(color[1]);
            float lerpValue = (float)TotalTests / 100.0f;
            const float c0[3] = {0, 1, 0}, c1[3] = {1, 1, 0}, c2[3] = {1, 0, 0};
            for (int k = 0; k < 3; k++) /* Lerp3 :115-125 */
                o[k] = lerpValue < 0.5f ? lerp(c0[k], c1[k], lerpValue * 2.0f) : lerp(c1[k], c2[k], (lerpValue - 0.5f) * 2.0f);
            for (int k = 0; k < 3; k++) o[k] = o[k] * pc->ExposureMultiplier;
            Tonemap(pc->TonemapType, o);
            break; }
        }
        if (outRgba) { outRgba[i * 4] = o[0]; outRgba[i * 4 + 1] = o[1]; outRgba[i * 4 + 2] = o[2]; outRgba[i * 4 + 3] = 1.0f; }
        if (outRgba8) { /* R8G8B8A8_UNORM conversion of the back buffer: saturate, scale, add 0.5, truncate */
            for (int k = 0; k < 3; k++) outRgba8[i * 4 + k] = (uint8_t)f2u(tb_saturate(o[k]) * 255.0f + 0.5f);
            outRgba8[i * 4 + 3] = 255;
        }
    }
}
