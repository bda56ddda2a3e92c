/* post_kernels.hip -- the output stage: luminance histogram -> averaged luminance -> PostProcessCS.
 *
 *   post_histogram_kernel   GenerateHistogramCS.hlsl:1-52      256 bins of log2 luminance, LDS histogram per 16x16 group
 *   post_average_kernel     CalculateAveragedLuminanceCS.hlsl  one group, integer weighted mean of the bin indices
 *   post_process_kernel     PostProcessCS.hlsl:23-195 + Tonemap.h:12-211   divide by the sample weight, exposure, one of
 *                           eight tonemappers, gamma; writes RGBA32F and the 8-bit back-buffer value
 *
 * All three are element-wise over the frame: 16 B read (+16 B write) per pixel, HBM/launch-latency bound; nothing is
 * kept between pixels, so the layout is simply the accumulation surface's (row-major float4, coalesced 16-B lanes).
 * fp32 arithmetic is spelled out operation by operation (no contraction, tb_math.h primitives) and mirrored by
 * oracle/post_ref.cpp, which the parity tests compare bit for bit. */
#include <hip/hip_runtime.h>
#include "tb_math.h"
#include "tb_vec.h"
#include "tb_abi.h"
#include "pt_launch.h"

namespace {

struct P3 { float x, y, z; };
__device__ __forceinline__ P3 p3(float x, float y, float z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }

__device__ __forceinline__ float luma709(P3 c) { return (c.x * 0.212671f + c.y * 0.715160f) + c.z * 0.072169f; } /* ColorToLuma, Tonemap.h:12-15 */
__device__ __forceinline__ float gamma1(float c) { return tb_pow(c, 1.0f / 2.2f); }                                /* GammaCorrect :153-156 */
__device__ __forceinline__ P3 gamma3(P3 c) { return p3(gamma1(c.x), gamma1(c.y), gamma1(c.z)); }
__device__ __forceinline__ float lerp1(float a, float b, float t) { return a + t * (b - a); }

/* mul(M, v): rows dotted with the column vector (Tonemap.h:44,49) */
__device__ __forceinline__ P3 mul_rows(const float m[9], P3 v)
{
    return p3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z, (m[6] * v.x + m[7] * v.y) + m[8] * v.z);
}
/* mul(v, M): row vector times the matrix (Tonemap.h:122) */
__device__ __forceinline__ P3 mul_cols(P3 v, const float m[9])
{
    return p3((v.x * m[0] + v.y * m[3]) + v.z * m[6], (v.x * m[1] + v.y * m[4]) + v.z * m[7], (v.x * m[2] + v.y * m[5]) + v.z * m[8]);
}

__device__ __forceinline__ float rrt_odt(float v) /* RRTAndODTFit :35-40 */
{
    float a = v * (v + 0.0245786f) - 0.000090537f;
    float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}

__device__ __forceinline__ P3 aces_fitted(P3 c) /* :42-55 */
{
    const float in[9] = {0.59719f, 0.35458f, 0.04823f, 0.07600f, 0.90834f, 0.01566f, 0.02840f, 0.13383f, 0.83777f};
    const float out[9] = {1.60475f, -0.53108f, -0.07367f, -0.10208f, 1.10813f, -0.00605f, -0.00327f, -0.07276f, 1.07602f};
    c = mul_rows(in, c);
    c = p3(rrt_odt(c.x), rrt_odt(c.y), rrt_odt(c.z));
    c = mul_rows(out, c);
    return p3(tb_saturate(c.x), tb_saturate(c.y), tb_saturate(c.z));
}

__device__ __forceinline__ float uncharted_partial(float x) /* :64-73 */
{
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
__device__ __forceinline__ float uncharted_filmic(float v) /* :76-84 */
{
    float curr = uncharted_partial(v * 2.0f);
    float white = 1.0f / uncharted_partial(11.2f);
    return curr * white;
}

__device__ __forceinline__ P3 commerce(P3 c) /* CommerceToneMapping :87-104 */
{
    const float startCompression = 0.8f - 0.04f, desaturation = 0.15f;
    float x = tb_min(c.x, tb_min(c.y, c.z));
    float offset = x < 0.08f ? x - (6.25f * x) * x : 0.04f;
    c = p3(c.x - offset, c.y - offset, c.z - offset);
    float peak = tb_max(c.x, tb_max(c.y, c.z));
    if (peak < startCompression) return c;
    float d = 1.0f - startCompression;
    float newPeak = 1.0f - (d * d) / ((peak + d) - startCompression);
    float s = newPeak / peak;
    c = p3(c.x * s, c.y * s, c.z * s);
    float g = 1.0f - 1.0f / (desaturation * (peak - newPeak) + 1.0f);
    return p3(lerp1(c.x, newPeak * 1.0f, g), lerp1(c.y, newPeak * 1.0f, g), lerp1(c.z, newPeak * 1.0f, g));
}

__device__ __forceinline__ float agx_contrast(float x) /* agxDefaultContrastApproximation :106-110 */
{
    float x2 = x * x, x4 = x2 * x2;
    return ((((((15.5f * x4) * x2 - (40.14f * x4) * x) + 31.96f * x4) - (6.868f * x2) * x) + 0.4298f * x2) + 0.1191f * x) - 0.00232f;
}
__device__ __forceinline__ P3 agx(P3 c) /* :112-126 */
{
    const float m[9] = {0.842479062253094f, 0.0423282422610123f, 0.0423756549057051f, 0.0784335999999992f, 0.878468636469772f, 0.0784336f,
                        0.0792237451477643f, 0.0791661274605434f, 0.879142973793104f};
    const float minEv = -12.47393f, maxEv = 4.026069f;
    c = mul_cols(c, m);
    c = p3(tb_clamp(tb_log2(c.x), minEv, maxEv), tb_clamp(tb_log2(c.y), minEv, maxEv), tb_clamp(tb_log2(c.z), minEv, maxEv));
    c = p3((c.x - minEv) / (maxEv - minEv), (c.y - minEv) / (maxEv - minEv), (c.z - minEv) / (maxEv - minEv));
    return p3(agx_contrast(c.x), agx_contrast(c.y), agx_contrast(c.z));
}
__device__ __forceinline__ P3 agx_look(P3 v, bool punchy) /* :129-151 */
{
    float luma = (v.x * 0.2126f + v.y * 0.7152f) + v.z * 0.0722f;
    const float slope = 1.0f, offset = 0.0f, power = punchy ? 1.35f : 1.0f, sat = punchy ? 1.4f : 1.0f;
    v = p3(tb_pow(v.x * slope + offset, power), tb_pow(v.y * slope + offset, power), tb_pow(v.z * slope + offset, power));
    return p3(luma + sat * (v.x - luma), luma + sat * (v.y - luma), luma + sat * (v.z - luma));
}

__device__ __forceinline__ float gt_tonemap(float x) /* GTTonemap :158-176 */
{
    const float m = 0.22f, a = 1.0f, c = 1.33f, P = 1.0f, l = 0.4f;
    float l0 = ((P - m) * l) / a;
    float S0 = m + l0;
    float S1 = m + a * l0;
    float C2 = (a * P) / (P - S1);
    float L = m + a * (x - m);
    float T = m * tb_pow(x / m, c);
    float S = P - (P - S1) * tb_exp((-C2 * (x - S0)) / P);
    float t = tb_saturate((x - 0.0f) / (m - 0.0f));
    float w0 = 1.0f - (t * t) * (3.0f - 2.0f * t);
    float w2 = (x < m + l) ? 0.0f : 1.0f;
    float w1 = (1.0f - w0) - w2;
    return (T * w0 + L * w1) + S * w2;
}

__device__ __forceinline__ P3 tonemap(uint32_t type, P3 c) /* Tonemap :178-204 */
{
    switch (type) {
    case 0: return gamma3(p3(c.x / (1.0f + c.x), c.y / (1.0f + c.y), c.z / (1.0f + c.z)));                       /* Reinhard */
    case 7: return gamma3(p3(gt_tonemap(c.x), gt_tonemap(c.y), gt_tonemap(c.z)));                                 /* GT */
    case 1: return gamma3(aces_fitted(c));                                                                        /* ACES */
    case 3: return gamma3(p3(uncharted_filmic(c.x), uncharted_filmic(c.y), uncharted_filmic(c.z)));               /* Uncharted */
    case 4: return gamma3(commerce(c));                                                                           /* Khronos PBR neutral */
    case 5: return agx_look(agx(c), false);
    case 6: return agx_look(agx(c), true);
    default: return gamma3(p3(tb_saturate(c.x), tb_saturate(c.y), tb_saturate(c.z)));                             /* clamp */
    }
}

__device__ __forceinline__ P3 lerp3colors(P3 a, P3 b, P3 c, float t) /* Lerp3 :115-125 */
{
    if (t < 0.5f) { float s = t * 2.0f; return p3(lerp1(a.x, b.x, s), lerp1(a.y, b.y, s), lerp1(a.z, b.z, s)); }
    float s = (t - 0.5f) * 2.0f;
    return p3(lerp1(b.x, c.x, s), lerp1(b.y, c.y, s), lerp1(b.z, c.z, s));
}

/* GenerateHistogramCS.hlsl:19-31 */
__device__ __forceinline__ uint32_t luminance_bin(float luminance, float minLog, float oneOverRange)
{
    if (luminance < 0.00001f) return 0u;
    float l = tb_saturate((tb_log2(luminance) - minLog) * oneOverRange);
    return (uint32_t)(l * 254.0f + 1.0f);
}

__global__ __launch_bounds__(256) void post_histogram_kernel(const TbFloat4* in, uint32_t W, uint32_t H, float minLog, float oneOverRange, uint32_t* histogram)
{
    __shared__ uint32_t bins[256];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t groupsX = (W + 15u) / 16u;
    const uint32_t x = (blockIdx.x % groupsX) * 16u + (threadIdx.x & 15u), y = (blockIdx.x / groupsX) * 16u + (threadIdx.x >> 4);
    uint32_t bin = 0xffffffffu; /* no pixel */
    if (x < W && y < H) {
        const TbFloat4 a = in[(size_t)y * W + x];
        P3 c = p3(a.x / a.w, a.y / a.w, a.z / a.w);
        bin = luminance_bin(luma709(c), minLog, oneOverRange);
    }
    /* neighbouring pixels mostly share a bin: one LDS atomic per distinct bin of the wave instead of one per pixel */
    unsigned long long todo = __ballot(bin != 0xffffffffu);
    while (todo) {
        const uint32_t leaderBin = (uint32_t)__shfl((int)bin, __ffsll((long long)todo) - 1, 64);
        const unsigned long long same = __ballot(bin == leaderBin);
        if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)todo) - 1)) atomicAdd(&bins[leaderBin], (uint32_t)__popcll(same));
        todo &= ~same;
    }
    __syncthreads();
    if (bins[threadIdx.x]) atomicAdd(&histogram[threadIdx.x], bins[threadIdx.x]);
}

/* CalculateAveragedLuminanceCS.hlsl:15-34 (uint arithmetic; a zero divisor gives 0xffffffff like D3D's udiv) */
__global__ __launch_bounds__(256) void post_average_kernel(const uint32_t* histogram, uint32_t pixelCount, float logRange, float minLog, float* averaged)
{
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const uint32_t count = histogram[threadIdx.x];
    atomicAdd(&total, count * threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t den = pixelCount - count;
        const uint32_t q = den ? total / den : 0xffffffffu;
        float avg = ((float)q - 1.0f) / 254.0f;
        *averaged = tb_exp2(avg * logRange + minLog);
    }
}

__global__ __launch_bounds__(256) void post_process_kernel(TbPostConstants pc, const TbFloat4* in, const float* inR32, const TbFloat4* aux,
    const float* averaged,
                                                           TbFloat4* out, uint32_t* outRgba8)
{
    const uint32_t groupsX = (pc.W + 15u) / 16u;
    const uint32_t x = (blockIdx.x % groupsX) * 16u + (threadIdx.x & 15u), y = (blockIdx.x / groupsX) * 16u + (threadIdx.x >> 4);
    if (x >= pc.W || y >= pc.H) return;
    const size_t i = (size_t)y * pc.W + x;
    TbFloat4 color;
    if (inR32) color = TbFloat4{inR32[i], 0.0f, 0.0f, 1.0f}; else color = in[i]; /* a typed load of an R32_FLOAT texel is (r, 0, 0, 1) */
    P3 o;
    switch (pc.OutputType) {
    default: /* OUTPUT_TYPE_LIT, ProcessLit :23-47 */ {
        o = p3(color.x / color.w, color.y / color.w, color.z / color.w);
        float exposure = pc.UseAutoExposure ? tb_pow(0.5f, 2.2f) / *averaged : pc.ExposureMultiplier;
        o = tonemap(pc.TonemapType, p3(o.x * exposure, o.y * exposure, o.z * exposure));
        break; }
    case TB_OUTPUT_TYPE_ALBEDO: /* ProcessAlbedo :64-76 */
        o = tonemap(pc.TonemapType, p3(color.x * pc.ExposureMultiplier, color.y * pc.ExposureMultiplier, color.z * pc.ExposureMultiplier));
        if (pc.UseGammaCorrection) o = gamma3(o);
        break;
    case TB_OUTPUT_TYPE_NORMAL: /* ProcessNormal :78-83 */ {
        uint32_t n = (uint32_t)color.w;
        if (n > 0) {
            tb3 v = tb3_normalize(tb3_make(color.x / (float)n, color.y / (float)n, color.z / (float)n));
            o = p3(tb_abs(v.x), tb_abs(v.y), tb_abs(v.z));
        } else o = p3(0, 0, 0);
        break; }
    case TB_OUTPUT_TYPE_DEPTH: case 7u: /* PassThroughColor :107-114 (depth, live pixels) */
        o = tonemap(pc.TonemapType, p3(color.x * pc.ExposureMultiplier, color.y * pc.ExposureMultiplier, color.z * pc.ExposureMultiplier));
        break;
    case 5u: /* OUTPUT_TYPE_LUMINANCE, ProcessLuminance :49-62 */ {
        o = p3(color.x / color.w, color.y / color.w, color.z / color.w);
        o = tonemap(pc.TonemapType, p3(o.x * pc.ExposureMultiplier, o.y * pc.ExposureMultiplier, o.z * pc.ExposureMultiplier));
        float l = luma709(o);
        o = p3(l, l, l);
        if (pc.UseGammaCorrection) o = gamma3(o);
        break; }
    case TB_OUTPUT_TYPE_HEATMAP: /* ProcessHeatmap :127-140 */ {
        uint32_t total = (uint32_t)color.x + (uint32_t)color.y;
        o = lerp3colors(p3(0, 1, 0), p3(1, 1, 0), p3(1, 0, 0), (float)total / 100.0f);
        o = tonemap(pc.TonemapType, p3(o.x * pc.ExposureMultiplier, o.y * pc.ExposureMultiplier, o.z * pc.ExposureMultiplier));
        break; }
    }
    (void)aux;
    if (out) out[i] = TbFloat4{o.x, o.y, o.z, 1.0f};
    if (outRgba8) { /* R8G8B8A8_UNORM store: clamp, scale, + 0.5, truncate */
        uint32_t r = (uint32_t)(tb_saturate(o.x) * 255.0f + 0.5f), g = (uint32_t)(tb_saturate(o.y) * 255.0f + 0.5f),
            b = (uint32_t)(tb_saturate(o.z) * 255.0f + 0.5f);
        outRgba8[i] = r | (g << 8) | (b << 16) | 0xff000000u;
    }
}

} // namespace

extern "C" hipError_t post_launch(hipStream_t stream, const TbPostConstants* pc, const TbFloat4* in, const float* inR32, const TbFloat4* aux,
                                  uint32_t* histogram, float* averaged, TbFloat4* out, uint32_t* outRgba8)
{
    const uint32_t groups = ((pc->W + 15u) / 16u) * ((pc->H + 15u) / 16u);
    if (pc->UseAutoExposure && pc->OutputType == TB_OUTPUT_TYPE_LIT) { /* TracerBoy.cpp:2948-3030: MinLogLuminance -10, LogLuminanceRange 16 */
        hipError_t e = hipMemsetAsync(histogram, 0, 256 * 4, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(post_histogram_kernel, dim3(groups), dim3(256), 0, stream, in, pc->W, pc->H, -10.0f, 1.0f / 16.0f, histogram);
        hipLaunchKernelGGL(post_average_kernel, dim3(1), dim3(256), 0, stream, (const uint32_t*)histogram, pc->W * pc->H, 16.0f, -10.0f, averaged);
    }
    hipLaunchKernelGGL(post_process_kernel, dim3(groups), dim3(256), 0, stream, *pc, in, inR32, aux, (const float*)averaged, out, outRgba8);
    return hipGetLastError();
}
