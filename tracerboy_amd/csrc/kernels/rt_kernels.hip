/* rt_kernels.hip -- the real-time chain behind the path tracer:
 *
 *   rt_temporal_kernel    TemporalAccumulationCS.hlsl:95-235   reprojection of the previous frame through the previous
 *                         camera, world-position history rejection, luminance moments -> variance in .w
 *   rt_denoise_kernel     DenoiserCS.hlsl:18-164               one 5x5 a-trous iteration (edge-stopping on luminance /
 *                         normal / world position), offsets dilated by OffsetMultiplier = 2^i
 *   rt_composite_kernel   CompositeAlbedoCS.hlsl:17-25         albedo * lighting * diffuse + lighting * specular + emissive
 *
 * Per-pixel gathers over RGBA32F surfaces that all fit the Infinity Cache at 1080p (33 MB each): 9 + 4 reads per pixel
 * in the temporal pass, 25 x 4 in a denoiser iteration; no reuse that LDS could capture once the taps are dilated, so the
 * kernels are plain coalesced-row gathers (a wave covers an 8x8 tile like the reference's groups).  Arithmetic is
 * spelled out per operation and mirrored by oracle/rt_ref.cpp (bit-exact parity tests). */
#include <hip/hip_runtime.h>
#include "tb_math.h"
#include "tb_vec.h"
#include "tb_abi.h"
#include "pt_launch.h"

namespace {

__device__ __forceinline__ tb3 xyz(const TbFloat4& v) { return tb3_make(v.x, v.y, v.z); }
__device__ __forceinline__ float luma709(tb3 c) { return (c.x * 0.212671f + c.y * 0.715160f) + c.z * 0.072169f; } /* Tonemap.h:12-15 */
__device__ __forceinline__ float dot3(tb3 a, tb3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float len3(tb3 a) { return tb_sqrt(dot3(a, a)); }
__device__ __forceinline__ tb3 norm3(tb3 a) { float l = len3(a); return tb3_make(a.x / l, a.y / l, a.z / l); }
__device__ __forceinline__ bool pixel_of(uint32_t W, uint32_t H, uint32_t& x, uint32_t& y)
{
    const uint32_t groupsX = (W + 7u) / 8u;
    x = (blockIdx.x % groupsX) * 8u + (threadIdx.x & 7u); y = (blockIdx.x / groupsX) * 8u + (threadIdx.x >> 3);
    return x < W && y < H;
}
/* Texture2D operator[] outside the resource returns 0 */
__device__ __forceinline__ TbFloat4 load_or_zero(const TbFloat4* t, uint32_t W, uint32_t H, uint32_t x, uint32_t y)
{
    return (x < W && y < H) ? t[(size_t)y * W + x] : TbFloat4{0.0f, 0.0f, 0.0f, 0.0f};
}
/* SampleLevel with a bilinear CLAMP sampler (exact fp32 weights; D3D hardware may quantise them) */
__device__ __forceinline__ TbFloat4 sample_bilinear_clamp(const TbFloat4* t, uint32_t W, uint32_t H, float u, float v)
{
    const float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    const float x0f = tb_floor(fx), y0f = tb_floor(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    auto clampi = [](float f, uint32_t n) { int i = (int)f; return (uint32_t)(i < 0 ? 0 : (i >= (int)n ? (int)n - 1 : i)); };
    const uint32_t x0 = clampi(x0f, W), x1 = clampi(x0f + 1.0f, W), y0 = clampi(y0f, H), y1 = clampi(y0f + 1.0f, H);
    const TbFloat4 a = t[(size_t)y0 * W + x0], b = t[(size_t)y0 * W + x1], c = t[(size_t)y1 * W + x0], d = t[(size_t)y1 * W + x1];
    TbFloat4 r;
    r.x = tb_lerp(tb_lerp(a.x, b.x, tx), tb_lerp(c.x, d.x, tx), ty); r.y = tb_lerp(tb_lerp(a.y, b.y, tx), tb_lerp(c.y, d.y, tx), ty);
    r.z = tb_lerp(tb_lerp(a.z, b.z, tx), tb_lerp(c.z, d.z, tx), ty); r.w = tb_lerp(tb_lerp(a.w, b.w, tx), tb_lerp(c.w, d.w, tx), ty);
    return r;
}

__global__ __launch_bounds__(64) void rt_temporal_kernel(TbTemporalConstants k, const TbFloat4* history, const TbFloat4* current, const TbFloat4* worldPos,
                                                         const TbFloat4* prevWorldPos, const TbFloat4* momentHistory, const TbFloat4* normals, TbFloat4* out,
                                                             TbFloat4* outMoment)
{
    uint32_t px, py;
    const uint32_t W = k.ResolutionX, H = k.ResolutionY;
    if (!pixel_of(W, H, px, py)) return;
    const size_t i = (size_t)py * W + px;
    const tb3 WorldPosition = xyz(worldPos[i]), WorldNormal = xyz(normals[i]);
    const bool bHitValid = WorldNormal.x != 0.0f || WorldNormal.y != 0.0f || WorldNormal.z != 0.0f;
    const float aspectRatio = (float)W / (float)H, lensHeight = k.CameraLensHeight, lensWidth = lensHeight * aspectRatio;
    const tb3 prevPos = tb3_make(k.PrevFrameCameraPosition[0], k.PrevFrameCameraPosition[1], k.PrevFrameCameraPosition[2]);
    const tb3 prevLook = tb3_make(k.PrevFrameCameraLookAt[0], k.PrevFrameCameraLookAt[1], k.PrevFrameCameraLookAt[2]);
    const tb3 prevRight = tb3_make(k.PrevFrameCameraRight[0], k.PrevFrameCameraRight[1], k.PrevFrameCameraRight[2]);
    const tb3 prevUp = tb3_make(k.PrevFrameCameraUp[0], k.PrevFrameCameraUp[1], k.PrevFrameCameraUp[2]);
    const tb3 PrevFrameCameraDir = norm3(prevLook - prevPos);
    const tb3 PrevFrameFocalPoint = prevPos - PrevFrameCameraDir * k.CameraFocalDistance;
    const tb3 PrevFrameRayDirection = norm3(WorldPosition - PrevFrameFocalPoint);
    const tb3 RawOutputColor = xyz(current[i]);

    tb3 nmin = WorldPosition, nmax = WorldPosition; /* WORLD_POSITION_HISTORY_REJECTION: box of the 3x3 neighbourhood */
    for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++) {
            const int cx = (int)px + x, cy = (int)py + y;
            const bool valid = cx > 0 && cy > 0 && cx < (int)W && cy < (int)H; /* `all(coord > 0)`: row and column 0 are left out, as written */
            if (valid && !(x == 0 && y == 0)) { const tb3 w = xyz(worldPos[(size_t)cy * W + cx]); nmin = tb3_min(nmin, w); nmax = tb3_max(nmax, w); }
        }

    tb3 PrevFrameColor = tb3_splat(0.0f), PrevMomentData = tb3_splat(0.0f);
    float t = -1.0f; /* PlaneIntersection :78-88 */
    { const float denom = dot3(PrevFrameCameraDir, PrevFrameRayDirection);
        if (tb_abs(denom) > 0.0f) t = dot3(prevPos - PrevFrameFocalPoint, PrevFrameCameraDir) / denom; }
    bool bValidHistory = false;
    if (!k.IgnoreHistory && t >= 0.0f && bHitValid) {
        const tb3 LensPosition = PrevFrameFocalPoint + PrevFrameRayDirection * t;
        const tb3 Offset = LensPosition - prevPos;
        float u = dot3(Offset, prevRight) / (lensWidth / 2.0f), v = dot3(Offset, prevUp) / (lensHeight / 2.0f);
        u = (u + 1.0f) / 2.0f; v = (v + 1.0f) / 2.0f; v = 1.0f - v;
        if (u >= 0.0f && u <= 1.0f && v >= 0.0f && v <= 1.0f) {
            const float distanceToNeighbor = len3(nmax - nmin);
            const float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
            float SummedWeight = 0.0f;
            for (uint32_t x = 0; x < 2; x++)
                for (uint32_t y = 0; y < 2; y++) {
                    const uint32_t ix = (uint32_t)((int)fx + (int)x), iy = (uint32_t)((int)fy + (int)y);
                    const tb3 pw = xyz(load_or_zero(prevWorldPos, W, H, ix, iy));
                    if (len3(pw - WorldPosition) < distanceToNeighbor) {
                        const float xw = x == 0 ? 1.0f - tb_frac(fx) : tb_frac(fx), yw = y == 0 ? 1.0f - tb_frac(fy) : tb_frac(fy);
                        const float weight = xw * yw;
                        const tb3 h = xyz(load_or_zero(history, W, H, ix, iy));
                        PrevFrameColor = PrevFrameColor + h * weight;
                        SummedWeight += weight;
                        if (k.OutputMomentInformation) PrevMomentData = PrevMomentData + xyz(load_or_zero(momentHistory, W, H, ix, iy)) * weight;
                    }
                }
            bValidHistory = SummedWeight > 0.0f;
            if (bValidHistory) { PrevFrameColor = PrevFrameColor / SummedWeight; PrevMomentData = PrevMomentData / SummedWeight; }
            if (k.OutputMomentInformation) PrevMomentData = xyz(sample_bilinear_clamp(momentHistory, W, H, u, v)); /* :204, replaces the weighted value */
        }
    }
    float outputAlpha = 1.0f;
    if (k.OutputMomentInformation) {
        const float luminance = luma709(RawOutputColor), luminanceSquared = luminance * luminance;
        const float sampleCount = PrevMomentData.z + 1.0f;
        const float lerpFactor = 1.0f / tb_min(sampleCount, 32.0f);
        const float m1 = tb_lerp(PrevMomentData.x, luminance, lerpFactor), m2 = tb_lerp(PrevMomentData.y, luminanceSquared, lerpFactor);
        outMoment[i] = TbFloat4{m1, m2, sampleCount, 0.0f};
        outputAlpha = tb_max(m2 - m1 * m1, 0.0f);
    }
    const float hw = bValidHistory ? k.HistoryWeight : 0.0f;
    out[i] = TbFloat4{tb_lerp(RawOutputColor.x, PrevFrameColor.x, hw), tb_lerp(RawOutputColor.y, PrevFrameColor.y, hw), tb_lerp(RawOutputColor.z,
        PrevFrameColor.z, hw), outputAlpha};
}

__global__ __launch_bounds__(64) void rt_denoise_kernel(TbDenoiserConstants k, const TbFloat4* input, const TbFloat4* normals, const TbFloat4* positions,
                                                        const TbFloat4* undenoised, TbFloat4* out)
{
    uint32_t px, py;
    const uint32_t W = k.ResolutionX, H = k.ResolutionY;
    if (!pixel_of(W, H, px, py)) return;
    const size_t i = (size_t)py * W + px;
    const float EPS = 0.0001f; /* SharedShaderStructs.h:3 */
    const tb3 normal = xyz(normals[i]);
    const TbFloat4 posData = positions[i];
    const tb3 position = xyz(posData);
    const float distanceToNeighborPixel = posData.w;
    const float luma = luma709(xyz(undenoised[i]));
    const float luminanceVariance = input[i].w;
    float weightedSum = 0.0f, accumulatedVariance = 0.0f;
    tb3 accumulatedColor = tb3_splat(0.0f);
    if (normal.x != 0.0f || normal.y != 0.0f || normal.z != 0.0f) {
        const int mult = (int)k.OffsetMultiplier;
        const float centerVarianceSqrt = tb_sqrt(luminanceVariance);
        for (int xo = -2; xo <= 2; xo++)
            for (int yo = -2; yo <= 2; yo++) {
                const int ox = xo * mult, oy = yo * mult;
                const int cx = (int)px + ox, cy = (int)py + oy;
                if (cx < 0 || cy < 0 || cx >= (int)W || cy >= (int)H) continue;
                const size_t c = (size_t)cy * W + cx;
                /* CalculateWeight :18-45 */
                const float l = luma709(xyz(undenoised[c]));
                const float lumaWeight = tb_exp(-tb_abs(l - luma) / tb_max(k.LumaWeightingMultiplier * centerVarianceSqrt, EPS));
                const float normalWeight = tb_pow(tb_max(0.0f, dot3(normal, xyz(normals[c]))), k.NormalWeightingExponential);
                const float distance = len3(xyz(positions[c]) - position);
                const float positionWeight = tb_exp(-distance / (k.IntersectionPositionWeightingMultiplier * tb_abs((float)ox * distanceToNeighborPixel +
                    (float)oy * distanceToNeighborPixel) + EPS));
                const float kw[3] = {3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
                const int ax = xo < 0 ? -xo : xo, ay = yo < 0 ? -yo : yo; /* |offset / OffsetMultiplier| of DenoiserCS: the tap's index in the 5x5 kernel */
                const float weight = (((lumaWeight * positionWeight) * normalWeight) * kw[ax]) * kw[ay];
                const TbFloat4 n = input[c];
                accumulatedColor = accumulatedColor + xyz(n) * weight;
                accumulatedVariance += (weight * weight) * n.w;
                weightedSum += weight;
            }
    } else {
        const TbFloat4 n = input[i];
        accumulatedVariance = n.w; accumulatedColor = xyz(n); weightedSum = 1.0f;
    }
    out[i] = TbFloat4{accumulatedColor.x / weightedSum, accumulatedColor.y / weightedSum, accumulatedColor.z / weightedSum,
        accumulatedVariance / (weightedSum * weightedSum)};
}

__global__ __launch_bounds__(64) void rt_composite_kernel(uint32_t W, uint32_t H, const TbFloat4* albedoTex, const TbFloat4* lighting,
    const TbFloat4* emissiveTex, TbFloat4* out)
{
    uint32_t px, py;
    if (!pixel_of(W, H, px, py)) return;
    const size_t i = (size_t)py * W + px;
    const TbFloat4 a = albedoTex[i]; const tb3 albedo = xyz(a), l = xyz(lighting[i]), e = xyz(emissiveTex[i]);
    const float diffuse = a.w, specular = 1.0f - diffuse;
    out[i] = TbFloat4{((albedo.x * l.x) * diffuse + l.x * specular) + e.x, ((albedo.y * l.y) * diffuse + l.y * specular) + e.y,
        ((albedo.z * l.z) * diffuse + l.z * specular) + e.z, 1.0f};
}

} // namespace

extern "C" hipError_t rt_launch_temporal(hipStream_t stream, const TbTemporalConstants* k, const TbFloat4* history, const TbFloat4* current,
    const TbFloat4* worldPos,
                                         const TbFloat4* prevWorldPos, const TbFloat4* momentHistory, const TbFloat4* normals, TbFloat4* out,
                                             TbFloat4* outMoment)
{
    const uint32_t groups = ((k->ResolutionX + 7u) / 8u) * ((k->ResolutionY + 7u) / 8u);
    hipLaunchKernelGGL(rt_temporal_kernel, dim3(groups), dim3(64), 0, stream, *k, history, current, worldPos, prevWorldPos, momentHistory, normals, out,
        outMoment);
    return hipGetLastError();
}
extern "C" hipError_t rt_launch_denoise(hipStream_t stream, const TbDenoiserConstants* k, const TbFloat4* input, const TbFloat4* normals,
    const TbFloat4* positions,
                                        const TbFloat4* undenoised, TbFloat4* out)
{
    const uint32_t groups = ((k->ResolutionX + 7u) / 8u) * ((k->ResolutionY + 7u) / 8u);
    hipLaunchKernelGGL(rt_denoise_kernel, dim3(groups), dim3(64), 0, stream, *k, input, normals, positions, undenoised, out);
    return hipGetLastError();
}
extern "C" hipError_t rt_launch_composite(hipStream_t stream, uint32_t W, uint32_t H, const TbFloat4* albedo, const TbFloat4* lighting,
    const TbFloat4* emissive, TbFloat4* out)
{
    const uint32_t groups = ((W + 7u) / 8u) * ((H + 7u) / 8u);
    hipLaunchKernelGGL(rt_composite_kernel, dim3(groups), dim3(64), 0, stream, W, H, albedo, lighting, emissive, out);
    return hipGetLastError();
}
