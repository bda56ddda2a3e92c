/* rt_ref.cpp -- CPU restatement of the reference's real-time chain (TEST INFRASTRUCTURE, see tb_oracle.h).
 *
 *   tbo_temporal    TracerBoy/TemporalAccumulationCS.hlsl:95-235 (NEIGHBORHOOD_CLAMPING 0, WORLD_POSITION_HISTORY_REJECTION 1)
 *   tbo_denoise     TracerBoy/DenoiserCS.hlsl:18-164 (USE_MEDIAN_FILTER 0), one a-trous iteration
 *   tbo_composite   TracerBoy/CompositeAlbedoCS.hlsl:17-25
 * Surfaces are W*H float4 arrays, row 0 = top.  fp32, one rounding per written operation, tb_math.h transcendentals.
 * Two choices the HLSL leaves to the hardware are fixed here and in rt_kernels.hip alike: the bilinear CLAMP sample of
 * the moment history uses exact fp32 weights, and a read outside a texture returns 0.  Parity unpinned by the reference
 * (no tests there); the closed-form checks are in tests/test_realtime_chain.py. */
#include "tb_oracle.h"
#include "../include/tb_math.h"
#include "../include/tb_vec.h"
#include <cstring>

namespace {

struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V3 ld(const float* t, size_t i) { return v3(t[4 * i], t[4 * i + 1], t[4 * i + 2]); }
inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 mul(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 divs(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float length(V3 a) { return tb_sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return divs(a, length(a)); }
inline V3 vmin(V3 a, V3 b) { return v3(tb_min(a.x, b.x), tb_min(a.y, b.y), tb_min(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(tb_max(a.x, b.x), tb_max(a.y, b.y), tb_max(a.z, b.z)); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline float ColorToLuma(V3 c) { return (c.x * 0.212671f + c.y * 0.715160f) + c.z * 0.072169f; }
inline V3 loadOrZero(const float* t, uint32_t W, uint32_t H, uint32_t x, uint32_t y) { return (x < W && y < H) ? ld(t, (size_t)y * W + x) : v3(0, 0, 0); }

void sampleBilinearClamp(const float* t, uint32_t W, uint32_t H, float u, float v, float out[4])
{
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    float x0f = tb_floor(fx), y0f = tb_floor(fy), tx = fx - x0f, ty = fy - y0f;
    auto clampi = [](float f, uint32_t n) { int i = (int)f; return (uint32_t)(i < 0 ? 0 : (i >= (int)n ? (int)n - 1 : i)); };
    uint32_t x0 = clampi(x0f, W), x1 = clampi(x0f + 1.0f, W), y0 = clampi(y0f, H), y1 = clampi(y0f + 1.0f, H);
    const float *a = t + 4 * ((size_t)y0 * W + x0), *b = t + 4 * ((size_t)y0 * W + x1), *c = t + 4 * ((size_t)y1 * W + x0), *d = t + 4 * ((size_t)y1 * W + x1);
    for (int k = 0; k < 4; k++) out[k] = lerp(lerp(a[k], b[k], tx), lerp(c[k], d[k], tx), ty);
}

} // namespace

extern "C" void tbo_temporal(const TbTemporalConstants* C, const float* TemporalHistory, const float* CurrentFrame, const float* WorldPositionTexture,
                             const float* PreviousFrameWorldPositionTexture, const float* MomentHistory, const float* WorldNormalTexture, float* OutputTexture,
                                 float* OutputMoment)
{
    const uint32_t W = C->ResolutionX, H = C->ResolutionY;
    for (uint32_t py = 0; py < H; py++) for (uint32_t px = 0; px < W; px++) {
        const size_t i = (size_t)py * W + px;
        V3 WorldPosition = ld(WorldPositionTexture, i), WorldNormal = ld(WorldNormalTexture, i);
        bool bHitValid = WorldNormal.x != 0.0f || WorldNormal.y != 0.0f || WorldNormal.z != 0.0f;
        float aspectRatio = (float)W / (float)H;
        float lensHeight = C->CameraLensHeight, lensWidth = lensHeight * aspectRatio;
        V3 PrevPos = v3(C->PrevFrameCameraPosition[0], C->PrevFrameCameraPosition[1], C->PrevFrameCameraPosition[2]);
        V3 PrevLookAt = v3(C->PrevFrameCameraLookAt[0], C->PrevFrameCameraLookAt[1], C->PrevFrameCameraLookAt[2]);
        V3 PrevRight = v3(C->PrevFrameCameraRight[0], C->PrevFrameCameraRight[1], C->PrevFrameCameraRight[2]);
        V3 PrevUp = v3(C->PrevFrameCameraUp[0], C->PrevFrameCameraUp[1], C->PrevFrameCameraUp[2]);
        V3 PrevFrameCameraDir = normalize(sub(PrevLookAt, PrevPos));
        V3 PrevFrameFocalPoint = sub(PrevPos, mul(PrevFrameCameraDir, C->CameraFocalDistance));
        V3 PrevFrameRayDirection = normalize(sub(WorldPosition, PrevFrameFocalPoint));
        V3 RawOutputColor = ld(CurrentFrame, i);

        V3 NeighborMinWorldPosition = WorldPosition, NeighborMaxWorldPosition = WorldPosition;
        for (int x = -1; x <= 1; x++) for (int y = -1; y <= 1; y++) { /* :123-146 */
            int cx = (int)px + x, cy = (int)py + y;
            bool bIsValidCoord = (cx > 0 && cy > 0) && cx < (int)W && cy < (int)H;
            bool bIsCenterCoord = x == 0 && y == 0;
            if (bIsValidCoord && !bIsCenterCoord) {
                V3 worldPosition = ld(WorldPositionTexture, (size_t)cy * W + cx);
                NeighborMinWorldPosition = vmin(NeighborMinWorldPosition, worldPosition);
                NeighborMaxWorldPosition = vmax(NeighborMaxWorldPosition, worldPosition);
            }
        }

        V3 PrevFrameColor = v3(0, 0, 0), PrevMomentData = v3(0, 0, 0);
        float t = -1.0f; /* PlaneIntersection(PrevFrameFocalPoint, PrevFrameRayDirection, PrevFrameCameraPosition, PrevFrameCameraDir) :78-88 */
        { float denom = dot(PrevFrameCameraDir, PrevFrameRayDirection);
            if (tb_abs(denom) > 0.0f) t = dot(sub(PrevPos, PrevFrameFocalPoint), PrevFrameCameraDir) / denom; }
        bool bValidHistory = false;
        if (!C->IgnoreHistory && t >= 0 && bHitValid) {
            V3 LensPosition = add(PrevFrameFocalPoint, mul(PrevFrameRayDirection, t));
            V3 OffsetFromCenter = sub(LensPosition, PrevPos);
            float UVx = dot(OffsetFromCenter, PrevRight) / (lensWidth / 2.0f), UVy = dot(OffsetFromCenter, PrevUp) / (lensHeight / 2.0f);
            UVx = (UVx + 1.0f) / 2.0f; UVy = (UVy + 1.0f) / 2.0f;
            UVy = 1.0f - UVy;
            if (UVx >= 0.0f && UVx <= 1.0f && UVy >= 0.0f && UVy <= 1.0f) {
                float distanceToNeighbor = length(sub(NeighborMaxWorldPosition, NeighborMinWorldPosition));
                float fx = UVx * (float)W - 0.5f, fy = UVy * (float)H - 0.5f;
                float SummedWeight = 0.0f;
                for (uint32_t x = 0; x < 2; x++) for (uint32_t y = 0; y < 2; y++) {
                    uint32_t ix = (uint32_t)((int)fx + (int)x), iy = (uint32_t)((int)fy + (int)y);
                    V3 PreviousFrameWorldPosition = loadOrZero(PreviousFrameWorldPositionTexture, W, H, ix, iy);
                    if (length(sub(PreviousFrameWorldPosition, WorldPosition)) < distanceToNeighbor) {
                        float xWeight = x == 0 ? 1.0f - tb_frac(fx) : tb_frac(fx);
                        float yWeight = y == 0 ? 1.0f - tb_frac(fy) : tb_frac(fy);
                        float weight = xWeight * yWeight;
                        PrevFrameColor = add(PrevFrameColor, mul(loadOrZero(TemporalHistory, W, H, ix, iy), weight));
                        SummedWeight += weight;
                        if (C->OutputMomentInformation) PrevMomentData = add(PrevMomentData, mul(loadOrZero(MomentHistory, W, H, ix, iy), weight));
                    }
                }
                bValidHistory = SummedWeight > 0.0f;
                if (bValidHistory) { PrevFrameColor = divs(PrevFrameColor, SummedWeight); PrevMomentData = divs(PrevMomentData, SummedWeight); }
                /* :204 */
                if (C->OutputMomentInformation) { float m[4]; sampleBilinearClamp(MomentHistory, W, H, UVx, UVy, m); PrevMomentData = v3(m[0], m[1], m[2]); }
            }
        }
        float outputAlpha = 1.0f;
        if (C->OutputMomentInformation) { /* :213-224 */
            float luminance = ColorToLuma(RawOutputColor);
            float luminanceSquared = luminance * luminance;
            float sampleCount = PrevMomentData.z + 1.0f;
            float lerpFactor = 1.0f / tb_min(sampleCount, 32.0f);
            float m1 = lerp(PrevMomentData.x, luminance, lerpFactor), m2 = lerp(PrevMomentData.y, luminanceSquared, lerpFactor);
            OutputMoment[4 * i] = m1; OutputMoment[4 * i + 1] = m2; OutputMoment[4 * i + 2] = sampleCount; OutputMoment[4 * i + 3] = 0.0f;
            float variance = tb_max(m2 - m1 * m1, 0.0f);
            outputAlpha = variance;
        }
        float w = bValidHistory ? C->HistoryWeight : 0.0f;
        OutputTexture[4 * i] = lerp(RawOutputColor.x, PrevFrameColor.x, w); OutputTexture[4 * i + 1] = lerp(RawOutputColor.y, PrevFrameColor.y, w);
        OutputTexture[4 * i + 2] = lerp(RawOutputColor.z, PrevFrameColor.z, w); OutputTexture[4 * i + 3] = outputAlpha;
    }
}

extern "C" void tbo_denoise(const TbDenoiserConstants* C, const float* InputTexture, const float* AOVNormals, const float* AOVIntersectPosition,
    const float* UndenoisedTexture,
                            float* OutputTexture)
{
    const uint32_t W = C->ResolutionX, H = C->ResolutionY;
    const float EPSILON = 0.0001f; /* SharedShaderStructs.h:3 */
    const int KERNEL_WIDTH = 5;
    for (uint32_t py = 0; py < H; py++) for (uint32_t px = 0; px < W; px++) {
        const size_t i = (size_t)py * W + px;
        V3 normal = ld(AOVNormals, i);
        V3 intersectedPosition = ld(AOVIntersectPosition, i);
        float distanceToNeighborPixel = AOVIntersectPosition[4 * i + 3];
        float luma = ColorToLuma(ld(UndenoisedTexture, i));
        float luminanceVariance = InputTexture[4 * i + 3];
        float weightedSum = 0.0f, accumulatedVariance = 0.0f;
        V3 accumulatedColor = v3(0, 0, 0);
        if (normal.x != 0.0f || normal.y != 0.0f || normal.z != 0.0f) { /* ValidNormal */
            const int mult = (int)C->OffsetMultiplier;
            for (int xOffset = -KERNEL_WIDTH / 2; xOffset <= KERNEL_WIDTH / 2; xOffset++) for (int yOffset =
                -KERNEL_WIDTH / 2; yOffset <= KERNEL_WIDTH / 2; yOffset++) {
                int ox = xOffset * mult, oy = yOffset * mult;
                int cx = (int)px + ox, cy = (int)py + oy;
                if (cx < 0 || cy < 0 || cx >= (int)W || cy >= (int)H) continue;
                const size_t c = (size_t)cy * W + cx;
                /* CalculateWeight :18-45 */
                float l = ColorToLuma(ld(UndenoisedTexture, c));
                float centerVarianceSqrt = tb_sqrt(luminanceVariance);
                float lumaWeight = tb_exp(-tb_abs(l - luma) / tb_max(C->LumaWeightingMultiplier * centerVarianceSqrt, EPSILON));
                float normalWeight = tb_pow(tb_max(0.0f, dot(normal, ld(AOVNormals, c))), C->NormalWeightingExponential);
                float distance = length(sub(ld(AOVIntersectPosition, c), intersectedPosition));
                float positionWeight = tb_exp(-distance / (C->IntersectionPositionWeightingMultiplier * tb_abs((float)ox * distanceToNeighborPixel +
                    (float)oy * distanceToNeighborPixel) + EPSILON));
                const float weights[3] = {3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
                int ax = ox / mult; if (ax < 0) ax = -ax;
                int ay = oy / mult; if (ay < 0) ay = -ay;
                float weight = (((lumaWeight * positionWeight) * normalWeight) * weights[ax]) * weights[ay];
                V3 NeighborColor = ld(InputTexture, c); float NeighborVariance = InputTexture[4 * c + 3];
                accumulatedColor = add(accumulatedColor, mul(NeighborColor, weight));
                accumulatedVariance += (weight * weight) * NeighborVariance;
                weightedSum += weight;
            }
        } else {
            accumulatedVariance = InputTexture[4 * i + 3]; accumulatedColor = ld(InputTexture, i); weightedSum = 1.0f;
        }
        OutputTexture[4 * i] = accumulatedColor.x / weightedSum; OutputTexture[4 * i + 1] = accumulatedColor.y / weightedSum;
            OutputTexture[4 * i + 2] = accumulatedColor.z / weightedSum;
        OutputTexture[4 * i + 3] = accumulatedVariance / (weightedSum * weightedSum);
    }
}

extern "C" void tbo_composite(uint32_t W, uint32_t H, const float* AlbedoTexture, const float* IndirectLightingTexture, const float* EmissiveTexture,
    float* OutputTexture)
{
    for (size_t i = 0; i < (size_t)W * H; i++) {
        V3 albedo = ld(AlbedoTexture, i); float diffuseContribution = AlbedoTexture[4 * i + 3], specularContribution = 1.0f - diffuseContribution;
        V3 indirectLighting = ld(IndirectLightingTexture, i), emissive = ld(EmissiveTexture, i);
        OutputTexture[4 * i] = ((albedo.x * indirectLighting.x) * diffuseContribution + indirectLighting.x * specularContribution) + emissive.x;
        OutputTexture[4 * i + 1] = ((albedo.y * indirectLighting.y) * diffuseContribution + indirectLighting.y * specularContribution) + emissive.y;
        OutputTexture[4 * i + 2] = ((albedo.z * indirectLighting.z) * diffuseContribution + indirectLighting.z * specularContribution) + emissive.z;
        OutputTexture[4 * i + 3] = 1.0f;
    }
}
