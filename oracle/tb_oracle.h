/* tb_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This directory is a scalar CPU restatement of the reference's
 * per-pixel path-tracing hot path (SoftwareRayTraceCS -> RayTraceCommon -> PathTrace -> Trace ->
 * SoftwareRayQuery::Traverse).  It may be imported / linked / executed only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg, and only as the checker.  Nothing
 * under tracerboy_amd/ links or calls it.
 *
 * PARITY UNPINNED by the reference: TracerBoy has no tests, no golden radiance, its RNG is seeded
 * from the wall clock and its shaders cannot be compiled in this image (no dxc / D3D12).  The
 * oracle is pinned instead by (a) line-by-line citations of the reference, (b) the known-answer
 * material listed in SURVEY.md 8c (tests/test_oracle_known_answers.py) and (c) scene fixtures
 * dumped by the reference's own PBRTParser built under oracle/_ref.
 */
#ifndef TB_ORACLE_H
#define TB_ORACLE_H

#include "../include/tb_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct TboAovs { /* all nullable; W*H entries each */
    TbFloat4* normals;        /* AOVNormals u2 */
    TbFloat4* worldPosition0; /* AOVWorldPosition0 u3 (even frames) */
    TbFloat4* worldPosition1; /* AOVWorldPosition1 u4 (odd frames)  */
    TbFloat4* customOutput;   /* AOVCustomOutput u5: first-hit albedo, or heatmap counters */
    float* depth;             /* AOVDepth u6 */
    TbFloat4* emissive;       /* AOVEmissive u7 */
} TboAovs;

/* Renders frames [firstFrame, firstFrame+numFrames) of rows [y0,y1) into output/jittered
 * (RGBA32F, W*H, accumulated exactly like RayGenCommon.h:721-727).  constants->GlobalFrameCount
 * is overridden per frame.  numThreads <= 1 runs on the calling thread. */
int tbo_render(const TbSceneView* scene, const TbPerFrameConstants* constants, uint32_t width,
               uint32_t height, uint32_t y0, uint32_t y1, uint32_t firstFrame, uint32_t numFrames,
               TbFloat4* output, TbFloat4* jittered, const TboAovs* aovs, TbRayStats* stats,
               int numThreads);

/* One sample of one pixel; returns (rgb*w, w) before the NaN filter. */
void tbo_sample_pixel(const TbSceneView* scene, const TbPerFrameConstants* constants, uint32_t width,
                      uint32_t height, uint32_t x, uint32_t y, float out[4], float* seedAfter,
                      TbRayStats* stats);

/* Closest hit of n rays through the layout-A BVH + hit attributes
 * (IntersectWithMaxDistance, RayGenCommon.h:365-414).  Arrays of n: t (-1 on miss),
 * materialIndex (-1 on miss), bary[2n], primitiveIndex, geometryIndex, normal[3n], uv[2n],
 * boxes/tris tested. */
void tbo_trace_closest(const TbSceneView* scene, uint32_t n, const float* origins, const float* dirs,
                       float* t, int32_t* materialIndex, float* bary, uint32_t* primitiveIndex,
                       uint32_t* geometryIndex, float* normal, float* uv, uint32_t* boxesTested,
                       uint32_t* trianglesTested);

/* Small pieces exposed for known-answer tests. */
float tbo_hash13(float x, float y, float z);
void tbo_rand_stream(float seed, float time, uint32_t n, float* out);
/* BSDF sampling pieces for the furnace / pdf known-answer tests: ImportanceSampleGGXPDF with the half vector the throughput update
 * forms (kernel.glsl:1084-1094,1701-1706); kind 0 ImportanceSampleGGX (:1066-1082), 1 the refraction lobe (:1048-1064), 2 cosine (:1025-1046) */
float tbo_ggx_pdf(const float* normal, const float* incoming, const float* outgoing, float roughness);
void tbo_sample_directions(int kind, float seed, float time, const float* incoming, const float* normal, float roughness, uint32_t n, float* outDirs,
    float* outPdf);
/* StatsBuffer +8 / +12 of the selected pixel after the given frames (RayGenCommon.h:632-648); returns 0 if nothing was written */
int tbo_selected_pixel(const TbSceneView* scene, const TbPerFrameConstants* constants, uint32_t width, uint32_t height, uint32_t firstFrame, uint32_t numFrames,
                       float* distance, int32_t* materialId);
/* Output stage (post_ref.cpp): GenerateHistogramCS + CalculateAveragedLuminanceCS (when pc->UseAutoExposure and the
 * output type is LIT) then PostProcessCS on `in` (W*H float4, or W*H floats when inIsR32).  All outputs nullable. */
void tbo_post_process(const TbPostConstants* pc, const float* in, int inIsR32, float* outRgba, uint8_t* outRgba8, float* averagedOut, uint32_t* histogramOut);

/* Real-time chain (rt_ref.cpp); surfaces are W*H float4, row 0 = top; momentHistory / outMoment may be NULL when
 * OutputMomentInformation is 0 */
void tbo_temporal(const TbTemporalConstants* constants, const float* history, const float* current, const float* worldPos, const float* prevWorldPos,
                  const float* momentHistory, const float* normals, float* out, float* outMoment);
void tbo_denoise(const TbDenoiserConstants* constants, const float* input, const float* normals, const float* positions, const float* undenoised, float* out);
void tbo_composite(uint32_t W, uint32_t H, const float* albedo, const float* lighting, const float* emissive, float* out);

/* IsValidHit alpha test on candidate hits of non-opaque geometry (off by default, like the reference's software path) */
void tbo_set_alpha_test(int enabled);

/* serial restatement of the fallback layer's top-level build (oracle/bvh_ref.cpp); returns the image size or < 0 */
int64_t tbo_build_tlas(const float* objectToWorld /* 12 per instance */, const float* rootBoxes /* min xyz, max xyz per instance */, const uint32_t* blasIndex,
                       const uint32_t* hitGroupBase, uint32_t numInstances, uint8_t* out, uint64_t capacity);
float tbo_math(int fn, float a, float b); /* 0 sin 1 cos 2 acos 3 atan2 4 exp 5 log 6 pow 7 sqrt 8 exp2 9 log2 10 asin */
void tbo_math_array(int fn, uint32_t n, const float* a, const float* b /* nullable */, float* out); /* + 14 min 15 max 16 frac 17 floor 18 rcp */
void tbo_camera_ray(const TbPerFrameConstants* constants, float lensHeight, uint32_t width, uint32_t height,
                    float pixelX, float pixelY, float jitterX, float jitterY, float origin[3], float dir[3]);

/* Serial restatement of the fallback layer's LBVH build (oracle/bvh_ref.cpp). Writes a layout-A
 * image into out (capacity bytes); returns bytes written or <0. */
int64_t tbo_build_lbvh(const float* positions /*3 per vertex, already world space*/,
                       const uint32_t* triVertexIndex /*3 per tri, absolute*/,
                       const uint32_t* triGeometry, const uint32_t* triPrimitive,
                       const uint32_t* triFlags, uint32_t numTriangles, uint8_t* out, uint64_t capacity);

/* The same with the fallback layer's treelet passes (TreeletReorder.cpp:38-109) between hierarchy and fit;
 * treeletPasses = 3 is the tree a PREFER_FAST_TRACE build gives the software traversal. */
int64_t tbo_build_lbvh2(const float* positions, const uint32_t* triVertexIndex, const uint32_t* triGeometry,
                        const uint32_t* triPrimitive, const uint32_t* triFlags, uint32_t numTriangles,
                        uint32_t treeletPasses, uint8_t* out, uint64_t capacity);

/* BVHValidator restatement (BVHValidator.cpp:60-190): 0 if valid, else a negative code. */
int tbo_validate_bvh(const uint8_t* bvh, uint32_t bvhBytes, const float* positions,
                     const uint32_t* triVertexIndex, uint32_t numTriangles, uint32_t* maxDepth);

#ifdef __cplusplus
}
#endif
#endif
