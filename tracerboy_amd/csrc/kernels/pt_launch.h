/* pt_launch.h -- C launch interface between the host context and the kernel translation units. */
#pragma once
#include <hip/hip_runtime_api.h>
#include "pt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif
hipError_t pt_launch_persistent(hipStream_t stream, const TbDeviceScene* ds, const TbPerFrameConstants* pf, const TbDeviceTargets* tg, uint32_t W, uint32_t H,
                                uint32_t firstFrame, uint32_t numFrames, const TbTileMap* tiles, int sceneInLds, int countRays);
hipError_t pt_launch_trace_closest(hipStream_t stream, const TbDeviceScene* ds, uint32_t n, const float* origins, const float* dirs, float* outT, int* outMat,
                                   float* outBary, uint32_t* outPrim, uint32_t* outGeom, float* outNormal, float* outUV, uint32_t* outBoxes, uint32_t* outTris);
/* TbDeviceTargets::regionOrder (1 + regions x numGroups words) from TbDeviceTargets::regionCost; keys: regions words of scratch; lateFrom: the first
 * position of the usual list from which a counted region's items move to the front (pt_kernels.hip region_order_kernel) */
hipError_t pt_launch_region_order(hipStream_t stream, const uint32_t* cost, const TbTileMap* tiles, uint32_t W, uint32_t H, uint32_t regions,
                                  uint32_t numGroups, uint32_t lateFrom, uint32_t* order, uint32_t* keys);
hipError_t pt_launch_accumulate_samples(hipStream_t stream, const TbFloat4* samples, uint32_t W, uint32_t H, uint32_t firstFrame, uint32_t numFrames,
    const TbTileMap* tiles,
                                        TbFloat4* output, TbFloat4* jittered);
hipError_t pt_launch_device_math(hipStream_t stream, int fn, uint32_t n, const float* a, const float* b, float* out);
hipError_t pt_launch_unpack_gathered(hipStream_t stream, const TbFloat4* gathered, size_t capacity, TbFloat4* full, uint32_t W, uint32_t H, uint32_t world,
    uint32_t tileW, uint32_t tileH);
hipError_t pt_launch_pack_owned(hipStream_t stream, const TbFloat4* full, TbFloat4* packed, uint32_t W, uint32_t H, const TbTileMap* tiles,
    uint32_t numOwnedTiles);
/* GPU LBVH build (bvh_kernels.hip); every pointer is a device pointer */
size_t bvh_gpu_scratch_bytes(uint32_t N);
hipError_t bvh_gpu_build(hipStream_t stream, const float* positions, const uint32_t* triVertexIndex, const uint32_t* triGeometry, const uint32_t* triPrimitive,
                         const uint32_t* triFlags, uint32_t N, uint32_t treeletPasses, uint8_t* scratch, size_t scratchBytes, uint8_t* bvhA, TbNodeB* nodesB,
                             TbTriB* trisB,
                         uint32_t* rootHeight);
/* top level over instances on the GPU (two-level scenes); rootRef / rootHeight are device words */
size_t bvh_gpu_tlas_scratch_bytes(uint32_t M);
hipError_t bvh_gpu_build_tlas(hipStream_t stream, uint32_t M, const float* objectToWorld, const float* worldToObject, const uint32_t* blasIndex,
    const uint32_t* hitGroupBase,
                              const float* blasBoxes, uint8_t* scratch, size_t scratchBytes, uint8_t* tlasA, TbNodeB* topNodes, uint32_t* rootRef,
                                  uint32_t* rootHeight);
/* real-time chain (rt_kernels.hip): temporal accumulation, one a-trous denoiser iteration, albedo composite */
hipError_t rt_launch_temporal(hipStream_t stream, const TbTemporalConstants* k, const TbFloat4* history, const TbFloat4* current, const TbFloat4* worldPos,
                              const TbFloat4* prevWorldPos, const TbFloat4* momentHistory, const TbFloat4* normals, TbFloat4* out, TbFloat4* outMoment);
hipError_t rt_launch_denoise(hipStream_t stream, const TbDenoiserConstants* k, const TbFloat4* input, const TbFloat4* normals, const TbFloat4* positions,
                             const TbFloat4* undenoised, TbFloat4* out);
hipError_t rt_launch_composite(hipStream_t stream, uint32_t W, uint32_t H, const TbFloat4* albedo, const TbFloat4* lighting, const TbFloat4* emissive,
    TbFloat4* out);
/* output stage (post_kernels.hip): optional histogram + average (auto exposure), then PostProcessCS */
hipError_t post_launch(hipStream_t stream, const TbPostConstants* pc, const TbFloat4* in, const float* inR32, const TbFloat4* aux,
                       uint32_t* histogram, float* averaged, TbFloat4* out, uint32_t* outRgba8);
#ifdef __cplusplus
}
#endif
