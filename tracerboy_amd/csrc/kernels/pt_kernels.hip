/* pt_kernels.hip -- gfx950 kernels of the path tracer and their launchers.
 *
 * pt_persistent<SCENE_LDS, COUNT>: the replacement of SoftwareRayTraceCS::main
 * (/root/reference/TracerBoy/SoftwareRayTraceCS.hlsl:9-51).  One lane owns one pixel for ALL
 * frames of the launch ("persistent" lanes with path regeneration): when its path ends it
 * accumulates the sample exactly like RayGenCommon.h:704-727 and immediately starts the next
 * frame's path, so a wavefront only idles at the very end of the launch instead of at the end of
 * every frame's longest path.  Per-pixel sample order is frame order, so the fp32 accumulation is
 * bit-identical to the reference's one-dispatch-per-frame loop.
 *   - every loop iteration casts exactly one ray per live lane (bounce, shadow or SSS ray), through
 *     ONE traversal instance -> no duplicated traversal code, ray types share the wave;
 *   - wave = 8x8 pixel tile (the reference's thread-group shape), block = 2x2 tiles;
 *   - traversal stack in LDS, [entry][lane] layout (bank-conflict free, unlike the reference's
 *     lane*16+entry, TraverseFunction.hlsli:147-154);
 *   - SCENE_LDS: scenes whose whole kernel-visible image (nodes, triangles, hit-group records,
 *     indices, vertices, materials, lights) fits the LDS budget are copied into LDS once per block
 *     with coalesced 16-B loads; all traversal and shading fetches are then ds_reads.
 */
#include <hip/hip_runtime.h>
#include "pt_common.hpp"
#include "pt_launch.h"

namespace {

/* Closest-hit batch for the traversal parity tests (IntersectWithMaxDistance, RayGenCommon.h:365-414); NODEC: through the compact
 * layout-C nodes (one-level scenes) */
template <bool NODEC>
__global__ __launch_bounds__(BLOCK) void trace_closest_kernel(TbDeviceScene ds, uint32_t n, const float* origins, const float* dirs, float* outT, int* outMat,
                                                               float* outBary, uint32_t* outPrim, uint32_t* outGeom, float* outNormal, float* outUV,
                                                               uint32_t* outBoxes, uint32_t* outTris)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* stack = (uint32_t*)smem + threadIdx.x;
    SceneRefs sc; make_refs<false>(sc, ds, nullptr);
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    tb3 o = ld3(origins + 3 * i), d = ld3(dirs + 3 * i);
    Hit h; uint32_t nb = 0, nt = 0;
    bool hit;
    if (NODEC) hit = traverse<true, true, false, true>(sc, ds, o, d, h, stack, BLOCK, nb, nt);
    else hit = ds.numInstances ? traverse_instanced<true, true>(sc, ds, o, d, h, stack, BLOCK, nb, nt) : traverse<true, true>(sc, ds, o, d, h, stack, BLOCK,
        nb, nt);
    outT[i] = hit ? h.t : -1.0f;
    if (outBary) { outBary[2 * i] = hit ? h.u : 0.0f; outBary[2 * i + 1] = hit ? h.v : 0.0f; }
    if (outPrim) outPrim[i] = hit ? h.prim : 0xffffffffu;
    if (outGeom) outGeom[i] = hit ? h.geom : 0xffffffffu;
    if (outBoxes) outBoxes[i] = nb;
    if (outTris) outTris[i] = nt;
    Surface s; s.normal = tb3_splat(0.0f); s.u = s.v = 0.0f; s.material = -1;
    if (hit) fetch_surface(sc, h, s, false);
    if (outMat) outMat[i] = s.material;
    if (outNormal) { outNormal[3 * i] = s.normal.x; outNormal[3 * i + 1] = s.normal.y; outNormal[3 * i + 2] = s.normal.z; }
    if (outUV) { outUV[2 * i] = s.u; outUV[2 * i + 1] = s.v; }
}

__global__ void device_math_kernel(int fn, uint32_t n, const float* a, const float* b, float* out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
    switch (fn) {
    case 0: r = tb_sin(x); break; case 1: r = tb_cos(x); break; case 2: r = tb_acos(x); break; case 3: r = tb_atan2(x, y); break;
    case 4: r = tb_exp(x); break; case 5: r = tb_log(x); break; case 6: r = tb_pow(x, y); break; case 7: r = tb_sqrt(x); break;
    case 8: r = tb_exp2(x); break; case 9: r = tb_log2(x); break; case 10: r = tb_asin(x); break;
    case 11: r = x / y; break; case 12: r = tb_frac(tb_sin(x + y) * 43758.5453123f); break; case 13: r = hash13(x, y, 0.0f); break;
    case 14: r = tb_min(x, y); break; case 15: r = tb_max(x, y); break; case 16: r = tb_frac(x); break; case 17: r = tb_floor(x); break;
        case 18: r = tb_rcp(x); break;
    default: break;
    }
    out[i] = r;
}

/* tile-major pack of the pixels this rank owns (tb_pack_owned_device) */
__global__ void pack_owned_kernel(const TbFloat4* full, TbFloat4* packed, uint32_t W, uint32_t H, TbTileMap tiles, uint32_t numOwnedTiles)
{
    uint32_t tilesX = (W + tiles.tileW - 1) / tiles.tileW;
    uint32_t local = blockIdx.x; /* index among owned tiles */
    if (local >= numOwnedTiles) return;
    uint32_t t = tiles.rank + local * tiles.world;
    uint32_t tx = t % tilesX, ty = t / tilesX;
    uint32_t x0 = tx * tiles.tileW, y0 = ty * tiles.tileH;
    uint32_t tw = min(tiles.tileW, W - x0), th = min(tiles.tileH, H - y0);
    size_t base = (size_t)local * tiles.tileW * tiles.tileH;
    for (uint32_t i = threadIdx.x; i < tw * th; i += blockDim.x) {
        uint32_t lx = i % tw, ly = i / tw;
        packed[base + i] = full[(size_t)(y0 + ly) * W + (x0 + lx)];
    }
}

/* rank 0 of a tile split: the inverse of pack_owned_kernel over the gathered buffers of all ranks (tb_unpack_gathered_device).
 * gathered = world x capacity pixels, rank r's packed tiles at r * capacity; one workgroup per tile of the frame. */
__global__ void unpack_gathered_kernel(const TbFloat4* gathered, size_t capacity, TbFloat4* full, uint32_t W, uint32_t H, uint32_t world, uint32_t tileW,
    uint32_t tileH)
{
    const uint32_t tilesX = (W + tileW - 1) / tileW, t = blockIdx.x;
    const uint32_t rank = t % world, local = t / world;
    const uint32_t x0 = (t % tilesX) * tileW, y0 = (t / tilesX) * tileH;
    const uint32_t tw = min(tileW, W - x0), th = min(tileH, H - y0);
    const TbFloat4* src = gathered + (size_t)rank * capacity + (size_t)local * tileW * tileH;
    for (uint32_t i = threadIdx.x; i < tw * th; i += blockDim.x) {
        const uint32_t lx = i % tw, ly = i / tw;
        full[(size_t)(y0 + ly) * W + (x0 + lx)] = src[i];
    }
}

/* Ordered sum of the frame-group mode's sample buffer (TbDeviceTargets::samples): RayGenCommon.h:721-727 per pixel,
 * frames in order.  One lane per owned pixel; reads numFrames x 16 B, HBM-bound. */
__global__ __launch_bounds__(256) void accumulate_samples_kernel(const TbFloat4* samples, uint32_t W, uint32_t H, uint32_t firstFrame, uint32_t numFrames,
    TbTileMap tiles,
                                                                 TbFloat4* output, TbFloat4* jittered)
{
    const uint32_t n = W * H;
    for (uint32_t pix = blockIdx.x * 256u + threadIdx.x; pix < n; pix += gridDim.x * 256u) {
        if (tiles.world > 1) {
            const uint32_t x = pix % W, y = pix / W, tx = (W + tiles.tileW - 1) / tiles.tileW;
            if ((((y / tiles.tileH) * tx + (x / tiles.tileW)) % tiles.world) != tiles.rank) continue;
        }
        TbFloat4 acc = {0, 0, 0, 0}, jacc = {0, 0, 0, 0};
        if (firstFrame > 0) { acc = output[pix]; jacc = jittered[pix]; }
        for (uint32_t f = 0; f < numFrames; f++) {
            const TbFloat4 s = samples[(size_t)f * n + pix];
            const uint32_t frame = firstFrame + f;
            const float o3 = tb_abs(s.w);
            const bool coinLow = (__float_as_uint(s.w) >> 31) != 0;
            if (frame == 0) acc = TbFloat4{0, 0, 0, 0};
            acc = TbFloat4{s.x + acc.x, s.y + acc.y, s.z + acc.z, o3 + acc.w};
            if (frame == 0 || coinLow) {
                if (frame == 0) jacc = TbFloat4{0, 0, 0, 0};
                jacc = TbFloat4{s.x + jacc.x, s.y + jacc.y, s.z + jacc.z, o3 + jacc.w};
            }
        }
        output[pix] = acc; jittered[pix] = jacc;
    }
}

} // namespace

/* TbDeviceTargets::regionOrder from TbDeviceTargets::regionCost (pt_scene.h): one workgroup of 1024.  order[0] = items moved to the front, order[1 ...] =
 * a permutation of the launch's items (group << 20 | region, claim_work_item's usual list: group by group, regions ascending): first the items of
 * COUNTED regions that the usual list holds at position lateFrom or later, then everything else, both parts in their usual order -- neighbours stay
 * neighbours in the list (a first version sorted counted regions by count with atomics: what ran side by side was then scattered over the frame,
 * and the whole-frame van-class step lost 1.2 % to the caches), and a launch much longer than its longest path keeps its usual order except for
 * its end.  keys (regions words of scratch) holds the ONE reading of every count that all passes use: other launches keep counting while this
 * runs, and an item read as moved in one pass and as not moved in the next would be handed out twice or never. */
__global__ __launch_bounds__(1024) void region_order_kernel(const uint32_t* __restrict__ cost, TbTileMap tiles, uint32_t W, uint32_t H,
    uint32_t regions, uint32_t numGroups, uint32_t lateFrom, uint32_t* __restrict__ order, uint32_t* __restrict__ keys)
{
    __shared__ uint32_t waveSum[16], base[2], moved;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, items = regions * numGroups;
    if (t == 0) moved = 0;
    for (uint32_t i = t; i < regions; i += 1024u) {
        uint32_t bx, by; block_region(tiles, W, H, i, bx, by);
        keys[i] = __hip_atomic_load(cost + ((by & 1023u) << 10 | (bx & 1023u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
    }
    __syncthreads(); /* (a workgroup's own global writes are visible to it after the barrier) */
    uint32_t mine = 0;
    for (uint32_t i = lateFrom + t; i < items; i += 1024u) mine += keys[i % regions];
    if (mine) atomicAdd(&moved, mine);
    __syncthreads();
    if (t == 0) { order[0] = moved; base[1] = 0; base[0] = moved; }
    __syncthreads();
    for (uint32_t i0 = 0; i0 < items; i0 += 1024u) {
        const uint32_t i = i0 + t, group = i < items ? i / regions : 0u, region = i - group * regions;
        const bool in = i < items, front = in && i >= lateFrom && keys[region] != 0u;
        const unsigned long long m = __ballot(front);
        const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) waveSum[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t frontBefore = before, total = 0;
        for (uint32_t w = 0; w < 16u; w++) { const uint32_t v = waveSum[w]; if (w < wave) frontBefore += v; total += v; }
        /* (system scope, as claim_work_item reads it) */
        if (front) __hip_atomic_store(order + 1u + base[1] + frontBefore, group << 20 | region, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (in) __hip_atomic_store(order + 1u + base[0] + (t - frontBefore), group << 20 | region, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        if (t == 0) { base[1] += total; base[0] += 1024u - total; }
        __syncthreads();
    }
}

extern "C" {

hipError_t pt_launch_accumulate_samples(hipStream_t stream, const TbFloat4* samples, uint32_t W, uint32_t H, uint32_t firstFrame, uint32_t numFrames,
    const TbTileMap* tiles,
                                        TbFloat4* output, TbFloat4* jittered)
{
    hipLaunchKernelGGL(accumulate_samples_kernel, dim3(2048), dim3(256), 0, stream, samples, W, H, firstFrame, numFrames, *tiles, output, jittered);
    return hipGetLastError();
}


hipError_t pt_launch_region_order(hipStream_t stream, const uint32_t* cost, const TbTileMap* tiles, uint32_t W, uint32_t H, uint32_t regions,
                                  uint32_t numGroups, uint32_t lateFrom, uint32_t* order, uint32_t* keys)
{
    hipLaunchKernelGGL(region_order_kernel, dim3(1), dim3(1024), 0, stream, cost, *tiles, W, H, regions, numGroups, lateFrom, order, keys);
    return hipGetLastError();
}

hipError_t pt_launch_trace_closest(hipStream_t stream, const TbDeviceScene* ds, uint32_t n, const float* origins, const float* dirs, float* outT, int* outMat,
                                   float* outBary, uint32_t* outPrim, uint32_t* outGeom, float* outNormal, float* outUV, uint32_t* outBoxes, uint32_t* outTris)
{
    size_t lds = (size_t)ds->stackDepth * BLOCK * 4;
    if (ds->nodesC && ds->numInstances) return hipErrorInvalidValue; /* layout C is built for one-level scenes */
    const void* fn = ds->nodesC ? (const void*)trace_closest_kernel<true> : (const void*)trace_closest_kernel<false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (ds->nodesC) hipLaunchKernelGGL(trace_closest_kernel<true>, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), lds, stream, *ds, n, origins, dirs, outT,
        outMat, outBary, outPrim, outGeom, outNormal, outUV, outBoxes, outTris);
    else hipLaunchKernelGGL(trace_closest_kernel<false>, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), lds, stream, *ds, n, origins, dirs, outT, outMat, outBary,
        outPrim, outGeom, outNormal, outUV, outBoxes, outTris);
    return hipGetLastError();
}

hipError_t pt_launch_device_math(hipStream_t stream, int fn, uint32_t n, const float* a, const float* b, float* out)
{
    hipLaunchKernelGGL(device_math_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, n, a, b, out);
    return hipGetLastError();
}

hipError_t pt_launch_unpack_gathered(hipStream_t stream, const TbFloat4* gathered, size_t capacity, TbFloat4* full, uint32_t W, uint32_t H, uint32_t world,
    uint32_t tileW, uint32_t tileH)
{
    const uint32_t tilesTotal = ((W + tileW - 1) / tileW) * ((H + tileH - 1) / tileH);
    if (tilesTotal == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_gathered_kernel, dim3(tilesTotal), dim3(256), 0, stream, gathered, capacity, full, W, H, world, tileW, tileH);
    return hipGetLastError();
}

hipError_t pt_launch_pack_owned(hipStream_t stream, const TbFloat4* full, TbFloat4* packed, uint32_t W, uint32_t H, const TbTileMap* tiles,
    uint32_t numOwnedTiles)
{
    if (numOwnedTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_owned_kernel, dim3(numOwnedTiles), dim3(256), 0, stream, full, packed, W, H, *tiles, numOwnedTiles);
    return hipGetLastError();
}

} // extern "C"
