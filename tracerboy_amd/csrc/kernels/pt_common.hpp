/* pt_common.hpp -- helpers shared by the kernel translation units. */
#pragma once
#include <hip/hip_runtime.h>
#include "pt_device.hpp"

namespace {
using namespace pt;

constexpr int BLOCK = 256;

template <bool SCENE_LDS>
__device__ __forceinline__ void make_refs(SceneRefs& sc, const TbDeviceScene& ds, const uint8_t* blob)
{
    if (SCENE_LDS) {
        sc.nodes = blob + ds.offNodes; sc.tris = (const TbTriB*)__builtin_assume_aligned(blob + ds.offTris, 16); sc.trisPermuted = 1;
        sc.hitGroups = (const TbDevHitGroup*)(blob + ds.offHitGroups); sc.indices = (const uint32_t*)(blob + ds.offIndices);
        sc.vertices = (const float*)(blob + ds.offVertices); sc.materials = (const TbDevMaterial*)(blob + ds.offMaterials);
        sc.lights = (const TbDevLight*)(blob + ds.offLights);
    } else {
        sc.nodes = (const uint8_t*)ds.nodes; sc.tris = ds.tris; sc.trisPermuted = 0; sc.hitGroups = ds.hitGroups; sc.indices = ds.indexBuffer; sc.vertices = ds.vertexBuffer;
        sc.materials = ds.materials; sc.lights = ds.lights;
    }
    sc.numHitGroups = ds.numHitGroups; sc.numIndices = ds.numIndices; sc.numVertexFloats = ds.numVertexFloats;
    sc.numMaterials = ds.numMaterials; sc.numLights = ds.numLights;
}

/* the 16x16 region of this workgroup (tb_persistent_grid, pt_scene.h); false when it lies outside the frame */
__device__ __forceinline__ bool block_region(const TbTileMap& tiles, uint32_t W, uint32_t H, uint32_t region, uint32_t& bx, uint32_t& by)
{
    if (tiles.world <= 1) { const uint32_t blocksX = (W + 15u) / 16u; bx = region % blocksX; by = region / blocksX; return true; }
    const uint32_t subX = tiles.tileW / 16u, perTile = subX * (tiles.tileH / 16u);
    const uint32_t k = region / perTile, sub = region % perTile;
    const uint32_t tilesX = (W + tiles.tileW - 1) / tiles.tileW, t = tiles.rank + k * tiles.world;
    bx = (t % tilesX) * subX + sub % subX; by = (t / tilesX) * (tiles.tileH / 16u) + sub / subX;
    return true;
}

/* Frame-group mode of pt_persistent: work items (region x frame group) come from 8 lists, a workgroup starting at list
 * blockIdx % 8 -- the XCD it runs on -- and moving on when a list is empty; one counter for the whole device would serialise
 * every claim on a single address.  banded = 0: item = 8 * count + list (every list is spread over the whole frame);
 * banded = 1: list q owns the q-th contiguous eighth of the regions, all frame groups of it, so that an XCD's L2 keeps seeing the
 * same part of the scene until its band is done and only then helps the others.  Returns group << 20 | region, or ~0 when
 * nothing is left.  Out of line: it runs once per few thousand samples and must not cost the path loop any registers. */
__device__ __noinline__ uint32_t claim_work_item(uint32_t* counters, uint32_t regions, uint32_t numGroups, uint32_t banded)
{
    for (uint32_t t = 0; t < 8; t++) {
        /* system scope: list q is counted mostly by the workgroups of XCD q, but a workgroup whose own list is empty takes from the
         * others'.  (Agent-scope counters were suspected when work items went missing and were not the cause -- the ray counters of
         * the counting launches are agent-scope adds from every XCD and equal the oracle's; the cause was the order in which slots
         * were bound, pt_persistent.inc bind_next.  The wider scope stays: a claim is made once per thousand samples.) */
        const uint32_t q = (blockIdx.x + t) & 7u, c = __hip_atomic_fetch_add(counters + q * 16u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (banded) {
            const uint32_t b0 = (uint32_t)(((unsigned long long)regions * q) >> 3), n = (uint32_t)(((unsigned long long)regions * (q + 1u)) >> 3) - b0;
            if (n && c < n * numGroups) { const uint32_t group = c / n; return group << 20 | (b0 + (c - group * n)); }
        } else {
            const uint32_t item = c * 8u + q;
            if (item < regions * numGroups) { const uint32_t group = item / regions; return group << 20 | (item - group * regions); }
        }
    }
    return 0xffffffffu;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}


} // namespace
