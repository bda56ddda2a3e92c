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

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}


} // namespace
