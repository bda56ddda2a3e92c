/* pt_scene.h -- device-side view of the scene and launch parameters shared by every kernel. */
#pragma once
#include "../../../include/tb_abi.h"
#include <stdint.h>

/* What the kernels read, as device pointers (the HIP replacement of the root signature at
 * /root/reference/TracerBoy/TracerBoy.cpp:568-664 / SharedRaytracing.h:3-53). */
/* Layout-B nodes are 64 B; in the LDS image they are stored 80 B apart: with a 64-B stride the 16-B pieces of different
 * nodes fall on only 4 distinct bank groups (ds_read_b128 serves 16 lanes per cycle), with 80 B on 16. */
#define TB_LDS_NODE_STRIDE 80u
/* Child refs in the DEVICE images are byte offsets / 16 instead of indices, so that an address is one shift-add: an
 * inner ref is node index * 4 (global image) or * 5 (LDS image), a leaf ref is TB_BVH_LEAF_FLAG | triangle index * 3. */
#define TB_DEVICE_REF_MASK 0x7fffffffu
/* The LDS image stores every triangle six times, once per axis order (kx, ky, kz) the watertight test can ask for
 * (copy kz*2 + (d[kz] < 0)): the per-lane component selects of RayTriangleIntersect (18 v_cndmask per test) disappear.
 * A leaf ref of the LDS image is TB_BVH_LEAF_FLAG | triangle index * 18; the lane adds copy * 3. */
#define TB_LDS_TRI_COPIES 6u

/* Device copies of the shading records, padded / trimmed to 16-B multiples so that each is fetched with 16-B loads
 * (a scattered 4-B load costs the vector memory pipe as much as a 16-B one).  The byte model of DESIGN.md keeps the
 * reference's record sizes. */
struct __attribute__((aligned(16))) TbDevHitGroup { uint32_t MaterialIndex, vFirst, iFirst, pad; }; /* the fields of HitGroupShaderRecord the path reads; offsets in elements */
struct __attribute__((aligned(16))) TbDevMaterial { TbMaterial m; uint32_t pad[3]; };               /* 84 -> 96 B */
struct __attribute__((aligned(16))) TbDevLight { TbLight l; uint32_t pad[2]; };                     /* 104 -> 112 B */

struct TbDeviceScene {
    const TbNodeB* nodes;        /* layout B, breadth-first order: the first `ldsNodes` are the top of the tree */
    const TbTriB* tris;
    uint32_t rootRef;            /* child-ref of the root */
    uint32_t numNodes, numTris;
    float rootCenter[3], rootHalf[3];
    const TbDevHitGroup* hitGroups;      uint32_t numHitGroups;
    const uint32_t* indexBuffer;         uint32_t numIndices;
    const float* vertexBuffer;           uint32_t numVertexFloats;
    const TbDevMaterial* materials;      uint32_t numMaterials;
    const TbTextureData* textureData;    uint32_t numTextureData;
    const TbDevLight* lights;            uint32_t numLights;
    const TbImageDesc* images;           uint32_t numImages;
    const TbFloat4* texelPool;
    const TbFloat4* envMap;              uint32_t envWidth, envHeight;
    const TbFloat4* blueNoise0;
    const TbFloat4* blueNoise1;
    TbConfigConstants config;
    uint32_t alphaTest;          /* option "alpha_test": IsValidHit filter on candidate hits of non-opaque geometry (full variant only) */
    uint32_t parkMin;            /* while-while scheduling of traverse(): leave the inner-node loop when fewer lanes than this still descend */
    uint32_t stackDepth;         /* entries per lane of the traversal stack (bvh max depth + 2) */
    /* whole-scene-in-LDS image (small scenes): byte offsets inside one contiguous device blob */
    const uint8_t* ldsBlob;      uint32_t ldsBlobBytes;
    uint32_t offNodes, offTris, offHitGroups, offIndices, offVertices, offMaterials, offLights;
};

/* Output surfaces of one dispatch (u0..u7, u10 of SharedRaytracing.h:13-23) */
struct TbDeviceTargets {
    TbFloat4* output;      /* u0 */
    TbFloat4* jittered;    /* u1 */
    TbFloat4* aovNormals;  /* u2, nullable */
    TbFloat4* aovWorldPos0, *aovWorldPos1; /* u3,u4 */
    TbFloat4* aovCustom;   /* u5 */
    float* aovDepth;       /* u6 */
    TbFloat4* aovEmissive; /* u7 */
    uint32_t* stats;       /* u10: [0]=ActiveWaves(groups) [1]=ActivePixels [2]=SelectedPixelDistance [3]=SelectedMaterialID */
    unsigned long long* rayStats; /* 7 x u64 (TbRayStats), nullable */
};

struct TbTileMap { /* multi-GPU tile ownership: tile t is rendered iff t % world == rank */
    uint32_t rank, world, tileW, tileH;
};
