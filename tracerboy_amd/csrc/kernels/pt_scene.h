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
/* the fields of HitGroupShaderRecord the path reads; offsets in elements */
struct __attribute__((aligned(16))) TbDevHitGroup { uint32_t MaterialIndex, vFirst, iFirst, pad; };
struct __attribute__((aligned(16))) TbDevMaterial { TbMaterial m; uint32_t pad[3]; };               /* 84 -> 96 B */
struct __attribute__((aligned(16))) TbDevLight { TbLight l; uint32_t pad[2]; };                     /* 104 -> 112 B */

struct TbDeviceScene {
    const TbNodeB* nodes;        /* layout B, breadth-first order: the first `ldsNodes` are the top of the tree */
    const TbTriB* tris;
    const TbNodeC* nodesC;       /* layout C (tb_abi.h): the same tree, same storage order, 32-B nodes with boxes on a 16-bit grid; null unless built */
    TbQuantFrame quant;          /* the grid of nodesC */
    uint32_t rootRef;            /* child-ref of the root (two-level scenes: of the top level, whose leaf refs are LEAF | instance index) */
    const TbInstanceB* instances; uint32_t numInstances; /* two-level scenes (flatten_instances = 0), else null / 0 */
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
    /* what the scene's materials can ask of a hit's vertices beyond their normals: bit 0 = some material has an albedo / emissive / specular /
     * alpha texture (the interpolated uv is read), bit 1 = some material has a normal map (the tangent is).  A feature set with textures compiled in
     * serves scenes without any (a glass scene needs the `sss` set): path_on_closest then fetches 16 B per vertex instead of 32. */
    uint32_t textureUse;
    uint32_t parkMin;            /* while-while scheduling of traverse(): leave the inner-node loop when fewer lanes than this still descend */
    uint32_t stackDepth;         /* entries per lane of the traversal stack held in LDS (= bvh max depth, unless the stack is split) */
    /* split stack (HYBRID kernels, frame-group launches of the higher-occupancy copies on trees too deep for their LDS share):
     * entries >= stackDepth live in global memory, entry e of lane L at stackOverflow[(e - stackDepth) * stackOverflowLanes + L],
     * L = blockIdx.x * 256 + threadIdx.x */
    uint32_t* stackOverflow;     uint32_t stackOverflowLanes;
    /* whole-scene-in-LDS image (small scenes): byte offsets inside one contiguous device blob */
    const uint8_t* ldsBlob;      uint32_t ldsBlobBytes;
    uint32_t offNodes, offTris, offHitGroups, offIndices, offVertices, offMaterials, offLights;
};

/* Output surfaces of one dispatch (u0..u7, u10 of SharedRaytracing.h:13-23) */
#ifndef TB_COSTLY_STEP
#define TB_COSTLY_STEP 8 /* an interior walk counts for its region (TbDeviceTargets::regionCost) when it reaches this step */
#endif
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
    /* Frame-group mode (nullable): with `samples` set the launch is a resident grid whose workgroups draw work items
     * (16x16 region x frameGroup consecutive frames) from *workCounter and whose lanes draw (pixel, frame) pairs of the
     * workgroup's items from a counter in LDS (pt_persistent.inc); every finished sample is written to
     * samples[(frame - firstFrame) * W * H + pixel] as (rgb*w, +-w; sign bit = jitter coin < 0.5) and
     * accumulate_samples_kernel then sums them in frame order, which keeps the fp32 accumulation of RayGenCommon.h:704-727
     * bit for bit while no lane waits for its neighbours' longer paths. */
    TbFloat4* samples; uint32_t frameGroup; uint32_t* workCounter;
    /* 1: the work items of a region shrink towards the end of the launch (tb_fg_groups below): half of the frames in groups of frameGroup, half of
     * the rest in groups half that size, ... single frames last -- what is left bound to slow workgroups when the lists run dry is then a few
     * hundred samples instead of two whole groups (docs/experiments/r6.md: the end of a launch that has the chip to itself) */
    uint32_t fgGuided;
    /* frame-group mode: slotLogCap entries per workgroup of the launch (at most 16 per CU), zeroed by the launcher; every slot a workgroup binds is recorded
     * here (pt_persistent.inc) */
    unsigned long long* slotLog; uint32_t slotLogCap;
    uint32_t bandedItems;  /* claim_work_item (pt_common.hpp): every XCD's list covers a contiguous eighth of the regions instead of every eighth region */
    /* Primary-visibility pre-pass (frame-group mode, nullable).  The camera ray of a sample depends on (x, y, frame) alone, and the
     * rays of an 8x8 pixel tile at one frame walk almost the same nodes, whereas inside the lock-step kernel a lane's primary ray is
     * walked together with its neighbours' incoherent bounce rays.  With primaryHits set, pt_primary (pt_persistent.inc) has walked
     * every camera ray of the launch, one tile per wave, and left the closest hit in record i = (frame - firstFrame) * W * H + pixel --
     * four 8-byte words: (t or -1, u), (v, primitive), (hit-group index, launchEpoch), the XOR of the three -- where the lane that draws the sample picks it
     * up instead of walking.  Same camera ray, same walk, same hit, same bits; one lock-step trip less per path. */
    unsigned long long* primaryHits;
    uint32_t* debugCounters; /* nullable; [0] = hit records of the pre-pass that failed their epoch / check word and were walked again (pt_persistent.inc) */
    /* frame-group mode: a number no other launch on these buffers has had (host counter); stamps the slot-log entries (low byte) and the hit records */
    uint32_t launchEpoch;
    /* 1: primaryHits holds FIRST-BOUNCE STATE records -- 96 B per sample, written by pt_first (pt_persistent.inc), which runs a sample's whole
     * first bounce with one 8x8 pixel tile per wave: camera ray, shading of the first hit, the feeler of that hit (a quarter of all rays and,
     * being shadow feelers, 29 % of all walk steps of the van-class scene: they start next to each other and converge on the same lights)
     * and the scatter.  The lock-step kernel takes the state where pt_first left it.  Same functions in the same order: same bits. */
    uint32_t firstBounce;
    /* Compact hit records (nonzero hitStamp): 16 B per sample -- (t or -1, u, v, stamp << (hitPrimBits + hitGeomBits) | hit group << hitPrimBits |
     * primitive) -- one load for the lane that draws the sample instead of two (the lock-step kernels are bound by the vector-memory issue rate).
     * hitStamp is never 0 (a cleared buffer is nobody's record) and changes with every launch; the 16-B piece is written and read whole, so it
     * needs no check word.  A hit whose indices do not fit their fields is stored with stamp 0: its lane walks the camera ray itself. */
    uint32_t hitStamp, hitPrimBits, hitGeomBits;
    /* What path_begin computes from the launch's camera and frame size alone, computed once on the host with the same functions (tb_vec.h: IEEE
     * division and square root on both sides, same bits): the pinhole, 1 / W, 1 / H, W / H -- four divisions and a square root of ~250 instructions
     * that the lock-step kernel ran for the few lanes of a wave that had just finished a path.  camPre = 0: not filled, path_begin computes them. */
    float camFocal[3], camInvResX, camInvResY, camAspect; uint32_t camPre;
    /* Costly regions first (frame-group mode of the feature sets with interior walks, nullable).  A launch ends when its longest path does, and a
     * path that walks a hundred steps inside glass takes milliseconds alone on its wave: started with the last items of the list it is most of a
     * small launch's time (docs/experiments/r6.md: a rank of 8's 8-spp launch of vw-van is dry after 3.2 ms and ends after 10.1).  Paths are long
     * where they were long a frame ago, so the kernels count, per 16x16 region of the FRAME, the interior walks that reach step TB_COSTLY_STEP
     * (regionCost[ry << 10 | rx]: 2^20 words, a region's coordinates have 10 bits each; cleared with the scene), and region_order_kernel
     * (pt_kernels.hip) turns the counts into the order in which the NEXT launch hands its items out: regionOrder[1 + i] = group << 20 | region
     * of the i-th claim, a permutation of the usual list in which the items of counted regions that would come late are first.  Any permutation
     * gives the same picture (a sample depends on pixel and frame alone); only the end of the launch moves. */
    uint32_t* regionCost; const uint32_t* regionOrder;
};

/* Split-role kernel (pipeline 4, pt_split.inc): a workgroup is `travWaves` traversal waves followed by `shadeWaves` shading waves.
 * Shading waves own the paths (state in registers), write each pending ray -- origin, direction and the five quotients of
 * GetRayData -- into the lane's slot in LDS and queue the slot's number; traversal waves draw rays from that queue whenever lanes
 * fall idle, walk them (resumable per-lane walk, one body per step chosen by majority) and leave the closest hit in the slot. */
struct TbSplitParams {
    uint32_t travWaves, shadeWaves;
    uint32_t readyMin;      /* a shading wave goes on when this many of its lanes have all their rays back (or every lane that waits) */
    uint32_t refillMin;     /* a traversal wave asks the queue for rays when this many of its lanes are idle */
    uint32_t innerWeight, leafWeight; /* a step runs the inner-node body when lanesAtInnerNodes * leafWeight >= lanesAtLeaves * innerWeight */
    /* the traversal waves are the LAST waves of the workgroup (younger: they yield issue slots to the shading waves); s_setprio 3 for the shading waves */
    uint32_t travLast, shadePrio;
    uint32_t ringCap;       /* entries of the ray queue: a power of two >= 2 x 128 x shadeWaves */
    uint32_t spinLimit;     /* every wait is bounded: a wave that has slept this often raises *abortFlag and the launch winds down */
    uint32_t* abortFlag;    /* host-visible word, 0 while all is well */
    unsigned long long* prof; /* option split_profile: 16 counters (pt_split.inc SP_*), filled by the profiling copy of the kernel; else null */
};

struct TbTileMap { /* multi-GPU tile ownership: tile t is rendered iff t % world == rank; tileW, tileH multiples of 16 */
    uint32_t rank, world, tileW, tileH;
};

/* Workgroup -> 16x16 pixel region of the persistent kernels.  One GPU: row-major over the frame.  Tile split
 * (world > 1): ONLY the rank's own tiles are launched, tile-major -- the k-th owned tile is rank + k * world and a
 * tile is tileW/16 x tileH/16 consecutive workgroups.  The hardware deals consecutive workgroups round-robin to the 8
 * XCDs, so every XCD gets real work; launching the whole frame and letting foreign workgroups exit left half of the
 * XCDs idle for every even world size (a rank's tiles then all sit in columns of one parity: measured 51 / 42 / 32 %
 * efficiency at 2 / 4 / 8 ranks, scripts/tile_split_timing.py). */
/* Frame groups of a region in a frame-group launch of F frames.  Uniform (guided = 0, or a launch shorter than two groups): ceil(F / 2^lg0)
 * groups of 2^lg0 frames.  Guided: equal groups up to the last two to three groups' worth of frames, over which the sizes halve -- one more
 * whole group, then groups half that size over half of what is left, and so on; the last stretch is single frames.  With group == ~0u
 * returns the number of groups; otherwise writes the group's first frame (relative to the launch) and log2 of its frame count.  Host (plan,
 * slot logs, grid) and device (fg_bind_next) share it. */
#if defined(__HIPCC__) || defined(__cplusplus)
static inline
#if defined(__HIPCC__)
__host__ __device__
#endif
uint32_t tb_fg_groups(uint32_t F, uint32_t lg0, uint32_t guided, uint32_t group, uint32_t* frame0, uint32_t* lg)
{
    if (!guided || F < (2u << lg0)) { if (group != 0xffffffffu) { *frame0 = group << lg0; *lg = lg0; } return (F + (1u << lg0) - 1u) >> lg0; }
    const uint32_t uniform = (F - (2u << lg0)) >> lg0; /* whole groups before the stretch that shrinks (2 to 3 groups' worth of frames) */
    if (group != 0xffffffffu && group < uniform) { *frame0 = group << lg0; *lg = lg0; return 0u; }
    uint32_t f = uniform << lg0, total = uniform, l = lg0;
    while (f < F) {
        const uint32_t left = F - f;
        const uint32_t n = l == 0u ? left : ((left / 2u) >> l); /* single frames run to the end; larger groups fill half of what is left */
        if (n == 0u) { l--; continue; }
        if (group != 0xffffffffu && group < total + n) { *frame0 = f + ((group - total) << l); *lg = l; return 0u; }
        total += n; f += n << l;
        if (l > 0u) l--;
    }
    return total;
}
#endif

#if defined(__HIPCC__) || defined(__cplusplus)
static inline
#if defined(__HIPCC__)
__host__ __device__
#endif
uint32_t tb_persistent_grid(uint32_t W, uint32_t H, const TbTileMap& t)
{
    if (t.world <= 1) return ((W + 15u) / 16u) * ((H + 15u) / 16u);
    const uint32_t tiles = ((W + t.tileW - 1) / t.tileW) * ((H + t.tileH - 1) / t.tileH);
    const uint32_t owned = t.rank < tiles ? (tiles - t.rank + t.world - 1) / t.world : 0u;
    return owned * (t.tileW / 16u) * (t.tileH / 16u);
}
#endif
