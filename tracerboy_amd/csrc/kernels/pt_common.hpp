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
        sc.nodes = (const uint8_t*)ds.nodes; sc.tris = ds.tris; sc.trisPermuted = 0; sc.hitGroups = ds.hitGroups; sc.indices = ds.indexBuffer;
            sc.vertices = ds.vertexBuffer;
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
__device__ __noinline__ uint32_t claim_work_item(uint32_t* counters, uint32_t regions, uint32_t numGroups, uint32_t banded, const uint32_t* order = nullptr)
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
            if (item < regions * numGroups) {
                /* TbDeviceTargets::regionOrder: the launch's items in the order region_order_kernel made for it.  System scope like the slot log: the
                 * table is rewritten before every launch, and an XCD's L2 may still hold a line of it as the launch before last read it */
                if (order) return __hip_atomic_load(order + 1u + item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const uint32_t group = item / regions; return group << 20 | (item - group * regions);
            }
        }
    }
    return 0xffffffffu;
}

/* Frame-group bookkeeping of pt_persistent (the comments there say what and why): the RARE parts, out of line like claim_work_item and fed
 * with values rather than with the kernel's argument structs.  Inlined, this code -- run once per few thousand samples -- kept its
 * pointers, the log's capacity and the epoch alive through the path loop and cost every frame-group kernel 16-48 B of scratch per lane
 * and 3 % of its speed.  (Moving the per-path part out of line as well -- the ring lookup of every drawn sample -- was measured too: its
 * LDS reads become flat loads through the texture addresser and the 870 k scene lost 2.7 %.)
 * groupConst: regions, log2(frames per group: of the LARGEST group when the groups shrink, tb_fg_groups), frame groups per region, "a claim has found
 * nothing", tag base ((launch epoch & 0xff) << 16), frames of the launch | guided << 31;
 * bindState: binders under way, slots bound so far; both and the 8-entry ring live in the workgroup's LDS.
 * A slot's entry: tag << 40 | first frame of the group (relative, 15 bits) << 24 | log2(frames of the group) << 20 | region y << 10 | region x. */
__device__ __noinline__ void fg_bind_next(uint32_t* groupConst, uint32_t* bindState, unsigned long long* slotTable, uint32_t* workCounter,
    unsigned long long* logRow,
                                          uint32_t logCap, uint32_t banded, TbTileMap tiles, uint32_t W, uint32_t H, const uint32_t* order = nullptr)
{
    atomicAdd(&bindState[0], 1u);
    if (!((volatile uint32_t*)groupConst)[3]) {
        const uint32_t regions = ((volatile uint32_t*)groupConst)[0], lg = ((volatile uint32_t*)groupConst)[1], numGroups = ((volatile uint32_t*)groupConst)[2];
        uint32_t item = 0xffffffffu;
        /* room in the log for every binder under way (the host sizes a row for 8x the workgroup's fair share; a full row retires the workgroup) */
        if (((volatile uint32_t*)bindState)[1] + ((volatile uint32_t*)bindState)[0] <= logCap) item = claim_work_item(workCounter, regions, numGroups, banded, order);
        if (item == 0xffffffffu) ((volatile uint32_t*)groupConst)[3] = 1u;
        else {
            const uint32_t slot = atomicAdd(&bindState[1], 1u), group = item >> 20, region = item & 0xfffffu;
            uint32_t rx, ry;
            block_region(tiles, W, H, region, rx, ry);
            const uint32_t fg = ((volatile uint32_t*)groupConst)[5];
            uint32_t frame0 = group << lg, lgS = lg;
            if (fg >> 31) (void)tb_fg_groups(fg & 0x7fffffffu, lg, 1u, group, &frame0, &lgS); /* groups that shrink towards the end of the launch */
            const unsigned long long e = (unsigned long long)(((volatile uint32_t*)groupConst)[4] | ((slot + 1u) & 0xffffu)) << 40 |
                ((unsigned long long)frame0 << 24) | (unsigned long long)(lgS << 20) | (unsigned long long)(ry << 10) | rx;
            ((volatile unsigned long long*)slotTable)[slot & 7u] = e;
            __hip_atomic_store(logRow + slot, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __threadfence_block();
    atomicSub(&bindState[0], 1u);
}

/* a slot whose ring entry is not (or no longer) its own: the slot's entry, or 0 = ask again later, 1 = nothing left for this workgroup
 * (an entry has its tag in the top bits: never 0 or 1; by value -- an out-parameter would put the caller's copy on the stack) */
__device__ __noinline__ unsigned long long fg_resolve_slow(uint32_t s, uint32_t tag, const uint32_t* groupConst, const uint32_t* bindState,
    const unsigned long long* logRow)
{
    if (s < ((volatile const uint32_t*)bindState)[1]) {
        /* bound, but the ring has gone round since (or the binder is between taking the number and writing the entry): the log knows.
         * System scope, both sides: an agent-scope load is served by the XCD's L2, which may still hold the row as an earlier launch
         * left it (the launcher's memset went through another XCD's L2) */
        const unsigned long long v = __hip_atomic_load(logRow + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return (uint32_t)(v >> 40) != tag ? 0ull : v;
    }
    /* not bound: for good once a claim has failed, no binder is under way and the count still says so -- read in this order (a binder
     * announces itself before it looks at the flag) */
    if (!((volatile const uint32_t*)groupConst)[3]) return 0ull;
    __threadfence_block();
    if (((volatile const uint32_t*)bindState)[0] != 0) return 0ull;
    __threadfence_block();
    return s < ((volatile const uint32_t*)bindState)[1] ? 0ull : 1ull;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}


} // namespace
