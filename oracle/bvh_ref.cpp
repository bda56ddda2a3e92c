/* bvh_ref.cpp -- serial CPU restatement of the fallback layer's bottom-level LBVH build and of its
 * BVH validator.  TEST INFRASTRUCTURE ONLY (see tb_oracle.h); the product's builder
 * (tracerboy_amd/csrc/host/bvh_build.cpp) is checked bit-for-bit against this one.
 *
 * Pipeline restated (paths relative to /root/reference/D3D12RaytracingFallback/src/):
 *   LoadPrimitives        BottomLevelLoadTriangles.hlsli:88-130      triangle -> Primitive
 *   scene AABB            CalculateSceneAABBFromPrimitives.hlsl:16-41
 *   Morton codes          CalculateMortonCodesForPrimitives.hlsl:17-30, CalculateMortonCodesBindings.h:116-149
 *   sort                  BitonicSort.cpp (key = Morton code; the bitonic network is not stable, so
 *                         the order of equal keys is implementation-defined there; this build
 *                         defines it: ties broken by original triangle index)
 *   hierarchy             BuildBVHSplits.hlsli:18-171 (Karras 2012)
 *   AABB fit              ComputeAABBs.hlsli:69-172, RayTracingHelper.hlsli:229-285
 *                         (smaller subtree on the left; on equal counts the reference's result
 *                         depends on which thread arrives second -- this build keeps Karras order)
 *   treelet passes        ClearBuffers.hlsl, FindTreelets.hlsl:27-88, TreeletReorder.hlsl:38-311, TreeletReorder.cpp:38-109,
 *                         TreeletReorderBindings.h:50-111 (Karras & Aila 2013, treelets of 7 leaves; 3 passes with
 *                         MinTrianglesPerTreelet 7, 14, 28 for PREFER_FAST_TRACE, which TracerBoy.cpp:1970 asks for).
 *                         Run between the hierarchy and the fit when treeletPasses > 0 (tbo_build_lbvh2).  The one
 *                         race in the reference that can change the result is fixed by rule: a group gives up after
 *                         33 treelets (`while (i++ < 32)`, TreeletReorder.hlsl:288-310) and which of two groups meeting
 *                         at a node goes on is whichever arrives second; here the one that has done fewer goes on.
 */
#include "tb_oracle.h"
#include "../include/tb_vec.h"

#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

struct Box { tb3 center, halfDim; };

inline int clz32(uint32_t v) { return v == 0 ? 32 : __builtin_clz(v); }

/* CalculateMortonCodesBindings.h:116-149 */
uint32_t MortonFromUnit(tb3 unit)
{
    const float maxCoord = 1024.0f;
    float ax = tb_min(tb_max(unit.x * maxCoord, 0.0f), maxCoord - 1);
    float ay = tb_min(tb_max(unit.y * maxCoord, 0.0f), maxCoord - 1);
    float az = tb_min(tb_max(unit.z * maxCoord, 0.0f), maxCoord - 1);
    uint32_t coords[3] = {(uint32_t)ay, (uint32_t)ax, (uint32_t)az};
    uint32_t code = 0;
    for (uint32_t bit = 0; bit < 10; bit++)
        for (uint32_t axis = 0; axis < 3; axis++)
            if (coords[axis] & (1u << bit)) code |= 1u << (bit * 3 + axis);
    return code;
}

struct Builder {
    uint32_t n;
    std::vector<uint32_t> codes; /* sorted */
    int lcp(int64_t a, int64_t b) const /* BuildBVHSplits.hlsli:33-54 */
    {
        if (a < 0 || b < 0 || a >= (int64_t)n || b >= (int64_t)n) return -1;
        uint32_t ca = codes[(size_t)a], cb = codes[(size_t)b];
        if (ca != cb) return clz32(ca ^ cb);
        return clz32((uint32_t)a ^ (uint32_t)b) + 31;
    }
};

/* RayTracingHelper.hlsli:229-235 */
inline Box AABBtoBox(tb3 mn, tb3 mx)
{
    Box b;
    b.center = (mn + mx) * 0.5f;
    b.halfDim = mx - b.center;
    return b;
}

} // namespace

/* ---- treelet reordering ------------------------------------------------------------------------------------------ */
namespace {

struct AABB { tb3 mn, mx; };
/* TreeletReorderBindings.h:104-110 */
inline AABB CombineAABB(const AABB& a, const AABB& b) { AABB r; r.mn = tb3_min(a.mn, b.mn); r.mx = tb3_max(a.mx, b.mx); return r; }
inline float ComputeBoxSurfaceArea(const AABB& a) /* TreeletReorderBindings.h:98-102 */
{
    tb3 dim = a.mx - a.mn;
    return 2.0f * (dim.x * dim.y + dim.x * dim.z + dim.y * dim.z);
}

const uint32_t FullTreeletSize = 7, NumInternalTreeletNodes = 6, NumTreeletSplitPermutations = 128, FullPartitionMask = 127;

struct Treelets {
    uint32_t N;
    std::vector<uint32_t>&left, &right, &parent;
    std::vector<AABB> aabb; /* AABBBuffer: every node */
    bool isLeaf(uint32_t x) const { return x >= N - 1; }
    uint32_t& L(uint32_t x) { return left[x]; }
    uint32_t& R(uint32_t x) { return right[x]; }

    /* one trip of TreeletReorder.hlsl main's loop body for the treelet rooted at nodeIndex (without TraverseToParent) */
    void reorder(uint32_t nodeIndex)
    {
        uint32_t treeletToReorder[FullTreeletSize], internalNodes[NumInternalTreeletNodes];
        float optimalCost[NumTreeletSplitPermutations]; uint32_t optimalPartition[NumTreeletSplitPermutations];
        /* FormTreelet :38-80: grow from the root's two children by opening the node of largest surface area */
        internalNodes[0] = nodeIndex;
        treeletToReorder[0] = L(nodeIndex); treeletToReorder[1] = R(nodeIndex);
        for (uint32_t treeletSize = 2; treeletSize < FullTreeletSize; treeletSize++) {
            float largestSurfaceArea = 0.0f; uint32_t nodeIndexToTraverse = 0, indexOfNodeIndexToTraverse = 0;
            for (uint32_t i = 0; i < treeletSize; i++) {
                uint32_t treeletNodeIndex = treeletToReorder[i];
                if (!isLeaf(treeletNodeIndex)) {
                    float surfaceArea = ComputeBoxSurfaceArea(aabb[treeletNodeIndex]);
                    if (surfaceArea > largestSurfaceArea) { largestSurfaceArea = surfaceArea; nodeIndexToTraverse = treeletNodeIndex;
                        indexOfNodeIndexToTraverse = i; }
                }
            }
            internalNodes[treeletSize - 1] = nodeIndexToTraverse;
            treeletToReorder[indexOfNodeIndexToTraverse] = L(nodeIndexToTraverse);
            treeletToReorder[treeletSize] = R(nodeIndexToTraverse);
        }
        /* FindOptimalPartitions :82-172.  Surface area of every subset's box first (:96-118) ... */
        for (uint32_t treeletBitmask = 1; treeletBitmask < NumTreeletSplitPermutations; treeletBitmask++) {
            AABB box; box.mn = tb3_splat(3.402823466e+38f); box.mx = tb3_splat(-3.402823466e+38f);
            for (uint32_t i = 0; i < FullTreeletSize; i++) if ((1u << i) & treeletBitmask) box = CombineAABB(box, aabb[treeletToReorder[i]]);
            optimalCost[treeletBitmask] = ComputeBoxSurfaceArea(box);
        }
        /* ... single leaves cost their area relative to the treelet root's (:121-127, CalculateCost :22-26) ... */
        float rootAABBSurfaceArea = ComputeBoxSurfaceArea(aabb[nodeIndex]);
        for (uint32_t i = 0; i < FullTreeletSize; i++) optimalCost[1u << i] = 1.0f * ComputeBoxSurfaceArea(aabb[treeletToReorder[i]]) / rootAABBSurfaceArea;
        /* ... then subsets by growing size: best split into two smaller subsets (:131-171) */
        for (uint32_t subsetSize = 2; subsetSize <= FullTreeletSize; subsetSize++) {
            for (uint32_t treeletBitmask = 1; treeletBitmask < NumTreeletSplitPermutations; treeletBitmask++) {
                if ((uint32_t)__builtin_popcount(treeletBitmask) != subsetSize) continue;
                float lowestCost = 3.402823466e+38f; uint32_t bestPartition = 0;
                uint32_t delta = (treeletBitmask - 1) & treeletBitmask;
                uint32_t partitionBitmask = (0u - delta) & treeletBitmask;
                do {
                    const float cost = optimalCost[partitionBitmask] + optimalCost[treeletBitmask ^ partitionBitmask];
                    if (cost < lowestCost) { lowestCost = cost; bestPartition = partitionBitmask; }
                    partitionBitmask = (partitionBitmask - delta) & treeletBitmask;
                } while (partitionBitmask != 0);
                optimalCost[treeletBitmask] = 1.0f * optimalCost[treeletBitmask] + lowestCost;
                optimalPartition[treeletBitmask] = bestPartition;
            }
        }
        /* ReformTree :174-233 */
        struct PartitionEntry { uint32_t Mask, NodeIndex; };
        uint32_t nodesAllocated = 1, partitionStackSize = 1;
        PartitionEntry partitionStack[FullTreeletSize];
        partitionStack[0].Mask = FullPartitionMask; partitionStack[0].NodeIndex = internalNodes[0];
        while (partitionStackSize > 0) {
            PartitionEntry partition = partitionStack[--partitionStackSize];
            PartitionEntry leftEntry; leftEntry.Mask = optimalPartition[partition.Mask];
            if (__builtin_popcount(leftEntry.Mask) > 1) { leftEntry.NodeIndex = internalNodes[nodesAllocated++];
                partitionStack[partitionStackSize++] = leftEntry; }
            else leftEntry.NodeIndex = treeletToReorder[__builtin_ctz(leftEntry.Mask)];
            PartitionEntry rightEntry; rightEntry.Mask = partition.Mask ^ leftEntry.Mask;
            if (__builtin_popcount(rightEntry.Mask) > 1) { rightEntry.NodeIndex = internalNodes[nodesAllocated++];
                partitionStack[partitionStackSize++] = rightEntry; }
            else rightEntry.NodeIndex = treeletToReorder[__builtin_ctz(rightEntry.Mask)];
            L(partition.NodeIndex) = leftEntry.NodeIndex; R(partition.NodeIndex) = rightEntry.NodeIndex;
            parent[leftEntry.NodeIndex] = partition.NodeIndex; parent[rightEntry.NodeIndex] = partition.NodeIndex;
        }
        for (int j = (int)NumInternalTreeletNodes - 1; j >= 0; j--) { uint32_t n = internalNodes[j]; aabb[n] = CombineAABB(aabb[L(n)], aabb[R(n)]); }
    }

    /* ClearBuffers + FindTreelets + TreeletReorder dispatches of one pass, serially: children before parents */
    void pass(uint32_t minTrianglesPerTreelet, const std::vector<AABB>& leafBox)
    {
        const uint32_t numNodes = 2 * N - 1;
        std::vector<uint32_t> order; order.reserve(numNodes);
        { std::vector<uint32_t> st; st.push_back(0);
          while (!st.empty()) { uint32_t x = st.back(); st.pop_back(); order.push_back(x); if (!isLeaf(x)) { st.push_back(L(x)); st.push_back(R(x)); } }
          std::reverse(order.begin(), order.end()); }
        std::vector<uint32_t> numTriangles(numNodes, 0), trips(numNodes, 0); /* trips: how many treelets the group standing on a node has done */
        for (uint32_t x : order) {
            if (isLeaf(x)) { aabb[x] = leafBox[x - (N - 1)]; numTriangles[x] = 1; continue; } /* FindTreelets.hlsl:44-48 */
            const uint32_t l = L(x), r = R(x);
            aabb[x] = CombineAABB(aabb[l], aabb[r]); /* FindTreelets.hlsl:50-55, TreeletReorder.hlsl:258-262 */
            numTriangles[x] = numTriangles[l] + numTriangles[r];
            if (numTriangles[x] < minTrianglesPerTreelet) continue;
            const bool lBig = numTriangles[l] >= minTrianglesPerTreelet, rBig = numTriangles[r] >= minTrianglesPerTreelet;
            if (!lBig && !rBig) trips[x] = 1; /* a base treelet, FindTreelets.hlsl:61-67 */
            else { /* reached by climbing groups, TreeletReorder.hlsl:235-268: every big child must have sent one */
                uint32_t fewest = 0xffffffffu; bool all = true;
                if (lBig) { if (trips[l] == 0) all = false; else fewest = std::min(fewest, trips[l]); }
                if (rBig) { if (trips[r] == 0) all = false; else fewest = std::min(fewest, trips[r]); }
                if (all && fewest < 33) trips[x] = fewest + 1;
            }
            if (trips[x]) reorder(x);
        }
    }
};

} // namespace

static int64_t build_lbvh_impl(const float* positions, const uint32_t* triVertexIndex, const uint32_t* triGeometry,
                                  const uint32_t* triPrimitive, const uint32_t* triFlags, uint32_t N, uint32_t treeletPasses, uint8_t* out, uint64_t capacity)
{
    if (N == 0) return -1;
    const uint64_t numNodes = 2ull * N - 1;
    const uint64_t offBoxes = 16, offPrims = offBoxes + 32 * numNodes, offMeta = offPrims + 40ull * N, total = offMeta + 12ull * N;
    if (total > capacity || total > 0xffffffffull) return -2;

    auto vert = [&](uint32_t t, int k) { const float* p = positions + 3ull * triVertexIndex[3ull * t + k]; return tb3_make(p[0], p[1], p[2]); };

    /* scene AABB */
    tb3 smin = tb3_splat(3.402823466e+38f), smax = tb3_splat(-3.402823466e+38f);
    for (uint32_t t = 0; t < N; t++) {
        tb3 v0 = vert(t, 0), v1 = vert(t, 1), v2 = vert(t, 2);
        smin = tb3_min(tb3_min(tb3_min(v0, smin), v1), v2);
        smax = tb3_max(tb3_max(tb3_max(v0, smax), v1), v2);
    }
    /* Morton codes */
    std::vector<std::pair<uint32_t, uint32_t>> keyed(N);
    tb3 dim = tb3_max(smax - smin, tb3_splat(0.00001f));
    for (uint32_t t = 0; t < N; t++) {
        tb3 c = (vert(t, 0) + vert(t, 1) + vert(t, 2)) / 3.0f;
        tb3 unit = (c - smin) / dim;
        keyed[t] = std::make_pair(MortonFromUnit(unit), t);
    }
    std::sort(keyed.begin(), keyed.end());

    Builder b; b.n = N; b.codes.resize(N);
    for (uint32_t i = 0; i < N; i++) b.codes[i] = keyed[i].first;

    /* Karras hierarchy: internal i in [0,N-1), leaf k is node N-1+k */
    std::vector<uint32_t> left(N > 1 ? N - 1 : 0), right(N > 1 ? N - 1 : 0), parent(numNodes, 0xffffffffu);
    for (int64_t idx = 0; idx + 1 < (int64_t)N; idx++) {
        /* DetermineRange :56-80 */
        int d = b.lcp(idx, idx + 1) - b.lcp(idx, idx - 1);
        d = d < -1 ? -1 : (d > 1 ? 1 : d);
        int minPrefix = b.lcp(idx, idx - d);
        int64_t maxLength = 2;
        while (b.lcp(idx, idx + maxLength * d) > minPrefix) maxLength *= 4;
        int64_t length = 0;
        for (int64_t t = maxLength / 2; t > 0; t /= 2)
            if (b.lcp(idx, idx + (length + t) * d) > minPrefix) length = length + t;
        int64_t j = idx + length * d;
        int64_t first = std::min(idx, j), last = std::max(idx, j);
        /* FindSplit :83-103 */
        int commonPrefix = b.lcp(first, last);
        int64_t split = first, step = last - first;
        do {
            step = (step + 1) >> 1;
            int64_t newSplit = split + step;
            if (newSplit < last) {
                int splitPrefix = b.lcp(first, newSplit);
                if (splitPrefix > commonPrefix) split = newSplit;
            }
        } while (step > 1);
        /* GenerateHierarchy :105-131 */
        uint32_t leafOff = N - 1;
        uint32_t a = (split == first) ? leafOff + (uint32_t)split : (uint32_t)split;
        uint32_t c = (split + 1 == last) ? leafOff + (uint32_t)split + 1 : (uint32_t)split + 1;
        left[(size_t)idx] = a; right[(size_t)idx] = c;
        parent[a] = (uint32_t)idx; parent[c] = (uint32_t)idx;
    }

    /* TreeletReorder::Optimize, TreeletReorder.cpp:38-109 */
    if (treeletPasses > 0 && N >= FullTreeletSize) {
        std::vector<AABB> leafBox(N);
        for (uint32_t k = 0; k < N; k++) { /* FindTreelets.hlsl:14-27: the leaf's centre/half-extent box turned back into min/max */
            uint32_t t = keyed[k].second;
            tb3 v0 = vert(t, 0), v1 = vert(t, 1), v2 = vert(t, 2);
            tb3 mn = tb3_min(tb3_min(v0, v1), v2), mx = tb3_max(tb3_max(v0, v1), v2);
            mn = tb3_min(mn, mx - tb3_splat(0.001f));
            Box bx = AABBtoBox(mn, mx);
            leafBox[k].mn = bx.center - bx.halfDim; leafBox[k].mx = bx.center + bx.halfDim; /* RayTracingHelper.hlsli:237-243 */
        }
        Treelets tr{N, left, right, parent, std::vector<AABB>((size_t)numNodes)};
        uint32_t minTrianglesPerTreelet = FullTreeletSize;
        for (uint32_t i = 0; i < treeletPasses; i++) {
            if (minTrianglesPerTreelet > N) break;
            tr.pass(minTrianglesPerTreelet, leafBox);
            minTrianglesPerTreelet *= 2;
        }
    }

    memset(out, 0, (size_t)total);
    TbBvhHeader hdr = {(uint32_t)offBoxes, (uint32_t)offPrims, (uint32_t)offMeta, (uint32_t)total};
    memcpy(out, &hdr, 16);
    TbAabbNode* nodes = (TbAabbNode*)(out + offBoxes);
    TbPrimitive* prims = (TbPrimitive*)(out + offPrims);
    TbPrimitiveMeta* meta = (TbPrimitiveMeta*)(out + offMeta);

    for (uint32_t k = 0; k < N; k++) {
        uint32_t t = keyed[k].second;
        tb3 v0 = vert(t, 0), v1 = vert(t, 1), v2 = vert(t, 2);
        TbPrimitive p; p.PrimitiveType = 1;
        p.v0[0] = v0.x; p.v0[1] = v0.y; p.v0[2] = v0.z; p.v1[0] = v1.x; p.v1[1] = v1.y; p.v1[2] = v1.z; p.v2[0] = v2.x; p.v2[1] = v2.y; p.v2[2] = v2.z;
        memcpy(&prims[k], &p, sizeof p);
        meta[k].GeometryContributionToHitGroupIndex = triGeometry ? triGeometry[t] : 0;
        meta[k].PrimitiveIndex = triPrimitive ? triPrimitive[t] : t;
        meta[k].GeometryFlags = triFlags ? triFlags[t] : 1u; /* D3D12_RAYTRACING_GEOMETRY_FLAG_OPAQUE */
    }

    /* AABB fit, bottom-up in dependency order (children before parents) */
    std::vector<uint32_t> count(numNodes, 0);
    std::vector<uint32_t> order; order.reserve((size_t)numNodes);
    {
        std::vector<uint32_t> st; st.push_back(0);
        while (!st.empty()) {
            uint32_t x = st.back(); st.pop_back();
            order.push_back(x);
            if (x < N - 1) { st.push_back(left[x]); st.push_back(right[x]); }
        }
        std::reverse(order.begin(), order.end());
    }
    auto writeNode = [&](uint32_t idx, const Box& bx, uint32_t fx, uint32_t fy) {
        TbAabbNode nd;
        nd.center[0] = bx.center.x; nd.center[1] = bx.center.y; nd.center[2] = bx.center.z; nd.flags = fx;
        nd.halfDim[0] = bx.halfDim.x; nd.halfDim[1] = bx.halfDim.y; nd.halfDim[2] = bx.halfDim.z; nd.rightNodeIndex = fy;
        nodes[idx] = nd;
    };
    auto readBox = [&](uint32_t idx) {
        Box bx; bx.center = tb3_make(nodes[idx].center[0], nodes[idx].center[1], nodes[idx].center[2]);
        bx.halfDim = tb3_make(nodes[idx].halfDim[0], nodes[idx].halfDim[1], nodes[idx].halfDim[2]);
        return bx;
    };
    for (uint32_t x : order) {
        if (x >= N - 1) { /* leaf: GetBoxDataFromTriangle, RayTracingHelper.hlsli:251-263 */
            uint32_t k = x - (N - 1);
            tb3 v0 = tb3_make(prims[k].v0[0], prims[k].v0[1], prims[k].v0[2]);
            tb3 v1 = tb3_make(prims[k].v1[0], prims[k].v1[1], prims[k].v1[2]);
            tb3 v2 = tb3_make(prims[k].v2[0], prims[k].v2[1], prims[k].v2[2]);
            tb3 mn = tb3_min(tb3_min(v0, v1), v2), mx = tb3_max(tb3_max(v0, v1), v2);
            mn = tb3_min(mn, mx - tb3_splat(0.001f)); /* AABB_Min_Padding */
            writeNode(x, AABBtoBox(mn, mx), k | TB_BVH_LEAF_FLAG, 1);
            count[x] = 1;
        } else { /* ComputeAABBs.hlsli:105-156, GetBoxFromChildBoxes RayTracingHelper.hlsli:275-285 */
            uint32_t l = left[x], r = right[x];
            if (count[l] > count[r]) { uint32_t t = l; l = r; r = t; } /* smaller subtree on the left */
            Box lb = readBox(l), rb = readBox(r);
            tb3 mn = tb3_min(lb.center - lb.halfDim, rb.center - rb.halfDim);
            tb3 mx = tb3_max(lb.center + lb.halfDim, rb.center + rb.halfDim);
            writeNode(x, AABBtoBox(mn, mx), l & TB_BVH_INDEX_MASK, r);
            count[x] = count[l] + count[r];
        }
    }
    return (int64_t)total;
}

/* ---- top level (two-level scenes) ---------------------------------------------------------------------------------------
 * Serial restatement of the fallback layer's top-level build: TopLevelLoadAABBs.hlsli:62-105 (per instance: root box of its
 * bottom-level structure, InverseAffineTransform of ObjectToWorld, TransformAABB, leaf node + BVHMetadata),
 * CalculateSceneAABBFromBVHs.hlsl:16-41 (scene box from the stored centre / half-extent boxes), CalculateMortonCodesForAABBs.hlsl
 * (code of the box centre), the same sort / BuildBVHSplits / ComputeAABBs passes as a bottom level; no treelet pass for
 * Level::Top (GpuBVH2Builder.cpp:498-501).  mul(float3x4, float4) is pinned as one fma chain per row like everywhere in this build.
 * instances: objectToWorld rows (12 floats each); rootBoxes: min xyz, max xyz of each instance's bottom-level root box. */
namespace {
float Determinant34(const float* t) /* RayTracingHelper.hlsli:287-295 */
{
#define M(r, c) t[(r) * 4 + (c)]
    return M(0, 0) * M(1, 1) * M(2, 2) - M(0, 0) * M(2, 1) * M(1, 2) - M(1, 0) * M(0, 1) * M(2, 2) + M(1, 0) * M(2, 1) * M(0, 2) + M(2, 0) * M(0, 1) * M(1,
        2) - M(2, 0) * M(1, 1) * M(0, 2);
}
void InverseAffine34(const float* t, float* o) /* RayTracingHelper.hlsli:297-316, term by term */
{
    const float invDet = 1.0f / Determinant34(t);
#define O(r, c) o[(r) * 4 + (c)]
    O(0, 0) = invDet * (M(1, 1) * (M(2, 2) * 1.0f - 0.0f * M(2, 3)) + M(2, 1) * (0.0f * M(1, 3) - M(1, 2) * 1.0f) + 0.0f * (M(1, 2) * M(2, 3) - M(2, 2) * M(1,
        3)));
    O(1, 0) = invDet * (M(1, 2) * (M(2, 0) * 1.0f - 0.0f * M(2, 3)) + M(2, 2) * (0.0f * M(1, 3) - M(1, 0) * 1.0f) + 0.0f * (M(1, 0) * M(2, 3) - M(2, 0) * M(1,
        3)));
    O(2, 0) = invDet * (M(1, 3) * (M(2, 0) * 0.0f - 0.0f * M(2, 1)) + M(2, 3) * (0.0f * M(1, 1) - M(1, 0) * 0.0f) + 1.0f * (M(1, 0) * M(2, 1) - M(2, 0) * M(1,
        1)));
    O(0, 1) = invDet * (M(2, 1) * (M(0, 2) * 1.0f - 0.0f * M(0, 3)) + 0.0f * (M(2, 2) * M(0, 3) - M(0, 2) * M(2, 3)) + M(0, 1) * (0.0f * M(2, 3) - M(2,
        2) * 1.0f));
    O(1, 1) = invDet * (M(2, 2) * (M(0, 0) * 1.0f - 0.0f * M(0, 3)) + 0.0f * (M(2, 0) * M(0, 3) - M(0, 0) * M(2, 3)) + M(0, 2) * (0.0f * M(2, 3) - M(2,
        0) * 1.0f));
    O(2, 1) = invDet * (M(2, 3) * (M(0, 0) * 0.0f - 0.0f * M(0, 1)) + 1.0f * (M(2, 0) * M(0, 1) - M(0, 0) * M(2, 1)) + M(0, 3) * (0.0f * M(2, 1) - M(2,
        0) * 0.0f));
    O(0, 2) = invDet * (0.0f * (M(0, 2) * M(1, 3) - M(1, 2) * M(0, 3)) + M(0, 1) * (M(1, 2) * 1.0f - 0.0f * M(1, 3)) + M(1, 1) * (0.0f * M(0, 3) - M(0,
        2) * 1.0f));
    O(1, 2) = invDet * (0.0f * (M(0, 0) * M(1, 3) - M(1, 0) * M(0, 3)) + M(0, 2) * (M(1, 0) * 1.0f - 0.0f * M(1, 3)) + M(1, 2) * (0.0f * M(0, 3) - M(0,
        0) * 1.0f));
    O(2, 2) = invDet * (1.0f * (M(0, 0) * M(1, 1) - M(1, 0) * M(0, 1)) + M(0, 3) * (M(1, 0) * 0.0f - 0.0f * M(1, 1)) + M(1, 3) * (0.0f * M(0, 1) - M(0,
        0) * 0.0f));
    O(0, 3) = invDet * (M(0, 1) * (M(2, 2) * M(1, 3) - M(1, 2) * M(2, 3)) + M(1, 1) * (M(0, 2) * M(2, 3) - M(2, 2) * M(0, 3)) + M(2, 1) * (M(1, 2) * M(0,
        3) - M(0, 2) * M(1, 3)));
    O(1, 3) = invDet * (M(0, 2) * (M(2, 0) * M(1, 3) - M(1, 0) * M(2, 3)) + M(1, 2) * (M(0, 0) * M(2, 3) - M(2, 0) * M(0, 3)) + M(2, 2) * (M(1, 0) * M(0,
        3) - M(0, 0) * M(1, 3)));
    O(2, 3) = invDet * (M(0, 3) * (M(2, 0) * M(1, 1) - M(1, 0) * M(2, 1)) + M(1, 3) * (M(0, 0) * M(2, 1) - M(2, 0) * M(0, 1)) + M(2, 3) * (M(1, 0) * M(0,
        1) - M(0, 0) * M(1, 1)));
#undef O
#undef M
}
inline tb3 MulPoint(const float* m, tb3 v)
{
    return tb3_make(tb_fma(m[2], v.z, tb_fma(m[1], v.y, tb_fma(m[0], v.x, m[3]))), tb_fma(m[6], v.z, tb_fma(m[5], v.y, tb_fma(m[4], v.x, m[7]))),
                    tb_fma(m[10], v.z, tb_fma(m[9], v.y, tb_fma(m[8], v.x, m[11]))));
}
} // namespace

extern "C" int64_t tbo_build_tlas(const float* objectToWorld, const float* rootBoxes, const uint32_t* blasIndex, const uint32_t* hitGroupBase,
                                  uint32_t M, uint8_t* out, uint64_t capacity)
{
    if (M == 0) return -1;
    const uint64_t numNodes = 2ull * M - 1, offBoxes = 16, offMeta = offBoxes + 32 * numNodes, total = offMeta + 116ull * M;
    if (total > capacity) return -2;
    std::vector<Box> leaf(M);
    std::vector<float> w2o(12ull * M);
    for (uint32_t i = 0; i < M; i++) {
        const float* o2w = objectToWorld + 12ull * i;
        InverseAffine34(o2w, &w2o[12ull * i]);
        const tb3 mn = tb3_make(rootBoxes[6 * i], rootBoxes[6 * i + 1], rootBoxes[6 * i + 2]), mx = tb3_make(rootBoxes[6 * i + 3], rootBoxes[6 * i + 4],
            rootBoxes[6 * i + 5]);
        /* TransformAABB :318-344: the eight corners in the order of the listing (the min / max of a set does not depend on it) */
        const tb3 corner[8] = {mn, tb3_make(mn.x, mn.y, mx.z), tb3_make(mn.x, mx.y, mx.z), tb3_make(mn.x, mx.y, mn.z), tb3_make(mx.x, mn.y, mn.z),
                               tb3_make(mx.x, mx.y, mn.z), tb3_make(mx.x, mn.y, mx.z), mx};
        tb3 tmn = tb3_splat(3.402823466e+38f), tmx = tb3_splat(-3.402823466e+38f);
        for (int k = 0; k < 8; k++) { const tb3 v = MulPoint(o2w, corner[k]); tmn = tb3_min(tmn, v); tmx = tb3_max(tmx, v); }
        leaf[i] = AABBtoBox(tmn, tmx);
    }
    tb3 smin = tb3_splat(3.402823466e+38f), smax = tb3_splat(-3.402823466e+38f);
    for (uint32_t i = 0; i < M; i++) { smin = tb3_min(leaf[i].center - leaf[i].halfDim, smin); smax = tb3_max(leaf[i].center + leaf[i].halfDim, smax); }
    const tb3 dim = tb3_max(smax - smin, tb3_splat(0.00001f));
    std::vector<std::pair<uint32_t, uint32_t>> keyed(M);
    for (uint32_t i = 0; i < M; i++) keyed[i] = std::make_pair(MortonFromUnit((leaf[i].center - smin) / dim), i);
    std::sort(keyed.begin(), keyed.end());
    Builder b; b.n = M; b.codes.resize(M);
    for (uint32_t i = 0; i < M; i++) b.codes[i] = keyed[i].first;
    std::vector<uint32_t> left(M > 1 ? M - 1 : 0), right(M > 1 ? M - 1 : 0);
    for (int64_t idx = 0; idx + 1 < (int64_t)M; idx++) { /* BuildBVHSplits.hlsli:56-131, as in build_lbvh_impl */
        int d = b.lcp(idx, idx + 1) - b.lcp(idx, idx - 1);
        d = d < -1 ? -1 : (d > 1 ? 1 : d);
        const int minPrefix = b.lcp(idx, idx - d);
        int64_t maxLength = 2;
        while (b.lcp(idx, idx + maxLength * d) > minPrefix) maxLength *= 4;
        int64_t length = 0;
        for (int64_t t = maxLength / 2; t > 0; t /= 2) if (b.lcp(idx, idx + (length + t) * d) > minPrefix) length = length + t;
        const int64_t j = idx + length * d, first = std::min(idx, j), last = std::max(idx, j);
        const int commonPrefix = b.lcp(first, last);
        int64_t split = first, step = last - first;
        do { step = (step + 1) >> 1; const int64_t ns = split + step; if (ns < last && b.lcp(first, ns) > commonPrefix) split = ns; } while (step > 1);
        left[(size_t)idx] = (split == first) ? (M - 1) + (uint32_t)split : (uint32_t)split;
        right[(size_t)idx] = (split + 1 == last) ? (M - 1) + (uint32_t)split + 1 : (uint32_t)split + 1;
    }
    memset(out, 0, (size_t)total);
    const TbBvhHeader hdr = {(uint32_t)offBoxes, (uint32_t)offMeta, (uint32_t)offMeta, (uint32_t)total};
    memcpy(out, &hdr, 16);
    TbAabbNode* nodes = (TbAabbNode*)(out + offBoxes);
    for (uint32_t k = 0; k < M; k++) { /* BVHMetadata of sorted leaf k (TopLevelLoadAABBs.hlsli:94-104) */
        const uint32_t i = keyed[k].second;
        TbBvhMetadata md; memset(&md, 0, sizeof md);
        memcpy(md.WorldToObject, &w2o[12ull * i], 48); memcpy(md.ObjectToWorld, objectToWorld + 12ull * i, 48);
        md.InstanceIDAndMask = 1u << 24; md.InstanceContributionToHitGroupIndexAndFlags = hitGroupBase[i] & 0x00ffffffu;
        md.BlasIndex = blasIndex[i]; md.InstanceIndex = i;
        memcpy(out + offMeta + 116ull * k, &md, 116);
    }
    std::vector<uint32_t> count((size_t)numNodes, 0), order; order.reserve((size_t)numNodes);
    { std::vector<uint32_t> st; st.push_back(0);
      while (!st.empty()) { uint32_t x = st.back(); st.pop_back(); order.push_back(x); if (M > 1 && x < M - 1) { st.push_back(left[x]); st.push_back(right[x]);
          } }
      std::reverse(order.begin(), order.end()); }
    auto writeNode = [&](uint32_t idx, const Box& bx, uint32_t fx, uint32_t fy) {
        TbAabbNode nd; nd.center[0] = bx.center.x; nd.center[1] = bx.center.y; nd.center[2] = bx.center.z; nd.flags = fx;
        nd.halfDim[0] = bx.halfDim.x; nd.halfDim[1] = bx.halfDim.y; nd.halfDim[2] = bx.halfDim.z; nd.rightNodeIndex = fy; nodes[idx] = nd;
    };
    auto readBox = [&](uint32_t idx) { Box bx; bx.center = tb3_make(nodes[idx].center[0], nodes[idx].center[1], nodes[idx].center[2]);
                                        bx.halfDim = tb3_make(nodes[idx].halfDim[0], nodes[idx].halfDim[1], nodes[idx].halfDim[2]); return bx; };
    for (uint32_t x : order) {
        /* TopLevelComputeAABBs.hlsl:16-33 */
        if (x >= M - 1) { const uint32_t k = x - (M - 1); writeNode(x, leaf[keyed[k].second], k | TB_BVH_LEAF_FLAG, 1); count[x] = 1; }
        else {
            uint32_t l = left[x], r = right[x];
            if (count[l] > count[r]) { uint32_t t = l; l = r; r = t; }
            const Box lb = readBox(l), rb = readBox(r);
            writeNode(x, AABBtoBox(tb3_min(lb.center - lb.halfDim, rb.center - rb.halfDim), tb3_max(lb.center + lb.halfDim, rb.center + rb.halfDim)),
                l & TB_BVH_INDEX_MASK, r);
            count[x] = count[l] + count[r];
        }
    }
    return (int64_t)total;
}

extern "C" int64_t tbo_build_lbvh(const float* positions, const uint32_t* triVertexIndex, const uint32_t* triGeometry,
                                  const uint32_t* triPrimitive, const uint32_t* triFlags, uint32_t N, uint8_t* out, uint64_t capacity)
{
    return build_lbvh_impl(positions, triVertexIndex, triGeometry, triPrimitive, triFlags, N, 0, out, capacity);
}

extern "C" int64_t tbo_build_lbvh2(const float* positions, const uint32_t* triVertexIndex, const uint32_t* triGeometry,
                                   const uint32_t* triPrimitive, const uint32_t* triFlags, uint32_t N, uint32_t treeletPasses, uint8_t* out, uint64_t capacity)
{
    return build_lbvh_impl(positions, triVertexIndex, triGeometry, triPrimitive, triFlags, N, treeletPasses, out, capacity);
}

/* BVHValidator.cpp:60-190 restated as invariants checked in one pass:
 *   - every child box is inside its parent box within TEST_EPSILON (1e-3)            (:114-134)
 *   - every node is reached exactly once from the root, no reference back to node 0 (:112,125)
 *   - every input triangle is matched by exactly one leaf whose box contains it      (:139-153,:174) */
extern "C" int tbo_validate_bvh(const uint8_t* bvh, uint32_t bvhBytes, const float* positions, const uint32_t* triVertexIndex,
                                uint32_t N, uint32_t* maxDepthOut)
{
    const float EPS = 0.001f;
    if (bvhBytes < 16) return -1;
    TbBvhHeader h; memcpy(&h, bvh, 16);
    /* A tree over PRE-SPLIT references (the build's option "presplit"; not a thing the reference's builders make) has more leaves than the scene
     * has triangles: every leaf still holds a whole input triangle, several leaves may hold the same one, and a leaf's box bounds the PART of
     * its triangle it stands for.  The invariants become: child boxes inside parent boxes, every node reached once, every leaf's triangle
     * is an input triangle and its box lies inside that triangle's bounds, every input triangle is held by at least one leaf, and the boxes of
     * a triangle's leaves together cover it (checked on 13 points of the triangle: vertices, edge midpoints, centroid, six more inside). */
    if (h.offsetToPrimitiveMetaData > h.offsetToVertices && (h.offsetToPrimitiveMetaData - h.offsetToVertices) % 40u == 0 &&
        (h.offsetToPrimitiveMetaData - h.offsetToVertices) / 40u > N) {
        const uint32_t L = (h.offsetToPrimitiveMetaData - h.offsetToVertices) / 40u;
        const uint64_t nn = 2ull * L - 1;
        if (h.offsetToBoxes != 16 || h.offsetToVertices != 16 + 32 * nn || h.totalSize != h.offsetToPrimitiveMetaData + 12ull * L || h.totalSize > bvhBytes) return -2;
        const TbAabbNode* nodes = (const TbAabbNode*)(bvh + 16);
        const uint8_t* prims = bvh + h.offsetToVertices;
        auto hash9 = [](const float* f) { uint64_t hsh = 1469598103934665603ull; for (int i = 0; i < 9; i++) { uint32_t u; memcpy(&u, f + i, 4);
            hsh = (hsh ^ u) * 1099511628211ull; } return hsh; };
        std::unordered_map<uint64_t, uint32_t> first; /* 9 floats -> first input triangle with them */
        std::vector<uint32_t> same(N);                 /* triangle -> the first triangle with the same vertices */
        for (uint32_t t = 0; t < N; t++) {
            float f[9]; for (int k = 0; k < 3; k++) memcpy(f + 3 * k, positions + 3ull * triVertexIndex[3ull * t + k], 12);
            auto it = first.emplace(hash9(f), t); same[t] = it.first->second;
        }
        std::vector<std::vector<uint32_t>> leavesOf(N); /* (first) triangle -> its leaf nodes */
        std::vector<uint8_t> seen((size_t)nn, 0), leafSeen(L, 0);
        struct Item { uint32_t node, depth; };
        std::vector<Item> st; st.push_back({0, 1});
        uint32_t maxDepth = 0; uint64_t visited = 0;
        while (!st.empty()) {
            Item it = st.back(); st.pop_back();
            if (it.node >= nn) return -3;
            if (seen[it.node]) return -4;
            seen[it.node] = 1; visited++;
            if (it.depth > maxDepth) maxDepth = it.depth;
            const TbAabbNode& nd = nodes[it.node];
            float pmin[3], pmax[3];
            for (int a = 0; a < 3; a++) { pmin[a] = nd.center[a] - nd.halfDim[a]; pmax[a] = nd.center[a] + nd.halfDim[a]; }
            if (nd.flags & TB_BVH_LEAF_FLAG) {
                const uint32_t k = nd.flags & TB_BVH_INDEX_MASK;
                if (k >= L || leafSeen[k]) return -5;
                leafSeen[k] = 1;
                float f[9]; memcpy(f, prims + 40ull * k + 4, 36);
                auto q = first.find(hash9(f));
                if (q == first.end()) return -7;
                for (int a = 0; a < 3; a++) { /* the part's box lies inside its triangle's bounds */
                    const float tmin = std::min(f[a], std::min(f[3 + a], f[6 + a])), tmax = std::max(f[a], std::max(f[3 + a], f[6 + a]));
                    if (!(pmin[a] + EPS >= tmin - EPS && pmax[a] - EPS <= tmax + EPS)) return -6;
                }
                leavesOf[q->second].push_back(it.node);
            } else {
                uint32_t ch[2] = {nd.flags & TB_BVH_INDEX_MASK, nd.rightNodeIndex};
                for (int c = 0; c < 2; c++) {
                    if (ch[c] == 0 || ch[c] >= nn) return -8;
                    const TbAabbNode& cn = nodes[ch[c]];
                    for (int a = 0; a < 3; a++) {
                        float cmin = cn.center[a] - cn.halfDim[a], cmax = cn.center[a] + cn.halfDim[a];
                        if (!(pmin[a] - EPS <= cmin && pmax[a] + EPS >= cmax)) return -9;
                    }
                    st.push_back({ch[c], it.depth + 1});
                }
            }
        }
        if (visited != nn) return -10;
        static const float W[13][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {.5f, .5f, 0}, {0, .5f, .5f}, {.5f, 0, .5f}, {1 / 3.f, 1 / 3.f, 1 / 3.f}, {.6f, .2f, .2f},
            {.2f, .6f, .2f}, {.2f, .2f, .6f}, {.8f, .1f, .1f}, {.1f, .8f, .1f}, {.1f, .1f, .8f}};
        for (uint32_t t = 0; t < N; t++) {
            const std::vector<uint32_t>& lv = leavesOf[same[t]];
            if (lv.empty()) return -11;
            if (same[t] != t) continue;
            float f[9]; for (int k = 0; k < 3; k++) memcpy(f + 3 * k, positions + 3ull * triVertexIndex[3ull * t + k], 12);
            for (int w = 0; w < 13; w++) {
                float pt[3]; for (int a = 0; a < 3; a++) pt[a] = W[w][0] * f[a] + W[w][1] * f[3 + a] + W[w][2] * f[6 + a];
                bool in = false;
                for (uint32_t nodeIdx : lv) {
                    const TbAabbNode& nd = nodes[nodeIdx]; bool ok = true;
                    for (int a = 0; a < 3; a++) if (!(pt[a] + EPS >= nd.center[a] - nd.halfDim[a] && pt[a] - EPS <= nd.center[a] + nd.halfDim[a])) ok = false;
                    if (ok) { in = true; break; }
                }
                if (!in) return -12; /* a point of the triangle that none of its parts' boxes holds */
            }
        }
        if (maxDepthOut) *maxDepthOut = maxDepth;
        return 0;
    }
    const uint64_t numNodes = 2ull * N - 1;
    if (h.offsetToBoxes != 16 || h.offsetToVertices != 16 + 32 * numNodes || h.offsetToPrimitiveMetaData != h.offsetToVertices + 40ull * N ||
        h.totalSize != h.offsetToPrimitiveMetaData + 12ull * N || h.totalSize > bvhBytes) return -2;
    const TbAabbNode* nodes = (const TbAabbNode*)(bvh + 16);
    const uint8_t* prims = bvh + h.offsetToVertices;
    std::vector<uint8_t> seen((size_t)numNodes, 0), leafSeen(N, 0);
    struct Item { uint32_t node, depth; };
    std::vector<Item> st; st.push_back({0, 1});
    uint32_t maxDepth = 0; uint64_t visited = 0;
    std::unordered_multimap<uint64_t, uint32_t> want; /* hash of 9 floats -> input triangle */
    auto hash9 = [](const float* f) { uint64_t hsh = 1469598103934665603ull; for (int i = 0; i < 9; i++) { uint32_t u; memcpy(&u, f + i, 4);
        hsh = (hsh ^ u) * 1099511628211ull; } return hsh; };
    std::vector<uint8_t> triMatched(N, 0);
    for (uint32_t t = 0; t < N; t++) {
        float f[9];
        for (int k = 0; k < 3; k++) memcpy(f + 3 * k, positions + 3ull * triVertexIndex[3ull * t + k], 12);
        want.emplace(hash9(f), t);
    }
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        if (it.node >= numNodes) return -3;
        if (seen[it.node]) return -4;
        seen[it.node] = 1; visited++;
        if (it.depth > maxDepth) maxDepth = it.depth;
        const TbAabbNode& nd = nodes[it.node];
        float pmin[3], pmax[3];
        for (int a = 0; a < 3; a++) { pmin[a] = nd.center[a] - nd.halfDim[a]; pmax[a] = nd.center[a] + nd.halfDim[a]; }
        if (nd.flags & TB_BVH_LEAF_FLAG) {
            uint32_t k = nd.flags & TB_BVH_INDEX_MASK;
            if (k >= N || leafSeen[k]) return -5;
            leafSeen[k] = 1;
            float f[9]; memcpy(f, prims + 40ull * k + 4, 36);
            for (int v = 0; v < 3; v++) for (int a = 0; a < 3; a++)
                if (!(f[3 * v + a] + EPS >= pmin[a] && f[3 * v + a] - EPS <= pmax[a])) return -6;
            auto range = want.equal_range(hash9(f));
            bool ok = false;
            for (auto q = range.first; q != range.second; ++q) if (!triMatched[q->second]) { triMatched[q->second] = 1; ok = true; break; }
            if (!ok) return -7;
        } else {
            uint32_t ch[2] = {nd.flags & TB_BVH_INDEX_MASK, nd.rightNodeIndex};
            for (int c = 0; c < 2; c++) {
                if (ch[c] == 0 || ch[c] >= numNodes) return -8; /* "Circular reference to root node" */
                const TbAabbNode& cn = nodes[ch[c]];
                for (int a = 0; a < 3; a++) {
                    float cmin = cn.center[a] - cn.halfDim[a], cmax = cn.center[a] + cn.halfDim[a];
                    if (!(pmin[a] - EPS <= cmin && pmax[a] + EPS >= cmax)) return -9;
                }
                st.push_back({ch[c], it.depth + 1});
            }
        }
    }
    if (visited != numNodes) return -10;
    for (uint32_t t = 0; t < N; t++) if (!triMatched[t]) return -11;
    if (maxDepthOut) *maxDepthOut = maxDepth;
    return 0;
}
