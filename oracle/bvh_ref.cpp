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
 * Not restated: TreeletReorder (3 passes, n = 7) -- see DESIGN.md "BVH quality".
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

extern "C" int64_t tbo_build_lbvh(const float* positions, const uint32_t* triVertexIndex, const uint32_t* triGeometry,
                                  const uint32_t* triPrimitive, const uint32_t* triFlags, uint32_t N, uint8_t* out, uint64_t capacity)
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
    auto hash9 = [](const float* f) { uint64_t hsh = 1469598103934665603ull; for (int i = 0; i < 9; i++) { uint32_t u; memcpy(&u, f + i, 4); hsh = (hsh ^ u) * 1099511628211ull; } return hsh; };
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
