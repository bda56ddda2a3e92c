/* bvh_kernels.hip -- LBVH construction on the GPU (option "bvh_builder" = 2).
 *
 * The same tree, bit for bit, as the host builder 0 (bvh_build.cpp) and the oracle's restatement (oracle/bvh_ref.cpp):
 *   bvh_bounds      scene box over all triangle vertices                          SceneAABBCalculator / CalculateSceneAABB
 *   bvh_morton      30-bit Morton code of the centroid (y,x,z interleave)         CalculateMortonCodesBindings.h:116-149
 *   rocprim sort    64-bit key = code << 32 | triangle index (ties by index)      BitonicSort in the reference
 *   bvh_hierarchy   Karras-2012 split search, one lane per inner node             BuildBVHSplits.hlsli:33-131
 *   bvh_treelet_*   (treeletPasses > 0, option "bvh_builder" = 4) the fallback layer's treelet passes, one wave per
 *                   treelet, byte-identical to host builder 3                       ClearBuffers.hlsl, FindTreelets.hlsl:27-88,
 *                                                                                   TreeletReorder.hlsl:38-311, TreeletReorder.cpp:38-109
 *   bvh_fit_*       leaves (1 triangle, 0.001 thin-box padding), then bottom-up:  RayTracingHelper.hlsli:251-263,
 *                   a node is fitted from its children's boxes, smaller subtree left ComputeAABBs.hlsli:152-156
 * Everything bottom-up is LEVEL-SYNCHRONOUS (round 3): one launch per wave-front of nodes whose children are finished, results of a
 * wave-front become visible to the next one at the kernel boundary.  Rounds 1-2 climbed inside one kernel -- one lane per leaf, the
 * second arrival at a node going on, an agent-scope __threadfence() on either side of every arrival counter -- and spent 170 of the
 * 171 ms of a 3 M-triangle build in those fences (each writes back / invalidates the XCD's L2): profiles/r2/c5_kernel_stats.csv.
 * Output: the reference's bottom-level memory image (layout A: header, 32-B nodes, 40-B primitives, 12-B metadata) plus
 * the layout-B arrays the kernels fetch.  Boxes are min/max of exact inputs, so the fit order cannot change a bit.
 * Integer / bit work apart from the box arithmetic: one pass over the triangles per stage, HBM-bound; the sort is
 * rocPRIM's radix sort (the library primitive for a plain key sort). */
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "tb_math.h"
#include "tb_vec.h"
#include "tb_abi.h"
#include "pt_launch.h"

namespace {

constexpr int BLOCK = 256;

__device__ __forceinline__ tb3 vertex(const float* positions, const uint32_t* triVertexIndex, uint32_t tri, int k)
{
    const float* p = positions + 3ull * triVertexIndex[3ull * tri + k];
    return tb3_make(p[0], p[1], p[2]);
}

/* float min/max through order-preserving integer keys (no NaNs in vertex data) */
__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__global__ __launch_bounds__(BLOCK) void bvh_bounds(const float* positions, const uint32_t* triVertexIndex, uint32_t N, uint32_t* ordMin, uint32_t* ordMax)
{
    tb3 mn = tb3_splat(3.402823466e+38f), mx = tb3_splat(-3.402823466e+38f);
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < N; i += gridDim.x * BLOCK)
        for (int k = 0; k < 3; k++) { tb3 v = vertex(positions, triVertexIndex, i, k); mn = tb3_min(v, mn); mx = tb3_max(v, mx); }
    float lo[3] = {mn.x, mn.y, mn.z}, hi[3] = {mx.x, mx.y, mx.z};
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) { lo[a] = tb_min(lo[a], __shfl_down(lo[a], o, 64)); hi[a] = tb_max(hi[a], __shfl_down(hi[a], o, 64)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&ordMin[a], f2ord(lo[a])); atomicMax(&ordMax[a], f2ord(hi[a])); }
    }
}

__device__ __forceinline__ uint32_t expand10(uint32_t v)
{
    v &= 0x3ff; v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249;
    return v;
}

__global__ __launch_bounds__(BLOCK) void bvh_morton(const float* positions, const uint32_t* triVertexIndex, uint32_t N, const uint32_t* ordMin,
    const uint32_t* ordMax,
                                                    unsigned long long* keys)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= N) return;
    const tb3 smin = tb3_make(ord2f(ordMin[0]), ord2f(ordMin[1]), ord2f(ordMin[2])), smax = tb3_make(ord2f(ordMax[0]), ord2f(ordMax[1]), ord2f(ordMax[2]));
    const tb3 dim = tb3_max(smax - smin, tb3_splat(0.00001f));
    const tb3 c = (vertex(positions, triVertexIndex, i, 0) + vertex(positions, triVertexIndex, i, 1) + vertex(positions, triVertexIndex, i, 2)) / 3.0f;
    const tb3 u = (c - smin) / dim;
    const float ax = tb_min(tb_max(u.x * 1024.0f, 0.0f), 1023.0f), ay = tb_min(tb_max(u.y * 1024.0f, 0.0f), 1023.0f), az = tb_min(tb_max(u.z * 1024.0f, 0.0f),
        1023.0f);
    const uint32_t code = expand10((uint32_t)ay) | (expand10((uint32_t)ax) << 1) | (expand10((uint32_t)az) << 2); /* axis 0 <- y, 1 <- x, 2 <- z */
    keys[i] = ((unsigned long long)code << 32) | (unsigned long long)i;
}

__device__ __forceinline__ int delta(const unsigned long long* keys, uint32_t N, long long a, long long b)
{
    if (b < 0 || b >= (long long)N) return -1;
    const uint32_t x = (uint32_t)(keys[a] >> 32) ^ (uint32_t)(keys[b] >> 32);
    if (x) return __clz((int)x);
    const uint32_t y = (uint32_t)a ^ (uint32_t)b;
    return (y ? __clz((int)y) : 32) + 31;
}

__global__ __launch_bounds__(BLOCK) void bvh_hierarchy(const unsigned long long* keys, uint32_t N, uint32_t* left, uint32_t* right, uint32_t* parent)
{
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (long long)N - 1) return;
    int d = delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1); d = (d > 0) - (d < 0);
    const int dmin = delta(keys, N, i, i - d);
    long long lmax = 2; while (delta(keys, N, i, i + lmax * d) > dmin) lmax *= 4;
    long long l = 0; for (long long st = lmax / 2; st > 0; st /= 2) if (delta(keys, N, i, i + (l + st) * d) > dmin) l += st;
    const long long j = i + l * d, first = i < j ? i : j, last = i < j ? j : i;
    const int dn = delta(keys, N, first, last);
    long long split = first, step = last - first;
    do { step = (step + 1) >> 1; const long long ns = split + step; if (ns < last && delta(keys, N, first, ns) > dn) split = ns; } while (step > 1);
    const uint32_t lc = (split == first) ? (N - 1) + (uint32_t)split : (uint32_t)split;
    const uint32_t rc = (split + 1 == last) ? (N - 1) + (uint32_t)split + 1 : (uint32_t)split + 1;
    left[i] = lc; right[i] = rc; parent[lc] = (uint32_t)i; parent[rc] = (uint32_t)i;
}

/* ---- bottom-up passes, level-synchronous ------------------------------------------------------------------------------------
 * stamp[i] (inner nodes; leaves count as finished) = the launch t >= 1 that finished node i, 0 = not yet.  Launch t finishes the
 * nodes whose children carry stamps in [1, t): a stamp written by the same launch does not count, because the data behind it need
 * not be visible yet; at the next launch it is.  Every launch adds the number of nodes it finished to *progress; the host stops
 * when all N - 1 are done (it reads the counter back every few launches).  Tree height launches in all. */
struct TbBox6 { float mn[3], mx[3]; };

__device__ __forceinline__ float box_area(const TbBox6& b)
{
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}
__device__ __forceinline__ TbBox6 box_union(const TbBox6& a, const TbBox6& b)
{
    TbBox6 r;
    for (int k = 0; k < 3; k++) { r.mn[k] = tb_min(a.mn[k], b.mn[k]); r.mx[k] = tb_max(a.mx[k], b.mx[k]); }
    return r;
}

__device__ __forceinline__ bool children_finished(const uint32_t* stamp, uint32_t N, uint32_t l, uint32_t r, uint32_t t)
{
    const uint32_t sl = l >= N - 1 ? 1u : stamp[l], sr = r >= N - 1 ? 1u : stamp[r];
    return sl != 0u && sr != 0u && (l >= N - 1 || sl < t) && (r >= N - 1 || sr < t);
}
__device__ __forceinline__ void count_progress(uint32_t* progress, bool did)
{
    const unsigned long long m = __ballot(did);
    if (m && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) atomicAdd(progress, (uint32_t)__popcll(m));
}

/* ---- treelet passes (Karras & Aila 2013 as the fallback layer runs it; host twin: bvh_build.cpp TreeletPass) ------------
 * Per pass (MinTrianglesPerTreelet m = 7, 14, 28):
 *   bvh_treelet_leaves   min / max box and triangle count 1 of every leaf (first pass only; later passes keep the arrays)
 *   bvh_treelet_up       level-synchronous: box = union of the children's, numTris = sum, for every inner node (first pass; a
 *                        treelet rebuild keeps both arrays exact for the six nodes it rewires, so later passes reuse them)
 *   bvh_treelet_classify a node with numTris >= m waits for as many groups as it has children with numTris >= m; one with none
 *                        is the root of a first treelet (FindTreelets.hlsl: the lowest nodes holding >= m triangles)
 *   bvh_treelet_rebuild  one wave per listed node: lane 0 grows the 7-leaf treelet below it, the 64 lanes price the 127 leaf
 *                        subsets (2 per lane, subset sizes in turn), lane 0 rewires the six inner nodes and leaves the number
 *                        of treelets its chain has rebuilt in trips[node]
 *   bvh_treelet_advance  per rebuilt node: the parent goes on the next list once every child it waits for has been rebuilt and
 *                        the shortest of their chains is below 33 (the reference's groups stop after 33 treelets; which of two
 *                        meeting groups goes on is a race there -- here, as in the host builder and the oracle, the smaller count)
 * rebuild / advance alternate until a list comes up empty: as many rounds as the longest chain of dependent treelets. */
struct TreeletBufs {
    uint32_t* left; uint32_t* right; uint32_t* parent; /* hierarchy (node ids; leaf k = N-1+k) */
    TbBox6* boxes;                                      /* min/max box per node */
    uint32_t* numTris;                                  /* triangles below every node (2N - 1 entries) */
    uint32_t* trips;                                    /* per inner node: treelets rebuilt by the chain that rebuilt this node, 0 = not rebuilt in this pass */
    uint32_t* chain;                                    /* per inner node: what trips[] becomes when the node is rebuilt (set when it is listed) */
    uint32_t* queued;                                   /* per inner node: already on a list in this pass */
    uint32_t* stamp;                                    /* per inner node: bvh_treelet_up */
};

__global__ __launch_bounds__(BLOCK) void bvh_treelet_leaves(const float* positions, const uint32_t* triVertexIndex, const unsigned long long* keys, uint32_t N,
    TreeletBufs b)
{
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= N) return;
    const uint32_t tri = (uint32_t)keys[k];
    const tb3 v0 = vertex(positions, triVertexIndex, tri, 0), v1 = vertex(positions, triVertexIndex, tri, 1), v2 = vertex(positions, triVertexIndex, tri, 2);
    tb3 mn = tb3_min(tb3_min(v0, v1), v2); const tb3 mx = tb3_max(tb3_max(v0, v1), v2);
    mn = tb3_min(mn, mx - tb3_splat(0.001f));
    const tb3 c = (mn + mx) * 0.5f, h = mx - c, lo = c - h, hi = c + h; /* the leaf's centre / half-extent box turned back into min / max */
    TbBox6 bx; bx.mn[0] = lo.x; bx.mn[1] = lo.y; bx.mn[2] = lo.z; bx.mx[0] = hi.x; bx.mx[1] = hi.y; bx.mx[2] = hi.z;
    b.boxes[(N - 1) + k] = bx; b.numTris[(N - 1) + k] = 1u;
}

__global__ __launch_bounds__(BLOCK) void bvh_treelet_up(uint32_t N, TreeletBufs b, uint32_t t, uint32_t* progress)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    bool did = false;
    if (i < N - 1 && b.stamp[i] == 0u) {
        const uint32_t l = b.left[i], r = b.right[i];
        if (children_finished(b.stamp, N, l, r, t)) {
            b.boxes[i] = box_union(b.boxes[l], b.boxes[r]); b.numTris[i] = b.numTris[l] + b.numTris[r];
            b.stamp[i] = t; did = true;
        }
    }
    count_progress(progress, did);
}

__global__ __launch_bounds__(BLOCK) void bvh_treelet_classify(uint32_t N, uint32_t minTris, TreeletBufs b, uint32_t* listCount, uint32_t* list)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= N - 1) return;
    b.trips[i] = 0u; b.queued[i] = 0u; b.chain[i] = 0u;
    if (b.numTris[i] < minTris) return;
    if (b.numTris[b.left[i]] < minTris && b.numTris[b.right[i]] < minTris) { b.queued[i] = 1u; b.chain[i] = 1u; list[atomicAdd(listCount, 1u)] = i; }
}

__global__ __launch_bounds__(64) void bvh_treelet_rebuild(uint32_t N, uint32_t minTris, TreeletBufs b, const uint32_t* listCount, const uint32_t* list)
{
    __shared__ uint32_t sLeaf[7], sInner[6], sLeafTris[7];
    __shared__ TbBox6 sLeafBox[7];
    __shared__ float sRootArea;
    __shared__ float sCost[128];
    __shared__ uint32_t sCut[128];
    const uint32_t lane = threadIdx.x;
    const uint32_t count = *listCount;
    for (uint32_t at0 = blockIdx.x; at0 < count; at0 += gridDim.x) {
        const uint32_t root = list[at0];
        if (lane == 0) { /* grow the treelet: open the inner node with the largest box, five times */
            float area[7]; uint32_t node[7];
            node[0] = b.left[root]; node[1] = b.right[root];
            area[0] = node[0] >= N - 1 ? -1.0f : box_area(b.boxes[node[0]]);
            area[1] = node[1] >= N - 1 ? -1.0f : box_area(b.boxes[node[1]]);
            sInner[0] = root;
            for (uint32_t n = 2; n < 7; n++) {
                float best = 0.0f; uint32_t at = 0, open = 0;
                for (uint32_t i = 0; i < n; i++) if (area[i] > best) { best = area[i]; at = i; open = node[i]; }
                sInner[n - 1] = open;
                const uint32_t l = b.left[open], r = b.right[open];
                node[at] = l; area[at] = l >= N - 1 ? -1.0f : box_area(b.boxes[l]);
                node[n] = r; area[n] = r >= N - 1 ? -1.0f : box_area(b.boxes[r]);
            }
            for (int i = 0; i < 7; i++) sLeaf[i] = node[i];
            sRootArea = box_area(b.boxes[root]);
        }
        __syncthreads();
        if (lane < 7) { sLeafBox[lane] = b.boxes[sLeaf[lane]]; sLeafTris[lane] = b.numTris[sLeaf[lane]]; }
        __syncthreads();
        for (uint32_t m = lane; m < 128; m += 64) { /* box area of every leaf subset; a single leaf costs area / root area */
            if (m == 0) continue;
            const uint32_t low = (uint32_t)__ffs((int)m) - 1u;
            TbBox6 u = sLeafBox[low];
            for (uint32_t i = low + 1; i < 7; i++) if (m & (1u << i)) u = box_union(u, sLeafBox[i]);
            const float a = box_area(u);
            sCost[m] = (m & (m - 1)) ? a : a / sRootArea;
        }
        __syncthreads();
        for (int size = 2; size <= 7; size++) { /* least cost of splitting a subset in two, by growing subset size */
            for (uint32_t m = lane; m < 128; m += 64) {
                if (__popc(m) != size) continue;
                const uint32_t d = (m - 1) & m;
                float least = 3.402823466e+38f; uint32_t arg = 0, p = (0u - d) & m;
                do { const float c = sCost[p] + sCost[m ^ p]; if (c < least) { least = c; arg = p; } p = (p - d) & m; } while (p);
                sCost[m] += least; sCut[m] = arg;
            }
            __syncthreads();
        }
        if (lane == 0) {
            /* hand the six inner ids out again: parent first, left child's id before the right child's, right subtree first */
            uint32_t used = 1, n = 0, todoMask[7], todoNode[7];
            todoMask[n] = 127u; todoNode[n++] = root;
            while (n) {
                --n; const uint32_t mask = todoMask[n], node = todoNode[n];
                const uint32_t lm = sCut[mask], rm = mask ^ lm; uint32_t ln, rn;
                if (lm & (lm - 1)) { ln = sInner[used++]; todoMask[n] = lm; todoNode[n++] = ln; } else ln = sLeaf[__ffs((int)lm) - 1];
                if (rm & (rm - 1)) { rn = sInner[used++]; todoMask[n] = rm; todoNode[n++] = rn; } else rn = sLeaf[__ffs((int)rm) - 1];
                b.left[node] = ln; b.right[node] = rn; b.parent[ln] = node; b.parent[rn] = node;
                uint32_t tris = 0; for (uint32_t i = 0; i < 7; i++) if (mask & (1u << i)) tris += sLeafTris[i];
                b.numTris[node] = tris; /* the root's is unchanged; the five nodes below it now hold other subsets */
            }
            for (int j = 5; j >= 0; j--) { const uint32_t x = sInner[j]; b.boxes[x] = box_union(b.boxes[b.left[x]], b.boxes[b.right[x]]); }
            b.trips[root] = b.chain[root]; /* rebuilt: the length of the chain of treelets that ends here (set when the node was listed) */
        }
        __syncthreads();
    }
}

/* After a round of rebuilds: which parents are ready?  trips[] is written by the rebuild launches only, so everything read here was
 * finished by an earlier launch.  The parent link of a rebuilt node and the child links of its parent are still the ones the pass
 * started with: only a treelet rooted at an ancestor rewires them, and ancestors come strictly later. */
__global__ __launch_bounds__(BLOCK) void bvh_treelet_advance(uint32_t N, uint32_t minTris, TreeletBufs b,
                                                            const uint32_t* listCount, const uint32_t* list, uint32_t* nextCount, uint32_t* next)
{
    const uint32_t at = blockIdx.x * BLOCK + threadIdx.x;
    if (at >= *listCount) return;
    const uint32_t x = list[at];
    if (x == 0u) return;
    const uint32_t up = b.parent[x];
    /* every child of `up` that holds >= minTris triangles must have been rebuilt; the shortest chain goes on */
    uint32_t fewest = 0xffffffffu; bool all = true;
    const uint32_t kids[2] = {b.left[up], b.right[up]};
    for (int k = 0; k < 2; k++) {
        const uint32_t c = kids[k];
        if (c >= N - 1 || b.numTris[c] < minTris) continue;
        const uint32_t tr = b.trips[c];
        if (tr == 0u) all = false; else fewest = tr < fewest ? tr : fewest;
    }
    if (!all || fewest >= 33u) return;
    if (atomicExch(&b.queued[up], 1u) != 0u) return; /* both children finished in the same round: one of them lists the parent */
    b.chain[up] = fewest + 1u;
    next[atomicAdd(nextCount, 1u)] = up;
}

struct FitOut {
    TbAabbNode* nodesA; uint8_t* primsA; TbPrimitiveMeta* metaA; /* layout A */
    TbNodeB* nodesB; TbTriB* trisB;                               /* layout B */
    uint32_t* count; uint32_t* height; uint32_t* stamp;           /* per node scratch */
    /* top level: sorted leaf k -> instance index (a layout-B leaf ref names the instance); null for triangles */
    const uint32_t* leafMap;
};

__device__ __forceinline__ void put_node(TbAabbNode* nodes, uint32_t i, tb3 mn, tb3 mx, uint32_t fx, uint32_t fy)
{
    const tb3 c = (mn + mx) * 0.5f, h = mx - c;
    TbAabbNode n;
    n.center[0] = c.x; n.center[1] = c.y; n.center[2] = c.z; n.flags = fx;
    n.halfDim[0] = h.x; n.halfDim[1] = h.y; n.halfDim[2] = h.z; n.rightNodeIndex = fy;
    nodes[i] = n;
}

__global__ __launch_bounds__(BLOCK) void bvh_fit_leaves(const float* positions, const uint32_t* triVertexIndex, const uint32_t* triGeometry,
    const uint32_t* triPrimitive,
                                                        const uint32_t* triFlags, const unsigned long long* keys, uint32_t N, FitOut o)
{
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= N) return;
    const uint32_t tri = (uint32_t)keys[k]; /* sorted position k -> input triangle */
    const tb3 v0 = vertex(positions, triVertexIndex, tri, 0), v1 = vertex(positions, triVertexIndex, tri, 1), v2 = vertex(positions, triVertexIndex, tri, 2);
    float* p = (float*)(o.primsA + 40ull * k); /* 40-B packed primitive: type + 9 floats */
    ((uint32_t*)p)[0] = 1u;
    p[1] = v0.x; p[2] = v0.y; p[3] = v0.z; p[4] = v1.x; p[5] = v1.y; p[6] = v1.z; p[7] = v2.x; p[8] = v2.y; p[9] = v2.z;
    TbPrimitiveMeta m; m.GeometryContributionToHitGroupIndex = triGeometry[tri]; m.PrimitiveIndex = triPrimitive[tri]; m.GeometryFlags = triFlags[tri];
    o.metaA[k] = m;
    TbTriB t;
    t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z; t.geometryIndex = m.GeometryContributionToHitGroupIndex;
    t.v1[0] = v1.x; t.v1[1] = v1.y; t.v1[2] = v1.z; t.primitiveIndex = m.PrimitiveIndex;
    t.v2[0] = v2.x; t.v2[1] = v2.y; t.v2[2] = v2.z; t.geometryFlags = m.GeometryFlags;
    o.trisB[k] = t;
    const uint32_t x = (N - 1) + k;
    tb3 mn = tb3_min(tb3_min(v0, v1), v2), mx = tb3_max(tb3_max(v0, v1), v2);
    mn = tb3_min(mn, mx - tb3_splat(0.001f));
    put_node(o.nodesA, x, mn, mx, k | TB_BVH_LEAF_FLAG, 1);
    o.count[x] = 1; o.height[x] = 1;
}

/* one wave-front of the bottom-up fit: the parent box comes from the children's STORED centre / half-extent boxes (ComputeAABBs.hlsli), so
 * the rounding of one level feeds the next and the order is a true dependency; smaller subtree left */
__global__ __launch_bounds__(BLOCK) void bvh_fit_up(uint32_t N, const uint32_t* left, const uint32_t* right, FitOut o, uint32_t t, uint32_t* progress)
{
    const uint32_t up = blockIdx.x * BLOCK + threadIdx.x;
    bool did = false;
    if (up < N - 1 && o.stamp[up] == 0u) {
        uint32_t l = left[up], r = right[up];
        if (children_finished(o.stamp, N, l, r, t)) {
            const uint32_t cl = o.count[l], cr = o.count[r];
            if (cl > cr) { const uint32_t x = l; l = r; r = x; }
            const TbAabbNode nl = o.nodesA[l], nr = o.nodesA[r];
            const tb3 lcen = tb3_make(nl.center[0], nl.center[1], nl.center[2]), lhal = tb3_make(nl.halfDim[0], nl.halfDim[1], nl.halfDim[2]);
            const tb3 rcen = tb3_make(nr.center[0], nr.center[1], nr.center[2]), rhal = tb3_make(nr.halfDim[0], nr.halfDim[1], nr.halfDim[2]);
            const tb3 mn = tb3_min(lcen - lhal, rcen - rhal), mx = tb3_max(lcen + lhal, rcen + rhal);
            put_node(o.nodesA, up, mn, mx, l & TB_BVH_INDEX_MASK, r);
            o.count[up] = cl + cr;
            const uint32_t hl = o.height[l], hr = o.height[r];
            o.height[up] = 1u + (hl > hr ? hl : hr);
            TbNodeB nb;
            nb.cx[0] = lcen.x; nb.cy[0] = lcen.y; nb.cz[0] = lcen.z; nb.hx[0] = lhal.x; nb.hy[0] = lhal.y; nb.hz[0] = lhal.z;
            nb.cx[1] = rcen.x; nb.cy[1] = rcen.y; nb.cz[1] = rcen.z; nb.hx[1] = rhal.x; nb.hy[1] = rhal.y; nb.hz[1] = rhal.z;
            nb.left = l >= N - 1 ? (TB_BVH_LEAF_FLAG | (o.leafMap ? o.leafMap[l - (N - 1)] : l - (N - 1))) : l;
            nb.right = r >= N - 1 ? (TB_BVH_LEAF_FLAG | (o.leafMap ? o.leafMap[r - (N - 1)] : r - (N - 1))) : r;
            nb.pad[0] = nb.pad[1] = 0;
            o.nodesB[up] = nb;
            o.stamp[up] = t; did = true;
        }
    }
    count_progress(progress, did);
}

/* ---- top level over instances (two-level scenes; fallback layer: TopLevelLoadAABBs.hlsli:62-105, CalculateSceneAABBFromBVHs.hlsl,
 * CalculateMortonCodesForAABBs.hlsl, then the same sort / BuildBVHSplits / ComputeAABBs passes as a bottom level and no treelet pass:
 * GpuBVH2Builder.cpp:498-501).  Host twin: bvh_build.cpp BuildTlas; oracle: tbo_build_tlas. */
struct TlasIn { const float* objectToWorld; const float* worldToObject; const uint32_t* blasIndex; const uint32_t* hitGroupBase; const float* blasBoxes; };

__device__ __forceinline__ tb3 xfm_point34_b(const float* m, tb3 v) /* the pinned dp4 order of TransformAABB: one fma chain per row */
{
    return tb3_make(tb_fma(m[2], v.z, tb_fma(m[1], v.y, tb_fma(m[0], v.x, m[3]))), tb_fma(m[6], v.z, tb_fma(m[5], v.y, tb_fma(m[4], v.x, m[7]))),
                    tb_fma(m[10], v.z, tb_fma(m[9], v.y, tb_fma(m[8], v.x, m[11]))));
}

__global__ __launch_bounds__(BLOCK) void tlas_leaf_boxes(uint32_t M, TlasIn in, float* leafC, float* leafH, uint32_t* ordMin, uint32_t* ordMax)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= M) return;
    const float* b = in.blasBoxes + 6ull * in.blasIndex[i]; /* min xyz, max xyz of the structure's root box */
    tb3 mn = tb3_splat(3.402823466e+38f), mx = tb3_splat(-3.402823466e+38f);
    for (int k = 0; k < 8; k++) { /* TransformAABB (RayTracingHelper.hlsli:318-344): the eight corners */
        const tb3 v = xfm_point34_b(in.objectToWorld + 12ull * i, tb3_make((k & 4) ? b[3] : b[0], (k & 2) ? b[4] : b[1], (k & 1) ? b[5] : b[2]));
        mn = tb3_min(mn, v); mx = tb3_max(mx, v);
    }
    const tb3 c = (mn + mx) * 0.5f, h = mx - c;
    leafC[3 * i] = c.x; leafC[3 * i + 1] = c.y; leafC[3 * i + 2] = c.z; leafH[3 * i] = h.x; leafH[3 * i + 1] = h.y; leafH[3 * i + 2] = h.z;
    const tb3 lo = c - h, hi = c + h; /* the scene box is taken from the STORED centre / half-extent boxes (RawDataToAABB) */
    atomicMin(&ordMin[0], f2ord(lo.x)); atomicMin(&ordMin[1], f2ord(lo.y)); atomicMin(&ordMin[2], f2ord(lo.z));
    atomicMax(&ordMax[0], f2ord(hi.x)); atomicMax(&ordMax[1], f2ord(hi.y)); atomicMax(&ordMax[2], f2ord(hi.z));
}

__global__ __launch_bounds__(BLOCK) void tlas_morton(uint32_t M, const float* leafC, const uint32_t* ordMin, const uint32_t* ordMax, unsigned long long* keys)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= M) return;
    const tb3 smin = tb3_make(ord2f(ordMin[0]), ord2f(ordMin[1]), ord2f(ordMin[2])), smax = tb3_make(ord2f(ordMax[0]), ord2f(ordMax[1]), ord2f(ordMax[2]));
    const tb3 dim = tb3_max(smax - smin, tb3_splat(0.00001f));
    const tb3 u = (tb3_make(leafC[3 * i], leafC[3 * i + 1], leafC[3 * i + 2]) - smin) / dim;
    const float ax = tb_min(tb_max(u.x * 1024.0f, 0.0f), 1023.0f), ay = tb_min(tb_max(u.y * 1024.0f, 0.0f), 1023.0f), az = tb_min(tb_max(u.z * 1024.0f, 0.0f),
        1023.0f);
    const uint32_t code = expand10((uint32_t)ay) | (expand10((uint32_t)ax) << 1) | (expand10((uint32_t)az) << 2);
    keys[i] = ((unsigned long long)code << 32) | (unsigned long long)i;
}

__global__ __launch_bounds__(BLOCK) void tlas_fit_leaves(uint32_t M, const unsigned long long* keys, const float* leafC, const float* leafH, TlasIn in,
                                                         TbAabbNode* nodesA, uint8_t* metaA, uint32_t* count, uint32_t* height, uint32_t* order)
{
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= M) return;
    const uint32_t i = (uint32_t)keys[k]; /* sorted leaf k = instance i */
    order[k] = i;
    TbAabbNode n;
    n.center[0] = leafC[3 * i]; n.center[1] = leafC[3 * i + 1]; n.center[2] = leafC[3 * i + 2]; n.flags = k | TB_BVH_LEAF_FLAG;
    n.halfDim[0] = leafH[3 * i]; n.halfDim[1] = leafH[3 * i + 1]; n.halfDim[2] = leafH[3 * i + 2]; n.rightNodeIndex = 1;
    nodesA[(M - 1) + k] = n; count[(M - 1) + k] = 1; height[(M - 1) + k] = 1;
    TbBvhMetadata md; memset(&md, 0, sizeof md);
    for (int j = 0; j < 12; j++) { md.WorldToObject[j] = in.worldToObject[12ull * i + j]; md.ObjectToWorld[j] = in.objectToWorld[12ull * i + j]; }
    md.InstanceIDAndMask = (0u & 0x00ffffffu) | (1u << 24);                          /* InstanceID 0, InstanceMask 1 (TracerBoy.cpp:2049) */
    md.InstanceContributionToHitGroupIndexAndFlags = in.hitGroupBase[i] & 0x00ffffffu; /* flags 0 */
    md.BlasIndex = in.blasIndex[i]; md.InstanceIndex = i;
    memcpy(metaA + 116ull * k, &md, 116);
}

} // namespace

#define BVH_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

/* All pointers are device pointers.  bvhA receives the 16-B header + nodes + primitives + metadata (layout A image,
 * 16 + 32(2N-1) + 52N bytes), nodesB N-1 (at least 1) layout-B nodes, trisB N records; rootHeight the tree depth in
 * nodes (= HostScene::bvhMaxDepth).  scratch: bvh_gpu_scratch_bytes(N) bytes. */
static size_t round256(size_t b) { return (b + 255) / 256 * 256; }
static size_t treelet_scratch_bytes(uint32_t N)
{
    const size_t nodes = 2ull * N - 1;
    return round256(sizeof(TbBox6) * nodes) /* boxes */ + round256(4 * nodes) /* numTris */ + 3 * round256(4ull * N) /* trips, chain, queued */
           + 2 * round256(4ull * (N / 7 + 2)) /* two lists */;
}
extern "C" size_t bvh_gpu_scratch_bytes(uint32_t N)
{
    size_t sortTmp = 0;
    unsigned long long* nullKeys = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, sortTmp, nullKeys, nullKeys, (size_t)N, 0, 62, (hipStream_t)0);
    const size_t nodes = 2ull * N - 1;
    return 2 * round256(8ull * N) /* keys in/out */ + 3 * round256(4 * nodes) /* parent, count, height */ + 3 * round256(4ull * N) /* left, right, stamp */
           + round256(64) /* bounds */ + round256(256) /* counters */ + round256(sortTmp) + treelet_scratch_bytes(N);
}

/* Runs `launch(t)` for t = 1, 2, ... until the device counter *progress says that all `total` inner nodes are finished; the counter
 * is read back every `batch` launches (a launch that finds nothing to do costs a few microseconds, a read-back a round trip). */
/* one pinned host word per thread for the counters the host reads back between launches (a copy into pageable memory goes through a
 * staging kernel: 90 us each, 13 ms of a 3 M-triangle build before this) */
static uint32_t* pinned_word()
{
    static thread_local uint32_t* w = nullptr;
    if (!w && hipHostMalloc((void**)&w, 64, hipHostMallocDefault) != hipSuccess) w = nullptr;
    return w;
}
static hipError_t read_word(hipStream_t stream, const uint32_t* device, uint32_t& out)
{
    uint32_t* w = pinned_word();
    if (!w) { BVH_TRY(hipMemcpyAsync(&out, device, 4, hipMemcpyDeviceToHost, stream)); return hipStreamSynchronize(stream); }
    BVH_TRY(hipMemcpyAsync(w, device, 4, hipMemcpyDeviceToHost, stream));
    BVH_TRY(hipStreamSynchronize(stream));
    out = *(volatile uint32_t*)w;
    return hipSuccess;
}

template <class L>
static hipError_t run_levels(hipStream_t stream, uint32_t* progress, uint32_t total, L launch)
{
    BVH_TRY(hipMemsetAsync(progress, 0, 4, stream));
    uint32_t done = 0, t = 1;
    const uint32_t batch = 8;
    while (done < total) {
        if (t > 65536u) return hipErrorLaunchFailure; /* a cycle in the hierarchy: cannot happen for a tree */
        for (uint32_t k = 0; k < batch; k++, t++) launch(t);
        BVH_TRY(read_word(stream, progress, done));
    }
    return hipSuccess;
}

extern "C" hipError_t bvh_gpu_build(hipStream_t stream, const float* positions, const uint32_t* triVertexIndex, const uint32_t* triGeometry,
    const uint32_t* triPrimitive,
                                    const uint32_t* triFlags, uint32_t N, uint32_t treeletPasses, uint8_t* scratch, size_t scratchBytes, uint8_t* bvhA,
                                        TbNodeB* nodesB,
                                    TbTriB* trisB, uint32_t* rootHeight)
{
    if (N == 0) return hipErrorInvalidValue;
    const size_t nodes = 2ull * N - 1;
    uint8_t* at = scratch;
    auto take = [&](size_t bytes) { uint8_t* p = at; at += (bytes + 255) / 256 * 256; return p; };
    unsigned long long* keysIn = (unsigned long long*)take(8ull * N);
    unsigned long long* keysOut = (unsigned long long*)take(8ull * N);
    uint32_t* parent = (uint32_t*)take(4 * nodes);
    uint32_t* count = (uint32_t*)take(4 * nodes);
    uint32_t* height = (uint32_t*)take(4 * nodes);
    uint32_t* left = (uint32_t*)take(4ull * N);
    uint32_t* right = (uint32_t*)take(4ull * N);
    uint32_t* stamp = (uint32_t*)take(4ull * N);
    uint32_t* bounds = (uint32_t*)take(64);
    uint32_t* counters = (uint32_t*)take(256); /* [0] progress of a level-synchronous pass, [1] / [2] the two treelet lists' lengths */
    size_t sortTmp = 0;
    BVH_TRY(rocprim::radix_sort_keys(nullptr, sortTmp, keysIn, keysOut, (size_t)N, 0, 62, stream));
    uint8_t* sortScratch = take(sortTmp);
    TreeletBufs tb;
    tb.left = left; tb.right = right; tb.parent = parent; tb.stamp = stamp;
    tb.boxes = (TbBox6*)take(sizeof(TbBox6) * nodes); tb.numTris = (uint32_t*)take(4 * nodes);
    tb.trips = (uint32_t*)take(4ull * N); tb.chain = (uint32_t*)take(4ull * N); tb.queued = (uint32_t*)take(4ull * N);
    uint32_t* lists[2]; lists[0] = (uint32_t*)take(4ull * (N / 7 + 2)); lists[1] = (uint32_t*)take(4ull * (N / 7 + 2));
    if ((size_t)(at - scratch) > scratchBytes) return hipErrorInvalidValue;

    BVH_TRY(hipMemsetAsync(bounds, 0xff, 12, stream));      /* ordMin = 0xffffffff */
    BVH_TRY(hipMemsetAsync(bounds + 4, 0x00, 12, stream));  /* ordMax = 0 */
    const uint32_t blocksN = (N + BLOCK - 1) / BLOCK, blocksInner = N > 1 ? (N - 1 + BLOCK - 1) / BLOCK : 1;
    hipLaunchKernelGGL(bvh_bounds, dim3(blocksN < 2048u ? blocksN : 2048u), dim3(BLOCK), 0, stream, positions, triVertexIndex, N, bounds, bounds + 4);
    hipLaunchKernelGGL(bvh_morton, dim3(blocksN), dim3(BLOCK), 0, stream, positions, triVertexIndex, N, (const uint32_t*)bounds, (const uint32_t*)(bounds + 4),
        keysIn);
    BVH_TRY(rocprim::radix_sort_keys(sortScratch, sortTmp, keysIn, keysOut, (size_t)N, 0, 62, stream));
    if (N > 1) hipLaunchKernelGGL(bvh_hierarchy, dim3(blocksInner), dim3(BLOCK), 0, stream, (const unsigned long long*)keysOut, N, left, right, parent);
    /* TreeletReorder::Optimize (TreeletReorder.cpp:38-109): MinTrianglesPerTreelet 7, 14, 28 for the three PREFER_FAST_TRACE passes */
    for (uint32_t pass = 0, minTris = 7; pass < treeletPasses && minTris <= N; pass++, minTris *= 2) {
        if (pass == 0) { /* boxes and triangle counts of the LBVH; the rebuilds keep both exact from here on */
            hipLaunchKernelGGL(bvh_treelet_leaves, dim3(blocksN), dim3(BLOCK), 0, stream, positions, triVertexIndex, (const unsigned long long*)keysOut, N, tb);
            BVH_TRY(hipMemsetAsync(stamp, 0, 4ull * N, stream));
            BVH_TRY(run_levels(stream, counters, N - 1, [&](uint32_t t) { hipLaunchKernelGGL(bvh_treelet_up, dim3(blocksInner), dim3(BLOCK), 0, stream, N, tb,
                t, counters); }));
        }
        uint32_t cur = 0, n = 0;
        BVH_TRY(hipMemsetAsync(counters + 1, 0, 8, stream));
        hipLaunchKernelGGL(bvh_treelet_classify, dim3(blocksInner), dim3(BLOCK), 0, stream, N, minTris, tb, counters + 1, lists[0]);
        BVH_TRY(read_word(stream, counters + 1, n));
        while (n) { /* one round per link of the longest chain of dependent treelets */
            if (n > N / 7 + 1) return hipErrorLaunchFailure;
            hipLaunchKernelGGL(bvh_treelet_rebuild, dim3(n), dim3(64), 0, stream, N, minTris, tb, (const uint32_t*)(counters + 1 + cur),
                (const uint32_t*)lists[cur]);
            BVH_TRY(hipMemsetAsync(counters + 1 + (cur ^ 1u), 0, 4, stream));
            hipLaunchKernelGGL(bvh_treelet_advance, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, N, minTris, tb,
                (const uint32_t*)(counters + 1 + cur), (const uint32_t*)lists[cur],
                               counters + 1 + (cur ^ 1u), lists[cur ^ 1u]);
            cur ^= 1u;
            BVH_TRY(read_word(stream, counters + 1 + cur, n));
        }
    }
    const uint64_t offBoxes = 16, offPrims = offBoxes + 32 * nodes, offMeta = offPrims + 40ull * N, total = offMeta + 12ull * N;
    const TbBvhHeader hdr = {(uint32_t)offBoxes, (uint32_t)offPrims, (uint32_t)offMeta, (uint32_t)total};
    BVH_TRY(hipMemcpyAsync(bvhA, &hdr, 16, hipMemcpyHostToDevice, stream));
    FitOut o;
    o.nodesA = (TbAabbNode*)(bvhA + offBoxes); o.primsA = bvhA + offPrims; o.metaA = (TbPrimitiveMeta*)(bvhA + offMeta);
    o.nodesB = nodesB; o.trisB = trisB; o.count = count; o.height = height; o.stamp = stamp; o.leafMap = nullptr;
    hipLaunchKernelGGL(bvh_fit_leaves, dim3(blocksN), dim3(BLOCK), 0, stream, positions, triVertexIndex, triGeometry, triPrimitive, triFlags,
        (const unsigned long long*)keysOut, N, o);
    if (N > 1) {
        BVH_TRY(hipMemsetAsync(stamp, 0, 4ull * N, stream));
        BVH_TRY(run_levels(stream, counters, N - 1, [&](uint32_t t) { hipLaunchKernelGGL(bvh_fit_up, dim3(blocksInner), dim3(BLOCK), 0, stream, N,
            (const uint32_t*)left, (const uint32_t*)right, o, t, counters); }));
    }
    BVH_TRY(hipMemcpyAsync(rootHeight, height, 4, hipMemcpyDeviceToDevice, stream)); /* node 0 is the root (the only leaf when N == 1) */
    BVH_TRY(hipStreamSynchronize(stream)); /* hdr lives on this stack frame */
    return hipGetLastError();
}

/* Top level on the GPU.  Device inputs: per instance objectToWorld / worldToObject (12 floats each), structure index, first hit-group
 * record; per structure its root box (min xyz, max xyz).  Outputs: the layout-A image (16 + 32 (2M - 1) + 116 M bytes), M - 1 layout-B
 * nodes whose leaf refs name instances, rootRef (LEAF | instance when M == 1), the depth in nodes.  scratch: bvh_gpu_tlas_scratch_bytes(M). */
extern "C" size_t bvh_gpu_tlas_scratch_bytes(uint32_t M)
{
    size_t sortTmp = 0;
    unsigned long long* nullKeys = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, sortTmp, nullKeys, nullKeys, (size_t)M, 0, 62, (hipStream_t)0);
    const size_t nodes = 2ull * M - 1;
    return 2 * round256(8ull * M) + 3 * round256(4 * nodes) + 4 * round256(4ull * M) + 2 * round256(12ull * M) + round256(64) + round256(256) +
        round256(sortTmp);
}

extern "C" hipError_t bvh_gpu_build_tlas(hipStream_t stream, uint32_t M, const float* objectToWorld, const float* worldToObject, const uint32_t* blasIndex,
    const uint32_t* hitGroupBase,
                                         const float* blasBoxes, uint8_t* scratch, size_t scratchBytes, uint8_t* tlasA, TbNodeB* topNodes,
                                             uint32_t* rootRefOut, uint32_t* rootHeight)
{
    if (M == 0) return hipErrorInvalidValue;
    const size_t nodes = 2ull * M - 1;
    uint8_t* at = scratch;
    auto take = [&](size_t bytes) { uint8_t* p = at; at += (bytes + 255) / 256 * 256; return p; };
    unsigned long long* keysIn = (unsigned long long*)take(8ull * M);
    unsigned long long* keysOut = (unsigned long long*)take(8ull * M);
    uint32_t* parent = (uint32_t*)take(4 * nodes);
    uint32_t* count = (uint32_t*)take(4 * nodes);
    uint32_t* height = (uint32_t*)take(4 * nodes);
    uint32_t* left = (uint32_t*)take(4ull * M);
    uint32_t* right = (uint32_t*)take(4ull * M);
    uint32_t* stamp = (uint32_t*)take(4ull * M);
    uint32_t* order = (uint32_t*)take(4ull * M);
    float* leafC = (float*)take(12ull * M);
    float* leafH = (float*)take(12ull * M);
    uint32_t* bounds = (uint32_t*)take(64);
    uint32_t* counters = (uint32_t*)take(256);
    size_t sortTmp = 0;
    BVH_TRY(rocprim::radix_sort_keys(nullptr, sortTmp, keysIn, keysOut, (size_t)M, 0, 62, stream));
    uint8_t* sortScratch = take(sortTmp);
    if ((size_t)(at - scratch) > scratchBytes) return hipErrorInvalidValue;
    const TlasIn in = {objectToWorld, worldToObject, blasIndex, hitGroupBase, blasBoxes};
    BVH_TRY(hipMemsetAsync(bounds, 0xff, 12, stream));
    BVH_TRY(hipMemsetAsync(bounds + 4, 0x00, 12, stream));
    const uint32_t blocksM = (M + BLOCK - 1) / BLOCK, blocksInner = M > 1 ? (M - 1 + BLOCK - 1) / BLOCK : 1;
    hipLaunchKernelGGL(tlas_leaf_boxes, dim3(blocksM), dim3(BLOCK), 0, stream, M, in, leafC, leafH, bounds, bounds + 4);
    hipLaunchKernelGGL(tlas_morton, dim3(blocksM), dim3(BLOCK), 0, stream, M, (const float*)leafC, (const uint32_t*)bounds, (const uint32_t*)(bounds + 4),
        keysIn);
    BVH_TRY(rocprim::radix_sort_keys(sortScratch, sortTmp, keysIn, keysOut, (size_t)M, 0, 62, stream));
    if (M > 1) hipLaunchKernelGGL(bvh_hierarchy, dim3(blocksInner), dim3(BLOCK), 0, stream, (const unsigned long long*)keysOut, M, left, right, parent);
    const uint64_t offBoxes = 16, offMeta = offBoxes + 32 * nodes, total = offMeta + 116ull * M;
    const TbBvhHeader hdr = {(uint32_t)offBoxes, (uint32_t)offMeta, (uint32_t)offMeta, (uint32_t)total};
    BVH_TRY(hipMemcpyAsync(tlasA, &hdr, 16, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(tlas_fit_leaves, dim3(blocksM), dim3(BLOCK), 0, stream, M, (const unsigned long long*)keysOut, (const float*)leafC, (const float*)leafH,
        in,
                       (TbAabbNode*)(tlasA + offBoxes), tlasA + offMeta, count, height, order);
    if (M > 1) {
        FitOut o; memset(&o, 0, sizeof o);
        o.nodesA = (TbAabbNode*)(tlasA + offBoxes); o.nodesB = topNodes; o.count = count; o.height = height; o.stamp = stamp; o.leafMap = order;
        BVH_TRY(hipMemsetAsync(stamp, 0, 4ull * M, stream));
        BVH_TRY(run_levels(stream, counters, M - 1, [&](uint32_t t) { hipLaunchKernelGGL(bvh_fit_up, dim3(blocksInner), dim3(BLOCK), 0, stream, M,
            (const uint32_t*)left, (const uint32_t*)right, o, t, counters); }));
        const uint32_t zero = 0;
        BVH_TRY(hipMemcpyAsync(rootRefOut, &zero, 4, hipMemcpyHostToDevice, stream));
    } else {
        uint32_t first = 0;
        BVH_TRY(read_word(stream, order, first));
        first |= TB_BVH_LEAF_FLAG;
        BVH_TRY(hipMemcpyAsync(rootRefOut, &first, 4, hipMemcpyHostToDevice, stream));
    }
    BVH_TRY(hipMemcpyAsync(rootHeight, height, 4, hipMemcpyDeviceToDevice, stream));
    BVH_TRY(hipStreamSynchronize(stream)); /* hdr / zero live on this stack frame */
    return hipGetLastError();
}
