/* bvh_build.cpp -- host BVH2 construction for the HIP traversal kernels.
 *
 * builder 0 (default, "LBVH"): the tree the reference's fallback layer would hand to
 *   SoftwareRayTraceCS minus its treelet pass -- 30-bit Morton codes (y,x,z interleave,
 *   /root/reference/D3D12RaytracingFallback/src/CalculateMortonCodesBindings.h:116-149), sort,
 *   Karras-2012 hierarchy (BuildBVHSplits.hlsli:33-131), bottom-up fit with one triangle per leaf,
 *   the 0.001 thin-box padding (RayTracingHelper.hlsli:251-263) and "smaller subtree on the left"
 *   (ComputeAABBs.hlsli:152-156).  Checked bit-for-bit against oracle/bvh_ref.cpp.
 * builder 1 ("SAH"): top-down 32-bin surface-area-heuristic build in the spirit of the reference's
 *   unused CpuBVH2Builder.cpp:249-516; same node numbering (inner 0..N-2, leaf N-1+k), same box
 *   arithmetic, so both layouts and the traversal code are shared.
 * builder 3 ("LBVH + treelets"): builder 0 followed by the fallback layer's treelet passes before the fit -- the tree a
 *   PREFER_FAST_TRACE bottom-level build (TracerBoy.cpp:1970) hands to SoftwareRayTraceCS: three passes over treelets of
 *   seven leaves with MinTrianglesPerTreelet 7, 14, 28 (TreeletReorder.cpp:38-109, TreeletReorder.hlsl:38-311,
 *   FindTreelets.hlsl:27-88).  Checked bit-for-bit against oracle/bvh_ref.cpp; builder 4 is the same on the GPU.
 * Both emit layout A (the fallback layer's memory image, used by the CPU checker and exported through
 * tb_host_scene_view) and layout B (what the kernels fetch, tb_abi.h).
 */
#include "host_scene.h"
#include "../../../include/tb_vec.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <exception>
#include <stdexcept>
#include <thread>
#include <mutex>

namespace tbhost {

namespace {

struct Bounds { tb3 mn, mx; };

struct Tree {
    uint32_t N = 0;                    /* leaves = references (one per triangle unless the SAH builder pre-split some, presplitReferences) */
    std::vector<uint32_t> order;       /* sorted position k -> reference (= input triangle where nothing was pre-split) */
    std::vector<uint32_t> left, right; /* children of inner node i (node ids; leaf k = N-1+k) */
    /* pre-split references (option "presplit", builder 1): reference r is the part of triangle refTri[r] inside refBox[r]; both empty = reference r is
     * triangle r with its own bounds */
    std::vector<uint32_t> refTri; std::vector<Bounds> refBox;
};

inline tb3 P(const HostScene& s, uint32_t tri, int k) { const float* p = &s.positions[3ull * s.triVertexIndex[3ull * tri + k]];
    return tb3_make(p[0], p[1], p[2]); }
inline uint32_t triOfRef(const Tree& t, uint32_t r) { return t.refTri.empty() ? r : t.refTri[r]; }
inline Bounds boundsOfRef(const HostScene& s, const Tree& t, uint32_t r)
{
    if (!t.refBox.empty()) return t.refBox[r];
    Bounds b; b.mn = tb3_splat(3.402823466e+38f); b.mx = tb3_splat(-3.402823466e+38f);
    for (int k = 0; k < 3; k++) { const tb3 v = P(s, r, k); b.mn = tb3_min(b.mn, v); b.mx = tb3_max(b.mx, v); }
    return b;
}

template <class F> void parallelFor(size_t n, F f)
{
    unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1;
    size_t nt = n < 65536 ? 1 : std::min<size_t>(hw, 16);
    if (nt == 1) { f((size_t)0, n); return; }
    /* an exception in a worker (bad_alloc) is carried to the caller instead of ending the process */
    std::vector<std::thread> th; size_t chunk = (n + nt - 1) / nt;
    std::exception_ptr failed; std::mutex fm;
    for (size_t t = 0; t < nt; t++) { size_t a = t * chunk, b = std::min(n, a + chunk);
        if (a < b) th.emplace_back([=, &failed, &fm]() { try { f(a, b); } catch (...) { std::lock_guard<std::mutex> g(fm); if (!failed) failed = std::current_exception(); } }); }
    for (auto& t : th) t.join();
    if (failed) std::rethrow_exception(failed);
}

inline uint32_t expand10(uint32_t v) { v &= 0x3ff; v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3;
    v = (v | (v << 2)) & 0x09249249; return v; }

/* 30-bit Morton code of a centroid inside the scene box (CalculateMortonCodesBindings.h:116-149): axis 0 <- y, axis 1 <- x, axis 2 <- z */
inline uint32_t mortonCode(tb3 c, tb3 smin, tb3 dim)
{
    tb3 u = (c - smin) / dim;
    float ax = tb_min(tb_max(u.x * 1024.0f, 0.0f), 1023.0f), ay = tb_min(tb_max(u.y * 1024.0f, 0.0f), 1023.0f), az = tb_min(tb_max(u.z * 1024.0f, 0.0f),
        1023.0f);
    return expand10((uint32_t)ay) | (expand10((uint32_t)ax) << 1) | (expand10((uint32_t)az) << 2);
}

/* sort by (code, element index) and build the Karras-2012 hierarchy over the sorted codes (BuildBVHSplits.hlsli:33-131) */
void sortAndSplit(std::vector<uint64_t>& keys, Tree& t)
{
    const uint32_t N = t.N;
    std::sort(keys.begin(), keys.end()); /* ties broken by element index: the build's definition */
    t.order.resize(N);
    std::vector<uint32_t> codes(N);
    for (uint32_t i = 0; i < N; i++) { t.order[i] = (uint32_t)keys[i]; codes[i] = (uint32_t)(keys[i] >> 32); }
    if (N < 2) return;
    t.left.resize(N - 1); t.right.resize(N - 1);
    auto delta = [&](int64_t a, int64_t b) -> int {
        if (b < 0 || b >= (int64_t)N) return -1;
        uint32_t x = codes[(size_t)a] ^ codes[(size_t)b];
        if (x) return __builtin_clz(x);
        uint32_t y = (uint32_t)a ^ (uint32_t)b;
        return (y ? __builtin_clz(y) : 32) + 31;
    };
    parallelFor(N - 1, [&](size_t a, size_t b) {
        for (int64_t i = (int64_t)a; i < (int64_t)b; i++) {
            int d = delta(i, i + 1) - delta(i, i - 1); d = (d > 0) - (d < 0);
            int dmin = delta(i, i - d);
            int64_t lmax = 2; while (delta(i, i + lmax * d) > dmin) lmax *= 4;
            int64_t l = 0; for (int64_t st = lmax / 2; st > 0; st /= 2) if (delta(i, i + (l + st) * d) > dmin) l += st;
            int64_t j = i + l * d, first = std::min(i, j), last = std::max(i, j);
            int dn = delta(first, last);
            int64_t split = first, step = last - first;
            do { step = (step + 1) >> 1; int64_t ns = split + step; if (ns < last && delta(first, ns) > dn) split = ns; } while (step > 1);
            t.left[(size_t)i] = (split == first) ? (N - 1) + (uint32_t)split : (uint32_t)split;
            t.right[(size_t)i] = (split + 1 == last) ? (N - 1) + (uint32_t)split + 1 : (uint32_t)split + 1;
        }
    });
}

void buildLbvh(const HostScene& s, Tree& t)
{
    const uint32_t N = t.N;
    tb3 smin = tb3_splat(3.402823466e+38f), smax = tb3_splat(-3.402823466e+38f);
    for (uint32_t i = 0; i < N; i++) for (int k = 0; k < 3; k++) { tb3 v = P(s, i, k); smin = tb3_min(v, smin); smax = tb3_max(v, smax); }
    tb3 dim = tb3_max(smax - smin, tb3_splat(0.00001f));
    std::vector<uint64_t> keys(N);
    parallelFor(N, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            tb3 c = (P(s, (uint32_t)i, 0) + P(s, (uint32_t)i, 1) + P(s, (uint32_t)i, 2)) / 3.0f;
            keys[i] = ((uint64_t)mortonCode(c, smin, dim) << 32) | (uint64_t)i;
        }
    });
    sortAndSplit(keys, t);
}

/* ---- binned SAH ---------------------------------------------------------------------------- */
inline Bounds emptyB() { Bounds b; b.mn = tb3_splat(3.402823466e+38f); b.mx = tb3_splat(-3.402823466e+38f); return b; }
inline void grow(Bounds& b, tb3 p) { b.mn = tb3_min(b.mn, p); b.mx = tb3_max(b.mx, p); }
inline void grow(Bounds& b, const Bounds& o) { b.mn = tb3_min(b.mn, o.mn); b.mx = tb3_max(b.mx, o.mx); }
inline float area(const Bounds& b) { tb3 d = b.mx - b.mn; if (d.x < 0 || d.y < 0 || d.z < 0) return 0.0f; return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x); }

/* Top-down binned SAH (32 bins per axis, one triangle per leaf; median split along the widest axis where no bin boundary separates the
 * centroids).  Numbering: inner nodes in preorder (a subtree over c triangles holds c - 1 of them, so the left child of inner node i is i + 1 and
 * the right child i + [triangles on the left]), leaf k = the k-th triangle of the final left-to-right order -- both follow from a range's place
 * alone, so subtrees are built by several threads (the large nodes at the top bin their triangles in parallel first) and the tree is the one a
 * single depth-first thread numbers (scripts/tree_digest.py).  min / max are exact and order-free (tb_math.h), the counts are integers. */
void buildSah(const HostScene& s, Tree& t)
{
    const uint32_t N = t.N;
    std::vector<Bounds> tb(N); std::vector<tb3> cen(N);
    parallelFor(N, [&](size_t a, size_t z) { for (size_t i = a; i < z; i++) { const Bounds b = boundsOfRef(s, t, (uint32_t)i);
        tb[i] = b; cen[i] = (b.mn + b.mx) * 0.5f; } });
    std::vector<uint32_t> ids(N); for (uint32_t i = 0; i < N; i++) ids[i] = i;
    if (N >= 2) { t.left.assign(N - 1, 0); t.right.assign(N - 1, 0); }
    constexpr int B = 32;
    struct Bins { Bounds bb[3][B]; uint32_t bc[3][B]; Bounds cb; };
    constexpr uint32_t WIDE = 1u << 17; /* ranges from here up are scanned by several threads */
    /* where [begin, end) is cut; reorders ids inside the range */
    bool topPhase = true; /* wide ranges are scanned by several threads only while ONE thread splits the top of the tree: a worker thread of the
                           * second phase that met a wide range used to start 16 more threads of its own */
    auto split = [&](uint32_t begin, uint32_t end) -> uint32_t {
        const uint32_t count = end - begin;
        const bool wide = topPhase && count >= WIDE;
        Bounds cb = emptyB();
        if (wide) {
            std::mutex m;
            parallelFor(count, [&](size_t a, size_t z) { Bounds l = emptyB(); for (size_t i = a; i < z; i++) grow(l, cen[ids[begin + i]]);
                std::lock_guard<std::mutex> g(m); grow(cb, l); });
        } else for (uint32_t i = begin; i < end; i++) grow(cb, cen[ids[i]]);
        const tb3 ext = cb.mx - cb.mn;
        const int axis = (ext.x >= ext.y && ext.x >= ext.z) ? 0 : (ext.y >= ext.z ? 1 : 2);
        uint32_t mid = begin + count / 2;
        const float e = tb3_get(ext, axis);
        bool cut = false;
        if (e > 0.0f && count > 2) {
            float k0[3], k1[3]; bool use[3];
            for (int ax = 0; ax < 3; ax++) { const float ea = tb3_get(ext, ax); use[ax] = ea > 0.0f; k0[ax] = tb3_get(cb.mn, ax);
                k1[ax] = use[ax] ? (float)B * (1.0f - 1e-6f) / ea : 0.0f; }
            auto binOf = [&](uint32_t id, int ax) { int b = (int)((tb3_get(cen[id], ax) - k0[ax]) * k1[ax]); return b < 0 ? 0 : (b >= B ? B - 1 : b); };
            Bounds bb[3][B]; uint32_t bc[3][B];
            for (int ax = 0; ax < 3; ax++) for (int i = 0; i < B; i++) { bb[ax][i] = emptyB(); bc[ax][i] = 0; }
            auto scan = [&](size_t a, size_t z, Bounds (*lb)[B], uint32_t (*lc)[B]) {
                for (size_t i = a; i < z; i++) { const uint32_t id = ids[begin + i];
                    for (int ax = 0; ax < 3; ax++) if (use[ax]) { const int b = binOf(id, ax); grow(lb[ax][b], tb[id]); lc[ax][b]++; } } };
            if (wide) {
                std::mutex m;
                parallelFor(count, [&](size_t a, size_t z) {
                    std::vector<Bounds> lbv(3 * B, emptyB()); std::vector<uint32_t> lcv(3 * B, 0);
                    scan(a, z, (Bounds(*)[B])lbv.data(), (uint32_t(*)[B])lcv.data());
                    std::lock_guard<std::mutex> g(m);
                    for (int ax = 0; ax < 3; ax++) for (int i = 0; i < B; i++) { grow(bb[ax][i], lbv[ax * B + i]); bc[ax][i] += lcv[ax * B + i]; } });
            } else scan(0, count, bb, bc);
            float bestCost = 3.402823466e+38f; int bestAxis = -1, bestBin = -1;
            for (int ax = 0; ax < 3; ax++) {
                if (!use[ax]) continue;
                float ra[B]; Bounds acc = emptyB(); uint32_t rc[B]; uint32_t c = 0;
                for (int i = B - 1; i > 0; i--) { grow(acc, bb[ax][i]); c += bc[ax][i]; ra[i] = area(acc); rc[i] = c; }
                acc = emptyB(); c = 0;
                for (int i = 0; i < B - 1; i++) { grow(acc, bb[ax][i]); c += bc[ax][i]; if (c == 0 || rc[i + 1] == 0) continue;
                    const float cost = area(acc) * (float)c + ra[i + 1] * (float)rc[i + 1]; if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestBin = i; } }
            }
            if (bestAxis >= 0) {
                auto it = std::stable_partition(ids.begin() + begin, ids.begin() + end, [&](uint32_t id) { return binOf(id, bestAxis) <= bestBin; });
                mid = (uint32_t)(it - ids.begin());
                cut = mid > begin && mid < end;
            }
        }
        if (!cut) { /* median along the widest axis (also the fallback for coincident centroids) */
            mid = begin + count / 2;
            std::stable_sort(ids.begin() + begin, ids.begin() + end, [&](uint32_t a, uint32_t b) { return tb3_get(cen[a], axis) < tb3_get(cen[b], axis); });
        }
        return mid;
    };
    struct Job { uint32_t begin, end, id; }; /* an inner node: its triangles and its number */
    auto expand = [&](const Job& j, std::vector<Job>& out) {
        const uint32_t mid = split(j.begin, j.end), nl = mid - j.begin, nr = j.end - mid;
        const uint32_t l = nl == 1 ? (N - 1) + j.begin : j.id + 1, r = nr == 1 ? (N - 1) + mid : j.id + nl;
        t.left[j.id] = l; t.right[j.id] = r;
        if (nr > 1) out.push_back(Job{mid, j.end, r});
        if (nl > 1) out.push_back(Job{j.begin, mid, l});
    };
    if (N >= 2) {
        /* the top of the tree, largest range first, until there is work for every thread; then one depth-first walk per remaining subtree */
        std::vector<Job> open; open.push_back(Job{0, N, 0});
        unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1;
        const size_t threads = N < 65536 ? 1 : std::min<size_t>(hw, 16);
        /* ... and until no open range is wide any more (the workers below scan their ranges alone) */
        for (;;) {
            size_t big = 0; for (size_t i = 1; i < open.size(); i++) if (open[i].end - open[i].begin > open[big].end - open[big].begin) big = i;
            if (open.empty()) break;
            const uint32_t bigCount = open[big].end - open[big].begin;
            if (!(threads > 1 && ((open.size() < 8 * threads && bigCount >= 4096) || bigCount >= WIDE))) break;
            const Job j = open[big]; open.erase(open.begin() + big);
            expand(j, open);
        }
        topPhase = false;
        std::sort(open.begin(), open.end(), [](const Job& a, const Job& b) { return a.end - a.begin > b.end - b.begin; });
        std::atomic<size_t> next{0};
        std::exception_ptr failed; std::mutex fm;
        auto worker = [&]() { std::vector<Job> stack;
            try {
                for (size_t k = next++; k < open.size(); k = next++) { stack.clear(); stack.push_back(open[k]);
                    while (!stack.empty()) { const Job j = stack.back(); stack.pop_back(); expand(j, stack); } }
            } catch (...) { std::lock_guard<std::mutex> g(fm); if (!failed) failed = std::current_exception(); next = open.size(); } };
        if (threads == 1) worker();
        else { std::vector<std::thread> th; for (size_t i = 0; i < threads; i++) th.emplace_back(worker); for (auto& x : th) x.join(); }
        if (failed) std::rethrow_exception(failed); /* surfaces through tb_load_scene as an error code */
    }
    t.order = ids; /* leaf k is the k-th triangle from the left */
}

/* ---- pre-split references (option "presplit" = percent of extra references allowed; builder 1) ---------------------------------------
 * A BVH with one triangle per leaf pays for every large or diagonal triangle with a box that is mostly empty, and so do all its
 * ancestors.  Before the top-down build the references whose boxes hold the most EMPTY surface are cut in two -- the triangle is clipped to
 * both halves and each half becomes a reference of its own with the clipped part's box (Ernst & Greiner 2007, "Early split clipping"; the
 * choice of what to cut and where after Karras & Aila 2013: largest box surface beyond what the clipped polygon needs first, the plane the
 * coarsest one of a power-of-two grid over the scene that crosses the box, so that neighbouring triangles are cut by the same planes) --
 * until `percent` % more references exist than triangles.  The builder then sees N' references; the leaves hold the WHOLE triangle
 * (vertices and indices copied per reference), only their boxes are the parts': a ray meets the triangle in whichever part's box it enters,
 * tests it like any other, and a second part's test of the same triangle finds t0 == best.t and commits nothing (`t0 < best`,
 * TraverseFunction.hlsli:297).  Closest hits are those of any other tree over the same triangles; the oracle walks the same image.
 * MEASURED AND NOT USED BY ANY WORKLOAD (round 6, docs/experiments/r6.md): pictures stay the same bits and triangle tests fall (vw-van 6.69 ->
 * 4.99 per sample) but box tests RISE (55.5 -> 61.2; Teapot 49.9 -> 64.1 from 18 extra references): the large triangles such a cut removes from
 * the top of the tree are what gives a near-first closest-hit walk its early, tight `closest`.  Off by default (option presplit = 0). */
struct Poly { int n; double v[10][3]; };
inline void clipPoly(Poly& p, int axis, double pos, bool keepBelow)
{
    Poly o; o.n = 0;
    for (int i = 0; i < p.n; i++) {
        const double* a = p.v[i]; const double* b = p.v[(i + 1) % p.n];
        const bool ia = keepBelow ? a[axis] <= pos : a[axis] >= pos, ib = keepBelow ? b[axis] <= pos : b[axis] >= pos;
        if (ia && o.n < 10) { memcpy(o.v[o.n++], a, 24); }
        if (ia != ib && o.n < 10) {
            const double t = (pos - a[axis]) / (b[axis] - a[axis]);
            for (int k = 0; k < 3; k++) o.v[o.n][k] = a[k] + t * (b[k] - a[k]);
            o.v[o.n][axis] = pos; o.n++;
        }
    }
    p = o;
}
inline Poly polyOfRef(const HostScene& s, uint32_t tri, const Bounds& b)
{
    Poly p; p.n = 3;
    for (int k = 0; k < 3; k++) { const tb3 v = P(s, tri, k); p.v[k][0] = v.x; p.v[k][1] = v.y; p.v[k][2] = v.z; }
    for (int a = 0; a < 3 && p.n >= 3; a++) { clipPoly(p, a, tb3_get(b.mn, a), false); if (p.n >= 3) clipPoly(p, a, tb3_get(b.mx, a), true); }
    return p;
}
/* bounds of a clipped polygon, rounded outwards and widened by a few ulps of the coordinates (the clip points are computed, not given),
 * inside `within` */
inline Bounds boundsOfPoly(const Poly& p, const Bounds& within)
{
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < p.n; i++) for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], p.v[i][k]); mx[k] = std::max(mx[k], p.v[i][k]); }
    Bounds b;
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        const double pad = 4.0 * 1.1920929e-7 * std::max(std::fabs(mn[k]), std::fabs(mx[k]));
        lo[k] = std::nextafterf((float)(mn[k] - pad), -3.402823466e+38f); hi[k] = std::nextafterf((float)(mx[k] + pad), 3.402823466e+38f);
    }
    b.mn = tb3_max(tb3_make(lo[0], lo[1], lo[2]), within.mn); b.mx = tb3_min(tb3_make(hi[0], hi[1], hi[2]), within.mx);
    return b;
}
inline double idealArea(const Poly& p) /* surface of the flattest box a polygon of this area vector could have: twice the sum of its projections */
{
    double ax = 0, ay = 0, az = 0;
    for (int i = 1; i + 1 < p.n; i++) {
        const double e1[3] = {p.v[i][0] - p.v[0][0], p.v[i][1] - p.v[0][1], p.v[i][2] - p.v[0][2]}, e2[3] = {p.v[i + 1][0] - p.v[0][0], p.v[i + 1][1] - p.v[0][1],
            p.v[i + 1][2] - p.v[0][2]};
        ax += e1[1] * e2[2] - e1[2] * e2[1]; ay += e1[2] * e2[0] - e1[0] * e2[2]; az += e1[0] * e2[1] - e1[1] * e2[0];
    }
    return std::fabs(ax) + std::fabs(ay) + std::fabs(az);
}

void presplitReferences(const HostScene& s, Tree& t, int percent)
{
    const uint32_t N = t.N;
    const uint64_t budget = (uint64_t)N * (uint64_t)std::max(percent, 0) / 100u;
    if (!budget || N < 2) return;
    t.refTri.resize(N); t.refBox.resize(N);
    Bounds scene = emptyB();
    for (uint32_t i = 0; i < N; i++) { t.refTri[i] = i; Bounds b = emptyB(); for (int k = 0; k < 3; k++) grow(b, P(s, i, k)); t.refBox[i] = b; grow(scene, b); }
    const tb3 sext = tb3_max(scene.mx - scene.mn, tb3_splat(1e-30f));
    auto areaD = [](const Bounds& b) { const double x = (double)b.mx.x - b.mn.x, y = (double)b.mx.y - b.mn.y, z = (double)b.mx.z - b.mn.z; return 2.0 * (x * y + y * z + z * x); };
    /* what a cut can remove: the box's surface beyond the clipped polygon's own */
    auto priority = [&](uint32_t r) { const Poly p = polyOfRef(s, t.refTri[r], t.refBox[r]); return p.n < 3 ? 0.0 : areaD(t.refBox[r]) - idealArea(p); };
    struct Item { double pri; uint32_t ref; bool operator<(const Item& o) const { return pri < o.pri || (pri == o.pri && ref > o.ref); } };
    std::vector<Item> heap; heap.reserve(N + (size_t)budget * 2);
    double totalArea = 0;
    for (uint32_t i = 0; i < N; i++) { totalArea += areaD(t.refBox[i]); heap.push_back(Item{priority(i), i}); }
    std::make_heap(heap.begin(), heap.end());
    /* Worth a reference of its own (and an inner node above it, whose box every ray through the neighbourhood then tests): a box many times
     * the scene's mean triangle box of which a good part is empty.  Cutting ordinary triangles only makes the tree deeper: with a plain budget
     * of +10 % the Teapot went from 49.9 to 68.7 box tests per sample. */
    constexpr double kBig = 16.0, kEmpty = 0.3;
    const double bigArea = kBig * totalArea / (double)N;
    uint64_t made = 0;
    while (made < budget && !heap.empty()) {
        std::pop_heap(heap.begin(), heap.end()); const Item it = heap.back(); heap.pop_back();
        if (!(it.pri > 0.0)) break;
        if (!(areaD(t.refBox[it.ref]) > bigArea && it.pri > kEmpty * areaD(t.refBox[it.ref]))) continue;
        const Bounds b = t.refBox[it.ref];
        const tb3 ext = b.mx - b.mn;
        int axis = (ext.x >= ext.y && ext.x >= ext.z) ? 0 : (ext.y >= ext.z ? 1 : 2);
        /* the coarsest plane of the scene's power-of-two grid strictly inside the box on that axis (the midpoint if none down to 2^-30) */
        const double lo = tb3_get(b.mn, axis), hi = tb3_get(b.mx, axis), s0 = tb3_get(scene.mn, axis), se = tb3_get(sext, axis);
        double plane = 0.5 * (lo + hi);
        for (int level = 1; level <= 30; level++) {
            const double cell = se / (double)(1u << level);
            const double k = std::floor((lo - s0) / cell) + 1.0, cand = s0 + k * cell;
            if (cand > lo && cand < hi) { plane = cand; break; }
        }
        const float pf = (float)plane;
        if (!(pf > tb3_get(b.mn, axis) && pf < tb3_get(b.mx, axis))) continue; /* too thin to cut in float */
        Poly whole = polyOfRef(s, t.refTri[it.ref], b);
        if (whole.n < 3) continue;
        Poly L = whole, R = whole; clipPoly(L, axis, (double)pf, true); clipPoly(R, axis, (double)pf, false);
        if (L.n < 3 || R.n < 3) continue;
        Bounds wl = b, wr = b;
        if (axis == 0) { wl.mx.x = pf; wr.mn.x = pf; } else if (axis == 1) { wl.mx.y = pf; wr.mn.y = pf; } else { wl.mx.z = pf; wr.mn.z = pf; }
        const Bounds bl = boundsOfPoly(L, wl), br = boundsOfPoly(R, wr);
        const uint32_t nr = (uint32_t)t.refTri.size();
        t.refBox[it.ref] = bl; t.refTri.push_back(t.refTri[it.ref]); t.refBox.push_back(br); made++;
        heap.push_back(Item{areaD(bl) - idealArea(L), it.ref}); std::push_heap(heap.begin(), heap.end());
        heap.push_back(Item{areaD(br) - idealArea(R), nr}); std::push_heap(heap.begin(), heap.end());
    }
    if (!made) { t.refTri.clear(); t.refBox.clear(); return; }
    t.N = (uint32_t)t.refTri.size();
}

/* ---- reinsertion passes (after Bittner, Hapala & Havran, "Fast insertion-based optimization of bounding volume
 * hierarchies", 2013): every subtree in turn is cut out and put back where the sum of the inner nodes' surface areas -- what
 * a random ray pays in box tests -- grows least, until a pass gains little.  The place is found by a branch-and-bound descent
 * from the root (the cost induced on the ancestors only grows on the way down, so a branch is dropped as soon as that alone
 * exceeds the best place found).  The top-down build's greedy early splits cost the most where large and small triangles
 * mix: cornell-box goes from 57.9 to 52.1 box tests per sample (+5 % frame rate), the procedural 6 k-triangle scene from
 * 32.9 to 26.7.  Moves that would deepen the tree beyond depthLimit are skipped (the traversal stack in LDS is as deep as
 * the tree).  Works on node ids: inner 0..N-2 with the root at 0, leaf N-1+k = order[k].  (Measured and dropped: accepting
 * moves by the simulated cost of the kernels' own walk over path-traced sample rays, child order included -- 47.9 box
 * tests on cornell-box, the same frame time.) */
void optimizeByReinsertion(const HostScene& s, Tree& t, int maxPasses, double minGain)
{
    const uint32_t N = t.N; if (N < 4) return;
    const uint32_t M = 2 * N - 1, NONE = 0xffffffffu;
    std::vector<Bounds> box(M); std::vector<float> sa(M); std::vector<uint32_t> parent(M, NONE); std::vector<uint16_t> height(M, 0);
    auto isLeaf = [&](uint32_t x) { return x >= N - 1; };
    auto unite = [&](const Bounds& a, const Bounds& b) { Bounds u = a; grow(u, b); return u; };
    for (uint32_t k = 0; k < N; k++) { Bounds b = boundsOfRef(s, t, t.order[k]);
        b.mn = tb3_min(b.mn, b.mx - tb3_splat(0.001f)); box[N - 1 + k] = b; sa[N - 1 + k] = area(b); }
    for (uint32_t i = 0; i + 1 < N; i++) { parent[t.left[i]] = i; parent[t.right[i]] = i; }
    auto pull = [&](uint32_t x) { const uint32_t l = t.left[x], r = t.right[x]; box[x] = unite(box[l], box[r]); sa[x] = area(box[x]);
        height[x] = (uint16_t)(1 + std::max(height[l], height[r])); };
    std::vector<uint32_t> st;
    { /* inner boxes and heights bottom-up */
        std::vector<uint32_t> pre; pre.reserve(M); st.push_back(0);
        while (!st.empty()) { uint32_t x = st.back(); st.pop_back(); pre.push_back(x); if (!isLeaf(x)) { st.push_back(t.left[x]); st.push_back(t.right[x]); } }
        for (size_t w = pre.size(); w-- > 0;) if (!isLeaf(pre[w])) pull(pre[w]);
    }
    /* the traversal stack in LDS is as deep as the tree (1 KB per level and workgroup): 31 levels leave room for five workgroups
     * per CU, 39 for four (context.cpp picks the kernel copy accordingly), so a tree is not allowed to grow across such a step */
    const uint32_t h0 = height[0] + 1u; /* in nodes, like HostScene::bvhMaxDepth */
    const uint32_t depthLimit = (h0 <= 31u ? 31u : (h0 <= 39u ? 39u : h0)) - 1u;
    auto refit = [&](uint32_t x) { for (; x != NONE; x = parent[x]) pull(x); };
    auto totalCost = [&]() { double c = 0; for (uint32_t i = 0; i + 1 < N; i++) c += sa[i]; return c; };
    auto replaceChild = [&](uint32_t p, uint32_t from, uint32_t to) { if (t.left[p] == from) t.left[p] = to; else t.right[p] = to; parent[to] = p; };
    double cost = totalCost();
    std::vector<uint32_t> cand(M - 1);
    struct Item { uint32_t node, depth; float induced; };
    std::vector<Item> todo;
    for (int pass = 0; pass < maxPasses; pass++) {
        for (uint32_t x = 1; x < M; x++) cand[x - 1] = x;
        std::stable_sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) { return sa[a] > sa[b]; });
        /* only the largest <share> percent of the subtrees (HostScene::reinsertionShare: option "reinsertion_share", the builder word of
         * tb_host_scene_load -- no environment override: a left-over variable used to change the tree without a trace) */
        size_t tried = 0;
        const double share = (double)s.reinsertionShare;
        const size_t limit = share >= 100.0 ? cand.size() : (size_t)((double)cand.size() * std::max(share, 0.0) / 100.0);
        for (uint32_t x : cand) {
            if (tried++ >= limit) break;
            const uint32_t p = parent[x]; if (p == NONE || p == 0) continue; /* children of the root stay: node 0 remains the root */
            const uint32_t g = parent[p], sib = t.left[p] == x ? t.right[p] : t.left[p];
            /* cut: the sibling takes the parent's place, the parent node p is kept to become the new junction */
            replaceChild(g, p, sib); parent[p] = NONE; refit(g);
            const Bounds bx = box[x]; const float ax = sa[x]; const uint32_t hx = height[x];
            /* going back next to the old sibling is the move to beat */
            uint32_t best = sib; float bestInc = area(unite(box[sib], bx));
            for (uint32_t a = parent[sib]; a != NONE; a = parent[a]) bestInc += area(unite(box[a], bx)) - sa[a];
            /* junction {y, x} in place of y costs area(y U x) plus what every ancestor of y grows by */
            todo.clear(); todo.push_back(Item{0, 0, 0.0f});
            while (!todo.empty()) {
                const Item it = todo.back(); todo.pop_back();
                if (it.induced + ax >= bestInc) continue;
                const uint32_t y = it.node;
                const float direct = area(unite(box[y], bx));
                if (y != 0 && it.induced + direct < bestInc && it.depth + 1u + std::max<uint32_t>(hx,
                    height[y]) <= depthLimit) { bestInc = it.induced + direct; best = y; }
                if (!isLeaf(y)) {
                    const float below = it.induced + direct - sa[y];
                    if (below + ax < bestInc) {
                        const uint32_t l = t.left[y], r = t.right[y];
                        /* the child that x enlarges less is searched first (popped last-in first-out) */
                        const float gl = area(unite(box[l], bx)) - sa[l], gr = area(unite(box[r], bx)) - sa[r];
                        if (gl <= gr) { todo.push_back(Item{r, it.depth + 1u, below}); todo.push_back(Item{l, it.depth + 1u, below}); }
                        else { todo.push_back(Item{l, it.depth + 1u, below}); todo.push_back(Item{r, it.depth + 1u, below}); }
                    }
                }
            }
            const uint32_t q = parent[best];
            replaceChild(q, best, p); t.left[p] = best; t.right[p] = x; parent[best] = p; parent[x] = p;
            refit(p);
        }
        const double now = totalCost();
        static const bool verbose = getenv("TB_REINSERT_VERBOSE") != nullptr; /* diagnostics only: changes no tree */
        if (verbose) fprintf(stderr, "reinsertion pass %d: SAH cost %.6g -> %.6g\n", pass, cost, now);
        const bool goOn = now < cost * (1.0 - minGain);
        cost = now;
        if (!goOn) break;
    }
}

/* ---- treelet passes (builder 3) -------------------------------------------------------------------------------------
 * Karras & Aila 2013 as the fallback layer runs it.  A pass starts at the lowest nodes that hold at least `minTris`
 * triangles and climbs; at every node on the way the 7-leaf treelet below it (grown by opening the largest box) is
 * rebuilt as the binary tree of least cost over all 2^7 leaf subsets, the six inner node ids being reused.  Cost of a
 * subset = area of its box + least cost of a split into two subsets; a single leaf costs area / root area
 * (TreeletReorder.hlsl:22-26,121-127,158-168 -- mixed units, kept as they are).  Boxes are min/max of exact inputs, so
 * the subset boxes here come from a running union (box[m] = box[m without its lowest leaf] U that leaf) instead of the
 * reference's seven-way loop: same bits.  A climbing group of the reference stops after 33 treelets and, where two
 * groups meet, the second to arrive goes on; the rule here (and in oracle/bvh_ref.cpp and bvh_kernels.hip): the group
 * that has done fewer treelets goes on. */
struct TreeletPass {
    const uint32_t N; std::vector<uint32_t>&left, &right;
    std::vector<Bounds> box;
    static float sarea(const Bounds& b) { const tb3 d = b.mx - b.mn; return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z); }
    bool leaf(uint32_t x) const { return x >= N - 1; }

    void rebuild(uint32_t root)
    {
        uint32_t leafNode[7], inner[6]; Bounds sub[128]; float cost[128]; uint8_t cut[128];
        inner[0] = root; leafNode[0] = left[root]; leafNode[1] = right[root];
        for (uint32_t n = 2; n < 7; n++) { /* open the inner node with the largest box */
            float best = 0.0f; uint32_t at = 0, node = 0;
            for (uint32_t i = 0; i < n; i++) if (!leaf(leafNode[i])) { const float a = sarea(box[leafNode[i]]); if (a > best) { best = a; at = i;
                node = leafNode[i]; } }
            inner[n - 1] = node; leafNode[at] = left[node]; leafNode[n] = right[node];
        }
        const float rootArea = sarea(box[root]);
        for (uint32_t m = 1; m < 128; m++) {
            const uint32_t low = (uint32_t)__builtin_ctz(m), rest = m & (m - 1);
            sub[m] = box[leafNode[low]]; if (rest) grow(sub[m], sub[rest]);
            cost[m] = rest ? sarea(sub[m]) : sarea(sub[m]) / rootArea;
        }
        static uint8_t bySize[8][35]; static uint8_t sizeCount[8]; static bool tables = false;
        if (!tables) { for (uint32_t m = 1; m < 128; m++) { const int k = __builtin_popcount(m); bySize[k][sizeCount[k]++] = (uint8_t)m; } tables = true; }
        for (int k = 2; k <= 7; k++) for (uint32_t q = 0; q < sizeCount[k]; q++) {
            const uint32_t m = bySize[k][q], d = (m - 1) & m;
            float least = 3.402823466e+38f; uint32_t arg = 0;
            uint32_t p = (0u - d) & m;
            do { const float c = cost[p] + cost[m ^ p]; if (c < least) { least = c; arg = p; } p = (p - d) & m; } while (p);
            cost[m] += least; cut[m] = (uint8_t)arg;
        }
        /* hand the inner ids out again: parent first, left child's id before the right child's, right subtree first */
        uint32_t used = 1; struct Todo { uint32_t mask, node; } todo[7]; uint32_t n = 0; todo[n++] = {127u, root};
        while (n) {
            const Todo t = todo[--n];
            const uint32_t lm = cut[t.mask], rm = t.mask ^ lm; uint32_t ln, rn;
            if (lm & (lm - 1)) { ln = inner[used++]; todo[n++] = {lm, ln}; } else ln = leafNode[__builtin_ctz(lm)];
            if (rm & (rm - 1)) { rn = inner[used++]; todo[n++] = {rm, rn}; } else rn = leafNode[__builtin_ctz(rm)];
            left[t.node] = ln; right[t.node] = rn;
        }
        for (int j = 5; j >= 0; j--) { const uint32_t x = inner[j]; box[x] = box[left[x]]; grow(box[x], box[right[x]]); }
    }

    void run(uint32_t minTris, const std::vector<Bounds>& leafBox)
    {
        const uint32_t M = 2 * N - 1;
        std::vector<uint32_t> pre; pre.reserve(M); { std::vector<uint32_t> st(1, 0u); while (!st.empty()) { const uint32_t x = st.back(); st.pop_back();
            pre.push_back(x); if (!leaf(x)) { st.push_back(left[x]); st.push_back(right[x]); } } }
        std::vector<uint32_t> tris(M, 0); std::vector<uint8_t> done(M, 0); /* done: treelets rebuilt by the group standing here, 0 = nobody came */
        for (size_t w = pre.size(); w-- > 0;) {
            const uint32_t x = pre[w];
            if (leaf(x)) { box[x] = leafBox[x - (N - 1)]; tris[x] = 1; continue; }
            const uint32_t l = left[x], r = right[x];
            box[x] = box[l]; grow(box[x], box[r]); tris[x] = tris[l] + tris[r];
            if (tris[x] < minTris) continue;
            uint32_t fewest = 0xffu; bool everyGroupCame = true, anyBig = false;
            for (uint32_t c : {l, r}) if (tris[c] >= minTris) { anyBig = true; if (!done[c]) everyGroupCame = false;
                else fewest = std::min<uint32_t>(fewest, done[c]); }
            if (!anyBig) done[x] = 1; else if (everyGroupCame && fewest < 33) done[x] = (uint8_t)(fewest + 1);
            if (done[x]) rebuild(x);
        }
    }
};

void treeletPasses(const HostScene& s, Tree& t, uint32_t passes)
{
    const uint32_t N = t.N; if (N < 7) return;
    std::vector<Bounds> leafBox(N);
    for (uint32_t k = 0; k < N; k++) { /* the leaf's centre / half-extent box turned back into min / max (FindTreelets.hlsl:14-27) */
        Bounds b = emptyB(); for (int v = 0; v < 3; v++) grow(b, P(s, t.order[k], v));
        b.mn = tb3_min(b.mn, b.mx - tb3_splat(0.001f));
        const tb3 c = (b.mn + b.mx) * 0.5f, h = b.mx - c;
        leafBox[k].mn = c - h; leafBox[k].mx = c + h;
    }
    TreeletPass tp{N, t.left, t.right, std::vector<Bounds>(2 * (size_t)N - 1)};
    uint32_t minTris = 7;
    for (uint32_t i = 0; i < passes && minTris <= N; i++, minTris *= 2) tp.run(minTris, leafBox);
}

/* one bottom-level structure over ALL triangles of s */
void BuildBvhSingle(HostScene& s, int builder)
{
    const uint64_t N64 = s.triGeometry.size();
    if (N64 == 0) throw std::runtime_error("BuildBvh: no triangles");
    if (N64 > 0x00ffffffull) throw std::runtime_error("BuildBvh: more than 2^24-1 triangles does not fit the 24-bit node indices of the reference layout");
    Tree t; t.N = (uint32_t)N64;
    if (builder == 1 && s.presplitPercent > 0) presplitReferences(s, t, s.presplitPercent);
    if (t.N > 0x00ffffffu) throw std::runtime_error("BuildBvh: more than 2^24-1 references after pre-splitting (lower option presplit)");
    const uint32_t N = t.N; /* leaves: references (= triangles unless pre-split) */
    if (builder == 1) {
        buildSah(s, t);
        const int passes = s.reinsertionPasses >= 0 ? s.reinsertionPasses : (N <= 4096 ? 16 : 3); /* option "reinsertion_passes" */
        if (passes > 0) optimizeByReinsertion(s, t, passes, N <= 4096 ? 1e-6 : 5e-3);
    } else {
        buildLbvh(s, t);
        if (builder == 3) treeletPasses(s, t, 3);
    }

    const uint64_t numNodes = 2ull * N - 1;
    const uint64_t offBoxes = 16, offPrims = offBoxes + 32 * numNodes, offMeta = offPrims + 40ull * N, total = offMeta + 12ull * N;
    if (total > 0xffffffffull) throw std::runtime_error("BuildBvh: BVH image exceeds 4 GiB");
    s.bvhA.assign((size_t)total, 0);
    TbBvhHeader hdr = {(uint32_t)offBoxes, (uint32_t)offPrims, (uint32_t)offMeta, (uint32_t)total};
    memcpy(s.bvhA.data(), &hdr, 16);
    TbAabbNode* nodes = (TbAabbNode*)(s.bvhA.data() + offBoxes);
    uint8_t* prims = s.bvhA.data() + offPrims;
    TbPrimitiveMeta* meta = (TbPrimitiveMeta*)(s.bvhA.data() + offMeta);
    s.trisB.resize(N);
    for (uint32_t k = 0; k < N; k++) {
        uint32_t tri = triOfRef(t, t.order[k]);
        TbPrimitive p; p.PrimitiveType = 1;
        TbTriB tbv;
        for (int v = 0; v < 3; v++) {
            tb3 q = P(s, tri, v);
            float* d = v == 0 ? p.v0 : (v == 1 ? p.v1 : p.v2); d[0] = q.x; d[1] = q.y; d[2] = q.z;
            float* e = v == 0 ? tbv.v0 : (v == 1 ? tbv.v1 : tbv.v2); e[0] = q.x; e[1] = q.y; e[2] = q.z;
        }
        memcpy(prims + 40ull * k, &p, 40);
        meta[k].GeometryContributionToHitGroupIndex = s.triGeometry[tri];
        meta[k].PrimitiveIndex = s.triPrimitive[tri];
        meta[k].GeometryFlags = s.triFlags[tri];
        tbv.geometryIndex = s.triGeometry[tri]; tbv.primitiveIndex = s.triPrimitive[tri]; tbv.geometryFlags = s.triFlags[tri];
        s.trisB[k] = tbv;
    }
    /* fit boxes bottom-up: visit order = reverse of a pre-order walk */
    std::vector<uint32_t> walk; walk.reserve((size_t)numNodes);
    std::vector<uint32_t> depth((size_t)numNodes, 0);
    { std::vector<uint32_t> st; st.push_back(0); depth[0] = 1; uint32_t maxD = 1;
      while (!st.empty()) { uint32_t x = st.back(); st.pop_back(); walk.push_back(x);
          if (N > 1 && x < N - 1) { uint32_t l = t.left[x], r = t.right[x]; depth[l] = depth[r] = depth[x] + 1; if (depth[l] > maxD) maxD = depth[l];
              st.push_back(l); st.push_back(r); } }
      s.bvhMaxDepth = maxD; }
    std::vector<uint32_t> count((size_t)numNodes, 0);
    auto center = [&](uint32_t i) { return tb3_make(nodes[i].center[0], nodes[i].center[1], nodes[i].center[2]); };
    auto half = [&](uint32_t i) { return tb3_make(nodes[i].halfDim[0], nodes[i].halfDim[1], nodes[i].halfDim[2]); };
    auto put = [&](uint32_t i, tb3 mn, tb3 mx, uint32_t fx, uint32_t fy) {
        tb3 c = (mn + mx) * 0.5f, h = mx - c;
        nodes[i].center[0] = c.x; nodes[i].center[1] = c.y; nodes[i].center[2] = c.z; nodes[i].flags = fx;
        nodes[i].halfDim[0] = h.x; nodes[i].halfDim[1] = h.y; nodes[i].halfDim[2] = h.z; nodes[i].rightNodeIndex = fy;
    };
    for (size_t w = walk.size(); w-- > 0;) {
        uint32_t x = walk[w];
        if (x >= N - 1) {
            uint32_t k = x - (N - 1);
            const Bounds rb = boundsOfRef(s, t, t.order[k]); /* the triangle's own bounds, or the pre-split part's */
            tb3 mn = rb.mn, mx = rb.mx;
            mn = tb3_min(mn, mx - tb3_splat(0.001f));
            put(x, mn, mx, k | TB_BVH_LEAF_FLAG, 1);
            count[x] = 1;
        } else {
            uint32_t l = t.left[x], r = t.right[x];
            if (count[l] > count[r]) std::swap(l, r);
            tb3 mn = tb3_min(center(l) - half(l), center(r) - half(r));
            tb3 mx = tb3_max(center(l) + half(l), center(r) + half(r));
            put(x, mn, mx, l & TB_BVH_INDEX_MASK, r);
            count[x] = count[l] + count[r];
        }
    }
    /* layout B */
    auto ref = [&](uint32_t node) -> uint32_t { return node >= N - 1 ? (TB_BVH_LEAF_FLAG | (node - (N - 1))) : node; };
    s.nodesB.assign(N > 1 ? N - 1 : 1, TbNodeB{});
    for (uint32_t i = 0; i + 1 < N; i++) {
        uint32_t l = nodes[i].flags & TB_BVH_INDEX_MASK, r = nodes[i].rightNodeIndex;
        TbNodeB nb; memset(&nb, 0, sizeof nb);
        const uint32_t child[2] = {l, r};
        for (int k = 0; k < 2; k++) {
            nb.cx[k] = nodes[child[k]].center[0]; nb.cy[k] = nodes[child[k]].center[1]; nb.cz[k] = nodes[child[k]].center[2];
            nb.hx[k] = nodes[child[k]].halfDim[0]; nb.hy[k] = nodes[child[k]].halfDim[1]; nb.hz[k] = nodes[child[k]].halfDim[2];
        }
        nb.left = ref(l); nb.right = ref(r);
        s.nodesB[i] = nb;
    }
    s.rootRefB = ref(0);
}

/* mul(AffineMatrix, float4(v, 1)) of TransformAABB (RayTracingHelper.hlsli:339); HLSL leaves the order of the dp4 open, the build
 * pins it as an fma chain (include/tb_vec.h) */
inline tb3 xfmPoint34(const float m[12], tb3 v)
{
    return tb3_make(tb_fma(m[2], v.z, tb_fma(m[1], v.y, tb_fma(m[0], v.x, m[3]))), tb_fma(m[6], v.z, tb_fma(m[5], v.y, tb_fma(m[4], v.x, m[7]))),
                    tb_fma(m[10], v.z, tb_fma(m[9], v.y, tb_fma(m[8], v.x, m[11]))));
}

/* Top level over the instances (fallback layer: TopLevelLoadAABBs.hlsli:62-105 leaf boxes + metadata, CalculateSceneAABBFromBVHs.hlsl,
 * CalculateMortonCodesForAABBs.hlsl, the same sort / BuildBVHSplits / ComputeAABBs passes as a bottom level; no treelet pass:
 * GpuBVH2Builder.cpp:498-501).  blasRoot[b] = root box (min, max) of structure b read back from its image. */
void BuildTlas(HostScene& s, const std::vector<Bounds>& blasRoot, std::vector<TbNodeB>& nodesOut /* M - 1 entries */, uint32_t& rootRef, uint32_t& depthOut,
               float rootCenter[3], float rootHalf[3])
{
    const uint32_t M = (uint32_t)s.instances.size();
    struct Box { tb3 c, h; };
    std::vector<Box> leaf(M);
    for (uint32_t i = 0; i < M; i++) { /* TransformAABB (:318-344): the eight corners, then AABBtoBoundingBox */
        const Bounds& b = blasRoot[s.instances[i].blas];
        Bounds w = emptyB();
        for (int k = 0; k < 8; k++) {
            const tb3 v = tb3_make((k & 4) ? b.mx.x : b.mn.x, (k & 2) ? b.mx.y : b.mn.y, (k & 1) ? b.mx.z : b.mn.z);
            grow(w, xfmPoint34(s.instances[i].objectToWorld, v));
        }
        leaf[i].c = (w.mn + w.mx) * 0.5f; leaf[i].h = w.mx - leaf[i].c;
    }
    Tree t; t.N = M;
    { /* scene box from the stored centre / half-extent boxes (RawDataToAABB: centre -+ half), Morton codes of the box centres */
        tb3 smin = tb3_splat(3.402823466e+38f), smax = tb3_splat(-3.402823466e+38f);
        for (uint32_t i = 0; i < M; i++) { smin = tb3_min(leaf[i].c - leaf[i].h, smin); smax = tb3_max(leaf[i].c + leaf[i].h, smax); }
        const tb3 dim = tb3_max(smax - smin, tb3_splat(0.00001f));
        std::vector<uint64_t> keys(M);
        for (uint32_t i = 0; i < M; i++) keys[i] = ((uint64_t)mortonCode(leaf[i].c, smin, dim) << 32) | (uint64_t)i;
        sortAndSplit(keys, t);
    }
    const uint64_t numNodes = 2ull * M - 1, offBoxes = 16, offMeta = offBoxes + 32 * numNodes, total = offMeta + 116ull * M;
    s.tlasA.assign((size_t)total, 0);
    const TbBvhHeader hdr = {(uint32_t)offBoxes, (uint32_t)offMeta, (uint32_t)offMeta, (uint32_t)total};
    memcpy(s.tlasA.data(), &hdr, 16);
    TbAabbNode* nodes = (TbAabbNode*)(s.tlasA.data() + offBoxes);
    for (uint32_t k = 0; k < M; k++) { /* metadata of sorted leaf k */
        const HostScene::Instance& in = s.instances[t.order[k]];
        TbBvhMetadata md; memset(&md, 0, sizeof md);
        memcpy(md.WorldToObject, in.worldToObject, 48); memcpy(md.ObjectToWorld, in.objectToWorld, 48);
        md.InstanceIDAndMask = (0u & 0x00ffffffu) | (1u << 24);                   /* InstanceID 0, InstanceMask 1 (TracerBoy.cpp:2049) */
        md.InstanceContributionToHitGroupIndexAndFlags = in.hitGroupBase & 0x00ffffffu; /* flags 0 */
        md.BlasIndex = in.blas; md.InstanceIndex = t.order[k];
        memcpy(s.tlasA.data() + offMeta + 116ull * k, &md, 116);
    }
    /* fit: same bottom-up pass as a bottom level (ComputeAABBs.hlsli), leaf = one "triangle" (flags.y = 1), no thin-box padding */
    std::vector<uint32_t> walk; walk.reserve((size_t)numNodes);
    std::vector<uint32_t> depth((size_t)numNodes, 0), count((size_t)numNodes, 0);
    { std::vector<uint32_t> st; st.push_back(0); depth[0] = 1; uint32_t maxD = 1;
      while (!st.empty()) { uint32_t x = st.back(); st.pop_back(); walk.push_back(x);
          if (M > 1 && x < M - 1) { uint32_t l = t.left[x], r = t.right[x]; depth[l] = depth[r] = depth[x] + 1; if (depth[l] > maxD) maxD = depth[l];
              st.push_back(l); st.push_back(r); } }
      depthOut = maxD; }
    auto center = [&](uint32_t i) { return tb3_make(nodes[i].center[0], nodes[i].center[1], nodes[i].center[2]); };
    auto half = [&](uint32_t i) { return tb3_make(nodes[i].halfDim[0], nodes[i].halfDim[1], nodes[i].halfDim[2]); };
    auto putCH = [&](uint32_t i, tb3 c, tb3 h, uint32_t fx, uint32_t fy) {
        nodes[i].center[0] = c.x; nodes[i].center[1] = c.y; nodes[i].center[2] = c.z; nodes[i].flags = fx;
        nodes[i].halfDim[0] = h.x; nodes[i].halfDim[1] = h.y; nodes[i].halfDim[2] = h.z; nodes[i].rightNodeIndex = fy;
    };
    for (size_t w = walk.size(); w-- > 0;) {
        const uint32_t x = walk[w];
        if (x >= M - 1) { const uint32_t k = x - (M - 1); putCH(x, leaf[t.order[k]].c, leaf[t.order[k]].h, k | TB_BVH_LEAF_FLAG, 1); count[x] = 1; }
        else {
            uint32_t l = t.left[x], r = t.right[x];
            if (count[l] > count[r]) std::swap(l, r);
            const tb3 mn = tb3_min(center(l) - half(l), center(r) - half(r)), mx = tb3_max(center(l) + half(l), center(r) + half(r));
            const tb3 c = (mn + mx) * 0.5f;
            putCH(x, c, mx - c, l & TB_BVH_INDEX_MASK, r);
            count[x] = count[l] + count[r];
        }
    }
    /* layout B: inner nodes 0..M-2 at the head of the shared node array, a leaf ref is LEAF | instance index */
    auto ref = [&](uint32_t node) -> uint32_t { return node >= M - 1 ? (TB_BVH_LEAF_FLAG | t.order[node - (M - 1)]) : node; };
    nodesOut.assign(M > 1 ? M - 1 : 0, TbNodeB{});
    for (uint32_t i = 0; i + 1 < M; i++) {
        const uint32_t child[2] = {nodes[i].flags & TB_BVH_INDEX_MASK, nodes[i].rightNodeIndex};
        TbNodeB nb; memset(&nb, 0, sizeof nb);
        for (int k = 0; k < 2; k++) {
            nb.cx[k] = nodes[child[k]].center[0]; nb.cy[k] = nodes[child[k]].center[1]; nb.cz[k] = nodes[child[k]].center[2];
            nb.hx[k] = nodes[child[k]].halfDim[0]; nb.hy[k] = nodes[child[k]].halfDim[1]; nb.hz[k] = nodes[child[k]].halfDim[2];
        }
        nb.left = ref(child[0]); nb.right = ref(child[1]);
        nodesOut[i] = nb;
    }
    rootRef = ref(0);
    memcpy(rootCenter, nodes[0].center, 12); memcpy(rootHalf, nodes[0].halfDim, 12);
}

} // namespace

void BuildBvh(HostScene& s, int builder) { BuildBvhWith(s, [builder](HostScene& one) { BuildBvhSingle(one, builder); }, nullptr); }

/* `single` builds one structure over ALL triangles of the scene it is handed (layout A + B, rootRefB, bvhMaxDepth); `tlas` (may be
 * null: the host's BuildTlas) builds the top level from the structures' root boxes.  The GPU builders pass their own (context.cpp). */
void BuildBvhWith(HostScene& s, const std::function<void(HostScene&)>& single, const TlasBuilder& tlas)
{
    if (s.blueNoise0.empty()) LoadBlueNoiseTiles(s); /* system textures are bound with the scene (TracerBoy.cpp:2126-2134) */
    if (s.instances.empty()) {
        single(s);
        s.blasOffsets.assign({0u, (uint32_t)s.bvhA.size()});
        return;
    }
    /* two-level: every bottom-level structure is built by the single-level builder over its own triangle range (object space); the
     * layout-A images go back to back into bvhA, the layout-B nodes and triangles into the shared arrays behind the top-level nodes,
     * their child refs rebased */
    const uint32_t M = (uint32_t)s.instances.size();
    const uint32_t tlasNodes = M > 1 ? M - 1 : 0;
    std::vector<uint8_t> allA; std::vector<TbNodeB> allNodes(tlasNodes); std::vector<TbTriB> allTris;
    std::vector<Bounds> blasRoot(s.blas.size());
    s.blasOffsets.clear();
    uint32_t maxBlasDepth = 0;
    for (size_t b = 0; b < s.blas.size(); b++) {
        HostScene::Blas& bl = s.blas[b];
        HostScene tmp;
        tmp.reinsertionPasses = s.reinsertionPasses; tmp.reinsertionShare = s.reinsertionShare; tmp.presplitPercent = s.presplitPercent;
        tmp.positions.swap(s.positions);
        tmp.triVertexIndex.assign(s.triVertexIndex.begin() + 3ull * bl.firstTri, s.triVertexIndex.begin() + 3ull * (bl.firstTri + bl.numTris));
        tmp.triGeometry.assign(s.triGeometry.begin() + bl.firstTri, s.triGeometry.begin() + bl.firstTri + bl.numTris);
        tmp.triPrimitive.assign(s.triPrimitive.begin() + bl.firstTri, s.triPrimitive.begin() + bl.firstTri + bl.numTris);
        tmp.triFlags.assign(s.triFlags.begin() + bl.firstTri, s.triFlags.begin() + bl.firstTri + bl.numTris);
        try { single(tmp); } catch (...) { tmp.positions.swap(s.positions); throw; }
        tmp.positions.swap(s.positions);
        while (allA.size() % 16) allA.push_back(0);
        bl.offsetA = (uint32_t)allA.size(); s.blasOffsets.push_back(bl.offsetA);
        allA.insert(allA.end(), tmp.bvhA.begin(), tmp.bvhA.end());
        const uint32_t nodeBase = (uint32_t)allNodes.size(), triBase = (uint32_t)allTris.size();
        auto rebase = [&](uint32_t ref) { return (ref & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((ref & ~TB_BVH_LEAF_FLAG) + triBase)) : ref + nodeBase; };
        if (tmp.trisB.size() > 1) for (TbNodeB nd : tmp.nodesB) { nd.left = rebase(nd.left); nd.right = rebase(nd.right); allNodes.push_back(nd); }
        allTris.insert(allTris.end(), tmp.trisB.begin(), tmp.trisB.end());
        bl.rootRefB = rebase(tmp.rootRefB); bl.depth = tmp.bvhMaxDepth;
        maxBlasDepth = std::max(maxBlasDepth, bl.depth);
        const TbAabbNode* root = (const TbAabbNode*)(tmp.bvhA.data() + 16);
        const tb3 c = tb3_make(root->center[0], root->center[1], root->center[2]), h = tb3_make(root->halfDim[0], root->halfDim[1], root->halfDim[2]);
        blasRoot[b].mn = c - h; blasRoot[b].mx = c + h; /* BoundingBoxToAABB (RayTracingHelper.hlsli:237-243) */
    }
    if ((uint64_t)allA.size() > 0xffffffffull) throw std::runtime_error("BuildBvh: BVH images exceed 4 GiB");
    s.blasOffsets.push_back((uint32_t)allA.size());
    std::vector<TbNodeB> top; uint32_t tlasDepth = 0; float rc[3], rh[3];
    if (tlas) {
        std::vector<float> boxes(6 * blasRoot.size());
        for (size_t b = 0; b < blasRoot.size(); b++) { boxes[6 * b] = blasRoot[b].mn.x; boxes[6 * b + 1] = blasRoot[b].mn.y;
            boxes[6 * b + 2] = blasRoot[b].mn.z; boxes[6 * b + 3] = blasRoot[b].mx.x; boxes[6 * b + 4] = blasRoot[b].mx.y; boxes[6 * b + 5] = blasRoot[b].mx.z;
            }
        tlas(s, boxes, top, s.rootRefB, tlasDepth);
    } else BuildTlas(s, blasRoot, top, s.rootRefB, tlasDepth, rc, rh);
    for (uint32_t i = 0; i < tlasNodes; i++) allNodes[i] = top[i];
    if (allNodes.empty()) allNodes.push_back(TbNodeB{});
    s.instancesB.resize(M);
    for (uint32_t i = 0; i < M; i++) {
        TbInstanceB ib; memset(&ib, 0, sizeof ib);
        memcpy(ib.worldToObject, s.instances[i].worldToObject, 48);
        ib.blasRootRef = s.blas[s.instances[i].blas].rootRefB; ib.hitGroupBase = s.instances[i].hitGroupBase; ib.instanceId = 0;
        s.instancesB[i] = ib;
    }
    s.bvhA.swap(allA); s.nodesB.swap(allNodes); s.trisB.swap(allTris);
    s.bvhMaxDepth = tlasDepth + maxBlasDepth; /* the walk's stack holds top-level entries below the bottom-level ones */
}

} // namespace tbhost
