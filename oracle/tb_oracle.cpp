/* tb_oracle.cpp -- scalar CPU restatement of TracerBoy's SoftwareRayTraceCS path.
 *
 * TEST INFRASTRUCTURE ONLY (see tb_oracle.h).  PARITY UNPINNED by the reference (no tests, wall-clock
 * RNG seed, shaders not compilable here); pinned by citations + known answers instead.
 *
 * One thread of the reference compute shader == one call of sample_pixel() here.  The code follows
 * the reference top-down in include order; every block cites the file:line it restates
 * (paths relative to /root/reference/).  "R" in comments = one rand() call; the number and order of
 * R calls is part of the contract.
 *
 * Build: g++ -O2 -ffp-contract=off (oracle/Makefile).  No SIMD intrinsics, not tuned: this is also
 * the "scalar C++ CPU transcription" that bench.py times as cpu_baseline (kind "port").
 */
#include "tb_oracle.h"
#include "../include/tb_vec.h"

#include <atomic>
#include <cstdio>
#include <mutex>
#include <string>
#include <cstdlib>
#include <thread>
#include <vector>

namespace {

/* kernel.glsl:1-10 (EPSILON is re-#defined after SharedShaderStructs.h:3, last one wins) */
const float EPSILON = 0.000001f;
const float PI = 3.1415926535f;
const float LARGE_NUMBER = 1e20f;
const float AIR_IOR = 1.0f;
const float MIN_ROUGHNESS = 0.04f;
const float MIN_ROUGHNESS_SQUARED = (float)(0.04 * 0.04);
const int INVALID_MATERIAL_ID = -1;
const float MIN_T = 0.001f; /* RayGenCommon.h:364 */

struct Ray { tb3 origin, direction; };

inline tb3 to3(const TbFloat3& f) { return tb3_make(f.x, f.y, f.z); }
inline tb3 to3(const float* f) { return tb3_make(f[0], f[1], f[2]); }

struct Ctx {
    const TbSceneView* scene;
    const TbPerFrameConstants* pf;
    uint32_t width, height, x, y;
    float seed;
    TbRayStats* stats;
    /* AOV side channel of one sample (RayGenCommon.h:524-654) */
    tb3 aovNormal, aovAlbedo, aovEmissive, aovWorldPos;
    float aovDistanceToNeighbor, aovDepth;
    bool aovDepthWritten, aovEmissiveWritten;
    uint32_t lastTris, lastBoxes;
    bool heatmapWritten;
    bool selWritten; float selDistance; int selMaterial; /* StatsBuffer +8 / +12 of the selected pixel (RayGenCommon.h:632-648) */
    char rayKind; /* ray-step log only: E bounce ray, S shadow feeler, W interior walk */
};

/* kernel.glsl:39-40  float rand() { return fract(sin(seed++ + GetTime())*43758.5453123); } */
inline float rnd(Ctx& c)
{
    float s = c.seed;
    c.seed = c.seed + 1.0f;
    return tb_frac(tb_sin(s + c.pf->Time) * 43758.5453123f);
}

/* RayGenCommon.h:662-667 */
inline float hash13(tb3 p3)
{
    p3 = tb3_make(tb_frac(p3.x * .1031f), tb_frac(p3.y * .1031f), tb_frac(p3.z * .1031f));
    /* dot() written out unfused: the known-answer seeds of SURVEY.md 8c are for this association */
    float d = (p3.x * (p3.y + 33.33f) + p3.y * (p3.z + 33.33f)) + p3.z * (p3.x + 33.33f);
    p3 = tb3_make(p3.x + d, p3.y + d, p3.z + d);
    return tb_frac((p3.x + p3.y) * p3.z);
}

/* ------------------------------------------------------------------------------------------
 * Software BVH traversal: D3D12RaytracingFallback/src/TraverseFunction.hlsli as configured by
 * RayGenCommon.h:355-362 (FAST_PATH 1, DISABLE_ANYHIT, DISABLE_PROCEDURAL_GEOMETRY), ray flags
 * RAY_FLAG_NONE, instanceFlags 0.  Reads the layout-A image exactly like RayTracingHelper.hlsli.
 * ------------------------------------------------------------------------------------------ */
struct RayData { /* TraverseFunction.hlsli:464-471 */
    tb3 InverseDirection, OriginTimesRayInverseDirection, Shear;
    int kx, ky, kz;
    tb3 AbsInverseDirection; /* abs(InverseDirection), TraverseFunction.hlsli:209, except on degenerate axes (below) */
};

/* The one place where the checker does not follow the reference's box test to the letter: for a ray with
 * d.k == 0 the reference's c*inv - o*inv is inf - inf = NaN on axis k and that axis never rejects a box
 * (TraverseFunction.hlsli:212-214).  Hits are unaffected, but the ray visits every node in its slab.
 * By default the constants of a degenerate axis are replaced (inv = 2^80, o*inv = o 2^80, |inv| = 2^80 (1 + 2^-10)),
 * which turns the same arithmetic into "the origin lies inside the box's slab widened by a thousandth of its
 * half-width" -- see pt_device.hpp.  TB_LITERAL_BOX_TEST=1 restores the literal behaviour (tests/test_host_scene.py
 * shows the images are bit-identical either way). */
static const bool g_literalBoxTest = getenv("TB_LITERAL_BOX_TEST") && atoi(getenv("TB_LITERAL_BOX_TEST")) != 0;
/* The second place (same switch): a ray with a NaN in its origin or direction.  Every product of the watertight triangle test then carries
 * the NaN into U, V and W together (each of the three is built from both sheared coordinates of two vertices), every comparison of
 * RayTriangleIntersect fails, and the ray can hit NOTHING -- but min / max drop the NaN operands of the slab test, so the literal walk
 * visits most or all of the tree to find that out: 1.39 million steps for one ray in 12 000 of the reference's own vw-van scene (its
 * glass materials carry index 0: tests/test_vw_van.py), 86 % of all the steps of the render and three orders of magnitude of a
 * lock-step wave's time.  By default such a ray is a miss at once; results are unchanged, only its BoxesTested differ from the literal walk. */
inline bool RayCannotHit(tb3 o, tb3 d) { return tb_isnan(o.x) || tb_isnan(o.y) || tb_isnan(o.z) || tb_isnan(d.x) || tb_isnan(d.y) || tb_isnan(d.z); }
static const float DEGEN_INV = 1.2089258196146292e24f, DEGEN_AINV = 1.2101064112353466e24f;

/* TraverseFunction.hlsli:431-445 */
inline int GetIndexOfBiggestChannel(tb3 v)
{
    if (v.x > v.y && v.x > v.z) return 0;
    else if (v.y > v.z) return 1;
    else return 2;
}

/* TraverseFunction.hlsli:473-495 (rcp == exact 1/x in this build, tb_math.h) */
inline RayData GetRayData(tb3 o, tb3 d)
{
    RayData r;
    r.InverseDirection = tb3_make(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.OriginTimesRayInverseDirection = o * r.InverseDirection;
    int z = GetIndexOfBiggestChannel(tb3_abs(d));
    r.kx = (z + 1) % 3;
    r.ky = (z + 2) % 3;
    r.kz = z;
    r.AbsInverseDirection = tb3_abs(r.InverseDirection);
    if (!g_literalBoxTest) {
        if (d.x == 0.0f) { r.InverseDirection.x = DEGEN_INV; r.AbsInverseDirection.x = DEGEN_AINV; r.OriginTimesRayInverseDirection.x = o.x * DEGEN_INV; }
        if (d.y == 0.0f) { r.InverseDirection.y = DEGEN_INV; r.AbsInverseDirection.y = DEGEN_AINV; r.OriginTimesRayInverseDirection.y = o.y * DEGEN_INV; }
        if (d.z == 0.0f) { r.InverseDirection.z = DEGEN_INV; r.AbsInverseDirection.z = DEGEN_AINV; r.OriginTimesRayInverseDirection.z = o.z * DEGEN_INV; }
    }
    if (tb3_get(d, r.kz) < 0.0f) { int t = r.kx; r.kx = r.ky; r.ky = t; }
    r.Shear = tb3_make(tb3_get(d, r.kx) / tb3_get(d, r.kz), tb3_get(d, r.ky) / tb3_get(d, r.kz),
                       1.0f / tb3_get(d, r.kz));
    return r;
}

/* TraverseFunction.hlsli:204-221 */
inline bool RayBoxTest(float& resultT, float closestT, const RayData& rd, tb3 c, tb3 h)
{
    /* not `precise` in the reference, so the shader compiler may contract; the build pins the contraction
     * (one fma per component per line), see include/tb_vec.h */
    const tb3 inv = rd.InverseDirection, oi = rd.OriginTimesRayInverseDirection;
    tb3 ai = rd.AbsInverseDirection;
    tb3 relativeMiddle = tb3_make(tb_fma(c.x, inv.x, -oi.x), tb_fma(c.y, inv.y, -oi.y), tb_fma(c.z, inv.z, -oi.z));
    tb3 maxL = tb3_make(tb_fma(h.x, ai.x, relativeMiddle.x), tb_fma(h.y, ai.y, relativeMiddle.y), tb_fma(h.z, ai.z, relativeMiddle.z));
    tb3 minL = tb3_make(tb_fma(-h.x, ai.x, relativeMiddle.x), tb_fma(-h.y, ai.y, relativeMiddle.y), tb_fma(-h.z, ai.z, relativeMiddle.z));
    float minT = tb_max(tb_max(minL.x, minL.y), minL.z);
    float maxT = tb_min(tb_min(maxL.x, maxL.y), maxL.z);
    resultT = tb_max(minT, 0.0f);
    return tb_max(minT, 0.0f) < tb_min(maxT, closestT);
}

/* TraverseFunction.hlsli:232-313, two-sided branch (:273-277, :295-307).  `precise` U,V,W: no
 * contraction, which the whole build guarantees.  Returns through hitT/bary only when accepted. */
inline void RayTriangleIntersect(float& hitT, float bary[2], tb3 o, const RayData& rd, tb3 v0, tb3 v1, tb3 v2)
{
    tb3 a0 = v0 - o, b0 = v1 - o, c0 = v2 - o;
    float Ax = tb3_get(a0, rd.kx), Ay = tb3_get(a0, rd.ky), Az = tb3_get(a0, rd.kz);
    float Bx = tb3_get(b0, rd.kx), By = tb3_get(b0, rd.ky), Bz = tb3_get(b0, rd.kz);
    float Cx = tb3_get(c0, rd.kx), Cy = tb3_get(c0, rd.ky), Cz = tb3_get(c0, rd.kz);
    Ax = tb_fma(-rd.Shear.x, Az, Ax); Ay = tb_fma(-rd.Shear.y, Az, Ay);
    Bx = tb_fma(-rd.Shear.x, Bz, Bx); By = tb_fma(-rd.Shear.y, Bz, By);
    Cx = tb_fma(-rd.Shear.x, Cz, Cx); Cy = tb_fma(-rd.Shear.y, Cz, Cy);
    /* `precise` (:260-262): U, V, W are never contracted */
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    float det = U + V + W;
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return;
    if (det == 0.0f) return;
    Az = rd.Shear.z * Az; Bz = rd.Shear.z * Bz; Cz = rd.Shear.z * Cz;
    const float T = tb_fma(W, Cz, tb_fma(V, Bz, U * Az));
    float signCorrectedT = tb_abs(T);
    if ((T > 0.0f) != (det > 0.0f)) signCorrectedT = -signCorrectedT;
    if (signCorrectedT < 0.0f || signCorrectedT > hitT * tb_abs(det)) return;
    const float rcpDet = 1.0f / det;
    bary[0] = V * rcpDet;
    bary[1] = W * rcpDet;
    hitT = T * rcpDet;
}

struct Committed { float t, bary[2]; uint32_t primitiveIndex, geometryIndex; };

inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline float ldf(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }

const int ORACLE_STACK = 256; /* reference: 16 with no overflow check (RayTracingHlslCompat.h:15) */

/* Optional ray-step log for scheduling studies (scripts/simd_sim.py): with TB_ORACLE_RAY_LOG=<file> every traversal
 * appends "x y frame steps" where steps is the sequence of I (inner node: both children tested) and L (leaf: one
 * triangle tested) events in visit order.  Off by default; not used by any test. */
static FILE* g_rayLog = nullptr;
static std::mutex g_rayLogMutex;
static thread_local std::string* g_raySteps = nullptr;

/* The any-hit alpha test of the reference's hardware path (IsValidHit, SharedHitGroup.h:157-179, applied to candidate
 * hits of non-opaque geometry by the RayQuery loop RayGenCommon.h:423-434).  The software path this file restates compiles
 * it out (DISABLE_ANYHIT, RayGenCommon.h:357); tbo_set_alpha_test(1) turns it on. */
static int g_alphaTest = 0;
bool IsValidHit(const TbSceneView* sc, uint32_t geometryIndex, uint32_t primitiveIndex, float b0, float b1);

/* TraverseFunction.hlsli:537-779 */
bool TraverseTwoLevel(const TbSceneView* sc, tb3 origin, tb3 direction, float TMin, float TMax, Committed& hit, uint32_t& trianglesTested,
    uint32_t& boxesTested);
bool Traverse(const TbSceneView* sc, tb3 origin, tb3 direction, float TMin, float TMax, Committed& hit,
              uint32_t& trianglesTested, uint32_t& boxesTested)
{
    if (sc->numInstances) return TraverseTwoLevel(sc, origin, direction, TMin, TMax, hit, trianglesTested, boxesTested);
    const uint8_t* bvh = sc->bvh;
    trianglesTested = 0; boxesTested = 0;
    hit.t = TMax; hit.bary[0] = hit.bary[1] = 0; hit.primitiveIndex = hit.geometryIndex = 0;
    if (!bvh || sc->numTriangles == 0) return false;
    if (!g_literalBoxTest && RayCannotHit(origin, direction)) return false;
    const uint32_t offBoxes = 16; /* RayTracingHelper.hlsli:69-75 */
    const uint32_t offPrims = ld32(bvh + 4), offMeta = ld32(bvh + 8);
    RayData rd = GetRayData(origin, direction);

    uint32_t stack[ORACLE_STACK];
    int top = 0;
    {
        const uint8_t* n = bvh + offBoxes;
        float unusedT;
        if (RayBoxTest(unusedT, hit.t, rd, to3((const float*)n), to3((const float*)(n + 16)))) stack[top++] = 0; /* :566-580 */
    }
    while (top != 0) {
        uint32_t node = stack[--top]; /* :589 */
        const uint8_t* n = bvh + offBoxes + 32u * node;
        uint32_t flagsX = ld32(n + 12), flagsY = ld32(n + 28);
        if (flagsX & TB_BVH_LEAF_FLAG) { /* :601, :641-709 */
            uint32_t leafIndex = flagsX & ~(TB_BVH_LEAF_FLAG | TB_BVH_PROCEDURAL_FLAG); /* RayTracingHelper.hlsli:51-54 */
            const uint8_t* m = bvh + offMeta + 12u * leafIndex;
            uint32_t geomContribution = ld32(m), primIdx = ld32(m + 4);
            trianglesTested++; /* :662 */
            if (g_raySteps) g_raySteps->push_back('L');
            uint32_t triId = flagsX & TB_BVH_INDEX_MASK; /* :331 */
            const uint8_t* p = bvh + offPrims + 40u * triId + 4; /* RayTracingHelper.hlsli:210-227 */
            tb3 v0 = tb3_make(ldf(p), ldf(p + 4), ldf(p + 8));
            tb3 v1 = tb3_make(ldf(p + 12), ldf(p + 16), ldf(p + 20));
            tb3 v2 = tb3_make(ldf(p + 24), ldf(p + 28), ldf(p + 32));
            float t0 = hit.t, b[2] = {0, 0};
            RayTriangleIntersect(t0, b, origin, rd, v0, v1, v2);
            bool valid = true;
            if (g_alphaTest && t0 < hit.t && t0 > TMin && !(ld32(m + 8) & 1u)) valid = IsValidHit(sc, geomContribution, primIdx, b[0], b[1]);
            if (valid && t0 < hit.t && t0 > TMin) { /* :420-426, commit :685-697 */
                hit.t = t0; hit.bary[0] = b[0]; hit.bary[1] = b[1];
                hit.primitiveIndex = primIdx; hit.geometryIndex = geomContribution;
            }
        } else { /* :717-766 */
            uint32_t l = flagsX & TB_BVH_INDEX_MASK, r = flagsY;
            const uint8_t* ln = bvh + offBoxes + 32u * l;
            const uint8_t* rn = bvh + offBoxes + 32u * r;
            float lt, rt;
            bool lh = RayBoxTest(lt, hit.t, rd, to3((const float*)ln), to3((const float*)(ln + 16)));
            bool rh = RayBoxTest(rt, hit.t, rd, to3((const float*)rn), to3((const float*)(rn + 16)));
            boxesTested += 2; /* :751 */
            if (g_raySteps) g_raySteps->push_back('I');
            if (top + 2 > ORACLE_STACK) return false; /* would have been UB in the reference */
            if (lh && rh) { /* :754-760, StackPush2 :163-173: far first, near on top; ties -> left first */
                bool rightFirst = rt < lt;
                stack[top++] = rightFirst ? l : r;
                stack[top++] = rightFirst ? r : l;
            } else if (lh || rh) {
                stack[top++] = rh ? r : l;
            }
        }
    }
    return hit.t < TMax; /* :776 */
}

/* TraverseFunction.hlsli:537-779 with FAST_PATH 0: the two-level walk of instanced scenes (the !FAST_PATH branch :603-640).
 * Top-level image sc->tlas (TopLevelLoadAABBs.hlsli layout: header, AABB nodes, one BVHMetadata per sorted leaf), bottom-level
 * images inside sc->bvh at sc->blasOffsets[BlasIndex].  One stack, a node counter per level (nodesToProcess[2], :549,:594):
 * a bottom level runs until its counter is zero, then the world ray data come back (:769-773).  The committed hit's hit-group
 * index is InstanceContributionToHitGroupIndex + GeometryContributionToHitGroupIndex (RayGenCommon.h:392 adds the instance INDEX,
 * which is the same number in the reference's own instance table, TracerBoy.cpp:2050). */
inline tb3 MulPoint34(const float* m, tb3 v) /* mul(float3x4, float4(v, 1)): the order of the dp4 pinned as one fma chain per row */
{
    return tb3_make(tb_fma(m[2], v.z, tb_fma(m[1], v.y, tb_fma(m[0], v.x, m[3]))), tb_fma(m[6], v.z, tb_fma(m[5], v.y, tb_fma(m[4], v.x, m[7]))),
                    tb_fma(m[10], v.z, tb_fma(m[9], v.y, tb_fma(m[8], v.x, m[11]))));
}
inline tb3 MulVector34(const float* m, tb3 v) /* mul(float3x4, float4(v, 0)) */
{
    return tb3_make(tb_fma(m[2], v.z, tb_fma(m[1], v.y, m[0] * v.x)), tb_fma(m[6], v.z, tb_fma(m[5], v.y, m[4] * v.x)), tb_fma(m[10], v.z, tb_fma(m[9], v.y,
        m[8] * v.x)));
}

bool TraverseTwoLevel(const TbSceneView* sc, tb3 origin, tb3 direction, float TMin, float TMax, Committed& hit,
                      uint32_t& trianglesTested, uint32_t& boxesTested)
{
    trianglesTested = 0; boxesTested = 0;
    hit.t = TMax; hit.bary[0] = hit.bary[1] = 0; hit.primitiveIndex = hit.geometryIndex = 0;
    const uint8_t* tlas = sc->tlas;
    if (!tlas || !sc->bvh || !sc->blasOffsets || sc->numInstances == 0) return false;
    if (!g_literalBoxTest && RayCannotHit(origin, direction)) return false;
    const uint32_t offBoxes = 16, offInstanceDescs = ld32(tlas + 4); /* GetOffsetToInstanceDesc, RayTracingHelper.hlsli:66-67 */
    RayData rd = GetRayData(origin, direction); /* :545 */
    enum { TOP = 0, BOTTOM = 1 };
    uint32_t nodesToProcess[2] = {0, 0};
    bool processingBottom = false;
    const uint8_t* current = tlas;
    uint32_t instanceOffset = 0;
    tb3 objectOrigin = origin;
    uint32_t stack[ORACLE_STACK];
    int top = 0;
    {
        const uint8_t* n = tlas + offBoxes;
        float unusedT;
        if (RayBoxTest(unusedT, hit.t, rd, to3((const float*)n), to3((const float*)(n + 16)))) { stack[top++] = 0; nodesToProcess[TOP]++; } /* :566-580 */
    }
    while (nodesToProcess[TOP] != 0) { /* :584 */
        do {
            const uint32_t node = stack[--top]; /* :589 */
            nodesToProcess[processingBottom ? BOTTOM : TOP]--;
            const uint8_t* n = current + offBoxes + 32u * node;
            const uint32_t flagsX = ld32(n + 12), flagsY = ld32(n + 28);
            if (flagsX & TB_BVH_LEAF_FLAG) {
                if (!processingBottom) { /* :603-640 */
                    const uint32_t leafIndex = flagsX & ~(TB_BVH_LEAF_FLAG | TB_BVH_PROCEDURAL_FLAG);
                    TbBvhMetadata md; memcpy(&md, tlas + offInstanceDescs + 116u * leafIndex, 116);
                    instanceOffset = md.InstanceContributionToHitGroupIndexAndFlags & 0x00ffffffu;
                    const bool validInstance = ((md.InstanceIDAndMask >> 24) & 0xffu /* ~0 inclusion mask */) != 0;
                    if (validInstance && md.BlasIndex < sc->numBlas) {
                        processingBottom = true;
                        if (top + 1 > ORACLE_STACK) return false;
                        stack[top++] = 0; /* the bottom level's root, untested (:625) */
                        current = sc->bvh + sc->blasOffsets[md.BlasIndex];
                        objectOrigin = MulPoint34(md.WorldToObject, origin);
                        const tb3 objectDirection = MulVector34(md.WorldToObject, direction);
                        rd = GetRayData(objectOrigin, objectDirection); /* :632-634 */
                        nodesToProcess[BOTTOM] = 1;
                    }
                } else { /* :641-709 */
                    const uint32_t offPrims = ld32(current + 4), offMeta = ld32(current + 8);
                    const uint32_t leafIndex = flagsX & ~(TB_BVH_LEAF_FLAG | TB_BVH_PROCEDURAL_FLAG);
                    const uint8_t* m = current + offMeta + 12u * leafIndex;
                    const uint32_t geomContribution = ld32(m), primIdx = ld32(m + 4);
                    trianglesTested++;
                    const uint8_t* p = current + offPrims + 40u * (flagsX & TB_BVH_INDEX_MASK) + 4;
                    const tb3 v0 = tb3_make(ldf(p), ldf(p + 4), ldf(p + 8)), v1 = tb3_make(ldf(p + 12), ldf(p + 16), ldf(p + 20)), v2 = tb3_make(ldf(p + 24),
                        ldf(p + 28), ldf(p + 32));
                    float t0 = hit.t, b[2] = {0, 0};
                    RayTriangleIntersect(t0, b, objectOrigin, rd, v0, v1, v2); /* ObjectRayOrigin / ObjectRayDirection, :667-668 */
                    bool valid = true; /* the any-hit filter sees CandidateInstanceIndex() + CandidateGeometryIndex(), RayGenCommon.h:427 */
                    if (g_alphaTest && t0 < hit.t && t0 > TMin && !(ld32(m + 8) & 1u)) valid = IsValidHit(sc, instanceOffset + geomContribution, primIdx, b[0],
                        b[1]);
                    if (valid && t0 < hit.t && t0 > TMin) {
                        hit.t = t0; hit.bary[0] = b[0]; hit.bary[1] = b[1];
                        hit.primitiveIndex = primIdx; hit.geometryIndex = instanceOffset + geomContribution;
                    }
                }
            } else { /* :717-766 */
                const uint32_t l = flagsX & TB_BVH_INDEX_MASK, r = flagsY;
                const uint8_t* ln = current + offBoxes + 32u * l;
                const uint8_t* rn = current + offBoxes + 32u * r;
                float lt, rt;
                const bool lh = RayBoxTest(lt, hit.t, rd, to3((const float*)ln), to3((const float*)(ln + 16)));
                const bool rh = RayBoxTest(rt, hit.t, rd, to3((const float*)rn), to3((const float*)(rn + 16)));
                boxesTested += 2;
                if (top + 2 > ORACLE_STACK) return false;
                if (lh && rh) { const bool rightFirst = rt < lt; stack[top++] = rightFirst ? l : r; stack[top++] = rightFirst ? r : l;
                    nodesToProcess[processingBottom ? BOTTOM : TOP] += 2; }
                else if (lh || rh) { stack[top++] = rh ? r : l; nodesToProcess[processingBottom ? BOTTOM : TOP] += 1; }
            }
        } while (nodesToProcess[processingBottom ? BOTTOM : TOP] != 0);
        processingBottom = false; /* :769-773 */
        rd = GetRayData(origin, direction);
        objectOrigin = origin;
        current = tlas;
    }
    return hit.t < TMax;
}

/* ------------------------------------------------------------------------------------------
 * Hit attributes: SharedHitGroup.h:48-151
 * ------------------------------------------------------------------------------------------ */
struct HitInfo { float uvx, uvy; tb3 normal, tangent; };

inline float vbf(const TbSceneView* sc, uint32_t i) { return i < sc->numVertexFloats ? sc->vertexBuffer[i] : 0.0f; }
inline uint32_t ibu(const TbSceneView* sc, uint32_t i) { return i < sc->numIndices ? sc->indexBuffer[i] : 0u; }

inline void GetHitInfo(const TbSceneView* sc, const TbHitGroupRecord& rec, uint32_t prim, float bx, float by, float bz, HitInfo& info)
{
    const uint32_t vFirst = rec.VertexBufferOffset / 4, iFirst = rec.IndexBufferOffset / 4; /* :48-62 */
    uint32_t i0 = ibu(sc, iFirst + prim * 3), i1 = ibu(sc, iFirst + prim * 3 + 1), i2 = ibu(sc, iFirst + prim * 3 + 2); /* :88-95 */
    const uint32_t s = 8; /* VertexStride :11 */
    auto f3 = [&](uint32_t vi, uint32_t off) { uint32_t b = s * vi + vFirst + off; return tb3_make(vbf(sc, b), vbf(sc, b + 1), vbf(sc, b + 2)); };
    /* uv :97-108 (offset 3) */
    {
        uint32_t b0 = s * i0 + vFirst + 3, b1 = s * i1 + vFirst + 3, b2 = s * i2 + vFirst + 3;
        info.uvx = tb_fma(bz, vbf(sc, b2), tb_fma(by, vbf(sc, b1), bx * vbf(sc, b0)));
        info.uvy = tb_fma(bz, vbf(sc, b2 + 1), tb_fma(by, vbf(sc, b1 + 1), bx * vbf(sc, b0 + 1)));
    }
    info.normal = tb3_normalize(tb3_bary(bx, by, bz, f3(i0, 0), f3(i1, 0), f3(i2, 0)));   /* :110-121 */
    info.tangent = tb3_normalize(tb3_bary(bx, by, bz, f3(i0, 5), f3(i1, 5), f3(i2, 5)));  /* :123-133 */
}

/* RayGenCommon.h:365-414,482-487: returns (t, materialIndex) or (-1,-1) */
inline void IntersectWithMaxDistance(Ctx& c, const Ray& ray, float maxT, float& resT, int& resMat, tb3& normal, tb3& tangent, float& uvx, float& uvy)
{
    Committed h;
    uint32_t tris, boxes;
    std::string steps;
    if (g_rayLog) g_raySteps = &steps;
    bool isHit = Traverse(c.scene, ray.origin, ray.direction, MIN_T, maxT, h, tris, boxes);
    if (g_rayLog) {
        g_raySteps = nullptr;
        std::lock_guard<std::mutex> lock(g_rayLogMutex);
        fprintf(g_rayLog, "%u %u %u %c %s\n", c.x, c.y, c.pf->GlobalFrameCount, c.rayKind ? c.rayKind : 'E', steps.empty() ? "-" : steps.c_str());
    }
    c.lastTris = tris; c.lastBoxes = boxes;
    if (c.stats) { c.stats->boxesTested += boxes; c.stats->trianglesTested += tris; c.stats->rays++; }
    normal = tb3_splat(0); tangent = tb3_splat(0); uvx = uvy = 0; /* payload init :386 */
    if (isHit) {
        resT = h.t;
        float bx = 1 - h.bary[0] - h.bary[1], by = h.bary[0], bz = h.bary[1]; /* GetBarycentrics3 SharedHitGroup.h:135-138 */
        uint32_t gi = 0 + h.geometryIndex; /* :392 */
        TbHitGroupRecord rec;
        if (gi < c.scene->numHitGroups) rec = c.scene->hitGroups[gi]; else memset(&rec, 0, sizeof rec);
        HitInfo hi;
        GetHitInfo(c.scene, rec, h.primitiveIndex, bx, by, bz, hi);
        uvx = hi.uvx; uvy = hi.uvy; normal = hi.normal; tangent = hi.tangent;
        resMat = (int)rec.MaterialIndex;
        if (c.stats) c.stats->hitsShaded++;
    } else {
        resT = -1; resMat = -1;
    }
    /* OutputRayStats :414,537-543 */
    if (c.pf->OutputMode == TB_OUTPUT_TYPE_HEATMAP) { c.heatmapWritten = true; }
}

/* ------------------------------------------------------------------------------------------
 * Textures: SharedRaytracing.h:55-137, Tonemap.h:208-211
 * ------------------------------------------------------------------------------------------ */
struct F4 { float x, y, z, w; };
inline F4 f4(float x, float y, float z, float w) { F4 r = {x, y, z, w}; return r; }

/* Bilinear, WRAP addressing, fp32 weights (D3D leaves filter precision to the hardware; the build
 * pins it: texel centres at (i+0.5)/N). */
inline F4 SampleBilinearWrap(const TbFloat4* tex, uint32_t w, uint32_t h, float u, float v)
{
    if (!tex || w == 0 || h == 0) return f4(0, 0, 0, 0);
    float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
    float x0f = tb_floor(fx), y0f = tb_floor(fy);
    float tx = fx - x0f, ty = fy - y0f;
    auto wrap = [](float f, uint32_t n) { float m = f - tb_floor(f / (float)n) * (float)n; int i = (int)m; if (i < 0) i = 0; if (i >= (int)n) i = (int)n - 1;
        return (uint32_t)i; };
    uint32_t x0 = wrap(x0f, w), x1 = wrap(x0f + 1.0f, w), y0 = wrap(y0f, h), y1 = wrap(y0f + 1.0f, h);
    const TbFloat4 &a = tex[y0 * w + x0], &b = tex[y0 * w + x1], &c = tex[y1 * w + x0], &d = tex[y1 * w + x1];
    auto bl = [&](float p, float q, float r, float s) { float top = tb_lerp(p, q, tx), bot = tb_lerp(r, s, tx); return tb_lerp(top, bot, ty); };
    return f4(bl(a.x, b.x, c.x, d.x), bl(a.y, b.y, c.y, d.y), bl(a.z, b.z, c.z, d.z), bl(a.w, b.w, c.w, d.w));
}

inline F4 GetTextureData_NonRecursive(const TbSceneView* sc, const TbTextureData& td, float u, float v)
{
    F4 data = f4(0, 0, 0, 0);
    switch (td.TextureType) {
    case TB_TEXTURE_TYPE_IMAGE:
        if (td.DescriptorHeapIndex < sc->numImages) {
            const TbImageDesc& im = sc->images[td.DescriptorHeapIndex];
            data = SampleBilinearWrap(sc->texelPool + im.texelOffset, im.width, im.height, u, v);
        }
        break;
    case TB_TEXTURE_TYPE_CHECKER: { /* SharedRaytracing.h:92-101 */
        float su = u * td.UScale, sv = v * td.VScale;
        data = f4(td.CheckerColor1.x, td.CheckerColor1.y, td.CheckerColor1.z, 1);
        if ((((int)su + (int)sv) % 2) == 0) data = f4(td.CheckerColor2.x, td.CheckerColor2.y, td.CheckerColor2.z, 1);
        break;
    }
    default: break;
    }
    if (td.TextureFlags & TB_TEXTURE_FLAG_NEEDS_GAMMA) { /* GammaToLinear = pow(c, 2.2) */
        data.x = tb_pow(data.x, 2.2f); data.y = tb_pow(data.y, 2.2f); data.z = tb_pow(data.z, 2.2f);
    }
    return data;
}

inline F4 GetTextureData(const TbSceneView* sc, uint32_t textureIndex, float u, float v)
{
    if (textureIndex == TB_INVALID_TEXTURE) return f4(0, 0, 0, 0);
    if (sc->config.FlipTextureUVs) { u = 0.0f + u * 1.0f; v = 1.0f + v * -1.0f; } /* :71-74 */
    if (textureIndex >= sc->numTextureData) return f4(0, 0, 0, 0);
    const TbTextureData& td = sc->textureData[textureIndex];
    if (td.TextureType == TB_TEXTURE_TYPE_SCALE) { /* :119-137 */
        TbTextureData z; memset(&z, 0, sizeof z);
        const TbTextureData& t1 = td.TextureIndex1 < sc->numTextureData ? sc->textureData[td.TextureIndex1] : z;
        const TbTextureData& t2 = td.TextureIndex2 < sc->numTextureData ? sc->textureData[td.TextureIndex2] : z;
        F4 c1 = GetTextureData_NonRecursive(sc, t1, u, v), c2 = GetTextureData_NonRecursive(sc, t2, u, v);
        return f4(c1.x * td.ScaleColor1.x + c2.x * td.ScaleColor2.x, c1.y * td.ScaleColor1.y + c2.y * td.ScaleColor2.y,
                  c1.z * td.ScaleColor1.z + c2.z * td.ScaleColor2.z, c1.w * 1.0f + c2.w * 1.0f);
    }
    return GetTextureData_NonRecursive(sc, td, u, v);
}

/* RayGenCommon.h:21-44 */
inline tb3 SampleEnvironmentMap(const TbSceneView* sc, tb3 v)
{
    const TbConfigConstants& cc = sc->config;
    tb3 vx = tb3_make(cc.EnvMapTransformVx.x, cc.EnvMapTransformVx.y, cc.EnvMapTransformVx.z);
    tb3 vy = tb3_make(cc.EnvMapTransformVy.x, cc.EnvMapTransformVy.y, cc.EnvMapTransformVy.z);
    tb3 vz = tb3_make(cc.EnvMapTransformVz.x, cc.EnvMapTransformVz.y, cc.EnvMapTransformVz.z);
    v = tb3_make(tb3_dot(v, vx), tb3_dot(v, vy), tb3_dot(v, vz));
    tb3 viewDir = tb3_normalize(v);
    float p = tb_atan2(viewDir.y, viewDir.x);
    p = p > 0 ? p : p + 6.28f; /* 2 * 3.14 */
    float u = p / 6.28f;
    float w = tb_acos(viewDir.z) / 3.14f;
    if (!sc->envMap) return tb3_splat(0.0f); /* black 1x1 when absent, TracerBoy.cpp:1919-1934 */
    F4 s = SampleBilinearWrap(sc->envMap, sc->envWidth, sc->envHeight, u, w);
    return tb3_make(s.x, s.y, s.z) * to3(cc.EnvironmentMapColorScale);
}

/* ------------------------------------------------------------------------------------------
 * Materials: RayGenCommon.h:273-341, kernel.glsl:1224-1246
 * ------------------------------------------------------------------------------------------ */
inline TbMaterial FetchMaterial(const TbSceneView* sc, uint32_t id)
{
    TbMaterial m;
    if (id < sc->numMaterials) m = sc->materials[id]; else memset(&m, 0, sizeof m); /* OOB structured read = 0 */
    return m;
}

inline bool IsValidTexture(uint32_t i) { return i != TB_INVALID_TEXTURE; }

inline TbMaterial GetMaterialInternal(Ctx& c, int MaterialID, float u, float v, bool IsBacksideOfGeometry)
{
    TbMaterial mat = FetchMaterial(c.scene, (uint32_t)MaterialID);
    if (c.stats) c.stats->materialFetches++;
    bool ShouldIgnoreEmissive = IsBacksideOfGeometry;
    if (ShouldIgnoreEmissive) { mat.emissive.x = mat.emissive.y = mat.emissive.z = 0; }
    if ((mat.Flags & TB_MAT_MIX) != 0) { /* :308-318, 1 R */
        if (rnd(c) < mat.albedo.z) return FetchMaterial(c.scene, (uint32_t)mat.albedo.x);
        else return FetchMaterial(c.scene, (uint32_t)mat.albedo.y);
    }
    if (IsValidTexture(mat.albedoIndex)) { F4 t = GetTextureData(c.scene, mat.albedoIndex, u, v); mat.albedo.x = t.x; mat.albedo.y = t.y; mat.albedo.z = t.z; }
    if (IsValidTexture(mat.emissiveIndex) && !ShouldIgnoreEmissive) { F4 t = GetTextureData(c.scene, mat.emissiveIndex, u, v); mat.emissive.x = t.x;
        mat.emissive.y = t.y; mat.emissive.z = t.z; }
    if (IsValidTexture(mat.specularMapIndex)) {
        F4 t = GetTextureData(c.scene, mat.specularMapIndex, u, v);
        mat.roughness = t.y;
        if (t.z > 0.5f) mat.Flags |= TB_MAT_METALLIC;
    }
    return mat;
}

/* kernel.glsl:1224-1233 */
inline void ArtistFriendlyAlbdeoToAbsorption(tb3 color, tb3 mfp, tb3& absorption, tb3& scattering)
{
    tb3 e = (-5.09406f * color + 2.61188f * color * color) - 4.31805f * color * color * color;
    tb3 alpha = tb3_splat(1.0f) - tb3_make(tb_exp(e.x), tb_exp(e.y), tb_exp(e.z));
    tb3 cm = color - tb3_splat(0.8f);
    tb3 s = (tb3_splat(1.9f) - color) + 3.5f * cm * cm;
    tb3 transmission = tb3_splat(1.0f) / (s * mfp);
    scattering = transmission * alpha;
    absorption = transmission - scattering;
}

/* kernel.glsl:1235-1246 */
inline TbMaterial GetMaterial(Ctx& c, int MaterialID, float u, float v, bool back)
{
    TbMaterial material = GetMaterialInternal(c, MaterialID, u, v, back);
    bool anyAlbedo = material.albedo.x != 0.0f || material.albedo.y != 0.0f || material.albedo.z != 0.0f;
    if ((material.Flags & TB_MAT_SUBSURFACE_SCATTER) != 0 && anyAlbedo) { /* `any(albedo) > 0.0f` */
        tb3 a, s;
        ArtistFriendlyAlbdeoToAbsorption(to3(material.albedo), tb3_splat(1.0f) / to3(material.scattering), a, s);
        material.absorption.x = a.x; material.absorption.y = a.y; material.absorption.z = a.z;
        material.scattering.x = s.x; material.scattering.y = s.y; material.scattering.z = s.z;
        material.albedo.x = material.albedo.y = material.albedo.z = 0;
    }
    return material;
}

/* RayGenCommon.h:273-295 */
inline tb3 GetDetailNormal(Ctx& c, const TbMaterial& mat, tb3 normal, tb3 tangent, float u, float v)
{
    if (IsValidTexture(mat.normalMapIndex) && c.pf->EnableNormalMaps) {
        tb3 bitangent = tb3_cross(tangent, normal);
        F4 nm = GetTextureData(c.scene, mat.normalMapIndex, u, v);
        float tx = (0.5f - nm.x) * 2.0f, ty = (0.5f - nm.y) * 2.0f;
        float tz = tb_sqrt(1.0f - (tx * tx + ty * ty));
        const float normalYClamp = 0.02f;
        return tb3_normalize(tangent * tx + bitangent * ty + normal * tb_max(tz, normalYClamp));
    }
    return normal;
}

bool IsValidHit(const TbSceneView* sc, uint32_t geometryIndex, uint32_t primitiveIndex, float b0, float b1) /* SharedHitGroup.h:157-179 */
{
    TbHitGroupRecord rec;
    if (geometryIndex < sc->numHitGroups) rec = sc->hitGroups[geometryIndex]; else memset(&rec, 0, sizeof rec);
    HitInfo hit;
    GetHitInfo(sc, rec, primitiveIndex, 1 - b0 - b1, b0, b1, hit);
    TbMaterial mat = FetchMaterial(sc, rec.MaterialIndex); /* GetMaterial_NonRecursive */
    if (IsValidTexture(mat.alphaIndex)) { float alpha = GetTextureData(sc, mat.alphaIndex, hit.uvx, hit.uvy).x; if (alpha < 0.9f) return false; }
    else if (IsValidTexture(mat.albedoIndex)) { float alpha = GetTextureData(sc, mat.albedoIndex, hit.uvx, hit.uvy).w; if (alpha < 0.9f) return false; }
    return true;
}

inline bool AllowsSpecular(const TbMaterial& m) { return (m.Flags & TB_MAT_NO_SPECULAR) == 0; }
inline bool IsMetallic(const TbMaterial& m) { return (m.Flags & TB_MAT_METALLIC) != 0; }
inline bool IsSubsurfaceScattering(const TbMaterial& m) { return (m.Flags & TB_MAT_SUBSURFACE_SCATTER) != 0; }
inline bool IsHairMaterial(const TbMaterial& m) { return (m.Flags & TB_MAT_HAIR) != 0; }
inline bool IsLight(const TbMaterial& m) { return (m.Flags & TB_MAT_LIGHT) != 0; }
inline bool UsePerfectSpecularOptimization(float roughness) { return roughness < 0.05f; }

/* ------------------------------------------------------------------------------------------
 * Light sampling: RayGenCommon.h:124-261
 * ------------------------------------------------------------------------------------------ */
inline TbLight FetchLight(const TbSceneView* sc, uint32_t i)
{
    TbLight l;
    if (i < sc->numLights) l = sc->lights[i]; else memset(&l, 0, sizeof l);
    return l;
}

inline tb3 GetRandomBarycentric(Ctx& c) /* :124-135, 2 R */
{
    float u = rnd(c);
    float v = rnd(c);
    if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
    return tb3_make(u, v, 1.0f - u - v);
}

inline float ColorToLuma(tb3 color) { return tb3_dot(color, tb3_make(0.212671f, 0.715160f, 0.072169f)); } /* Tonemap.h:12-15 */

inline float GetLightTargetPDF(const TbLight& light, tb3 b, tb3 P) /* :164-168 (sic: a / d * d) */
{
    tb3 lp = to3(light.P0) * b.x + to3(light.P1) * b.y + to3(light.P2) * b.z;
    float d = tb3_length(lp - P);
    return (light.SurfaceArea * ColorToLuma(to3(light.LightColor))) / d * d;
}

inline void GetOneLightSample(Ctx& c, tb3 P, tb3& LightDirection, tb3& LightColor, float& PDFValue, tb3& LightNormal, float& LightAttenuation)
{
    LightDirection = LightColor = LightNormal = tb3_splat(0);
    LightAttenuation = 0.0f;
    PDFValue = 0.0f;
    const uint32_t lightCount = c.pf->LightCount;
    if (lightCount > 0 && c.pf->EnableNextEventEstimation) {
        if (c.stats) c.stats->lightSamples++;
        if (c.pf->EnableSamplingImportanceResampling) { /* :180-211 */
            const uint32_t N = 16;
            uint32_t SelectedIndex = 0; tb3 SelectedBarycentric = tb3_splat(0); float WeightSum = 0.0f;
            for (uint32_t i = 0; i < N; i++) {
                uint32_t lightIndex = (uint32_t)(rnd(c) * (float)lightCount);
                TbLight light = FetchLight(c.scene, lightIndex);
                tb3 b = GetRandomBarycentric(c);
                float TargetPDF = GetLightTargetPDF(light, b, P);
                float proposalPDF = 1.0f / (float)lightCount;
                float Weight = TargetPDF / (proposalPDF * (float)N);
                WeightSum += Weight;
                if (rnd(c) < Weight / WeightSum) { SelectedIndex = lightIndex; SelectedBarycentric = b; }
            }
            TbLight light = FetchLight(c.scene, SelectedIndex);
            tb3 b = SelectedBarycentric;
            float sir = GetLightTargetPDF(light, b, P) / WeightSum;
            PDFValue = sir / light.SurfaceArea;
            tb3 lp = to3(light.P0) * b.x + to3(light.P1) * b.y + to3(light.P2) * b.z;
            LightDirection = lp - P;
            LightNormal = to3(light.N0) * b.x + to3(light.N1) * b.y + to3(light.N2) * b.z;
            LightColor = to3(light.LightColor);
        } else { /* :212-259, 3 R */
            uint32_t lightIndex = (uint32_t)(rnd(c) * (float)lightCount);
            TbLight light = FetchLight(c.scene, lightIndex);
            tb3 b = GetRandomBarycentric(c);
            switch (light.LightType) {
            case TB_LIGHT_TYPE_AREA: {
                tb3 lp = to3(light.P0) * b.x + to3(light.P1) * b.y + to3(light.P2) * b.z;
                LightDirection = lp - P;
                LightNormal = to3(light.N0) * b.x + to3(light.N1) * b.y + to3(light.N2) * b.z;
                float d = tb3_length(LightDirection);
                LightAttenuation = 1.0f / (d * d);
                LightDirection = LightDirection / d;
                break;
            }
            case TB_LIGHT_TYPE_DIRECTIONAL: {
                LightDirection = -to3(light.Direction);
                if (c.pf->DebugValue > 0.0f) { /* :236-241 (DebugValue defaults to 1, TracerBoy.h:301) */
                    LightDirection.x = tb_sin(c.pf->DebugValue);
                    LightDirection.y = tb_sin(c.pf->DebugValue2);
                    LightDirection = tb3_normalize(LightDirection);
                }
                LightNormal = -LightDirection;
                LightAttenuation = 1.0f;
                break;
            }
            default: break;
            }
            LightColor = to3(light.LightColor);
            PDFValue = 1.0f / (float)lightCount;
            if (light.LightType == TB_LIGHT_TYPE_AREA) PDFValue /= light.SurfaceArea;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * BSDF helpers: kernel.glsl:466-478, 541-546, 991-1099, 1186-1269
 * ------------------------------------------------------------------------------------------ */
inline float GGXNormalDistributionFunction(tb3 Normal, tb3 HalfVector, float RoughnessSquared)
{
    RoughnessSquared = tb_max(RoughnessSquared, MIN_ROUGHNESS_SQUARED);
    float a2 = RoughnessSquared * RoughnessSquared;
    float nDotH = tb3_dot(Normal, HalfVector);
    float Denominator = PI * tb_pow(nDotH * nDotH * (a2 - 1.0f) + 1.0f, 2.0f);
    return a2 / Denominator;
}

inline float DiffuseBRDF(tb3 L, tb3 N) { return tb_max(tb3_dot(L, N), 0.0f) / PI; }

inline tb3 GenerateRandomDirection(Ctx& c) /* :991-999 */
{
    float u1 = rnd(c); float u2 = rnd(c);
    float r = tb_sqrt(1.0f - u1 * u1);
    float phi = 6.28f * u2; /* 2.0 * 3.14 */
    return tb3_make(tb_cos(phi) * r, tb_sin(phi) * r, u1);
}

inline tb3 ReorientVectorAroundNormal(tb3 v, tb3 normal) /* :1001-1015 */
{
    tb3 tangent;
    if (tb_abs(normal.x) > tb_abs(normal.y)) tangent = tb3_make(-normal.z, 0, normal.x) / tb_sqrt(normal.x * normal.x + normal.z * normal.z);
    else tangent = tb3_make(0, normal.z, -normal.y) / tb_sqrt(normal.y * normal.y + normal.z * normal.z);
    tb3 bitangent = tb3_cross(normal, tangent);
    return tb3_normalize(v.x * tangent + v.y * normal + v.z * bitangent);
}

inline tb3 GenerateCosineWeightedDirection(tb3 normal, float rand0, float rand1, float& pdfValue) /* :1025-1041 */
{
    float r = tb_sqrt(rand0);
    float theta = (2.0f * PI) * rand1;
    float x = r * tb_cos(theta);
    float y = tb_sqrt(tb_max(EPSILON, 1.0f - rand0));
    float z = r * tb_sin(theta);
    pdfValue = y / PI;
    return ReorientVectorAroundNormal(tb3_make(x, y, z), normal);
}

inline tb3 GenerateImportanceSampledDirection(tb3 normal, float roughness, float rand0, float rand1, float& PDFValue) /* :1048-1064 */
{
    float lobeMultiplier = tb_pow(1.0f - roughness, 5.0f) * 1000.0f;
    float u1 = rand0, u2 = rand1;
    float theta = (2.0f * PI) * u2;
    float phi = tb_acos(tb_sqrt(tb_pow(u1, 1.0f / (lobeMultiplier + 1.0f))));
    tb3 direction = tb3_make(tb_sin(phi) * tb_cos(theta), tb_cos(phi), tb_sin(phi) * tb_sin(theta));
    PDFValue = (lobeMultiplier + 1.0f) * tb_pow(tb_cos(phi), lobeMultiplier) / (2.0f * PI);
    return ReorientVectorAroundNormal(direction, normal);
}

inline tb3 GenerateRandomImportanceSampledDirection(Ctx& c, tb3 normal, float roughness, float& PDFValue) /* :1096-1099 */
{
    float r0 = rnd(c); float r1 = rnd(c);
    return GenerateImportanceSampledDirection(normal, roughness, r0, r1, PDFValue);
}

inline tb3 ImportanceSampleGGX(Ctx& c, tb3 incomingRay, tb3 normal, float roughness) /* :1066-1082, 2 R */
{
    roughness = tb_max(MIN_ROUGHNESS, roughness);
    float a = roughness * roughness;
    float a2 = a * a;
    float u1 = rnd(c); float u2 = rnd(c);
    float theta = (2.0f * PI) * u2;
    float phi = tb_acos(tb_sqrt((1.0f - u1) / ((a2 - 1.0f) * u1 + 1.0f)));
    tb3 direction = tb3_make(tb_sin(phi) * tb_cos(theta), tb_cos(phi), tb_sin(phi) * tb_sin(theta));
    tb3 GGXSampledNormal = ReorientVectorAroundNormal(direction, normal);
    return tb3_reflect(incomingRay, GGXSampledNormal);
}

inline float ImportanceSampleGGXPDF(tb3 normal, tb3 outgoingRay, tb3 halfVector, float roughness) /* :1084-1094 */
{
    roughness = tb_max(MIN_ROUGHNESS, roughness);
    float a = roughness * roughness;
    float a2 = a * a;
    float cosTheta = tb_abs(tb3_dot(normal, halfVector));
    float e = (a2 - 1.0f) * cosTheta * cosTheta + 1.0f;
    if (e <= 0.0f) return LARGE_NUMBER;
    float d = a2 / (PI * e * e);
    return d * tb_abs(tb3_dot(halfVector, normal)) / (4.0f * tb_abs(tb3_dot(outgoingRay, halfVector)));
}

inline tb3 GetHalfVectorSafe(tb3 a, tb3 b, tb3 normal) /* :1258-1269 */
{
    float aDotB = tb3_dot(a, b);
    if (aDotB > (-1.0f + EPSILON)) return tb3_normalize(a + b);
    else return normal;
}

inline tb3 GetRayPoint(const Ray& r, float t) { return tb3_madd(r.direction, t, r.origin); }

/* ------------------------------------------------------------------------------------------
 * Trace: kernel.glsl:1278-1776
 * ------------------------------------------------------------------------------------------ */
struct BlueNoiseData { float PrimaryJitter[2], SecondaryRayDirection[2], AreaLightJitter[2], DOFJitter[2]; };

/* RayGenCommon.h:49-69 */
inline float Halton(int b, int i)
{
    float r = 0.0f, f = 1.0f;
    while (i > 0) {
        f = f / (float)b;
        r = r + f * (float)(i % b);
        i = (int)tb_floor((float)i / (float)b);
    }
    return r;
}

inline BlueNoiseData GetBlueNoise(Ctx& c) /* RayGenCommon.h:104-122 */
{
    BlueNoiseData d;
    if (!c.pf->UseBlueNoise) {
        d.PrimaryJitter[0] = rnd(c); d.PrimaryJitter[1] = rnd(c);
        d.SecondaryRayDirection[0] = rnd(c); d.SecondaryRayDirection[1] = rnd(c);
        d.AreaLightJitter[0] = rnd(c); d.AreaLightJitter[1] = rnd(c);
        d.DOFJitter[0] = rnd(c); d.DOFJitter[1] = rnd(c);
    } else {
        TbFloat4 z = {0, 0, 0, 0};
        uint32_t idx = (c.y % 256u) * 256u + (c.x % 256u);
        TbFloat4 n0 = c.scene->blueNoise0 ? c.scene->blueNoise0[idx] : z;
        TbFloat4 n1 = c.scene->blueNoise1 ? c.scene->blueNoise1[idx] : z;
        float h2 = Halton(2, (int)c.pf->GlobalFrameCount), h3 = Halton(3, (int)c.pf->GlobalFrameCount);
        d.PrimaryJitter[0] = tb_frac(n0.x + h2); d.PrimaryJitter[1] = tb_frac(n0.y + h3);
        d.SecondaryRayDirection[0] = tb_frac(n0.z + h2); d.SecondaryRayDirection[1] = tb_frac(n0.w + h3);
        d.AreaLightJitter[0] = tb_frac(n1.x + h2); d.AreaLightJitter[1] = tb_frac(n1.y + h3);
        d.DOFJitter[0] = tb_frac(n1.z + h2); d.DOFJitter[1] = tb_frac(n1.w + h3);
    }
    return d;
}

tb3 Trace(Ctx& c, Ray ray, Ray neighborRay)
{
    const TbPerFrameConstants& pf = *c.pf;
    tb3 accumulatedColor = tb3_splat(0.0f);
    tb3 accumulatedIndirectLightMultiplier = tb3_splat(1.0f);
    BlueNoiseData BlueNoise = GetBlueNoise(c); /* :1283, 8 R, values unused */
    (void)BlueNoise;
    bool bPrevRayWasPerfectlySpecular = false;

    for (int i = 0; i < (int)pf.MaxBounces; i++) {
        if (i >= 2) { /* Russian roulette :1288-1302 */
            float p = tb_max(tb_max(accumulatedIndirectLightMultiplier.x, accumulatedIndirectLightMultiplier.y), accumulatedIndirectLightMultiplier.z);
            p = tb_max(p, EPSILON);
            if (p < rnd(c)) break;
            else accumulatedIndirectLightMultiplier = accumulatedIndirectLightMultiplier * (1.0f / p);
        }
        bool bFirstRay = (i == 0);
        tb3 normal, tangent; float uvx, uvy;
        float resT; int resMat;
        c.rayKind = 'E';
        IntersectWithMaxDistance(c, ray, 999999.0f, resT, resMat, normal, tangent, uvx, uvy); /* :1312 */

        /* :1319-1326 */
        if (accumulatedIndirectLightMultiplier.x < EPSILON && accumulatedIndirectLightMultiplier.y < EPSILON && accumulatedIndirectLightMultiplier.z < EPSILON)
            break;

        if (resMat == INVALID_MATERIAL_ID) { /* :1328-1343 */
            accumulatedColor = accumulatedColor + accumulatedIndirectLightMultiplier * SampleEnvironmentMap(c.scene, ray.direction);
            if (bFirstRay) { c.aovEmissive = accumulatedColor; c.aovEmissiveWritten = true; }
            break;
        }
        tb3 RayPoint = GetRayPoint(ray, resT);
        ray.origin = RayPoint + normal * EPSILON; /* :1353 (unflipped normal) */
        float RayDirectionDotN = tb3_dot(normal, ray.direction);
        bool IsBacksideOfGeometry = RayDirectionDotN > 0.0f;
        TbMaterial material = GetMaterial(c, resMat, uvx, uvy, IsBacksideOfGeometry);
        tb3 detailNormal = GetDetailNormal(c, material, normal, tangent, uvx, uvy);
        if (i == 0) { /* :1365-1376 */
            tb3 NeighborRayPoint = GetRayPoint(neighborRay, resT);
            c.aovWorldPos = c.aovWorldPos + RayPoint;
            c.aovDistanceToNeighbor += tb3_length(NeighborRayPoint - RayPoint);
            c.aovNormal = detailNormal;
            c.aovDepth = tb_saturate(resT / pf.MaxZ); c.aovDepthWritten = true;
            /* IsSelectedPixel :593-596; OutputMaterial(int(result.y)) */
            if (c.x == pf.SelectedPixelX && c.y == pf.SelectedPixelY) { c.selWritten = true; c.selDistance = resT; c.selMaterial = resMat; }
            if (pf.OutputMode == TB_OUTPUT_TYPE_HEATMAP) break;
        }
        bool IsInsidePrimitve = IsBacksideOfGeometry;
        float CurrentIOR = IsInsidePrimitve ? material.IOR : AIR_IOR;
        float NewIOR = IsInsidePrimitve ? AIR_IOR : material.IOR;
        if (IsInsidePrimitve) { normal = -normal; RayDirectionDotN = -RayDirectionDotN; detailNormal = -detailNormal; }

        float ReflectionCoefficient = material.SpecularCoef;
        bool bSpecularRay = false;
        if (AllowsSpecular(material)) { /* :1400-1417 */
            if (IsMetallic(material) || IsHairMaterial(material)) bSpecularRay = true;
            else bSpecularRay = rnd(c) < 0.5f;
        }
        bool bUsePerfectSpecularOptimization = bSpecularRay && UsePerfectSpecularOptimization(material.roughness);
        if (bPrevRayWasPerfectlySpecular || bFirstRay || !IsLight(material) || !pf.EnableNextEventEstimation)
            accumulatedColor = accumulatedColor + accumulatedIndirectLightMultiplier * to3(material.emissive); /* :1425-1428 */
        if (IsLight(material)) break; /* :1430-1433 */

        float lightPDF, lightAttenuation; tb3 lightDirection, lightColor, lightNormal;
        GetOneLightSample(c, RayPoint, lightDirection, lightColor, lightPDF, lightNormal, lightAttenuation); /* :1437 */

        if (!bUsePerfectSpecularOptimization) { /* :1440-1517 */
            if (lightPDF > EPSILON && tb3_dot(lightDirection, lightNormal) < 0.0f) {
                tb3 ShadowMultiplier = tb3_splat(1.0f);
                Ray shadowFeeler; shadowFeeler.origin = RayPoint + normal * EPSILON; shadowFeeler.direction = lightDirection;
                tb3 sN, sT; float su, sv; float sResT; int sMat;
                c.rayKind = 'S';
                IntersectWithMaxDistance(c, shadowFeeler, 999999.0f, sResT, sMat, sN, sT, su, sv); /* :1455 */
                if (sMat != INVALID_MATERIAL_ID) {
                    float LightDirectionDotN = tb3_dot(sN, lightDirection);
                    bool sBack = LightDirectionDotN > 0.0f;
                    TbMaterial sMaterial = GetMaterial(c, sMat, su, sv, sBack); /* :1472 */
                    if (!IsLight(sMaterial)) ShadowMultiplier = tb3_splat(0.0f); /* :1474-1511 */
                }
                float lightMultiplier = lightAttenuation * DiffuseBRDF(lightDirection, detailNormal) * tb_abs(tb3_dot(lightNormal, lightDirection)) / lightPDF;
                /* :1514-1515 */
                accumulatedColor = accumulatedColor + accumulatedIndirectLightMultiplier * to3(material.albedo) * lightMultiplier * ShadowMultiplier *
                    lightColor;
            }
        }

        tb3 previousDirection = ray.direction;
        bPrevRayWasPerfectlySpecular = bUsePerfectSpecularOptimization;
        if (bSpecularRay) {
            ray.direction = ImportanceSampleGGX(c, ray.direction, normal, material.roughness); /* :1521-1526 */
        } else {
            if (IsSubsurfaceScattering(material)) { /* :1529-1691 */
                float nr = CurrentIOR / NewIOR;
                float discriminant = 1.0f - nr * nr * (1.0f - RayDirectionDotN * RayDirectionDotN);
                if (discriminant > EPSILON) {
                    tb3 refractionDirection = tb3_normalize(nr * (ray.direction - normal * RayDirectionDotN) - normal * tb_sqrt(discriminant));
                    if (bUsePerfectSpecularOptimization) { ray.direction = refractionDirection; bPrevRayWasPerfectlySpecular = true; }
                    else {
                        float PDFValue;
                        ray.direction = GenerateRandomImportanceSampledDirection(c, refractionDirection, material.roughness, PDFValue);
                        if (PDFValue < EPSILON) {
                            ray.direction = GenerateRandomImportanceSampledDirection(c, refractionDirection, material.roughness, PDFValue);
                            if (PDFValue < EPSILON) break; /* :1552 */
                        }
                    }
                } else {
                    ray.direction = tb3_reflect(ray.direction, normal);
                }
                const int MAX_SSS_BOUNCES = 100;
                bool noScatter = material.scattering.x < EPSILON; /* float3 -> bool truncation :1567 */
                float DistancePerScatter = 1.0f / ((material.scattering.x + material.scattering.y + material.scattering.z) / 3.0f);
                float maxTravelDistance = noScatter ? LARGE_NUMBER : DistancePerScatter;
                bool exittingPrimitive = (material.Flags & TB_MAT_SINGLE_SIDED) != 0;
                bool brokeOut = false;
                for (int j = 0; j < MAX_SSS_BOUNCES && !exittingPrimitive; j++) {
                    float travelDistance = tb_max(-tb_log(rnd(c)), 0.1f) * maxTravelDistance;
                    c.rayKind = 'W';
                    IntersectWithMaxDistance(c, ray, 999999.0f, resT, resMat, normal, tangent, uvx, uvy); /* :1607 */
                    bool bHitFound = resMat != INVALID_MATERIAL_ID;
                    if (!bHitFound) { accumulatedIndirectLightMultiplier = tb3_splat(0.0f); break; }
                    resT = tb_min(travelDistance, resT);
                    float distanceTravelledBeforeScatter = resT;
                    exittingPrimitive = resT < travelDistance || noScatter;
                    bool lastRay = (j == MAX_SSS_BOUNCES - 1);
                    if (lastRay && !exittingPrimitive) accumulatedIndirectLightMultiplier = tb3_splat(0.0f);
                    RayPoint = GetRayPoint(ray, resT);
                    ray.origin = RayPoint + normal * EPSILON;
                    tb3 ab = to3(material.absorption);
                    tb3 beerLambert = tb3_make(tb_exp(-distanceTravelledBeforeScatter * ab.x), tb_exp(-distanceTravelledBeforeScatter * ab.y),
                        tb_exp(-distanceTravelledBeforeScatter * ab.z));
                    accumulatedIndirectLightMultiplier = accumulatedIndirectLightMultiplier * beerLambert;
                    if (exittingPrimitive) {
                        RayDirectionDotN = tb3_dot(normal, ray.direction);
                        if (RayDirectionDotN >= 0.0f) { normal = -normal; RayDirectionDotN = -RayDirectionDotN; }
                        float nr2 = NewIOR / CurrentIOR;
                        float disc2 = 1.0f - nr2 * nr2 * (1.0f - RayDirectionDotN * RayDirectionDotN);
                        if (disc2 > EPSILON) {
                            tb3 refractionDirection = tb3_normalize(nr2 * (ray.direction - normal * RayDirectionDotN) - normal * tb_sqrt(disc2));
                            if (bUsePerfectSpecularOptimization) { ray.direction = refractionDirection; bPrevRayWasPerfectlySpecular = true; }
                            else {
                                float PDFValue;
                                ray.direction = GenerateRandomImportanceSampledDirection(c, refractionDirection, material.roughness, PDFValue);
                                if (PDFValue < EPSILON) {
                                    ray.direction = GenerateRandomImportanceSampledDirection(c, refractionDirection, material.roughness, PDFValue);
                                    if (PDFValue < EPSILON) { brokeOut = true; break; } /* :1666 leaves the walk only */
                                }
                            }
                        } else {
                            ray.direction = tb3_reflect(ray.direction, normal);
                            exittingPrimitive = false;
                        }
                        previousDirection = ray.direction;
                    } else {
                        float pdfValue = 1.0f; /* isotropic branch, kernel.glsl:1205-1209 */
                        ray.direction = GenerateRandomDirection(c);
                        accumulatedIndirectLightMultiplier = accumulatedIndirectLightMultiplier / pdfValue;
                    }
                }
                (void)brokeOut;
                continue; /* :1690 */
            } else {
                float PDFValue;
                float r0 = rnd(c); float r1 = rnd(c);
                ray.direction = GenerateCosineWeightedDirection(normal, r0, r1, PDFValue); /* :1695 */
            }
        }

        float DiffusePDF = tb3_dot(ray.direction, normal) / PI; /* :1699 */
        if (AllowsSpecular(material)) {
            tb3 halfVector = GetHalfVectorSafe(-previousDirection, ray.direction, normal);
            float DistributionPDF = ImportanceSampleGGXPDF(normal, ray.direction, halfVector, material.roughness);
            float SpecularPDF = DistributionPDF;
            float PDFLerpValue = 0.5f;
            float PDFValue = IsMetallic(material) ? SpecularPDF : tb_lerp(SpecularPDF, DiffusePDF, PDFLerpValue);
            accumulatedIndirectLightMultiplier = accumulatedIndirectLightMultiplier / PDFValue;
        } else {
            accumulatedIndirectLightMultiplier = accumulatedIndirectLightMultiplier / DiffusePDF;
        }

        if (bFirstRay) { c.aovEmissive = to3(material.emissive); c.aovEmissiveWritten = true; } /* :1720-1723 */
        /* :1725 IsLight -> break is dead: already left at :1430 */
        bool bRemoveAlbedo = pf.IsRealTime && bFirstRay;
        tb3 albedo = bRemoveAlbedo ? tb3_splat(1.0f) : to3(material.albedo);
        if (IsMetallic(material)) { /* :1734-1741 */
            tb3 halfVector = tb3_normalize(-previousDirection + ray.direction);
            float roughnessSquared = tb_max(material.roughness * material.roughness, MIN_ROUGHNESS_SQUARED);
            float specular = GGXNormalDistributionFunction(detailNormal, halfVector, roughnessSquared) /
                (4.0f * tb_abs(tb3_dot(-previousDirection, halfVector)) * tb_max(tb_abs(tb3_dot(-previousDirection, normal)), tb_abs(tb3_dot(ray.direction,
                    normal))));
            accumulatedIndirectLightMultiplier = accumulatedIndirectLightMultiplier * (specular * albedo * tb_saturate(tb3_dot(ray.direction, normal)));
        } else if (AllowsSpecular(material)) { /* :1744-1765 */
            tb3 halfVector = GetHalfVectorSafe(-previousDirection, ray.direction, normal);
            float fresnel = ReflectionCoefficient + (1.0f - ReflectionCoefficient) * tb_pow(tb_abs(1.0f - tb3_dot(-previousDirection, halfVector)), 5.0f);
            float diffuseMultiplier = (float)(28.0 / (23.0 * 3.1415926535))
                * (1.0f - ReflectionCoefficient)
                * (1.0f - tb_pow(1.0f - 0.5f * tb3_dot(-previousDirection, normal), 5.0f))
                * (1.0f - tb_pow(1.0f - 0.5f * tb3_dot(ray.direction, normal), 5.0f));
            tb3 diffuse = albedo * diffuseMultiplier;
            float roughnessSquared = tb_max(material.roughness * material.roughness, MIN_ROUGHNESS_SQUARED);
            float specular = GGXNormalDistributionFunction(detailNormal, halfVector, roughnessSquared) /
                (4.0f * tb_abs(tb3_dot(-previousDirection, halfVector)) * tb_max(tb_abs(tb3_dot(-previousDirection, normal)), tb_abs(tb3_dot(ray.direction,
                    normal))));
            tb3 IndirectLightMultiplier = (diffuse + tb3_splat(fresnel * specular)) * tb_saturate(tb3_dot(ray.direction, normal));
            accumulatedIndirectLightMultiplier = accumulatedIndirectLightMultiplier * IndirectLightMultiplier;
        } else { /* :1766-1769 */
            accumulatedIndirectLightMultiplier = accumulatedIndirectLightMultiplier * (albedo * DiffuseBRDF(ray.direction, detailNormal));
        }
        if (bFirstRay) c.aovAlbedo = to3(material.albedo); /* :1771 */
    }
    return accumulatedColor;
}

/* kernel.glsl:1800-1803 */
inline float Gaussian(float x, float mu, float sigma)
{
    float d = x - mu;
    return 1.0f / tb_sqrt(2.0f * PI * sigma * sigma) * tb_exp(-tb_pow(d, 2.0f) / (2.0f * sigma * sigma));
}

/* kernel.glsl:1786-1798 with GetRotationFactor() == 0.5 => view matrix is exactly the identity
 * (RayGenCommon.h:17, kernel.glsl:1778-1784: xRotation = 0, x*1 + y*0 + z*(-0) == x). */
inline tb3 GetLensPosition(const TbPerFrameConstants& pf, float lensHeight, float u, float v, float aspectRatio)
{
    tb3 lensPoint = to3(pf.CameraPosition);
    float lensWidth = lensHeight * aspectRatio;
    lensPoint = lensPoint + to3(pf.CameraRight) * (u * 2.0f - 1.0f) * lensWidth / 2.0f;
    lensPoint = lensPoint + to3(pf.CameraUp) * (v * 2.0f - 1.0f) * lensHeight / 2.0f;
    return lensPoint;
}

struct CameraRays { Ray camera, neighbor; float filterWeight; };

/* kernel.glsl:1805-1901 given the primary / DOF jitter */
inline CameraRays MakeCameraRays(const TbPerFrameConstants& pf, float lensHeight, uint32_t W, uint32_t H, float pixelCoordX, float pixelCoordY,
                                 float jx, float jy, float dofx, float dofy)
{
    CameraRays out;
    float resX = (float)W, resY = (float)H;
    float pixelUVSizeX = 1.0f / resX, pixelUVSizeY = 1.0f / resY;
    float u = pixelCoordX * pixelUVSizeX, v = pixelCoordY * pixelUVSizeY;
    if (pf.FixedPixelOffset.x >= 0.0f) { jx = pf.FixedPixelOffset.x; jy = pf.FixedPixelOffset.y; }
    float offX = jx - 0.5f, offY = jy - 0.5f;
    float pixelRadius = pf.FilterWidth / 2.0f;
    float filterWeight = 1.0f;
    switch (pf.FilterType) {
    case TB_FILTER_TYPE_TRIANGLE: filterWeight = tb_max(0.5f - tb_abs(offX), 0.5f - tb_abs(offY)); break;
    case TB_FILTER_TYPE_GAUSSIAN: {
        float sigma = 0.8f;
        float expX = Gaussian(1.0f, 0.0f, sigma), expY = Gaussian(1.0f, 0.0f, sigma);
        filterWeight = tb_max(0.0f, Gaussian(offX * 2.0f, 0.0f, sigma) - expX) * tb_max(0.0f, Gaussian(offY * 2.0f, 0.0f, sigma) - expY);
        break;
    }
    default: filterWeight = 1.0f; break;
    }
    u += offX * pixelUVSizeX * (pixelRadius * 2.0f);
    v += offY * pixelUVSizeY * (pixelRadius * 2.0f);
    float aspectRatio = resX / resY;
    tb3 camPos = to3(pf.CameraPosition);
    tb3 focalPoint = camPos - pf.FocalDistance * tb3_normalize(to3(pf.CameraLookAt) - camPos);
    tb3 lensPoint = GetLensPosition(pf, lensHeight, u, v, aspectRatio);
    tb3 neighborLensPoint = GetLensPosition(pf, lensHeight, u + pixelUVSizeX, v + pixelUVSizeY, aspectRatio);
    out.camera.origin = focalPoint; out.camera.direction = tb3_normalize(lensPoint - focalPoint);
    out.neighbor.origin = focalPoint; out.neighbor.direction = tb3_normalize(neighborLensPoint - focalPoint);
    if (pf.DOFFocusDistance > 0.0f) { /* :1890-1901 */
        tb3 FocusPoint = GetRayPoint(out.camera, pf.DOFFocusDistance);
        float Radius = tb_sqrt(dofx) * pf.DOFApertureWidth;
        float Theta = dofy * 2.0f * PI;
        float fjx = tb_cos(Theta) * Radius, fjy = tb_sin(Theta) * Radius;
        out.camera.origin = out.camera.origin + (fjx * to3(pf.CameraRight) + fjy * to3(pf.CameraUp));
        out.camera.direction = tb3_normalize(FocusPoint - out.camera.origin);
    }
    out.filterWeight = filterWeight;
    return out;
}

/* SoftwareRayTraceCS.hlsl:38-50 + RayGenCommon.h:690-709 for one pixel; returns (rgb*w, w) */
void sample_pixel(Ctx& c, float out[4])
{
    const TbPerFrameConstants& pf = *c.pf;
    c.aovNormal = c.aovAlbedo = c.aovEmissive = c.aovWorldPos = tb3_splat(0); /* ClearAOVs :650-654, :693-694 */
    c.aovDistanceToNeighbor = 0; c.aovDepth = 0; c.aovDepthWritten = c.aovEmissiveWritten = false; c.heatmapWritten = false;
    c.lastTris = c.lastBoxes = 0;
    c.seed = hash13(tb3_make((float)c.x, (float)c.y, (float)pf.GlobalFrameCount));
    float dux = ((float)c.x + 0.5f) / (float)c.width, duy = ((float)c.y + 0.5f) / (float)c.height; /* :696 */
    float uvx = 0.0f + dux * 1.0f, uvy = 1.0f + duy * -1.0f;                                         /* :697 */
    float pcx = uvx * (float)c.width, pcy = uvy * (float)c.height;                                    /* :703 */
    BlueNoiseData bn = GetBlueNoise(c); /* kernel.glsl:1830 */
    CameraRays cr = MakeCameraRays(pf, c.scene->config.CameraLensHeight, c.width, c.height, pcx, pcy, bn.PrimaryJitter[0], bn.PrimaryJitter[1],
        bn.DOFJitter[0], bn.DOFJitter[1]);
    tb3 color = Trace(c, cr.camera, cr.neighbor);
    if (pf.FireflyClampValue >= EPSILON) color = tb3_min(color, tb3_splat(pf.FireflyClampValue)); /* :1910-1913 */
    out[0] = color.x * cr.filterWeight; out[1] = color.y * cr.filterWeight; out[2] = color.z * cr.filterWeight; out[3] = cr.filterWeight;
    if (c.stats) c.stats->samples++;
}

void render_rows(const TbSceneView* scene, const TbPerFrameConstants* constants, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
                 uint32_t firstFrame, uint32_t numFrames, TbFloat4* output, TbFloat4* jittered, const TboAovs* aovs, TbRayStats* stats)
{
    for (uint32_t f = 0; f < numFrames; f++) {
        TbPerFrameConstants pf = *constants;
        pf.GlobalFrameCount = firstFrame + f;
        for (uint32_t y = y0; y < y1; y++) for (uint32_t x = 0; x < W; x++) {
            Ctx c; memset(&c, 0, sizeof c);
            c.scene = scene; c.pf = &pf; c.width = W; c.height = H; c.x = x; c.y = y; c.stats = stats;
            float s[4];
            sample_pixel(c, s);
            bool ok = !(tb_isnan(s[0]) || tb_isnan(s[1]) || tb_isnan(s[2]) || tb_isnan(s[3])); /* RayGenCommon.h:704-707 */
            float o[4] = {0, 0, 0, 0};
            if (ok) { o[0] += s[0]; o[1] += s[1]; o[2] += s[2]; o[3] += s[3]; }
            size_t p = (size_t)y * W + x;
            if (aovs) {
                TbFloat4 wp = {c.aovWorldPos.x, c.aovWorldPos.y, c.aovWorldPos.z, c.aovDistanceToNeighbor};
                if ((pf.GlobalFrameCount % 2) == 0) { if (aovs->worldPosition0) aovs->worldPosition0[p] = wp; }
                else { if (aovs->worldPosition1) aovs->worldPosition1[p] = wp; }
                if (aovs->normals) { TbFloat4 n = {c.aovNormal.x, c.aovNormal.y, c.aovNormal.z, 1.0f}; aovs->normals[p] = n; }
                if (aovs->customOutput) {
                    TbFloat4 a = {c.aovAlbedo.x, c.aovAlbedo.y, c.aovAlbedo.z, 1.0f};
                    if (c.heatmapWritten) { a.x = (float)c.lastTris; a.y = (float)c.lastBoxes; a.z = 0; a.w = 0; }
                    aovs->customOutput[p] = a;
                }
                if (aovs->depth && c.aovDepthWritten) aovs->depth[p] = c.aovDepth;
                if (aovs->emissive && c.aovEmissiveWritten) { TbFloat4 e = {c.aovEmissive.x, c.aovEmissive.y, c.aovEmissive.z, 1.0f}; aovs->emissive[p] = e; }
            }
            /* accumulate: RayGenCommon.h:721-727 */
            TbFloat4 acc = (pf.IsRealTime || pf.GlobalFrameCount == 0) ? TbFloat4{0, 0, 0, 0} : output[p];
            TbFloat4 r = {o[0] + acc.x, o[1] + acc.y, o[2] + acc.z, o[3] + acc.w};
            output[p] = r;
            float coin = rnd(c); /* HLSL `||` evaluates both sides; it is the last R either way */
            if (!pf.IsRealTime && (pf.GlobalFrameCount == 0 || coin < 0.5f)) {
                if (jittered) {
                    TbFloat4 ja = pf.GlobalFrameCount > 0 ? jittered[p] : TbFloat4{0, 0, 0, 0};
                    TbFloat4 jr = {o[0] + ja.x, o[1] + ja.y, o[2] + ja.z, o[3] + ja.w};
                    jittered[p] = jr;
                }
            }
        }
    }
}

} // namespace

extern "C" {

int tbo_render(const TbSceneView* scene, const TbPerFrameConstants* constants, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
               uint32_t firstFrame, uint32_t numFrames, TbFloat4* output, TbFloat4* jittered, const TboAovs* aovs, TbRayStats* stats, int numThreads)
{
    if (!scene || !constants || !output || y1 > H || y0 > y1) return -1;
    if (stats) memset(stats, 0, sizeof *stats);
    struct LogGuard { LogGuard() { const char* e = getenv("TB_ORACLE_RAY_LOG"); if (e && *e && !g_rayLog) g_rayLog = fopen(e, "a"); }
                      ~LogGuard() { if (g_rayLog) { fclose(g_rayLog); g_rayLog = nullptr; } } } logGuard;
    if (numThreads <= 1) { render_rows(scene, constants, W, H, y0, y1, firstFrame, numFrames, output, jittered, aovs, stats); return 0; }
    /* Strips of rows handed out dynamically; each pixel is touched by exactly one thread per frame and frames stay in order
     * inside a strip, so the image is identical to the serial one.  8 rows per strip, fewer when that would leave threads without
     * work (at least four strips per thread, down to single rows: 1080 rows on 256 threads are 1080 work items). */
    uint32_t strip = (y1 - y0) / (4u * (uint32_t)numThreads);
    strip = strip < 1u ? 1u : (strip > 8u ? 8u : strip);
    uint32_t nStrips = (y1 - y0 + strip - 1) / strip;
    std::atomic<uint32_t> next(0);
    std::vector<TbRayStats> local((size_t)numThreads);
    std::vector<std::thread> th;
    for (int t = 0; t < numThreads; t++) {
        th.emplace_back([&, t]() {
            TbRayStats* st = stats ? &local[(size_t)t] : nullptr;
            if (st) memset(st, 0, sizeof *st);
            for (;;) {
                uint32_t s = next.fetch_add(1);
                if (s >= nStrips) break;
                uint32_t a = y0 + s * strip, b = a + strip < y1 ? a + strip : y1;
                render_rows(scene, constants, W, H, a, b, firstFrame, numFrames, output, jittered, aovs, st);
            }
        });
    }
    for (auto& t : th) t.join();
    if (stats) for (auto& l : local) {
        stats->boxesTested += l.boxesTested; stats->trianglesTested += l.trianglesTested; stats->hitsShaded += l.hitsShaded;
        stats->materialFetches += l.materialFetches; stats->lightSamples += l.lightSamples; stats->samples += l.samples; stats->rays += l.rays;
    }
    return 0;
}

void tbo_sample_pixel(const TbSceneView* scene, const TbPerFrameConstants* constants, uint32_t W, uint32_t H, uint32_t x, uint32_t y,
                      float out[4], float* seedAfter, TbRayStats* stats)
{
    Ctx c; memset(&c, 0, sizeof c);
    c.scene = scene; c.pf = constants; c.width = W; c.height = H; c.x = x; c.y = y; c.stats = stats;
    sample_pixel(c, out);
    if (seedAfter) *seedAfter = c.seed;
}

void tbo_trace_closest(const TbSceneView* scene, uint32_t n, const float* origins, const float* dirs, float* t, int32_t* materialIndex,
                       float* bary, uint32_t* primitiveIndex, uint32_t* geometryIndex, float* normal, float* uv,
                       uint32_t* boxesTested, uint32_t* trianglesTested)
{
    for (uint32_t i = 0; i < n; i++) {
        Committed h; uint32_t tris, boxes;
        tb3 o = to3(origins + 3 * i), d = to3(dirs + 3 * i);
        bool hit = Traverse(scene, o, d, MIN_T, 999999.0f, h, tris, boxes);
        if (boxesTested) boxesTested[i] = boxes;
        if (trianglesTested) trianglesTested[i] = tris;
        t[i] = hit ? h.t : -1.0f;
        if (bary) { bary[2 * i] = hit ? h.bary[0] : 0; bary[2 * i + 1] = hit ? h.bary[1] : 0; }
        if (primitiveIndex) primitiveIndex[i] = hit ? h.primitiveIndex : 0xffffffffu;
        if (geometryIndex) geometryIndex[i] = hit ? h.geometryIndex : 0xffffffffu;
        int mat = -1; tb3 nrm = tb3_splat(0); float u = 0, v = 0;
        if (hit) {
            TbHitGroupRecord rec;
            if (h.geometryIndex < scene->numHitGroups) rec = scene->hitGroups[h.geometryIndex]; else memset(&rec, 0, sizeof rec);
            HitInfo hi;
            GetHitInfo(scene, rec, h.primitiveIndex, 1 - h.bary[0] - h.bary[1], h.bary[0], h.bary[1], hi);
            mat = (int)rec.MaterialIndex; nrm = hi.normal; u = hi.uvx; v = hi.uvy;
        }
        if (materialIndex) materialIndex[i] = mat;
        if (normal) { normal[3 * i] = nrm.x; normal[3 * i + 1] = nrm.y; normal[3 * i + 2] = nrm.z; }
        if (uv) { uv[2 * i] = u; uv[2 * i + 1] = v; }
    }
}

void tbo_set_alpha_test(int enabled) { g_alphaTest = enabled ? 1 : 0; }

float tbo_hash13(float x, float y, float z) { return hash13(tb3_make(x, y, z)); }

void tbo_rand_stream(float seed, float time, uint32_t n, float* out)
{
    TbPerFrameConstants pf; memset(&pf, 0, sizeof pf); pf.Time = time;
    Ctx c; memset(&c, 0, sizeof c); c.pf = &pf; c.seed = seed;
    for (uint32_t i = 0; i < n; i++) out[i] = rnd(c);
}

/* Known-answer probes of the BSDF sampling pieces (tests/test_oracle_furnace.py): the functions themselves, not restatements */
/* the pdf the throughput update uses, kernel.glsl:1701-1706 */
float tbo_ggx_pdf(const float* normal, const float* incoming, const float* outgoing, float roughness)
{
    const tb3 n = to3(normal), in = to3(incoming), out = to3(outgoing);
    return ImportanceSampleGGXPDF(n, out, GetHalfVectorSafe(-in, out, n), roughness);
}
void tbo_sample_directions(int kind, float seed, float time, const float* incoming, const float* normal, float roughness, uint32_t n, float* outDirs,
    float* outPdf)
{
    TbPerFrameConstants pf; memset(&pf, 0, sizeof pf); pf.Time = time;
    Ctx c; memset(&c, 0, sizeof c); c.pf = &pf; c.seed = seed;
    const tb3 nn = to3(normal), in = incoming ? to3(incoming) : tb3_splat(0);
    for (uint32_t i = 0; i < n; i++) {
        tb3 d; float pdf = 0.0f;
        if (kind == 0) d = ImportanceSampleGGX(c, in, nn, roughness);                                  /* :1066-1082 */
        else if (kind == 1) d = GenerateRandomImportanceSampledDirection(c, nn, roughness, pdf);        /* :1048-1064,1096-1099 */
        else { float r0 = rnd(c); float r1 = rnd(c); d = GenerateCosineWeightedDirection(nn, r0, r1, pdf); } /* :1025-1046 */
        outDirs[3 * i] = d.x; outDirs[3 * i + 1] = d.y; outDirs[3 * i + 2] = d.z;
        if (outPdf) outPdf[i] = pdf;
    }
}

/* SelectPixel -> ReadbackStats (TracerBoy.h:362-368): what StatsBuffer +8 / +12 hold after frames [firstFrame, firstFrame + numFrames)
 * of the pixel (constants->SelectedPixelX, Y): OutputDistanceToFirstHit / OutputMaterial of the last frame whose primary ray hit
 * (RayGenCommon.h:632-648, kernel.glsl:1370-1371).  Returns 0 when no frame wrote them. */
int tbo_selected_pixel(const TbSceneView* scene, const TbPerFrameConstants* constants, uint32_t W, uint32_t H, uint32_t firstFrame, uint32_t numFrames,
                       float* distance, int32_t* materialId)
{
    int written = 0;
    if (constants->SelectedPixelX >= W || constants->SelectedPixelY >= H) return 0;
    for (uint32_t f = 0; f < numFrames; f++) {
        TbPerFrameConstants pf = *constants; pf.GlobalFrameCount = firstFrame + f;
        Ctx c; memset(&c, 0, sizeof c);
        c.scene = scene; c.pf = &pf; c.width = W; c.height = H; c.x = pf.SelectedPixelX; c.y = pf.SelectedPixelY;
        float s[4]; sample_pixel(c, s);
        if (c.selWritten) { *distance = c.selDistance; *materialId = c.selMaterial; written = 1; }
    }
    return written;
}

float tbo_math(int fn, float a, float b)
{
    switch (fn) {
    case 0: return tb_sin(a); case 1: return tb_cos(a); case 2: return tb_acos(a); case 3: return tb_atan2(a, b);
    case 4: return tb_exp(a); case 5: return tb_log(a); case 6: return tb_pow(a, b); case 7: return tb_sqrt(a);
    case 8: return tb_exp2(a); case 9: return tb_log2(a); case 10: return tb_asin(a);
    default: return 0.0f;
    }
}

/* vector form for tests/test_math.py; codes 14-18: min, max, frac, floor, 1/x */
void tbo_math_array(int fn, uint32_t n, const float* a, const float* b, float* out)
{
    for (uint32_t i = 0; i < n; i++) {
        const float x = a[i], y = b ? b[i] : 0.0f;
        switch (fn) {
        case 14: out[i] = tb_min(x, y); break; case 15: out[i] = tb_max(x, y); break; case 16: out[i] = tb_frac(x); break;
        case 17: out[i] = tb_floor(x); break; case 18: out[i] = tb_rcp(x); break;
        default: out[i] = tbo_math(fn, x, y);
        }
    }
}

void tbo_camera_ray(const TbPerFrameConstants* constants, float lensHeight, uint32_t W, uint32_t H, float pixelX, float pixelY,
                    float jitterX, float jitterY, float origin[3], float dir[3])
{
    CameraRays cr = MakeCameraRays(*constants, lensHeight, W, H, pixelX, pixelY, jitterX, jitterY, 0.0f, 0.0f);
    origin[0] = cr.camera.origin.x; origin[1] = cr.camera.origin.y; origin[2] = cr.camera.origin.z;
    dir[0] = cr.camera.direction.x; dir[1] = cr.camera.direction.y; dir[2] = cr.camera.direction.z;
}

} // extern "C"
