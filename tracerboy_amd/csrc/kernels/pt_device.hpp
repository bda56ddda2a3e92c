/* pt_device.hpp -- device functions of the MI355X path tracer.
 *
 * The reference runs one thread per pixel through one monolithic shader
 * (/root/reference/TracerBoy/SoftwareRayTraceCS.hlsl -> RayGenCommon.h:690 RayTraceCommon ->
 * kernel.glsl:1805 PathTrace -> :1278 Trace).  Here the same per-path arithmetic is cut at every
 * ray cast into a small state machine, so that a lane (persistent megakernel) or a queue entry
 * (wavefront pipeline) always has exactly one pending ray:
 *
 *     path_begin()           camera ray                         kernel.glsl:1805-1906 + :1283
 *     path_pre_extend()      Russian roulette                   kernel.glsl:1288-1302
 *     -- traverse --                                           TraverseFunction.hlsli:537-779
 *     path_on_closest()      miss / material / emissive / NEE sample -> shadow ray   :1312-1455
 *     -- traverse --
 *     path_on_shadow()       visibility + NEE add               kernel.glsl:1460-1516
 *     path_scatter()         BSDF sample, throughput, (enter SSS walk)              :1519-1772
 *     -- traverse --  (SSS walk only)
 *     path_on_sss()          one interior random-walk step      kernel.glsl:1601-1687
 *
 * All arithmetic goes through include/tb_math.h / tb_vec.h with -ffp-contract=off, in the same
 * order as the CPU checker, which is what makes the images bit-identical.  "R" = one rnd() call.
 */
#pragma once
#include <hip/hip_runtime.h>
#include "../../../include/tb_vec.h"
#include "pt_scene.h"
#include "pt_device_features.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define TBD __device__ __forceinline__

namespace pt {

/* Pricing builds (scripts/c2_instruction_mix.sh; never shipped): -DTB_EXP_DOUBLE=k evaluates ONE phase of the path loop a second time on
 * operands the compiler cannot tell from new ones (they pass through an empty asm) and folds the copy's results into a compare that never
 * holds, so that nothing of it is shared with the real evaluation or removed.  The difference of SQ_INSTS_VALU between such a build and
 * the shipped one is the phase's dynamic VALU instruction count.  1 path_begin, 2 both walks whole, 3 the inner-node step, 4 the leaf step,
 * 5 path_on_closest, 6 path_scatter, 7 the ray set-up (GetRayData + root box), 8 every rnd(), 9 every IEEE division of ray set-up. */
#ifndef TB_EXP_DOUBLE
#define TB_EXP_DOUBLE 0
#endif
__device__ __forceinline__ float exp_opaque(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ bool exp_never(float a, float b, float c, float d) { return __float_as_uint(a) + 3u * __float_as_uint(b) + 5u * __float_as_uint(c) + 7u * __float_as_uint(d) == 0x7fc12345u && a == 1.2345e-33f; }

constexpr float EPSILON = 0.000001f;      /* kernel.glsl:1 */
constexpr float PI = 3.1415926535f;       /* kernel.glsl:2 */
constexpr float LARGE_NUMBER = 1e20f;
constexpr float AIR_IOR = 1.0f;
constexpr float MIN_ROUGHNESS = 0.04f;
constexpr float MIN_ROUGHNESS_SQUARED = (float)(0.04 * 0.04);
constexpr float MIN_T = 0.001f;           /* RayGenCommon.h:364 */
constexpr float MAX_T = 999999.0f;        /* kernel.glsl:1160 */
constexpr int MAX_SSS_BOUNCES = 100;      /* kernel.glsl:1565 */

/* Scene/setting features a kernel variant is compiled for.  The host picks the smallest compiled
 * superset of what the loaded scene and the output settings can ever exercise, so a stripped
 * branch is one that could never be taken: results are identical across variants. */
enum : uint32_t {
    FEAT_ENV = PT_FEAT_ENV, FEAT_SPECULAR = PT_FEAT_SPECULAR, FEAT_TEXTURES = PT_FEAT_TEXTURES, FEAT_SSS = PT_FEAT_SSS,
    FEAT_MIX = PT_FEAT_MIX, FEAT_EXT = PT_FEAT_EXT, FEAT_ALL = PT_FEAT_ALL,
};

/* Pointers the step functions read through; filled from global memory or from the LDS copy. */
struct SceneRefs {
    const uint8_t* nodes; /* layout-B nodes: 64 B apart in global memory, 80 B apart in the LDS copy (bank spread) */
    const TbTriB* tris; uint32_t trisPermuted; /* LDS image: six axis-permuted copies per triangle (pt_scene.h) */
    const TbDevHitGroup* hitGroups; const uint32_t* indices; const float* vertices;
    const TbDevMaterial* materials; const TbDevLight* lights;
    uint32_t numHitGroups, numIndices, numVertexFloats, numMaterials, numLights;
};

TBD tb3 ld3(const float* p) { return tb3_make(p[0], p[1], p[2]); }
TBD tb3 ld3(const TbFloat3& f) { return tb3_make(f.x, f.y, f.z); }

/* ---- RNG: kernel.glsl:39-40, RayGenCommon.h:662-667 ------------------------------------------ */
TBD float rnd(float& seed, float time)
{
    float s = seed;
    seed = seed + 1.0f;
#if TB_EXP_DOUBLE == 8
    { const float r2 = tb_frac(tb_sin(exp_opaque(s) + time) * 43758.5453123f); if (exp_never(r2, s, time, r2)) seed = seed + 1.0f; }
#endif
    return tb_frac(tb_sin(s + time) * 43758.5453123f);
}

TBD float hash13(float x, float y, float z)
{
    tb3 p = tb3_make(tb_frac(x * .1031f), tb_frac(y * .1031f), tb_frac(z * .1031f));
    float d = (p.x * (p.y + 33.33f) + p.y * (p.z + 33.33f)) + p.z * (p.x + 33.33f); /* unfused, like the checker */
    p = tb3_make(p.x + d, p.y + d, p.z + d);
    return tb_frac((p.x + p.y) * p.z);
}

/* ---- traversal ------------------------------------------------------------------------------- */
struct RayPre { tb3 inv, ainv, oinv, shear; int kx, ky, kz; tb3 o, operm; uint32_t permUnits; };

/* Axis-parallel rays.  With d.k == 0 the reference's slab arithmetic (c*inv - o*inv, TraverseFunction.hlsli:212-214)
 * yields inf - inf = NaN on axis k, which min/max ignore: the axis never rejects a box, so such a ray visits every
 * node inside its slab on the other axes (hundreds of thousands on a large scene) -- and the reference RNG produces
 * them routinely (rand() returns exactly 0 about once in 300 calls, giving a bounce direction equal to the normal).
 * The build adds the missing test without adding an instruction to the walk: on a degenerate axis the ray's constants
 * become inv = 2^80, o*inv = o * 2^80 (exact) and |inv| = 2^80 (1 + 2^-10), so the same fmas give (c -+ 1.001 h - o) 2^80
 * -- far below zero / far above `closest` when the origin is inside the box's slab (widened by a thousandth of its
 * half-width so that no box the reference would have needed is lost), far above `closest` on the entry side when it
 * is outside.  Hits are unchanged; only fewer boxes are visited.  oracle/tb_oracle.cpp does the same. */
#define TB_DEGEN_INV 1.2089258196146292e24f       /* 2^80 */
#define TB_DEGEN_AINV 1.2101064112353466e24f      /* 2^80 (1 + 2^-10) */

/* GetRayData in two halves: the five quotients (1 / d per axis, the two shear terms) and everything else.  The split-role kernel
 * (pt_split.inc) has the lane that SHADES a path divide, with the wave's other paths, and hands the quotients to the lane that walks
 * the ray, which assembles the rest from compares, products and selects; ray_prepare() is the two halves back to back. */
struct RayDiv { tb3 inv0; float sx, sy; };

TBD void ray_axes(tb3 d, int& kx, int& ky, int& kz, float& dz)
{
    tb3 a = tb3_abs(d);
    int z = (a.x > a.y && a.x > a.z) ? 0 : (a.y > a.z ? 1 : 2);
    kx = z == 2 ? 0 : z + 1; ky = kx == 2 ? 0 : kx + 1;
    dz = tb3_get(d, z);
    if (dz < 0.0f) { int t = kx; kx = ky; ky = t; }
    kz = z;
}

TBD RayDiv ray_divide(tb3 d)
{
    RayDiv q;
    q.inv0 = tb3_make(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int kx, ky, kz; float dz;
    ray_axes(d, kx, ky, kz, dz);
    q.sx = tb3_get(d, kx) / dz; q.sy = tb3_get(d, ky) / dz;
    return q;
}

TBD RayPre ray_assemble(tb3 o, tb3 d, const RayDiv& q)
{
    RayPre r;
    r.inv = q.inv0; /* inv0: before the degenerate-axis substitution: 1 / d[kz] below is one of these three quotients */
    r.oinv = o * r.inv; r.ainv = tb3_abs(r.inv);
    if (d.x == 0.0f) { r.inv.x = TB_DEGEN_INV; r.ainv.x = TB_DEGEN_AINV; r.oinv.x = o.x * TB_DEGEN_INV; }
    if (d.y == 0.0f) { r.inv.y = TB_DEGEN_INV; r.ainv.y = TB_DEGEN_AINV; r.oinv.y = o.y * TB_DEGEN_INV; }
    if (d.z == 0.0f) { r.inv.z = TB_DEGEN_INV; r.ainv.z = TB_DEGEN_AINV; r.oinv.z = o.z * TB_DEGEN_INV; }
    int kx, ky, z; float dz;
    ray_axes(d, kx, ky, z, dz);
    r.kx = kx; r.ky = ky; r.kz = z;
    r.operm = tb3_make(tb3_get(o, kx), tb3_get(o, ky), tb3_get(o, z)); /* for the axis-permuted triangle copies */
    r.permUnits = (uint32_t)(z * 2 + (dz < 0.0f ? 1 : 0)) * 3u;        /* copy index * 48 B / 16 */
    r.o = o;
    r.shear = tb3_make(q.sx, q.sy, tb3_get(q.inv0, z)); /* Shear.z = 1 / d[kz]: the same division as inv0[kz], not repeated */
    return r;
}

TBD RayPre ray_prepare(tb3 o, tb3 d) /* GetRayData, TraverseFunction.hlsli:473-495 */
{
    return ray_assemble(o, d, ray_divide(d));
}

/* A ray with a NaN in its origin or direction hits nothing: the NaN reaches U, V and W of the watertight test together and every one of
 * its comparisons fails.  The slab test's min / max DROP NaN operands, though, so the literal walk visits most or all of the tree before it
 * has hit nothing -- 1.39 million steps for one ray in 12 000 of the reference's vw-van scene, 86 % of the render's steps and, in a lock-step
 * wave, three orders of magnitude of its time (1.5 Msamples/s).  Such a ray is a miss at once, in the kernels and in the checker alike
 * (oracle/tb_oracle.cpp RayCannotHit; TB_LITERAL_BOX_TEST=1 walks it literally: tests/test_vw_van.py shows the same bits either way). */
/* three v_cmp_u_f32: unordered = either operand is a NaN */
TBD bool ray_cannot_hit(tb3 o, tb3 d) { return __builtin_isunordered(o.x, o.y) || __builtin_isunordered(o.z, d.x) || __builtin_isunordered(d.y, d.z); }

TBD bool box_test(float& tEntry, float closest, const RayPre& r, tb3 c, tb3 h) /* RayBoxTest :204-221 */
{
    /* (not `precise` in the reference: contraction allowed, pinned here as explicit fmas) */
    tb3 ai = r.ainv;
    tb3 mid = tb3_make(tb_fma(c.x, r.inv.x, -r.oinv.x), tb_fma(c.y, r.inv.y, -r.oinv.y), tb_fma(c.z, r.inv.z, -r.oinv.z));
    tb3 hi = tb3_make(tb_fma(h.x, ai.x, mid.x), tb_fma(h.y, ai.y, mid.y), tb_fma(h.z, ai.z, mid.z));
    tb3 lo = tb3_make(tb_fma(-h.x, ai.x, mid.x), tb_fma(-h.y, ai.y, mid.y), tb_fma(-h.z, ai.z, mid.z));
    float tmin = tb_max(tb_max(lo.x, lo.y), lo.z);
    float tmax = tb_min(tb_min(hi.x, hi.y), hi.z);
    tEntry = tb_max(tmin, 0.0f);
    return tb_max(tmin, 0.0f) < tb_min(tmax, closest);
}

/* Both children of a layout-B node at once: the same arithmetic as two box_test() calls, with the nine fmas of the
 * two boxes issued as packed fp32 (v_pk_fma_f32, IEEE fma per half, so the bits are those of the scalar form). */
typedef float tbf2 __attribute__((ext_vector_type(2)));
TBD tbf2 f2_splat(float x) { tbf2 r; r.x = x; r.y = x; return r; }
TBD tbf2 f2_ld(const float* p) { tbf2 r; r.x = p[0]; r.y = p[1]; return r; }

/* the records are 16-B aligned in global memory and in the LDS image; say so in the type so that they are fetched
 * with 16-B loads (global_load_dwordx4 / ds_read_b128) */
struct __attribute__((aligned(16))) NodeB16 { TbNodeB n; };
struct __attribute__((aligned(16))) TriB16 { TbTriB t; };

TBD TbNodeB load_node(const SceneRefs& sc, uint32_t ref)
{
    return ((const NodeB16*)(sc.nodes + ((size_t)ref << 4)))->n; /* device child refs are offsets in 16-B units (pt_scene.h) */
}

TBD TbTriB load_tri(const SceneRefs& sc, uint32_t leafRef) /* 48-B records from a 16-B aligned base: three aligned 16-B loads */
{
    return ((const TriB16*)((const uint8_t*)sc.tris + ((size_t)(leafRef & TB_DEVICE_REF_MASK) << 4)))->t;
}

TBD void box_test2(bool& lh, bool& rh, float& lt, float& rt, float closest, const RayPre& r, const TbNodeB& n)
{
    const tbf2 cx = f2_ld(n.cx), cy = f2_ld(n.cy), cz = f2_ld(n.cz), hx = f2_ld(n.hx), hy = f2_ld(n.hy), hz = f2_ld(n.hz);
    const tbf2 mx = __builtin_elementwise_fma(cx, f2_splat(r.inv.x), f2_splat(-r.oinv.x));
    const tbf2 my = __builtin_elementwise_fma(cy, f2_splat(r.inv.y), f2_splat(-r.oinv.y));
    const tbf2 mz = __builtin_elementwise_fma(cz, f2_splat(r.inv.z), f2_splat(-r.oinv.z));
    const tbf2 hix = __builtin_elementwise_fma(hx, f2_splat(r.ainv.x), mx), lox = __builtin_elementwise_fma(-hx, f2_splat(r.ainv.x), mx);
    const tbf2 hiy = __builtin_elementwise_fma(hy, f2_splat(r.ainv.y), my), loy = __builtin_elementwise_fma(-hy, f2_splat(r.ainv.y), my);
    const tbf2 hiz = __builtin_elementwise_fma(hz, f2_splat(r.ainv.z), mz), loz = __builtin_elementwise_fma(-hz, f2_splat(r.ainv.z), mz);
    const float tminL = tb_max(tb_max(lox.x, loy.x), loz.x), tmaxL = tb_min(tb_min(hix.x, hiy.x), hiz.x);
    const float tminR = tb_max(tb_max(lox.y, loy.y), loz.y), tmaxR = tb_min(tb_min(hix.y, hiy.y), hiz.y);
    lt = tb_max(tminL, 0.0f); rt = tb_max(tminR, 0.0f);
    lh = lt < tb_min(tmaxL, closest); rh = rt < tb_min(tmaxR, closest);
}

/* Layout C (tb_abi.h TbNodeC): both boxes as centre / half-extent on the 16-bit grid ds.quant, fetched with two aligned 16-B loads.
 * The ray's slab constants are taken into grid units once per ray (ray_to_grid), after which the arithmetic is box_test2()'s:
 * mid = c * inv - o * inv with c = origin + cell * q becomes q * (cell * inv) - (o - origin) * inv. */
struct __attribute__((aligned(16))) NodeC16 { uint32_t cx, cy, cz, hx, hy, hz, left, right; };
TBD NodeC16 load_node_c(const TbDeviceScene& ds, uint32_t ref) { return *(const NodeC16*)((const uint8_t*)ds.nodesC + ((size_t)ref << 4)); }
TBD tbf2 f2_u16x2(uint32_t w) { tbf2 r; r.x = (float)(w & 0xffffu); r.y = (float)(w >> 16); return r; } /* v_cvt_f32_u32 with an SDWA word select each */

TBD void ray_to_grid(RayPre& r, const TbDeviceScene& ds, tb3 o)
{
    const tb3 cell = ld3(ds.quant.cell), org = ld3(ds.quant.origin);
    r.oinv = (o - org) * r.inv;          /* r.inv already carries the degenerate-axis substitution (2^80: the product stays exact) */
    r.inv = r.inv * cell; r.ainv = r.ainv * cell;
}

TBD void box_test2_c(bool& lh, bool& rh, float& lt, float& rt, float closest, const RayPre& r, const NodeC16& n)
{
    const tbf2 cx = f2_u16x2(n.cx), cy = f2_u16x2(n.cy), cz = f2_u16x2(n.cz), hx = f2_u16x2(n.hx), hy = f2_u16x2(n.hy), hz = f2_u16x2(n.hz);
    const tbf2 mx = __builtin_elementwise_fma(cx, f2_splat(r.inv.x), f2_splat(-r.oinv.x));
    const tbf2 my = __builtin_elementwise_fma(cy, f2_splat(r.inv.y), f2_splat(-r.oinv.y));
    const tbf2 mz = __builtin_elementwise_fma(cz, f2_splat(r.inv.z), f2_splat(-r.oinv.z));
    const tbf2 hix = __builtin_elementwise_fma(hx, f2_splat(r.ainv.x), mx), lox = __builtin_elementwise_fma(-hx, f2_splat(r.ainv.x), mx);
    const tbf2 hiy = __builtin_elementwise_fma(hy, f2_splat(r.ainv.y), my), loy = __builtin_elementwise_fma(-hy, f2_splat(r.ainv.y), my);
    const tbf2 hiz = __builtin_elementwise_fma(hz, f2_splat(r.ainv.z), mz), loz = __builtin_elementwise_fma(-hz, f2_splat(r.ainv.z), mz);
    const float tminL = tb_max(tb_max(lox.x, loy.x), loz.x), tmaxL = tb_min(tb_min(hix.x, hiy.x), hiz.x);
    const float tminR = tb_max(tb_max(lox.y, loy.y), loz.y), tmaxR = tb_min(tb_min(hix.y, hiy.y), hiz.y);
    lt = tb_max(tminL, 0.0f); rt = tb_max(tminR, 0.0f);
    lh = lt < tb_min(tmaxL, closest); rh = rt < tb_min(tmaxR, closest);
}

struct Hit { float t, u, v; uint32_t prim, geom; };

/* Woop/Benthin/Wald watertight test, two-sided branch: RayTriangleIntersect :232-313 + :420-426 */
/* IsValidHit (SharedHitGroup.h:157-179), the any-hit alpha test of the reference's hardware path; defined after the
 * texture functions.  The software path the build mirrors compiles it out (DISABLE_ANYHIT, RayGenCommon.h:357); option
 * "alpha_test" turns it on as a filter on candidate hits of non-opaque geometry. */
TBD bool is_valid_hit(const SceneRefs& sc, const TbDeviceScene& ds, uint32_t geom, uint32_t prim, float u, float v);

template <bool ALPHA>
TBD void tri_test(Hit& best, float tMin, tb3 o, const RayPre& r, const TbTriB& tri, bool permuted, const SceneRefs& sc, const TbDeviceScene& ds,
    uint32_t geomBase = 0u)
{
    float Az, Bz, Cz, U, V, W;
    if (permuted) {
        /* the record already holds (v[kx], v[ky], v[kz]) for this ray's axis order: no per-lane component selects, and
         * the x/y halves go through packed fp32 ops (each product / fma rounded on its own, like the scalar form) */
        const tbf2 oxy = {r.operm.x, r.operm.y}, shear = {r.shear.x, r.shear.y};
        Az = tri.v0[2] - r.operm.z; Bz = tri.v1[2] - r.operm.z; Cz = tri.v2[2] - r.operm.z;
        const tbf2 A = __builtin_elementwise_fma(-shear, f2_splat(Az), f2_ld(tri.v0) - oxy);
        const tbf2 B = __builtin_elementwise_fma(-shear, f2_splat(Bz), f2_ld(tri.v1) - oxy);
        const tbf2 C = __builtin_elementwise_fma(-shear, f2_splat(Cz), f2_ld(tri.v2) - oxy);
        /* `precise` in the reference (TraverseFunction.hlsli:260-262): never contracted */
        const tbf2 pu = C * B.yx, pv = A * C.yx, pw = B * A.yx;
        U = pu.x - pu.y; V = pv.x - pv.y; W = pw.x - pw.y;
    } else {
        const tb3 a = ld3(tri.v0) - o, b = ld3(tri.v1) - o, c = ld3(tri.v2) - o;
        Az = tb3_get(a, r.kz); Bz = tb3_get(b, r.kz); Cz = tb3_get(c, r.kz);
        const float Ax = tb_fma(-r.shear.x, Az, tb3_get(a, r.kx)), Ay = tb_fma(-r.shear.y, Az, tb3_get(a, r.ky));
        const float Bx = tb_fma(-r.shear.x, Bz, tb3_get(b, r.kx)), By = tb_fma(-r.shear.y, Bz, tb3_get(b, r.ky));
        const float Cx = tb_fma(-r.shear.x, Cz, tb3_get(c, r.kx)), Cy = tb_fma(-r.shear.y, Cz, tb3_get(c, r.ky));
        /* `precise` in the reference (TraverseFunction.hlsli:260-262): never contracted */
        U = Cx * By - Cy * Bx;
        V = Ax * Cy - Ay * Cx;
        W = Bx * Ay - By * Ax;
    }
    float det = U + V + W;
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return;
    if (det == 0.0f) return;
    Az = r.shear.z * Az; Bz = r.shear.z * Bz; Cz = r.shear.z * Cz;
    float T = tb_fma(W, Cz, tb_fma(V, Bz, U * Az));
    float sT = tb_abs(T);
    if ((T > 0.0f) != (det > 0.0f)) sT = -sT;
    if (sT < 0.0f || sT > best.t * tb_abs(det)) return;
    float rcpDet = 1.0f / det;
    float t0 = T * rcpDet;
    if (t0 < best.t && t0 > tMin) {
        const float bu = V * rcpDet, bv = W * rcpDet;
        /* non-opaque candidate (D3D12_RAYTRACING_GEOMETRY_FLAG_OPAQUE clear): RayGenCommon.h:423-434 */
        /* geomBase: InstanceContributionToHitGroupIndex of the instance being walked (two-level scenes), :427 */
        if (ALPHA && ds.alphaTest && !(tri.geometryFlags & 1u) && !is_valid_hit(sc, ds, tri.geometryIndex + geomBase, tri.primitiveIndex, bu, bv)) return;
        best.t = t0; best.u = bu; best.v = bv;
        best.prim = tri.primitiveIndex; best.geom = tri.geometryIndex + geomBase;
    }
}

/* Stack-based near-first BVH2 walk over layout B (tb_abi.h).  Visit order, box arithmetic and the
 * `closest` used by each box test are exactly Traverse()'s (TraverseFunction.hlsli:584-768): both
 * children are tested when their parent is popped, the far child is pushed and the near child is
 * visited next (the reference pushes both and pops the near one straight back, :163-180), ties go
 * left.  stack: LDS, entry e of this lane at stack[e * stride]. */
/* Wave-occupancy profile (counting kernel only): for each phase, the number of times a wave executed it
 * ("trips", counted by the first active lane) and the number of lanes that were active in it. */
enum { PROF_INNER = 0, PROF_LEAF = 2, PROF_CLOSEST = 4, PROF_SHADOW = 6, PROF_SCATTER = 8, PROF_REGEN = 10, PROF_ITER = 12, PROF_SLOTS = 14 };
struct WaveProf { unsigned long long v[PROF_SLOTS]; };
TBD void prof_hit(WaveProf* p, int slot)
{
    if (!p) return;
    p->v[slot] += 1; /* active lane-executions */
    unsigned long long m = __ballot(1);
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) p->v[slot + 1] += 1; /* wave trips */
}

/* The values a walk reads at every step are made ITS OWN at its entry: passed through an empty asm they become new values as far as the
 * register allocator can tell, live from here to the end of the walk.  Without that the ray origin is the same value as Path::ro, which
 * lives from the start of the bounce to the end of shading; the kernels held to 96 registers spill that whole range and reload the origin
 * from scratch at every use -- at every triangle test, 2-3 scratch loads per leaf step in a loop whose texture addresser is the
 * bottleneck (scripts/isa_spill_map.py lists them).  Round 4: van- / bistro-class 4K +7 %, vw-van +15 %; no instruction added. */
TBD void walk_owns(tb3& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z)); }

/* RAY_COUNTERS: the ray's BoxesTested / TrianglesTested (TraverseFunction.hlsli:662,751) are kept -- counting launches, and the full
 * feature set, whose heatmap output reads them (RayGenCommon.h:537-543); ALPHA: the IsValidHit filter on non-opaque candidates is
 * compiled in (it still needs ds.alphaTest at run time); HYBRID: split stack; NODEC: fetch layout-C nodes (ds.nodesC) instead of layout B;
 * PROFILE: the wave-occupancy profile (counting launches only).  One flag per thing a kernel pays for: the first parameter used to stand
 * for "counters or the full feature set", one refactor away from changing who pays for what. */
template <bool RAY_COUNTERS, bool ALPHA, bool HYBRID = false, bool NODEC = false, bool PROFILE = false, bool OWN_DIR = false>
TBD bool traverse(const SceneRefs& sc, const TbDeviceScene& ds, tb3 o, tb3 d, Hit& best, uint32_t* stack, uint32_t stride,
                  uint32_t& boxes, uint32_t& tris, WaveProf* prof = nullptr, uint32_t* overflow = nullptr)
{
    best.t = MAX_T; best.u = best.v = 0.0f; best.prim = best.geom = 0u;
    if (ray_cannot_hit(o, d)) return false;
    walk_owns(o);
    /* OWN_DIR (the feature sets with interior walks, whose kernels are held to 6 / 4 waves per SIMD): the direction too -- ray_prepare alone reads
     * it, but as the path's own rd it lives across the walk, was spilled whole and reloaded at each of its uses, and in one kernel a component of
     * the ORIGIN went to scratch inside the walk instead.  Not for the other sets: the copy is three v_mov per ray, 0.6 % of the VALU-bound
     * cornell-box kernel's instructions (round 5: 7 001 against 7 065 Msamples/s with it there). */
    if (OWN_DIR) walk_owns(d);
    RayPre r = ray_prepare(o, d);
    float unusedT;
#if TB_EXP_DOUBLE == 7
    { const tb3 o2 = tb3_make(exp_opaque(o.x), exp_opaque(o.y), exp_opaque(o.z)), d2 = tb3_make(exp_opaque(d.x), exp_opaque(d.y), exp_opaque(d.z));
      const RayPre r2 = ray_prepare(o2, d2); float t2; const bool in2 = box_test(t2, best.t, r2, ld3(ds.rootCenter), ld3(ds.rootHalf));
      if (exp_never(r2.inv.x + r2.inv.y + r2.inv.z, r2.oinv.x + r2.oinv.y + r2.oinv.z + r2.ainv.x + r2.ainv.y + r2.ainv.z, r2.shear.x + r2.shear.y + r2.shear.z + r2.operm.x + r2.operm.y +
          r2.operm.z, t2 + (float)(r2.kx + 3 * r2.ky + 9 * r2.kz + (int)r2.permUnits)) && in2) best.prim ^= 1u; }
#endif
    if (!box_test(unusedT, best.t, r, ld3(ds.rootCenter), ld3(ds.rootHalf))) return false; /* :566-580 */
    if (NODEC) ray_to_grid(r, ds, o);
    /* "while-while" scheduling: a lane that reaches a leaf parks on it until the other lanes of the
     * wave have reached theirs (or fewer than PARK_MIN are still descending), then the wave runs the
     * triangle test once for everybody.  Only the interleaving ACROSS lanes changes; each lane's own
     * sequence of box and triangle tests is unchanged. */
    constexpr uint32_t DONE = 0xffffffffu; /* has the leaf bit set, so it also leaves the inner loop */
    const int PARK_MIN = (int)ds.parkMin; /* option "park_min", default 8 */
    uint32_t top = 0;
    uint32_t ref = ds.rootRef;
    auto pop = [&]() -> uint32_t {
        if (!top) return DONE;
        --top;
        if (HYBRID && top >= ds.stackDepth) return overflow[(size_t)(top - ds.stackDepth) * ds.stackOverflowLanes];
        return stack[top * stride];
    };
    while (ref != DONE) {
        while (!(ref & TB_BVH_LEAF_FLAG)) {
            if (PROFILE) prof_hit(prof, PROF_INNER);
            float lt, rt; bool lh, rh; uint32_t nl, nr;
            if (NODEC) { const NodeC16 n = load_node_c(ds, ref); box_test2_c(lh, rh, lt, rt, best.t, r, n); nl = n.left; nr = n.right; }
            else { const TbNodeB n = load_node(sc, ref); box_test2(lh, rh, lt, rt, best.t, r, n); nl = n.left; nr = n.right;
#if TB_EXP_DOUBLE == 3
                { RayPre r2 = r; r2.inv = tb3_make(exp_opaque(r.inv.x), exp_opaque(r.inv.y), exp_opaque(r.inv.z)); float lt2, rt2; bool lh2, rh2;
                  box_test2(lh2, rh2, lt2, rt2, exp_opaque(best.t), r2, n); if (exp_never(lt2, rt2, lh2 ? 1.0f : 2.0f, rh2 ? 3.0f : 5.0f)) best.prim ^= 1u; }
#endif
            }
            if (RAY_COUNTERS) boxes += 2;
            if (lh && rh) {
                bool rightFirst = rt < lt;
                const uint32_t far = rightFirst ? nl : nr;
                /* HYBRID: the first ds.stackDepth entries in LDS, deeper ones (rare) in the lane's global overflow column */
                if (HYBRID && top >= ds.stackDepth) overflow[(size_t)(top - ds.stackDepth) * ds.stackOverflowLanes] = far;
                else stack[top * stride] = far;
                top++;
                ref = rightFirst ? nr : nl;
            } else if (lh || rh) {
                ref = rh ? nr : nl;
            } else {
                ref = pop();
            }
            if (__popcll(__ballot(!(ref & TB_BVH_LEAF_FLAG))) < PARK_MIN) break;
        }
        if ((ref & TB_BVH_LEAF_FLAG) && ref != DONE) {
            if (PROFILE) prof_hit(prof, PROF_LEAF);
            const TbTriB tri = load_tri(sc, sc.trisPermuted ? ref + r.permUnits : ref);
            if (RAY_COUNTERS) tris++;
#if TB_EXP_DOUBLE == 4
            { Hit b2 = best; b2.t = exp_opaque(best.t); RayPre r2 = r; r2.operm = tb3_make(exp_opaque(r.operm.x), exp_opaque(r.operm.y), exp_opaque(r.operm.z));
              r2.shear = tb3_make(exp_opaque(r.shear.x), exp_opaque(r.shear.y), exp_opaque(r.shear.z));
              const tb3 o2 = tb3_make(exp_opaque(o.x), exp_opaque(o.y), exp_opaque(o.z));
              tri_test<ALPHA>(b2, MIN_T, o2, r2, tri, sc.trisPermuted != 0, sc, ds); if (exp_never(b2.t, b2.u, b2.v, b2.t) && b2.prim == 77u) best.geom ^= 1u; }
#endif
            tri_test<ALPHA>(best, MIN_T, o, r, tri, sc.trisPermuted != 0, sc, ds);
            ref = pop();
        }
    }
    return best.t < MAX_T;
}

/* Two-level walk for instanced scenes (option flatten_instances = 0): TraverseFunction.hlsli:537-779 with FAST_PATH 0 -- the
 * !FAST_PATH branch :603-640.  The top level is walked with the world ray; at a top-level leaf the ray is taken into the instance's
 * object space (mul(WorldToObject, float4(o, 1)) / float4(d, 0): the direction is NOT renormalised, so t is the same number in
 * both spaces and `closest` carries over), GetRayData is evaluated again, and the bottom-level structure is walked from its root
 * -- whose own box is never tested (:625: StackPush(0)) -- until the stack is back at the height it had on entry; then the world
 * ray data are recomputed (:769-773) and the top level goes on.  Visit order at both levels as in traverse(), and since round 3
 * its scheduling too: while-while parking, the split stack (HYBRID), and a place in the frame-group kernels of the env / sss / vol
 * feature sets (TWOLEVEL copies, pt_variant.inc) beside the full-feature kernels.
 * ALPHA: the IsValidHit filter (option alpha_test) on non-opaque candidates of the bottom levels, RayGenCommon.h:423-434. */
struct __attribute__((aligned(16))) InstB16 { TbInstanceB i; };
TBD tb3 xfm_point34(const float* m, tb3 v) /* pinned order of the dp4: one fma chain per row, as host_scene / bvh_build / the oracle */
{
    return tb3_make(tb_fma(m[2], v.z, tb_fma(m[1], v.y, tb_fma(m[0], v.x, m[3]))), tb_fma(m[6], v.z, tb_fma(m[5], v.y, tb_fma(m[4], v.x, m[7]))),
                    tb_fma(m[10], v.z, tb_fma(m[9], v.y, tb_fma(m[8], v.x, m[11]))));
}
TBD tb3 xfm_vector34(const float* m, tb3 v)
{
    return tb3_make(tb_fma(m[2], v.z, tb_fma(m[1], v.y, m[0] * v.x)), tb_fma(m[6], v.z, tb_fma(m[5], v.y, m[4] * v.x)), tb_fma(m[10], v.z, tb_fma(m[9], v.y,
        m[8] * v.x)));
}

template <bool RAY_COUNTERS, bool ALPHA, bool HYBRID = false>
TBD bool traverse_instanced(const SceneRefs& sc, const TbDeviceScene& ds, tb3 o, tb3 d, Hit& best, uint32_t* stack, uint32_t stride, uint32_t& boxes,
    uint32_t& tris,
                            uint32_t* overflow = nullptr)
{
    best.t = MAX_T; best.u = best.v = 0.0f; best.prim = best.geom = 0u;
    if (ray_cannot_hit(o, d)) return false;
    walk_owns(o); walk_owns(d); /* both are read again at every instance entered */
    RayPre r = ray_prepare(o, d);
    float unusedT;
    if (!box_test(unusedT, best.t, r, ld3(ds.rootCenter), ld3(ds.rootHalf))) return false; /* :566-580, the top level's root box */
    /* the world ray's slab constants, kept for the way back out of an instance (:769-773 recomputes GetRayData there: the same values;
     * the top level tests boxes only, so the triangle-test half of RayPre may stay the object ray's) */
    const tb3 wInv = r.inv, wOinv = r.oinv, wAinv = r.ainv;
    constexpr uint32_t DONE = 0xffffffffu;
    const int PARK_MIN = (int)ds.parkMin;
    uint32_t top = 0, floor = 0, hitBase = 0;
    bool inBottom = false;
    tb3 ro = o;
    uint32_t ref = ds.rootRef;
    auto pop = [&]() -> uint32_t {
        if (inBottom && top == floor) { inBottom = false; r.inv = wInv; r.oinv = wOinv; r.ainv = wAinv; } /* bottom level exhausted: back to the world ray */
        if (!top) return DONE;
        --top;
        if (HYBRID && top >= ds.stackDepth) return overflow[(size_t)(top - ds.stackDepth) * ds.stackOverflowLanes];
        return stack[top * stride];
    };
    /* while-while like traverse(): lanes that reach a leaf of either level -- an instance to enter or a triangle to test -- park
     * until fewer than PARK_MIN lanes still descend; each lane's own sequence of tests is unchanged */
    while (ref != DONE) {
        while (!(ref & TB_BVH_LEAF_FLAG)) {
            const TbNodeB n = load_node(sc, ref);
            float lt, rt; bool lh, rh;
            box_test2(lh, rh, lt, rt, best.t, r, n);
            if (RAY_COUNTERS) boxes += 2;
            if (lh && rh) {
                const bool rightFirst = rt < lt;
                const uint32_t far = rightFirst ? n.left : n.right;
                if (HYBRID && top >= ds.stackDepth) overflow[(size_t)(top - ds.stackDepth) * ds.stackOverflowLanes] = far;
                else stack[top * stride] = far;
                top++;
                ref = rightFirst ? n.right : n.left;
            } else if (lh || rh) ref = rh ? n.right : n.left;
            else ref = pop();
            if (__popcll(__ballot(!(ref & TB_BVH_LEAF_FLAG))) < PARK_MIN) break;
        }
        if ((ref & TB_BVH_LEAF_FLAG) && ref != DONE) {
            if (!inBottom) { /* top-level leaf = instance (:603-640; InstanceMask 1 & inclusion mask 0xff: always valid) */
                const TbInstanceB in = ((const InstB16*)((const uint8_t*)ds.instances + ((size_t)(ref & TB_DEVICE_REF_MASK) << 4)))->i;
                ro = xfm_point34(in.worldToObject, o);
                r = ray_prepare(ro, xfm_vector34(in.worldToObject, d));
                inBottom = true; floor = top; hitBase = in.hitGroupBase;
                ref = in.blasRootRef;
            } else {
                const TbTriB tri = load_tri(sc, ref);
                if (RAY_COUNTERS) tris++;
                /* committed record: InstanceContributionToHitGroupIndex + GeometryContributionToHitGroupIndex; the IsValidHit filter sees the
                 * same index (RayGenCommon.h:427: CandidateInstanceIndex() + CandidateGeometryIndex()) */
                tri_test<ALPHA>(best, MIN_T, ro, r, tri, false, sc, ds, hitBase);
                ref = pop();
            }
        }
    }
    return best.t < MAX_T;
}

/* Resumable form of the same walk for the streaming kernel: the traversal state lives in registers
 * across calls, so a lane whose ray has finished can wait for shading while the others keep walking. */
struct Trav { RayPre r; Hit best; uint32_t ref, top; uint32_t boxes, tris; };
constexpr uint32_t TRAV_DONE = 0xffffffffu;

TBD bool trav_begin(Trav& t, const TbDeviceScene& ds, tb3 o, tb3 d) /* returns false when the root box is missed */
{
    t.best.t = MAX_T; t.best.u = t.best.v = 0.0f; t.best.prim = t.best.geom = 0u;
    t.boxes = t.tris = 0; t.top = 0;
    t.r = ray_prepare(o, d);
    float unusedT;
    bool in = !ray_cannot_hit(o, d) && box_test(unusedT, t.best.t, t.r, ld3(ds.rootCenter), ld3(ds.rootHalf));
    t.ref = in ? ds.rootRef : TRAV_DONE;
    return in;
}

/* One while-while round for the lanes with `busy` set; clears `busy` when a lane's walk is complete. */
template <bool RAY_COUNTERS, bool ALPHA, int PARK_MIN, bool PROFILE = false>
TBD void trav_round(Trav& t, bool& busy, const SceneRefs& sc, const TbDeviceScene& ds, uint32_t* stack, uint32_t stride, WaveProf* prof)
{
    while (busy && !(t.ref & TB_BVH_LEAF_FLAG)) {
        if (PROFILE) prof_hit(prof, PROF_INNER);
        const TbNodeB n = load_node(sc, t.ref);
        float lt, rt; bool lh, rh;
        box_test2(lh, rh, lt, rt, t.best.t, t.r, n);
        if (RAY_COUNTERS) t.boxes += 2;
        if (lh && rh) {
            bool rightFirst = rt < lt;
            stack[(t.top++) * stride] = rightFirst ? n.left : n.right;
            t.ref = rightFirst ? n.right : n.left;
        } else if (lh || rh) {
            t.ref = rh ? n.right : n.left;
        } else {
            t.ref = t.top ? stack[(--t.top) * stride] : TRAV_DONE;
        }
        if (__popcll(__ballot(!(t.ref & TB_BVH_LEAF_FLAG))) < PARK_MIN) break;
    }
    if (busy && (t.ref & TB_BVH_LEAF_FLAG)) {
        if (t.ref != TRAV_DONE) {
            if (PROFILE) prof_hit(prof, PROF_LEAF);
            const TbTriB tri = load_tri(sc, sc.trisPermuted ? t.ref + t.r.permUnits : t.ref);
            if (RAY_COUNTERS) t.tris++;
            tri_test<ALPHA>(t.best, MIN_T, t.r.o, t.r, tri, sc.trisPermuted != 0, sc, ds);
            t.ref = t.top ? stack[(--t.top) * stride] : TRAV_DONE;
        }
        if (t.ref == TRAV_DONE) busy = false;
    }
}

/* ---- hit attributes: SharedHitGroup.h:48-151 --------------------------------------------------- */
struct Surface { tb3 normal, tangent; float u, v; int material; };

struct __attribute__((aligned(16))) Vertex8 { float f[8]; };   /* N.xyz, UV.xy, T.xyz (SharedHitGroup.h:11 VertexStride 8) */
struct Index3 { uint32_t a, b, c; };

/* out-of-range reads return 0 like the reference's raw buffer loads; a record is either wholly inside or treated as absent */
struct __attribute__((aligned(16))) Vertex4 { float f[4]; };
TBD Vertex8 load_vertex(const SceneRefs& sc, uint32_t firstFloat, bool second = true)
{
    /* two 16-B pieces: normal and u; v and the tangent (second false: not fetched, zeros).  A vertex is wholly inside the buffer or absent. */
    const bool inside = firstFloat + 8u <= sc.numVertexFloats;
    Vertex4 a, b;
    for (int k = 0; k < 4; k++) a.f[k] = b.f[k] = 0.0f;
    if (inside) a = *(const Vertex4*)(sc.vertices + firstFloat);
    if (inside && second) b = *(const Vertex4*)(sc.vertices + firstFloat + 4);
    Vertex8 v;
    for (int k = 0; k < 4; k++) { v.f[k] = a.f[k]; v.f[4 + k] = b.f[k]; }
    return v;
}

/* needUV false: the second 16-B piece of the vertices (v, tangent) is not fetched and u = v = 0 -- for callers that know no material of the scene
 * reads a texture (TbDeviceScene::textureUse) */
TBD void fetch_surface(const SceneRefs& sc, const Hit& h, Surface& s, bool needTangent, bool needUV = true)
{
    TbDevHitGroup rec;
    if (h.geom < sc.numHitGroups) rec = sc.hitGroups[h.geom];
    else { rec.MaterialIndex = 0; rec.vFirst = 0; rec.iFirst = 0; rec.pad = 0; }
    const uint32_t iAt = rec.iFirst + h.prim * 3;
    Index3 ix; ix.a = ix.b = ix.c = 0u;
    if (iAt + 3u <= sc.numIndices) ix = *(const Index3*)(sc.indices + iAt);
    const float bx = 1 - h.u - h.v, by = h.u, bz = h.v; /* GetBarycentrics3 :135-138 */
    const Vertex8 v0 = load_vertex(sc, 8 * ix.a + rec.vFirst, needUV), v1 = load_vertex(sc, 8 * ix.b + rec.vFirst, needUV),
                  v2 = load_vertex(sc, 8 * ix.c + rec.vFirst, needUV);
    s.u = tb_fma(bz, v2.f[3], tb_fma(by, v1.f[3], bx * v0.f[3]));
    s.v = tb_fma(bz, v2.f[4], tb_fma(by, v1.f[4], bx * v0.f[4]));
    tb3 n0 = tb3_make(v0.f[0], v0.f[1], v0.f[2]), n1 = tb3_make(v1.f[0], v1.f[1], v1.f[2]), n2 = tb3_make(v2.f[0], v2.f[1], v2.f[2]);
    s.normal = tb3_normalize(tb3_bary(bx, by, bz, n0, n1, n2));
    if (needTangent) {
        tb3 t0 = tb3_make(v0.f[5], v0.f[6], v0.f[7]), t1 = tb3_make(v1.f[5], v1.f[6], v1.f[7]), t2 = tb3_make(v2.f[5], v2.f[6], v2.f[7]);
        s.tangent = tb3_normalize(tb3_bary(bx, by, bz, t0, t1, t2));
    } else s.tangent = tb3_splat(0.0f);
    s.material = (int)rec.MaterialIndex;
}

/* OutputDistanceToFirstHit + OutputMaterial for the selected pixel (RayGenCommon.h:632-648, called at kernel.glsl:1370-1371 for a
 * primary ray that hit): StatsBuffer +8 = asuint(distance), +12 = the hit group's MaterialIndex (result.y, before any mix coin). */
TBD void output_selected_pixel(uint32_t* stats, const SceneRefs& sc, const Hit& h)
{
    stats[2] = __float_as_uint(h.t);
    stats[3] = h.geom < sc.numHitGroups ? sc.hitGroups[h.geom].MaterialIndex : 0u;
}

/* ---- textures: SharedRaytracing.h:55-137 -------------------------------------------------------- */
struct F4 { float x, y, z, w; };
TBD F4 f4(float x, float y, float z, float w) { F4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

TBD uint32_t wrap_texel(float f, uint32_t n)
{
    float m = f - tb_floor(f / (float)n) * (float)n;
    int i = (int)m; if (i < 0) i = 0; if (i >= (int)n) i = (int)n - 1;
    return (uint32_t)i;
}

TBD F4 sample_bilinear_wrap(const TbFloat4* tex, uint32_t w, uint32_t h, float u, float v)
{
    if (!tex || w == 0 || h == 0) return f4(0, 0, 0, 0);
    float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
    float x0f = tb_floor(fx), y0f = tb_floor(fy);
    float tx = fx - x0f, ty = fy - y0f;
    uint32_t x0 = wrap_texel(x0f, w), x1 = wrap_texel(x0f + 1.0f, w), y0 = wrap_texel(y0f, h), y1 = wrap_texel(y0f + 1.0f, h);
    const TbFloat4 a = tex[y0 * w + x0], b = tex[y0 * w + x1], c = tex[y1 * w + x0], d = tex[y1 * w + x1];
    F4 r;
    r.x = tb_lerp(tb_lerp(a.x, b.x, tx), tb_lerp(c.x, d.x, tx), ty);
    r.y = tb_lerp(tb_lerp(a.y, b.y, tx), tb_lerp(c.y, d.y, tx), ty);
    r.z = tb_lerp(tb_lerp(a.z, b.z, tx), tb_lerp(c.z, d.z, tx), ty);
    r.w = tb_lerp(tb_lerp(a.w, b.w, tx), tb_lerp(c.w, d.w, tx), ty);
    return r;
}

/* The out-of-line helpers take what they read BY VALUE: handing them `ds` (a by-value kernel argument) by reference
 * makes the compiler copy the whole 336-B struct to scratch and turns every scene pointer loaded back from there into a
 * generic one (flat_load instead of global_load for all BVH / geometry fetches of the kernel). */
struct TexArgs { const TbTextureData* textureData; const TbImageDesc* images; const TbFloat4* texelPool; uint32_t numTextureData, numImages; int flip; };
TBD TexArgs tex_args(const TbDeviceScene& ds)
{
    TexArgs a; a.textureData = ds.textureData; a.images = ds.images; a.texelPool = ds.texelPool;
    a.numTextureData = ds.numTextureData; a.numImages = ds.numImages; a.flip = (int)ds.config.FlipTextureUVs;
    return a;
}

TBD F4 texture_nonrecursive(const TexArgs& ds, const TbTextureData& td, float u, float v)
{
    F4 data = f4(0, 0, 0, 0);
    if (td.TextureType == TB_TEXTURE_TYPE_IMAGE) {
        if (td.DescriptorHeapIndex < ds.numImages) {
            const TbImageDesc im = ds.images[td.DescriptorHeapIndex];
            data = sample_bilinear_wrap(ds.texelPool + im.texelOffset, im.width, im.height, u, v);
        }
    } else if (td.TextureType == TB_TEXTURE_TYPE_CHECKER) {
        float su = u * td.UScale, sv = v * td.VScale;
        data = f4(td.CheckerColor1.x, td.CheckerColor1.y, td.CheckerColor1.z, 1);
        if ((((int)su + (int)sv) % 2) == 0) data = f4(td.CheckerColor2.x, td.CheckerColor2.y, td.CheckerColor2.z, 1);
    }
    if (td.TextureFlags & TB_TEXTURE_FLAG_NEEDS_GAMMA) { data.x = tb_pow(data.x, 2.2f); data.y = tb_pow(data.y, 2.2f); data.z = tb_pow(data.z, 2.2f); }
    return data;
}

__device__ __noinline__ F4 texture_fetch_impl(TexArgs ds, uint32_t textureIndex, float u, float v)
{
    if (textureIndex == TB_INVALID_TEXTURE) return f4(0, 0, 0, 0);
    if (ds.flip) { u = 0.0f + u * 1.0f; v = 1.0f + v * -1.0f; }
    if (textureIndex >= ds.numTextureData) return f4(0, 0, 0, 0);
    const TbTextureData td = ds.textureData[textureIndex];
    if (td.TextureType == TB_TEXTURE_TYPE_SCALE) {
        TbTextureData z; memset(&z, 0, sizeof z);
        const TbTextureData t1 = td.TextureIndex1 < ds.numTextureData ? ds.textureData[td.TextureIndex1] : z;
        const TbTextureData t2 = td.TextureIndex2 < ds.numTextureData ? ds.textureData[td.TextureIndex2] : z;
        F4 c1 = texture_nonrecursive(ds, t1, u, v), c2 = texture_nonrecursive(ds, t2, u, v);
        return f4(c1.x * td.ScaleColor1.x + c2.x * td.ScaleColor2.x, c1.y * td.ScaleColor1.y + c2.y * td.ScaleColor2.y,
                  c1.z * td.ScaleColor1.z + c2.z * td.ScaleColor2.z, c1.w * 1.0f + c2.w * 1.0f);
    }
    return texture_nonrecursive(ds, td, u, v);
}

TBD F4 texture_fetch(const TbDeviceScene& ds, uint32_t textureIndex, float u, float v) { return texture_fetch_impl(tex_args(ds), textureIndex, u, v); }

struct EnvArgs { const TbFloat4* map; uint32_t w, h; tb3 vx, vy, vz, scale; };
__device__ __noinline__ tb3 sample_environment_impl(EnvArgs e, tb3 v) /* RayGenCommon.h:21-44 */
{
    v = tb3_make(tb3_dot(v, e.vx), tb3_dot(v, e.vy), tb3_dot(v, e.vz));
    tb3 dir = tb3_normalize(v);
    float p = tb_atan2(dir.y, dir.x);
    p = p > 0 ? p : p + 6.28f;
    float u = p / 6.28f;
    float w = tb_acos(dir.z) / 3.14f;
    if (!e.map) return tb3_splat(0.0f);
    F4 s = sample_bilinear_wrap(e.map, e.w, e.h, u, w);
    return tb3_make(s.x, s.y, s.z) * e.scale;
}
TBD tb3 sample_environment(const TbDeviceScene& ds, tb3 v)
{
    const TbConfigConstants& cc = ds.config;
    EnvArgs e; e.map = ds.envMap; e.w = ds.envWidth; e.h = ds.envHeight;
    e.vx = tb3_make(cc.EnvMapTransformVx.x, cc.EnvMapTransformVx.y, cc.EnvMapTransformVx.z);
    e.vy = tb3_make(cc.EnvMapTransformVy.x, cc.EnvMapTransformVy.y, cc.EnvMapTransformVy.z);
    e.vz = tb3_make(cc.EnvMapTransformVz.x, cc.EnvMapTransformVz.y, cc.EnvMapTransformVz.z);
    e.scale = ld3(cc.EnvironmentMapColorScale);
    return sample_environment_impl(e, v);
}

/* ---- materials: RayGenCommon.h:298-341, kernel.glsl:1224-1246 ---------------------------------- */
TBD TbMaterial fetch_material(const SceneRefs& sc, uint32_t id)
{
    TbMaterial m;
    if (id < sc.numMaterials) m = sc.materials[id].m; else memset(&m, 0, sizeof m);
    return m;
}

/* Outcome of a shadow feeler's first hit when GetMaterial draws no random number (no mix materials): only the LIGHT
 * flag of the base material matters (kernel.glsl:1460-1516; texture overrides never touch that flag). */
TBD bool shadow_hit_is_light(const SceneRefs& sc, uint32_t geom)
{
    const uint32_t mi = geom < sc.numHitGroups ? sc.hitGroups[geom].MaterialIndex : 0u;
    const int flags = mi < sc.numMaterials ? sc.materials[mi].m.Flags : 0;
    return (flags & TB_MAT_LIGHT) != 0;
}

TBD bool is_valid_hit(const SceneRefs& sc, const TbDeviceScene& ds, uint32_t geom, uint32_t prim, float u, float v)
{
    Hit h; h.t = 0.0f; h.u = u; h.v = v; h.prim = prim; h.geom = geom;
    Surface s;
    fetch_surface(sc, h, s, false);                              /* GetHitInfo: the interpolated uv is what matters */
    const TbMaterial m = fetch_material(sc, (uint32_t)s.material); /* GetMaterial_NonRecursive */
    if (m.alphaIndex != TB_INVALID_TEXTURE) return !(texture_fetch(ds, m.alphaIndex, s.u, s.v).x < 0.9f);
    if (m.albedoIndex != TB_INVALID_TEXTURE) return !(texture_fetch(ds, m.albedoIndex, s.u, s.v).w < 0.9f);
    return true;
}

template <uint32_t F>
TBD TbMaterial get_material(const SceneRefs& sc, const TbDeviceScene& ds, float& seed, float time, int id, float u, float v, bool back)
{
    TbMaterial mat = fetch_material(sc, (uint32_t)id);
    if (back) { mat.emissive.x = mat.emissive.y = mat.emissive.z = 0.0f; }
    if ((F & FEAT_MIX) && (mat.Flags & TB_MAT_MIX) != 0) { /* 1 R */
        /* one record fetch on the index the coin selects (the two sides' fetches were issued one after the other for a divergent wave) */
        const uint32_t pick = rnd(seed, time) < mat.albedo.z ? (uint32_t)mat.albedo.x : (uint32_t)mat.albedo.y;
        mat = fetch_material(sc, pick);
    } else if (F & FEAT_TEXTURES) {
        if (mat.albedoIndex != TB_INVALID_TEXTURE) { F4 t = texture_fetch(ds, mat.albedoIndex, u, v); mat.albedo.x = t.x; mat.albedo.y = t.y;
            mat.albedo.z = t.z; }
        if (mat.emissiveIndex != TB_INVALID_TEXTURE && !back) { F4 t = texture_fetch(ds, mat.emissiveIndex, u, v); mat.emissive.x = t.x; mat.emissive.y = t.y;
            mat.emissive.z = t.z; }
        if (mat.specularMapIndex != TB_INVALID_TEXTURE) { F4 t = texture_fetch(ds, mat.specularMapIndex, u, v); mat.roughness = t.y;
            if (t.z > 0.5f) mat.Flags |= TB_MAT_METALLIC; }
    }
    bool anyAlbedo = mat.albedo.x != 0.0f || mat.albedo.y != 0.0f || mat.albedo.z != 0.0f;
    if ((F & FEAT_SSS) && (mat.Flags & TB_MAT_SUBSURFACE_SCATTER) != 0 && anyAlbedo) { /* ArtistFriendlyAlbdeoToAbsorption :1224-1233 */
        tb3 color = ld3(mat.albedo), mfp = tb3_splat(1.0f) / ld3(mat.scattering);
        tb3 e = (-5.09406f * color + 2.61188f * color * color) - 4.31805f * color * color * color;
        tb3 alpha = tb3_splat(1.0f) - tb3_make(tb_exp(e.x), tb_exp(e.y), tb_exp(e.z));
        tb3 cm = color - tb3_splat(0.8f);
        tb3 s = (tb3_splat(1.9f) - color) + 3.5f * cm * cm;
        tb3 transmission = tb3_splat(1.0f) / (s * mfp);
        tb3 scattering = transmission * alpha, absorption = transmission - scattering;
        mat.absorption.x = absorption.x; mat.absorption.y = absorption.y; mat.absorption.z = absorption.z;
        mat.scattering.x = scattering.x; mat.scattering.y = scattering.y; mat.scattering.z = scattering.z;
        mat.albedo.x = mat.albedo.y = mat.albedo.z = 0.0f;
    }
    return mat;
}

/* ---- BSDF helpers: kernel.glsl:466-478, 991-1099, 1258-1269 ------------------------------------ */
TBD float ggx_ndf(tb3 N, tb3 H, float r2)
{
    r2 = tb_max(r2, MIN_ROUGHNESS_SQUARED);
    float a2 = r2 * r2;
    float nDotH = tb3_dot(N, H);
    float den = PI * tb_pow(nDotH * nDotH * (a2 - 1.0f) + 1.0f, 2.0f);
    return a2 / den;
}
TBD float diffuse_brdf(tb3 L, tb3 N) { return tb_max(tb3_dot(L, N), 0.0f) / PI; }

TBD tb3 reorient(tb3 v, tb3 n) /* ReorientVectorAroundNormal :1001-1015 */
{
    /* (a select form -- one square root and three divisions on operands chosen per lane -- was measured in round 5, twice, on one box against
     * this form: within +-0.5 % on all six workloads) */
    tb3 t;
    if (tb_abs(n.x) > tb_abs(n.y)) t = tb3_make(-n.z, 0, n.x) / tb_sqrt(n.x * n.x + n.z * n.z);
    else t = tb3_make(0, n.z, -n.y) / tb_sqrt(n.y * n.y + n.z * n.z);
    tb3 b = tb3_cross(n, t);
    return tb3_normalize(v.x * t + v.y * n + v.z * b);
}

TBD tb3 lobe_direction(tb3 n, float roughness, float r0, float r1, float& pdf) /* GenerateImportanceSampledDirection :1048-1064 */
{
    float lobe = tb_pow(1.0f - roughness, 5.0f) * 1000.0f;
    float theta = (2.0f * PI) * r1;
    float phi = tb_acos(tb_sqrt(tb_pow(r0, 1.0f / (lobe + 1.0f))));
    tb3 d = tb3_make(tb_sin(phi) * tb_cos(theta), tb_cos(phi), tb_sin(phi) * tb_sin(theta));
    pdf = (lobe + 1.0f) * tb_pow(tb_cos(phi), lobe) / (2.0f * PI);
    return reorient(d, n);
}

TBD float ggx_pdf(tb3 n, tb3 outgoing, tb3 h, float roughness) /* ImportanceSampleGGXPDF :1084-1094 */
{
    roughness = tb_max(MIN_ROUGHNESS, roughness);
    float a = roughness * roughness, a2 = a * a;
    float c = tb_abs(tb3_dot(n, h));
    float e = (a2 - 1.0f) * c * c + 1.0f;
    if (e <= 0.0f) return LARGE_NUMBER;
    float d = a2 / (PI * e * e);
    return d * tb_abs(tb3_dot(h, n)) / (4.0f * tb_abs(tb3_dot(outgoing, h)));
}


/* ---- per-path state ------------------------------------------------------------------------------ */
enum : uint32_t {
    ST_DONE = 0,      /* sample finished: L/weight are final, needs accumulate */
    ST_EXTEND = 1,    /* pending ray is the bounce ray */
    ST_SHADOW = 2,    /* pending ray is the NEE shadow feeler */
    ST_SSS = 3,       /* pending ray is a step of the interior random walk */
    ST_SCATTER = 4,   /* no pending ray: path_scatter() must run next */
    ST_WAIT = 5,      /* frame-group mode: the lane holds a sample number whose work item is not published yet */
    ST_PRIMARY = 6,   /* frame-group mode with the primary-visibility pre-pass: a path just begun, its closest hit waits in the sample's slot */
};
enum : uint32_t {
    F_SPECULAR = 1u << 0, F_PERFECT = 1u << 1, F_PREV_PERFECT = 1u << 2, F_EXITING = 1u << 3, F_NOSCATTER = 1u << 4,
    F_AOV_DEPTH = 1u << 5, F_AOV_EMISSIVE = 1u << 6, F_HEATMAP = 1u << 7,
};

/* Fields that a feature set never touches are dead after inlining and cost no registers. */
struct Path {
    tb3 ro, rd;            /* pending ray */
    tb3 T, L;              /* accumulatedIndirectLightMultiplier, accumulatedColor */
    float seed, weight;
    uint32_t state, flags;
    int bounce;
    /* live from path_on_closest to path_scatter */
    tb3 N, Nd, prevDir, nextOrigin;
    tb3 albedo; float roughness, specCoef, curIOR, newIOR, nDotD; int matFlags;
    tb3 absorption, scattering;
    tb3 contrib;           /* NEE contribution waiting for visibility: ((T*albedo)*lm)*lightColor */
    /* SSS walk */
    int sssStep; float maxTravel;
    /* AOV side channel (first hit; FEAT_EXT only) */
    tb3 aovNormal, aovAlbedo, aovEmissive, aovWorldPos; float aovNeighbor, aovDepth;
    tb3 neighborDir;
    uint32_t lastBoxes, lastTris;
    uint32_t nMat, nLight; /* GetMaterial / GetOneLightSample calls of the current sample (byte model) */
};

/* kernel.glsl:1786-1798 with the identity view matrix (GetRotationFactor() == 0.5) */
TBD tb3 lens_position(const TbPerFrameConstants& pf, float lensHeight, float u, float v, float aspect)
{
    tb3 p = ld3(pf.CameraPosition);
    float lensWidth = lensHeight * aspect;
    p = p + ld3(pf.CameraRight) * (u * 2.0f - 1.0f) * lensWidth / 2.0f;
    p = p + ld3(pf.CameraUp) * (v * 2.0f - 1.0f) * lensHeight / 2.0f;
    return p;
}

TBD float gaussian(float x, float mu, float sigma)
{
    float d = x - mu;
    return 1.0f / tb_sqrt(2.0f * PI * sigma * sigma) * tb_exp(-tb_pow(d, 2.0f) / (2.0f * sigma * sigma));
}

TBD float halton(int b, int i) /* RayGenCommon.h:49-59 */
{
    float r = 0.0f, f = 1.0f;
    while (i > 0) { f = f / (float)b; r = r + f * (float)(i % b); i = (int)tb_floor((float)i / (float)b); }
    return r;
}

/* GetBlueNoise, RayGenCommon.h:104-122: 8 R, or 2 texel fetches + Halton23.  `need`: bit i set = out[i] is read by the caller;
 * a rand() whose value nobody reads only moves the seed on (the compiler does not manage to drop all of an unused sin by
 * itself: the range reduction and the quadrant test of five of them stayed in the regeneration path) */
template <uint32_t F>
TBD void blue_noise(const TbDeviceScene& ds, const TbPerFrameConstants& pf, uint32_t frame, float& seed, uint32_t x, uint32_t y, float out[8], uint32_t need)
{
    if (!pf.UseBlueNoise) { /* a uniform branch in every variant: blue noise is the reference's default (TracerBoy.h:354) */
        for (int i = 0; i < 8; i++) { if ((need >> i) & 1u) out[i] = rnd(seed, pf.Time); else { out[i] = 0.0f; seed = seed + 1.0f; } }
    } else {
        uint32_t idx = (y % 256u) * 256u + (x % 256u);
        TbFloat4 z = {0, 0, 0, 0};
        TbFloat4 n0 = ds.blueNoise0 ? ds.blueNoise0[idx] : z, n1 = ds.blueNoise1 ? ds.blueNoise1[idx] : z;
        float h2 = halton(2, (int)frame), h3 = halton(3, (int)frame);
        out[0] = tb_frac(n0.x + h2); out[1] = tb_frac(n0.y + h3); out[2] = tb_frac(n0.z + h2); out[3] = tb_frac(n0.w + h3);
        out[4] = tb_frac(n1.x + h2); out[5] = tb_frac(n1.y + h3); out[6] = tb_frac(n1.z + h2); out[7] = tb_frac(n1.w + h3);
    }
}

/* SoftwareRayTraceCS.hlsl:38-39 + RayGenCommon.h:693-703 + kernel.glsl:1805-1906,1280-1284.
 * `frame` is PerFrameConstants.GlobalFrameCount of this sample (a lane may be on a different frame
 * than its neighbours in the persistent kernel, so it is not read from pf). */
template <uint32_t F>
TBD void path_begin(Path& p, const TbDeviceScene& ds, const TbPerFrameConstants& pf, uint32_t frame, uint32_t W, uint32_t H, uint32_t x, uint32_t y,
                    const TbDeviceTargets* cam = nullptr /* TbDeviceTargets::camPre: the launch's camera constants, from the host */)
{
    p.flags = 0; p.bounce = 0; p.lastBoxes = p.lastTris = 0; p.nMat = p.nLight = 0;
    p.aovNormal = p.aovAlbedo = p.aovEmissive = p.aovWorldPos = tb3_splat(0.0f); p.aovNeighbor = 0.0f; p.aovDepth = 0.0f;
    p.seed = hash13((float)x, (float)y, (float)frame);
    float resX = (float)W, resY = (float)H;
    float dux = ((float)x + 0.5f) / resX, duy = ((float)y + 0.5f) / resY;
    float uvx = 0.0f + dux * 1.0f, uvy = 1.0f + duy * -1.0f;
    float pcx = uvx * resX, pcy = uvy * resY;
    float bn[8];
    blue_noise<F>(ds, pf, frame, p.seed, x, y, bn, (F & FEAT_EXT) ? 0xc3u : 0x03u); /* kernel.glsl:1830: the pixel jitter, with FEAT_EXT also the DOF jitter */
    const bool pre = cam && cam->camPre;
    float psx = pre ? cam->camInvResX : 1.0f / resX, psy = pre ? cam->camInvResY : 1.0f / resY;
    float u = pcx * psx, v = pcy * psy;
    float jx = bn[0], jy = bn[1];
    if ((F & FEAT_EXT) && pf.FixedPixelOffset.x >= 0.0f) { jx = pf.FixedPixelOffset.x; jy = pf.FixedPixelOffset.y; }
    float offX = jx - 0.5f, offY = jy - 0.5f;
    float pixelRadius = pf.FilterWidth / 2.0f;
    float w = 1.0f;
    if (F & FEAT_EXT) {
        if (pf.FilterType == TB_FILTER_TYPE_TRIANGLE) w = tb_max(0.5f - tb_abs(offX), 0.5f - tb_abs(offY));
        else if (pf.FilterType == TB_FILTER_TYPE_GAUSSIAN) {
            float sigma = 0.8f;
            float eX = gaussian(1.0f, 0.0f, sigma), eY = gaussian(1.0f, 0.0f, sigma);
            w = tb_max(0.0f, gaussian(offX * 2.0f, 0.0f, sigma) - eX) * tb_max(0.0f, gaussian(offY * 2.0f, 0.0f, sigma) - eY);
        }
    }
    p.weight = w;
    u += offX * psx * (pixelRadius * 2.0f);
    v += offY * psy * (pixelRadius * 2.0f);
    float aspect = pre ? cam->camAspect : resX / resY;
    tb3 camPos = ld3(pf.CameraPosition);
    tb3 focal = pre ? ld3(cam->camFocal) : camPos - pf.FocalDistance * tb3_normalize(ld3(pf.CameraLookAt) - camPos);
    float lensHeight = ds.config.CameraLensHeight;
    tb3 lens = lens_position(pf, lensHeight, u, v, aspect);
    p.ro = focal; p.rd = tb3_normalize(lens - focal);
    if (F & FEAT_EXT) {
        tb3 nlens = lens_position(pf, lensHeight, u + psx, v + psy, aspect);
        p.neighborDir = tb3_normalize(nlens - focal);
        if (pf.DOFFocusDistance > 0.0f) { /* :1890-1901 */
            tb3 focus = tb3_madd(p.rd, pf.DOFFocusDistance, p.ro);
            float radius = tb_sqrt(bn[6]) * pf.DOFApertureWidth;
            float theta = bn[7] * 2.0f * PI;
            float fx = tb_cos(theta) * radius, fy = tb_sin(theta) * radius;
            p.ro = p.ro + (fx * ld3(pf.CameraRight) + fy * ld3(pf.CameraUp));
            p.rd = tb3_normalize(focus - p.ro);
        }
    }
    /* Trace() prologue :1280-1284 (second GetBlueNoise: 8 R whose values are unused) */
    p.L = tb3_splat(0.0f); p.T = tb3_splat(1.0f);
    float unused[8];
    blue_noise<F>(ds, pf, frame, p.seed, x, y, unused, 0u);
    p.state = pf.MaxBounces > 0 ? ST_EXTEND : ST_DONE;
}

/* Russian roulette at the top of bounce i >= 2 (kernel.glsl:1288-1302); also the loop bound */
TBD void path_pre_extend(Path& p, const TbPerFrameConstants& pf)
{
    if (p.bounce >= (int)pf.MaxBounces) { p.state = ST_DONE; return; }
    if (p.bounce >= 2) {
        float q = tb_max(tb_max(p.T.x, p.T.y), p.T.z);
        q = tb_max(q, EPSILON);
        if (q < rnd(p.seed, pf.Time)) { p.state = ST_DONE; return; }
        p.T = p.T * (1.0f / q);
    }
    p.state = ST_EXTEND;
}

/* GetOneLightSample, RayGenCommon.h:170-261 */
TBD TbLight fetch_light(const SceneRefs& sc, uint32_t i)
{
    TbLight l;
    if (i < sc.numLights) l = sc.lights[i].l; else memset(&l, 0, sizeof l);
    return l;
}
TBD tb3 random_barycentric(float& seed, float time)
{
    float u = rnd(seed, time);
    float v = rnd(seed, time);
    if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
    return tb3_make(u, v, 1.0f - u - v);
}
TBD float light_target_pdf(const TbLight& l, tb3 b, tb3 P)
{
    tb3 lp = ld3(l.P0) * b.x + ld3(l.P1) * b.y + ld3(l.P2) * b.z;
    float d = tb3_length(lp - P);
    float luma = tb3_dot(ld3(l.LightColor), tb3_make(0.212671f, 0.715160f, 0.072169f));
    return (l.SurfaceArea * luma) / d * d;
}

template <uint32_t F>
TBD void one_light_sample(const SceneRefs& sc, const TbPerFrameConstants& pf, float& seed, tb3 P, tb3& dir, tb3& color, float& pdf, tb3& nrm, float& atten)
{
    dir = color = nrm = tb3_splat(0.0f); atten = 0.0f; pdf = 0.0f;
    const uint32_t lightCount = pf.LightCount;
    if (!(lightCount > 0 && pf.EnableNextEventEstimation)) return;
    if ((F & FEAT_EXT) && pf.EnableSamplingImportanceResampling) { /* :180-211 */
        uint32_t sel = 0; tb3 selB = tb3_splat(0.0f); float wsum = 0.0f;
        for (uint32_t i = 0; i < 16; i++) {
            uint32_t li = (uint32_t)(rnd(seed, pf.Time) * (float)lightCount);
            TbLight l = fetch_light(sc, li);
            tb3 b = random_barycentric(seed, pf.Time);
            float target = light_target_pdf(l, b, P);
            float proposal = 1.0f / (float)lightCount;
            float w = target / (proposal * 16.0f);
            wsum += w;
            if (rnd(seed, pf.Time) < w / wsum) { sel = li; selB = b; }
        }
        TbLight l = fetch_light(sc, sel);
        float sir = light_target_pdf(l, selB, P) / wsum;
        pdf = sir / l.SurfaceArea;
        tb3 lp = ld3(l.P0) * selB.x + ld3(l.P1) * selB.y + ld3(l.P2) * selB.z;
        dir = lp - P;
        nrm = ld3(l.N0) * selB.x + ld3(l.N1) * selB.y + ld3(l.N2) * selB.z;
        color = ld3(l.LightColor);
        return;
    }
    uint32_t li = (uint32_t)(rnd(seed, pf.Time) * (float)lightCount);
    TbLight l = fetch_light(sc, li);
    tb3 b = random_barycentric(seed, pf.Time);
    if (l.LightType == TB_LIGHT_TYPE_AREA) {
        tb3 lp = ld3(l.P0) * b.x + ld3(l.P1) * b.y + ld3(l.P2) * b.z;
        dir = lp - P;
        nrm = ld3(l.N0) * b.x + ld3(l.N1) * b.y + ld3(l.N2) * b.z;
        float d = tb3_length(dir);
        atten = 1.0f / (d * d);
        dir = dir / d;
    } else if ((F & FEAT_EXT) && l.LightType == TB_LIGHT_TYPE_DIRECTIONAL) {
        dir = -ld3(l.Direction);
        if (pf.DebugValue > 0.0f) { dir.x = tb_sin(pf.DebugValue); dir.y = tb_sin(pf.DebugValue2); dir = tb3_normalize(dir); }
        nrm = -dir;
        atten = 1.0f;
    }
    color = ld3(l.LightColor);
    pdf = 1.0f / (float)lightCount;
    if (l.LightType == TB_LIGHT_TYPE_AREA) pdf /= l.SurfaceArea;
}

TBD tb3 detail_normal(const TbDeviceScene& ds, const TbPerFrameConstants& pf, const TbMaterial& m, tb3 n, tb3 t, float u, float v) /* RayGenCommon.h:273-295 */
{
    if (m.normalMapIndex != TB_INVALID_TEXTURE && pf.EnableNormalMaps) {
        tb3 bt = tb3_cross(t, n);
        F4 nm = texture_fetch(ds, m.normalMapIndex, u, v);
        float tx = (0.5f - nm.x) * 2.0f, ty = (0.5f - nm.y) * 2.0f;
        float tz = tb_sqrt(1.0f - (tx * tx + ty * ty));
        return tb3_normalize(t * tx + bt * ty + n * tb_max(tz, 0.02f));
    }
    return n;
}

/* After the closest-hit traversal of bounce p.bounce: kernel.glsl:1314-1455.
 * Leaves the path in ST_SHADOW (shadow feeler pending), ST_SCATTER or ST_DONE. */
template <uint32_t F>
TBD void path_on_closest(Path& p, const SceneRefs& sc, const TbDeviceScene& ds, const TbPerFrameConstants& pf, bool isHit, const Hit& h)
{
    const bool first = p.bounce == 0;
    if (p.T.x < EPSILON && p.T.y < EPSILON && p.T.z < EPSILON) { p.state = ST_DONE; return; } /* :1319-1326 */
    if (!isHit) { /* :1328-1343 */
        tb3 env = (F & FEAT_ENV) ? sample_environment(ds, p.rd) : tb3_splat(0.0f); /* black 1x1 texture when no map is bound */
        p.L = p.L + p.T * env;
        if ((F & FEAT_EXT) && first) { p.aovEmissive = p.L; p.flags |= F_AOV_EMISSIVE; }
        p.state = ST_DONE; return;
    }
    Surface s;
    /* (scenes without textures served by a feature set that has them: TbDeviceScene::textureUse) */
    const bool needUV = (F & FEAT_TEXTURES) && (ds.textureUse & 3u) != 0u;
    fetch_surface(sc, h, s, (F & FEAT_TEXTURES) && pf.EnableNormalMaps != 0 && (ds.textureUse & 2u) != 0u, needUV);
    if (!needUV) s.u = 0.0f;
    tb3 RayPoint = tb3_madd(p.rd, h.t, p.ro);
    p.nextOrigin = RayPoint + s.normal * EPSILON; /* :1353, unflipped normal */
    float nDotD = tb3_dot(s.normal, p.rd);
    const bool back = nDotD > 0.0f;
    TbMaterial m = get_material<F>(sc, ds, p.seed, pf.Time, s.material, s.u, s.v, back);
    p.nMat++;
    /* detail_normal's own condition */
    tb3 Nd = (F & FEAT_TEXTURES) ? detail_normal(ds, pf, m, s.normal, s.tangent, s.u, s.v) : s.normal;
    if ((F & FEAT_EXT) && first) { /* :1365-1376 */
        tb3 camPos = ld3(pf.CameraPosition);
        tb3 focal = camPos - pf.FocalDistance * tb3_normalize(ld3(pf.CameraLookAt) - camPos); /* neighbour ray origin, kernel.glsl:1887 */
        tb3 npt = tb3_madd(p.neighborDir, h.t, focal);
        p.aovWorldPos = p.aovWorldPos + RayPoint;
        p.aovNeighbor += tb3_length(npt - RayPoint);
        p.aovNormal = Nd;
        p.aovDepth = tb_saturate(h.t / pf.MaxZ); p.flags |= F_AOV_DEPTH;
        if (pf.OutputMode == TB_OUTPUT_TYPE_HEATMAP) { p.flags |= F_HEATMAP; p.state = ST_DONE; return; }
    }
    tb3 N = s.normal;
    if (F & FEAT_SSS) { p.curIOR = back ? m.IOR : AIR_IOR; p.newIOR = back ? AIR_IOR : m.IOR; }
    if (back) { N = -N; nDotD = -nDotD; Nd = -Nd; }
    bool spec = false;
    if ((F & FEAT_SPECULAR) && (m.Flags & TB_MAT_NO_SPECULAR) == 0) { /* :1400-1417 */
        if ((m.Flags & TB_MAT_METALLIC) != 0 || (m.Flags & TB_MAT_HAIR) != 0) spec = true;
        else spec = rnd(p.seed, pf.Time) < 0.5f;
    }
    const bool perfect = spec && m.roughness < 0.05f;
    const bool isLight = (m.Flags & TB_MAT_LIGHT) != 0;
    if ((p.flags & F_PREV_PERFECT) || first || !isLight || !pf.EnableNextEventEstimation) p.L = p.L + p.T * ld3(m.emissive); /* :1425-1428 */
    if (isLight) { p.state = ST_DONE; return; } /* :1430-1433 */

    p.N = N; p.Nd = Nd; p.nDotD = nDotD; p.prevDir = p.rd;
    p.albedo = ld3(m.albedo); p.roughness = m.roughness; p.specCoef = m.SpecularCoef; p.matFlags = m.Flags;
    if (F & FEAT_SSS) { p.absorption = ld3(m.absorption); p.scattering = ld3(m.scattering); }
    if ((F & FEAT_EXT) && first) p.aovEmissive = ld3(m.emissive); /* value written at :1722 if the bounce completes */
    p.flags = (p.flags & ~(F_SPECULAR | F_PERFECT)) | (spec ? F_SPECULAR : 0u) | (perfect ? F_PERFECT : 0u);

    float lpdf, latten; tb3 ldir, lcol, lnrm;
    one_light_sample<F>(sc, pf, p.seed, RayPoint, ldir, lcol, lpdf, lnrm, latten); /* :1437 */
    if (pf.LightCount > 0 && pf.EnableNextEventEstimation) p.nLight++;
    if (!perfect && lpdf > EPSILON && tb3_dot(ldir, lnrm) < 0.0f) { /* :1440-1455 */
        float lm = latten * diffuse_brdf(ldir, Nd) * tb_abs(tb3_dot(lnrm, ldir)) / lpdf;
        /* :1514-1515 is (((T*albedo)*lm)*Shadow)*lightColor with Shadow in {0,1}: x*1 == x, so the lit
         * value is pend*lightColor; the shadowed value is taken as contrib*0 (see path_on_shadow) */
        p.contrib = (p.T * p.albedo * lm) * lcol;
        p.ro = RayPoint + N * EPSILON; p.rd = ldir;
        p.state = ST_SHADOW;
        return;
    }
    p.state = ST_SCATTER;
}

/* After the shadow feeler: kernel.glsl:1460-1516.  p.rd is still the light direction. */
template <uint32_t F>
TBD void path_on_shadow(Path& p, const SceneRefs& sc, const TbDeviceScene& ds, const TbPerFrameConstants& pf, bool isHit, const Hit& h)
{
    /* GetMaterial of the blocker, of which only two things reach the result: the mix coin it may flip (one R, iff the record has
     * MIX_MATERIAL_FLAG; a mix material takes no texture overrides, RayGenCommon.h:298-341) and the LIGHT flag of the record it ends with.
     * Hit attributes, back-face test, overrides and SSS coefficients of the full function are not computed: two or three small loads
     * instead of the index triple, three vertices and one or two 84-B records. */
    bool lit = true;
    if (isHit) {
        p.nMat++;
        const uint32_t mi = h.geom < sc.numHitGroups ? sc.hitGroups[h.geom].MaterialIndex : 0u;
        int flags = mi < sc.numMaterials ? sc.materials[mi].m.Flags : 0;
        if ((F & FEAT_MIX) && (flags & TB_MAT_MIX) != 0) { /* 1 R */
            const TbFloat3 mix = sc.materials[mi].m.albedo; /* (index A, index B, weight): mi is in range, its flags were just read */
            const uint32_t pick = rnd(p.seed, pf.Time) < mix.z ? (uint32_t)mix.x : (uint32_t)mix.y;
            flags = pick < sc.numMaterials ? sc.materials[pick].m.Flags : 0;
        }
        lit = (flags & TB_MAT_LIGHT) != 0;
    }
    (void)ds;
    /* shadowed: (pend*0)*lightColor == +-0 for finite operands and NaN otherwise; contrib*0 has the same
     * value except when pend*lightColor overflows fp32, which needs radiance ~1e38 (parity notes) */
    p.L = p.L + (lit ? p.contrib : p.contrib * 0.0f);
    p.state = ST_SCATTER;
}

/* The same visibility test for kernels that scatter BEFORE they trace the feeler (pt_persistent.inc, feature sets without mix
 * materials): lightDir is the feeler's direction, the path's own ray already is the next bounce.  Touches nothing but L. */
template <uint32_t F>
TBD void path_apply_shadow(Path& p, const SceneRefs& sc, const TbDeviceScene& ds, const TbPerFrameConstants& pf, bool isHit, const Hit& h, tb3 lightDir,
    tb3 contrib)
{
    static_assert(!(F & FEAT_MIX), "GetMaterial of the blocker draws a random number for mix materials: the feeler must be judged before the scatter");
    /* Without mix materials GetMaterial of the blocker draws no random number, and of everything it computes -- hit attributes, the back-face
     * test, texture overrides, the SSS coefficients -- only the LIGHT flag of the record is read here (an override can add METALLIC, nothing
     * else): two dword loads (hit group -> material index -> flags) instead of the index triple, three vertices and the 84-B record.
     * nMat still counts the reference's GetMaterial call (byte model). */
    bool lit = true;
    if (isHit) {
        p.nMat++;
        lit = shadow_hit_is_light(sc, h.geom);
    }
    (void)ds; (void)pf; (void)lightDir;
    p.L = p.L + (lit ? contrib : contrib * 0.0f);
}

TBD void finish_bounce(Path& p, const TbPerFrameConstants& pf)
{
    p.bounce++;
    path_pre_extend(p, pf);
}

/* Refraction / reflection at an SSS boundary; shared by entry (:1531-1563) and exit (:1645-1678).
 * Returns false when the path must stop the enclosing loop (`break`). */
TBD bool refract_or_reflect(Path& p, const TbPerFrameConstants& pf, tb3 normal, float nDotD, float nr, bool perfect, bool& reflected)
{
    reflected = false;
    float disc = 1.0f - nr * nr * (1.0f - nDotD * nDotD);
    if (disc > EPSILON) {
        tb3 refr = tb3_normalize(nr * (p.rd - normal * nDotD) - normal * tb_sqrt(disc));
        if (perfect) { p.rd = refr; p.flags |= F_PREV_PERFECT; }
        else {
            float pdf;
            float r0 = rnd(p.seed, pf.Time); float r1 = rnd(p.seed, pf.Time);
            p.rd = lobe_direction(refr, p.roughness, r0, r1, pdf);
            if (pdf < EPSILON) {
                r0 = rnd(p.seed, pf.Time); r1 = rnd(p.seed, pf.Time);
                p.rd = lobe_direction(refr, p.roughness, r0, r1, pdf);
                if (pdf < EPSILON) return false;
            }
        }
    } else {
        p.rd = tb3_reflect(p.rd, normal);
        reflected = true;
    }
    return true;
}

/* kernel.glsl:1519-1772: next direction + throughput.  Sets up the next pending ray. */
template <uint32_t F>
TBD void path_scatter(Path& p, const TbPerFrameConstants& pf)
{
    const bool first = p.bounce == 0;
    const bool spec = (F & FEAT_SPECULAR) && (p.flags & F_SPECULAR) != 0, perfect = (F & FEAT_SPECULAR) && (p.flags & F_PERFECT) != 0;
    const bool allowsSpec = (F & FEAT_SPECULAR) && (p.matFlags & TB_MAT_NO_SPECULAR) == 0,
        metallic = (F & FEAT_SPECULAR) && (p.matFlags & TB_MAT_METALLIC) != 0;
    p.rd = p.prevDir; p.ro = p.nextOrigin;
    p.flags = (p.flags & ~F_PREV_PERFECT) | (perfect ? F_PREV_PERFECT : 0u); /* :1520 */
    const bool interior = !spec && (F & FEAT_SSS) && (p.matFlags & TB_MAT_SUBSURFACE_SCATTER) != 0;
    if (!interior) {
        /* GGX reflection (:1521-1526, ImportanceSampleGGX :1066-1082) and the cosine-weighted hemisphere (:1695, :1025-1041) draw the same two
         * random numbers in the same order, turn the second into the same azimuth and end in the same ReorientVectorAroundNormal about N:
         * a plastic surface flips a coin between them, so a wave always holds both.  The lane's polar part (A = sin, B = cos of the polar
         * angle) is computed on its own side; the azimuth's sine and cosine, the three products and the reorientation once for the wave. */
        const float r0 = rnd(p.seed, pf.Time); const float r1 = rnd(p.seed, pf.Time);
        const float theta = (2.0f * PI) * r1;
        float A, B;
        if (spec) {
            const float roughness = tb_max(MIN_ROUGHNESS, p.roughness);
            const float a = roughness * roughness, a2 = a * a;
            const float phi = tb_acos(tb_sqrt((1.0f - r0) / ((a2 - 1.0f) * r0 + 1.0f)));
            A = tb_sin(phi); B = tb_cos(phi);
        } else {
            A = tb_sqrt(r0); B = tb_sqrt(tb_max(EPSILON, 1.0f - r0));
        }
        const tb3 w = reorient(tb3_make(A * tb_cos(theta), B, A * tb_sin(theta)), p.N);
        p.rd = spec ? tb3_reflect(p.prevDir, w) : w;
    } else { /* :1529-1600 */
        bool reflected;
        if (!refract_or_reflect(p, pf, p.N, p.nDotD, p.curIOR / p.newIOR, perfect, reflected)) { p.state = ST_DONE; return; } /* :1552 */
        bool noScatter = p.scattering.x < EPSILON;
        float perScatter = 1.0f / ((p.scattering.x + p.scattering.y + p.scattering.z) / 3.0f);
        p.maxTravel = noScatter ? LARGE_NUMBER : perScatter;
        bool exiting = (p.matFlags & TB_MAT_SINGLE_SIDED) != 0;
        p.flags = (p.flags & ~(F_EXITING | F_NOSCATTER)) | (exiting ? F_EXITING : 0u) | (noScatter ? F_NOSCATTER : 0u);
        p.sssStep = 0;
        if (exiting) { finish_bounce(p, pf); return; } /* loop body never runs, then `continue` :1690 */
        p.state = ST_SSS; /* first walk step's ray is (p.ro, p.rd) */
        return;
    }
    /* The lanes of a wave fall on all three sides of the reference's branches here (matte, plastic and metal share a scene), and a branching
     * form issues every side's divisions and pow()s for the whole wave.  What the sides have in common is therefore computed once, on operands
     * selected per lane: the half vector's normalisation, the three divisions of T by the pdf, the GGX term.  Each lane's operations and their
     * order are those of its own branch (kernel.glsl:1699-1769). */
    float diffusePdf = tb3_dot(p.rd, p.N) / PI; /* :1699 */
    const bool anySpec = allowsSpec || metallic;
    tb3 hvN = tb3_splat(0.0f), hvSafe = tb3_splat(0.0f);
    if (anySpec) {
        const tb3 toEye = -p.prevDir;
        hvN = tb3_normalize(toEye + p.rd);                                            /* the metallic branch's half vector, :1735 */
        hvSafe = tb3_dot(toEye, p.rd) > (-1.0f + EPSILON) ? hvN : p.N;               /* GetHalfVectorSafe :1258-1269 */
    }
    float pdf = diffusePdf;
    if (allowsSpec) {
        float specPdf = ggx_pdf(p.N, p.rd, hvSafe, p.roughness);
        pdf = metallic ? specPdf : tb_lerp(specPdf, diffusePdf, 0.5f);
    }
    p.T = p.T / pdf;
    if ((F & FEAT_EXT) && first) p.flags |= F_AOV_EMISSIVE; /* :1720-1723 (value stored in path_on_closest) */
    tb3 albedo = ((F & FEAT_EXT) && pf.IsRealTime && first) ? tb3_splat(1.0f) : p.albedo;
    if (anySpec) {
        const tb3 hv = metallic ? hvN : hvSafe;
        float r2 = tb_max(p.roughness * p.roughness, MIN_ROUGHNESS_SQUARED);
        float specular = ggx_ndf(p.Nd, hv, r2) / (4.0f * tb_abs(tb3_dot(-p.prevDir, hv)) * tb_max(tb_abs(tb3_dot(-p.prevDir, p.N)), tb_abs(tb3_dot(p.rd,
            p.N))));
        if (metallic) { /* :1734-1741 */
            p.T = p.T * (specular * albedo * tb_saturate(tb3_dot(p.rd, p.N)));
        } else { /* :1744-1765 */
            float fresnel = p.specCoef + (1.0f - p.specCoef) * tb_pow(tb_abs(1.0f - tb3_dot(-p.prevDir, hv)), 5.0f);
            float dm = (float)(28.0 / (23.0 * 3.1415926535)) * (1.0f - p.specCoef)
                * (1.0f - tb_pow(1.0f - 0.5f * tb3_dot(-p.prevDir, p.N), 5.0f))
                * (1.0f - tb_pow(1.0f - 0.5f * tb3_dot(p.rd, p.N), 5.0f));
            tb3 diffuse = albedo * dm;
            tb3 mult = (diffuse + tb3_splat(fresnel * specular)) * tb_saturate(tb3_dot(p.rd, p.N));
            p.T = p.T * mult;
        }
    } else { /* :1766-1769 */
        p.T = p.T * (albedo * diffuse_brdf(p.rd, p.Nd));
    }
    if ((F & FEAT_EXT) && first) p.aovAlbedo = p.albedo; /* :1771 */
    finish_bounce(p, pf);
}

/* One step of the interior random walk after its traversal: kernel.glsl:1601-1687.
 * The step's travelDistance R is drawn BEFORE the traversal in the reference (:1603); callers
 * draw it with sss_travel() when they issue the ray. */
TBD float sss_travel(Path& p, const TbPerFrameConstants& pf) { return tb_max(-tb_log(rnd(p.seed, pf.Time)), 0.1f) * p.maxTravel; }

TBD void path_on_sss(Path& p, const SceneRefs& sc, const TbPerFrameConstants& pf, bool isHit, const Hit& h, float travel)
{
    const bool perfect = (p.flags & F_PERFECT) != 0, noScatter = (p.flags & F_NOSCATTER) != 0;
    if (!isHit) { p.T = tb3_splat(0.0f); finish_bounce(p, pf); return; } /* :1610-1615 then :1690 */
    Surface s;
    fetch_surface(sc, h, s, false);
    tb3 normal = s.normal;
    float t = tb_min(travel, h.t);
    bool exiting = t < travel || noScatter;
    bool lastRay = p.sssStep == MAX_SSS_BOUNCES - 1;
    if (lastRay && !exiting) p.T = tb3_splat(0.0f);
    tb3 RayPoint = tb3_madd(p.rd, t, p.ro);
    p.ro = RayPoint + normal * EPSILON;
    p.T = p.T * tb3_make(tb_exp(-t * p.absorption.x), tb_exp(-t * p.absorption.y), tb_exp(-t * p.absorption.z));
    bool stop = false;
    if (exiting) {
        float nDotD = tb3_dot(normal, p.rd);
        if (nDotD >= 0.0f) { normal = -normal; nDotD = -nDotD; }
        bool reflected;
        if (!refract_or_reflect(p, pf, normal, nDotD, p.newIOR / p.curIOR, perfect, reflected)) stop = true; /* :1666 leaves the walk */
        else if (reflected) exiting = false;
    } else {
        float u1 = rnd(p.seed, pf.Time); float u2 = rnd(p.seed, pf.Time); /* GenerateRandomDirection :991-999 */
        float r = tb_sqrt(1.0f - u1 * u1);
        float phi = 6.28f * u2;
        p.rd = tb3_make(tb_cos(phi) * r, tb_sin(phi) * r, u1);
        p.T = p.T / 1.0f;
    }
    p.sssStep++;
    if (stop || exiting || p.sssStep >= MAX_SSS_BOUNCES) { finish_bounce(p, pf); return; } /* `continue` :1690 */
    p.state = ST_SSS;
}

} // namespace pt
