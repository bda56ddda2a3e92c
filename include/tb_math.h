/* tb_math.h -- the fp32 arithmetic contract of the path-tracing hot path.
 *
 * Why this file exists: the reference RNG is `fract(sin(seed++ + Time) * 43758.5453123)`
 * (/root/reference/TracerBoy/kernel.glsl:39-40).  One ulp of difference in sin() becomes a
 * different random number and flips `rand() < p` branches, so "identical RNG seeds" is only
 * meaningful if every transcendental is the SAME sequence of correctly rounded IEEE-754 binary32
 * operations on the host (oracle, g++) and on the device (HIP, gfx950).  This header pins that
 * sequence.  Rules:
 *   - only +,-,*,/ , sqrt, fma (explicit), floor/rint, comparisons and bit casts: all are
 *     correctly rounded / exact on x86-64 and on gfx950 (HIP's default
 *     -fhip-fp32-correctly-rounded-divide-sqrt) -- never a libm call, never a hardware approx op;
 *   - every translation unit that includes it is compiled with -ffp-contract=off; fused
 *     multiply-adds appear only where written as tb_fma();
 *   - min/max/abs are written as compare+select so NaN and signed-zero behaviour is defined
 *     (HLSL min/max return the non-NaN operand; so do these).
 *
 * HLSL lowering mirrored here (DXC emits exp2/log2 based forms):
 *   exp(x) = exp2(x * log2(e)),  log(x) = log2(x) * ln(2),  pow(x,y) = exp2(y * log2(x)),
 *   frac(x) = x - floor(x),  rcp(x) = 1/x,  normalize(v) = v * (1/sqrt(dot(v,v))).
 * Polynomials follow the public Cephes single-precision library (Moshier), which is what the
 * accuracy tests in tests/test_math.py check against double-precision libm.
 */
#ifndef TB_MATH_H
#define TB_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define TB_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define TB_HD inline
#endif

/* clang (hipcc) honours the pragma; g++ has no equivalent that keeps inlining, so the build
 * passes -ffp-contract=off on the command line (oracle/Makefile, tracerboy_amd/build.py) and
 * tests/test_math.py checks host-vs-device bit equality on the GPU. */
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define TB_PI_F 3.1415926535f /* kernel.glsl:3 `#define PI 3.1415926535` rounded to binary32 */

/* ---- bit casts ------------------------------------------------------------------------- */
TB_HD uint32_t tb_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
TB_HD float tb_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---- exact / correctly rounded primitives ----------------------------------------------- */
TB_HD float tb_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
TB_HD float tb_sqrt(float x) { return __builtin_sqrtf(x); }
TB_HD float tb_rcp(float x) { return 1.0f / x; }
TB_HD float tb_floor(float x) { return __builtin_floorf(x); }
TB_HD float tb_abs(float x) { return tb_u2f(tb_f2u(x) & 0x7fffffffu); }
/* min/max with the semantics of gfx950's v_min_f32 / v_max_f32 (IEEE minNum/maxNum as HLSL's min/max:
 * the non-NaN operand wins; -0 orders below +0).  The device uses the native instruction (and lets the
 * compiler form v_min3/v_max3); the host spells the same function out. */
TB_HD float tb_min(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fminf(a, b);
#else
    if (a < b) return a;
    if (b < a) return b;
    if (a != a) return b;
    if (b != b) return a;
    return tb_u2f(tb_f2u(a) | tb_f2u(b)); /* equal: -0 if either is -0 */
#endif
}
TB_HD float tb_max(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaxf(a, b);
#else
    if (a > b) return a;
    if (b > a) return b;
    if (a != a) return b;
    if (b != b) return a;
    return tb_u2f(tb_f2u(a) & tb_f2u(b)); /* equal: +0 if either is +0 */
#endif
}
TB_HD float tb_clamp(float x, float lo, float hi) { return tb_min(tb_max(x, lo), hi); }
TB_HD float tb_saturate(float x) { return tb_clamp(x, 0.0f, 1.0f); }
TB_HD float tb_frac(float x) { return x - tb_floor(x); }
TB_HD bool tb_isnan(float x) { return x != x; }

/* ---- sin / cos --------------------------------------------------------------------------
 * Range reduction in binary64 (two-term pi/2, explicit fma): exact on host and device and good
 * for |x| up to ~1e9, far beyond the seed values the RNG reaches.  Kernel polynomials on
 * [-pi/4, pi/4] are Cephes sinf/cosf. */
TB_HD void tb_sincos_reduce(float x, float* r, int* quadrant)
{
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632679489655800e+00;
    const double pio2_lo = 6.12323399573676603587e-17;
    double xd = (double)x;
    double k = __builtin_rint(xd * two_over_pi);
    double rd = __builtin_fma(-k, pio2_hi, xd);
    rd = __builtin_fma(-k, pio2_lo, rd);
    *r = (float)rd;
    /* k is integral and |k| <= 1e9 * 2 / pi < 2^31 (callers reject |x| >= 1e9): k mod 4 is the low two bits of the two's
     * complement int (three binary64 operations fewer per sin / cos than k - 4 floor(k / 4); same value) */
    *quadrant = (int)k & 3;
}

TB_HD float tb_sin_poly(float r)
{
    float z = r * r;
    float p = tb_fma(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = tb_fma(p, z, -1.6666654611e-1f);
    return tb_fma(p * z, r, r);
}

TB_HD float tb_cos_poly(float r)
{
    float z = r * r;
    float p = tb_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = tb_fma(p, z, 4.166664568298827e-2f);
    return tb_fma(p * z, z, tb_fma(-0.5f, z, 1.0f));
}

/* The guard against inf, nan and absurdly large arguments stays a BRANCH here, unlike the selects of exp2 / log2 / asin / acos / atan below:
 * every rand() goes through tb_sin, the cornell-box kernel is VALU-bound, and the select form's two extra vector instructions per call
 * cost it 0.6 % where the branch's exec-mask bookkeeping is scalar work that kernel has to spare (same-box A/B, profiles/r5/ab_rounds.json). */
TB_HD float tb_sin(float x)
{
    if (!(tb_abs(x) < 1.0e9f)) return tb_u2f(0x7fc00000u); /* inf, nan, absurdly large */
    float r; int q;
    tb_sincos_reduce(x, &r, &q);
    const float sp = tb_sin_poly(r), cp = tb_cos_poly(r); /* both, then a select: the quadrant differs from lane to lane */
    float s = (q & 1) ? cp : sp;
    return (q & 2) ? -s : s;
}

TB_HD float tb_cos(float x)
{
    if (!(tb_abs(x) < 1.0e9f)) return tb_u2f(0x7fc00000u);
    float r; int q;
    tb_sincos_reduce(x, &r, &q);
    const float sp = tb_sin_poly(r), cp = tb_cos_poly(r);
    float c = (q & 1) ? sp : cp;
    return ((q + 1) & 2) ? -c : c;
}

/* ---- asin / acos (Cephes asinf/acosf) ---------------------------------------------------- */
TB_HD float tb_asin_core(float a) /* 0 <= a <= 0.5 */
{
    float z = a * a;
    float p = tb_fma(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = tb_fma(p, z, 4.5470025998e-2f);
    p = tb_fma(p, z, 7.4953002686e-2f);
    p = tb_fma(p, z, 1.6666752422e-1f);
    return tb_fma(p * z, a, a);
}

/* select form (see tb_exp2): one evaluation of the core polynomial on the argument the case asks for, the case's affine map chosen
 * afterwards; each case's operations are those of the branching form */
TB_HD float tb_asin(float x)
{
    const float a = tb_abs(x);
    const bool ok = a <= 1.0f, hi = a > 0.5f;
    const float s = tb_sqrt(0.5f * (1.0f - a));
    const float c = tb_asin_core(hi ? s : a);
    float r = hi ? 1.5707963267948966f - 2.0f * c : c;
    r = (x < 0.0f) ? -r : r;
    return ok ? r : tb_u2f(0x7fc00000u);
}

TB_HD float tb_acos(float x)
{
    const float a = tb_abs(x);
    const bool ok = a <= 1.0f, hi = a > 0.5f;
    const float s = tb_sqrt(0.5f * (1.0f - a)); /* 1 - x for x > 0.5, 1 + x for x < -0.5: the same number */
    const float c = tb_asin_core(hi ? s : a);
    const float mid = 1.5707963267948966f - ((x < 0.0f) ? -c : c);                 /* |x| <= 0.5: pi/2 - asin(x) */
    const float out = (x > 0.5f) ? 2.0f * c : 3.14159265358979323846f - 2.0f * c;  /* x > 0.5 | x < -0.5 */
    const float r = hi ? out : mid;
    return ok ? r : tb_u2f(0x7fc00000u);
}

/* ---- atan / atan2 (Cephes atanf) --------------------------------------------------------- */
TB_HD float tb_atan(float xx)
{
    float x = tb_abs(xx);
    /* the three ranges as selects of ONE division's operands: -(1 / x) = (-1) / x and x = x / 1 exactly */
    const bool big = x > 2.414213562373095f, mid = x > 0.4142135623730950f;
    float y = big ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.0f);
    x = (big ? -1.0f : (mid ? x - 1.0f : x)) / (big ? x : (mid ? x + 1.0f : 1.0f));
    float z = x * x;
    float p = tb_fma(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = tb_fma(p, z, 1.99777106478e-1f);
    p = tb_fma(p, z, -3.33329491539e-1f);
    y = y + tb_fma(p * z, x, x);
    return (xx < 0.0f) ? -y : y;
}

TB_HD float tb_atan2(float y, float x)
{
    const float pi = 3.14159265358979323846f;
    if (x != x || y != y) return tb_u2f(0x7fc00000u);
    if (x == 0.0f) {
        if (y > 0.0f) return 0.5f * pi;
        if (y < 0.0f) return -0.5f * pi;
        return 0.0f;
    }
    float a = tb_atan(y / x);
    if (x < 0.0f) return (y < 0.0f) ? a - pi : a + pi;
    return a;
}

/* ---- exp2 / log2 (Cephes exp2f / log2f) and the HLSL-style derived forms ------------------ */
/* Special arguments are handled by selects on the way in (the main path then sees a harmless value) and on the way out, not by
 * early returns: on the device every early return is an exec-mask save / restore and a branch around code that a neighbouring lane
 * needs anyway (pow = exp2(y * log2(x)) alone had eight of them).  Every argument's value is that of the branching form. */
TB_HD float tb_exp2(float x)
{
    const bool isNan = x != x, big = x >= 128.0f, tiny = x < -150.0f;
    const float xm = (isNan || big || tiny) ? 0.0f : x;
    float n = __builtin_rintf(xm);
    float r = xm - n; /* exact, |r| <= 0.5 */
    float p = tb_fma(1.535336188319500e-4f, r, 1.339887440266574e-3f);
    p = tb_fma(p, r, 9.618437357674640e-3f);
    p = tb_fma(p, r, 5.550332471162809e-2f);
    p = tb_fma(p, r, 2.402264791363012e-1f);
    p = tb_fma(p, r, 6.931472028550421e-1f);
    p = tb_fma(p, r, 1.0f);
    /* scale by 2^n in two steps so results in the denormal range round once, the same way on
     * host and device */
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1;
    float s1 = tb_u2f((uint32_t)(n1 + 127) << 23);
    float s2 = tb_u2f((uint32_t)(n2 + 127) << 23);
    float v = (p * s1) * s2;
    v = tiny ? 0.0f : v;
    v = big ? tb_u2f(0x7f800000u) : v;
    return isNan ? x : v;
}

TB_HD float tb_log2(float x)
{
    const bool isNan = x != x, neg = x < 0.0f, zero = x == 0.0f, inf = x == tb_u2f(0x7f800000u);
    float xm = (isNan || neg || zero || inf) ? 1.0f : x;
    uint32_t u = tb_f2u(xm);
    const bool den = (u & 0x7f800000u) == 0; /* denormal: renormalise exactly */
    xm = den ? xm * 8388608.0f : xm;
    u = tb_f2u(xm);
    int e = den ? -23 : 0;
    e += (int)((u >> 23) & 0xff) - 126;
    float m = tb_u2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
    const bool low = m < 0.70710678118654752440f;
    e = low ? e - 1 : e;
    m = low ? m + m - 1.0f : m - 1.0f;
    float z = m * m;
    float p = tb_fma(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = tb_fma(p, m, 1.1676998740e-1f);
    p = tb_fma(p, m, -1.2420140846e-1f);
    p = tb_fma(p, m, 1.4249322787e-1f);
    p = tb_fma(p, m, -1.6668057665e-1f);
    p = tb_fma(p, m, 2.0000714765e-1f);
    p = tb_fma(p, m, -2.4999993993e-1f);
    p = tb_fma(p, m, 3.3333331174e-1f);
    float y = (p * z) * m;
    y = tb_fma(-0.5f, z, y); /* ln(1+m) - m */
    const float log2ea = 0.44269504088896340735992f; /* log2(e) - 1 */
    float r = y * log2ea;
    r = tb_fma(m, log2ea, r);
    r = r + y;
    r = r + m;
    float v = r + (float)e;
    v = inf ? x : v;
    v = zero ? tb_u2f(0xff800000u) : v;
    v = neg ? tb_u2f(0x7fc00000u) : v;
    return isNan ? x : v;
}

TB_HD float tb_exp(float x) { return tb_exp2(x * 1.4426950408889634f); }
TB_HD float tb_log(float x) { return tb_log2(x) * 0.6931471805599453f; }
/* HLSL pow: NaN for x < 0, exp2(y*log2(x)) otherwise (0^y>0 = 0, 0^y<0 = inf, 0^0 = NaN). */
TB_HD float tb_pow(float x, float y) { return tb_exp2(y * tb_log2(x)); }

#endif /* TB_MATH_H */
