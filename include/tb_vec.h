/* tb_vec.h -- float3 helpers with a pinned operation order (part of the arithmetic contract,
 * see tb_math.h).  HLSL leaves the association of dot()/normalize() to the driver; the build
 * fixes it: dot = (x*x' + y*y') + z*z', normalize = v * (1/sqrt(dot)), reflect = i - 2*n*dot(i,n),
 * lerp = a + t*(b-a). */
#ifndef TB_VEC_H
#define TB_VEC_H

#include "tb_math.h"

struct tb3 {
    float x, y, z;
};

TB_HD tb3 tb3_make(float x, float y, float z) { tb3 r; r.x = x; r.y = y; r.z = z; return r; }
TB_HD tb3 tb3_splat(float s) { return tb3_make(s, s, s); }
TB_HD tb3 operator+(tb3 a, tb3 b) { return tb3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
TB_HD tb3 operator-(tb3 a, tb3 b) { return tb3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
TB_HD tb3 operator*(tb3 a, tb3 b) { return tb3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
TB_HD tb3 operator/(tb3 a, tb3 b) { return tb3_make(a.x / b.x, a.y / b.y, a.z / b.z); }
TB_HD tb3 operator*(tb3 a, float s) { return tb3_make(a.x * s, a.y * s, a.z * s); }
TB_HD tb3 operator*(float s, tb3 a) { return tb3_make(s * a.x, s * a.y, s * a.z); }
TB_HD tb3 operator/(tb3 a, float s) { return tb3_make(a.x / s, a.y / s, a.z / s); }
TB_HD tb3 operator-(tb3 a) { return tb3_make(-a.x, -a.y, -a.z); }
/* HLSL leaves contraction to the driver compiler except where `precise` is written; the build pins it:
 * dot products, a*s+b and the barycentric blend are fused-multiply-add chains (one rounding per step,
 * identical on host and device), everything else is unfused. */
TB_HD float tb3_dot(tb3 a, tb3 b) { return tb_fma(a.z, b.z, tb_fma(a.y, b.y, a.x * b.x)); }
TB_HD tb3 tb3_madd(tb3 a, float s, tb3 b) { tb3 r; r.x = tb_fma(a.x, s, b.x); r.y = tb_fma(a.y, s, b.y); r.z = tb_fma(a.z, s, b.z); return r; } /* a*s + b */
TB_HD tb3 tb3_bary(float b0, float b1, float b2, tb3 v0, tb3 v1, tb3 v2)
{
    tb3 r;
    r.x = tb_fma(b2, v2.x, tb_fma(b1, v1.x, b0 * v0.x));
    r.y = tb_fma(b2, v2.y, tb_fma(b1, v1.y, b0 * v0.y));
    r.z = tb_fma(b2, v2.z, tb_fma(b1, v1.z, b0 * v0.z));
    return r;
}
TB_HD tb3 tb3_cross(tb3 a, tb3 b)
{
    return tb3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
TB_HD float tb3_length(tb3 a) { return tb_sqrt(tb3_dot(a, a)); }
TB_HD tb3 tb3_normalize(tb3 a) { float inv = 1.0f / tb_sqrt(tb3_dot(a, a)); return a * inv; }
TB_HD tb3 tb3_reflect(tb3 i, tb3 n) { float d = tb3_dot(i, n); return i - n * (2.0f * d); }
TB_HD tb3 tb3_abs(tb3 a) { return tb3_make(tb_abs(a.x), tb_abs(a.y), tb_abs(a.z)); }
TB_HD tb3 tb3_min(tb3 a, tb3 b) { return tb3_make(tb_min(a.x, b.x), tb_min(a.y, b.y), tb_min(a.z, b.z)); }
TB_HD tb3 tb3_max(tb3 a, tb3 b) { return tb3_make(tb_max(a.x, b.x), tb_max(a.y, b.y), tb_max(a.z, b.z)); }
TB_HD float tb3_get(tb3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
TB_HD float tb_lerp(float a, float b, float t) { return a + t * (b - a); }

#endif /* TB_VEC_H */
