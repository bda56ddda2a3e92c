"""include/tb_math.h -- the fp32 arithmetic contract shared by the HIP kernels and the oracle -- against float64 libm (numpy).

Oracle and kernels include the same header, so an error in it is invisible to every parity test; this file is what bounds it.
Bounds are in units in the last place of the correctly rounded binary32 result, over the ranges the path uses:
the RNG calls sin() on seed + Time (seed < ~1e4, tested to 1e6), the BSDF code cos/acos/pow/exp on [0, 2 pi], [-1, 1], [0, 1].
exp and pow are the DXC lowering exp2(x log2 e) / exp2(y log2 x): their error grows with the magnitude of the exponent, which
is the reference's own behaviour on a GPU (hardware exp2 of a rounded product), so the bound is stated as a function of it.
The GPU half (-m gpu) checks host == device bit for bit, including NaN / -0 / denormal operands of min, max, frac, floor, 1/x."""
import numpy as np
import pytest

import oracle_lib as ol

SIN, COS, ACOS, ATAN2, EXP, LOG, POW, SQRT, EXP2, LOG2, ASIN, DIV, RAND, HASH13, MIN, MAX, FRAC, FLOOR, RCP = range(19)
N = 300000


def ulp_error(got, exact):
    """|got - exact| in ulps of the binary32 number nearest to `exact` (denormal spacing below 2^-126)."""
    e32 = np.abs(exact.astype(np.float32))
    ulp = np.maximum(np.spacing(e32).astype(np.float64), 2.0 ** -149)
    return np.abs(got.astype(np.float64) - exact) / ulp


def f32(a):
    return np.asarray(a, np.float32)


@pytest.fixture(scope="module")
def rng(built):
    return np.random.default_rng(20260)


@pytest.mark.parametrize("fn, ref, lo, hi, bound", [
    (SIN, np.sin, -10.0, 10.0, 2.0), (SIN, np.sin, 0.0, 8192.0, 2.0), (SIN, np.sin, -1e6, 1e6, 2.0),   # measured 1.52
    (COS, np.cos, -10.0, 10.0, 2.0), (COS, np.cos, -1e6, 1e6, 2.0),                                   # measured 1.53
    (ACOS, np.arccos, -1.0, 1.0, 2.0),                                                                 # measured 1.27
    (ASIN, np.arcsin, -1.0, 1.0, 3.0),                                                                 # measured 2.36
    (EXP2, np.exp2, -126.0, 127.0, 1.0),                                                               # measured 0.93
])
def test_transcendentals_within_stated_ulps(rng, fn, ref, lo, hi, bound):
    x = f32(rng.uniform(lo, hi, N))
    got = ol.math_fn(fn, x)
    err = ulp_error(got, ref(x.astype(np.float64)))
    assert err.max() <= bound, (fn, lo, hi, float(err.max()), float(x[err.argmax()]))
    assert err.mean() < 0.5


def test_log2_log_sqrt_rcp_div(rng):
    x = f32(np.exp(rng.uniform(-87, 88, N)))
    assert ulp_error(ol.math_fn(LOG2, x), np.log2(x.astype(np.float64))).max() <= 1.5            # measured 1.07
    assert ulp_error(ol.math_fn(LOG, x), np.log(x.astype(np.float64))).max() <= 2.0              # measured 1.44
    near1 = f32(rng.uniform(0.5, 2.0, N))                                                        # log2 -> 0: relative accuracy must hold (m + y form)
    assert ulp_error(ol.math_fn(LOG2, near1), np.log2(near1.astype(np.float64))).max() <= 1.5
    # correctly rounded primitives: identical to numpy's binary32 results
    assert np.array_equal(ol.math_fn(SQRT, x), np.sqrt(x))
    assert np.array_equal(ol.math_fn(RCP, x), np.float32(1.0) / x)
    den = f32(np.exp(rng.uniform(-100, -87, 1000)))                                              # denormal operands
    assert ulp_error(ol.math_fn(LOG2, den), np.log2(den.astype(np.float64))).max() <= 1.5


def test_atan2_all_quadrants(rng):
    y, x = f32(rng.normal(0, 3, N)), f32(rng.normal(0, 3, N))
    err = ulp_error(ol.math_fn(ATAN2, y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64)))
    assert err.max() <= 4.0                                                                      # measured 3.0
    # axis cases exactly as the header defines them (the environment lookup's atan2(dir.y, dir.x), RayGenCommon.h:27)
    pi = np.float32(3.14159265358979)
    cases = [((1, 0), np.float32(0.5) * pi), ((-1, 0), -np.float32(0.5) * pi), ((0, 0), 0.0), ((0, 1), 0.0), ((0, -1), pi)]
    for (yy, xx), want in cases:
        assert ol.math_fn(ATAN2, f32([yy]), f32([xx]))[0] == np.float32(want), (yy, xx)


def test_exp_and_pow_error_grows_with_the_exponent_only(rng):
    """exp(x) = exp2(x log2 e), pow(x, y) = exp2(y log2 x) (the DXC lowering, tb_math.h): one rounding of the product t costs
    up to |t| ln 2 half-ulps of the result.  Bound: 1.5 + 1.5 |t| ulps for exp, 1.5 + 2.5 |t| for pow (log2 x adds its own ulp); on the ranges the path uses that is a few ulps."""
    x = f32(rng.uniform(-87, 88, N))
    t = np.abs(x.astype(np.float64)) * 1.4426950408889634
    err = ulp_error(ol.math_fn(EXP, x), np.exp(x.astype(np.float64)))
    assert np.all(err <= 1.5 + 1.5 * t)
    small = f32(rng.uniform(-10, 2, N))                                                         # Beer-Lambert exp(-t sigma), the Gaussian filter, Burley fit
    assert ulp_error(ol.math_fn(EXP, small), np.exp(small.astype(np.float64))).max() <= 12.0
    for y in (2.0, 5.0, 2.2, 0.5, 1.0 / 1001.0, 1000.0):                                          # GGX / Schlick / gamma / lobe exponents of kernel.glsl
        b = f32(rng.uniform(0.0, 1.0, N)); yy = np.full(N, y, np.float32)
        exact = np.power(b.astype(np.float64), np.float64(np.float32(y)))
        keep = exact > 1e-35
        tt = np.abs(np.float64(np.float32(y)) * np.log2(np.maximum(b.astype(np.float64), 1e-300)))
        err = ulp_error(ol.math_fn(POW, b, yy)[keep], exact[keep])
        assert np.all(err <= 1.5 + 2.5 * tt[keep]), (y, float((err / (1.5 + tt[keep])).max()))
        if y in (2.0, 0.5, 1.0 / 1001.0):                                                         # bases that are not vanishingly small: a few ulps
            big = b[keep] >= 0.01
            assert err[big].max() <= 12.0, (y, float(err[big].max()))


def test_special_values_follow_the_hlsl_rules():
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    one = lambda fn, a, b=None: ol.math_fn(fn, f32([a]), None if b is None else f32([b]))[0]
    assert np.isnan(one(SIN, inf)) and np.isnan(one(COS, -inf)) and np.isnan(one(SIN, nan)) and np.isnan(one(SIN, 2.5e9))
    assert one(SIN, 0.0) == 0.0 and one(COS, 0.0) == 1.0
    assert np.isnan(one(ACOS, 1.0000001)) and np.isnan(one(ASIN, -1.0000001)) and one(ACOS, 1.0) == 0.0 and one(ASIN, 0.0) == 0.0
    assert one(ACOS, -1.0) == np.float32(3.14159265358979323846)
    assert one(EXP2, 128.0) == inf and one(EXP2, -151.0) == 0.0 and one(EXP2, 0.0) == 1.0 and one(EXP2, 10.0) == 1024.0 and np.isnan(one(EXP2, nan))
    assert one(EXP2, -140.0) == np.float32(2.0 ** -140) and one(EXP2, -149.0) == np.float32(2.0 ** -149)   # denormal results, rounded once
    assert one(LOG2, 0.0) == -inf and np.isnan(one(LOG2, -1.0)) and one(LOG2, inf) == inf and one(LOG2, 1.0) == 0.0 and one(LOG2, 8.0) == 3.0
    assert abs(float(one(LOG2, 1e-40)) - np.log2(float(np.float32(1e-40)))) < 1e-5
    # HLSL pow: NaN for x < 0, 0^y = 0 for y > 0, inf for y < 0, NaN for 0^0 (exp2(0 * -inf))
    assert np.isnan(one(POW, -1.0, 2.0)) and one(POW, 0.0, 2.0) == 0.0 and one(POW, 0.0, -1.0) == inf and np.isnan(one(POW, 0.0, 0.0))
    assert one(POW, 1.0, 1e30) == 1.0 and one(POW, 2.0, 3.0) == 8.0
    assert one(SQRT, 0.0) == 0.0 and np.isnan(one(SQRT, -1.0)) and one(RCP, 0.0) == inf


def test_min_max_frac_floor_semantics():
    """HLSL min/max return the non-NaN operand; -0 orders below +0 (v_min_f32 / v_max_f32).  frac(x) = x - floor(x)."""
    nan = np.float32(np.nan)
    a = f32([1.0, nan, 2.0, nan, 0.0, -0.0, -1.0, np.inf, -np.inf, 1e-45])
    b = f32([2.0, 3.0, nan, nan, -0.0, 0.0, -1.0, 1.0, 1.0, -1e-45])
    mn, mx = ol.math_fn(MIN, a, b), ol.math_fn(MAX, a, b)
    assert list(mn[:3]) == [1.0, 3.0, 2.0] and np.isnan(mn[3]) and list(mx[:3]) == [2.0, 3.0, 2.0] and np.isnan(mx[3])
    assert np.signbit(mn[4]) and np.signbit(mn[5]) and not np.signbit(mx[4]) and not np.signbit(mx[5])
    assert mn[7] == 1.0 and mx[7] == np.inf and mn[8] == -np.inf and mx[8] == 1.0 and mn[9] == np.float32(-1e-45) and mx[9] == np.float32(1e-45)
    x = f32([0.25, -0.25, 3.0, -3.0, 43758.5453123, -1e-30, 8388609.0, 1e30])
    fl, fr = ol.math_fn(FLOOR, x), ol.math_fn(FRAC, x)
    assert np.array_equal(fl, np.floor(x)) and np.array_equal(fr, x - np.floor(x))
    assert fr[5] == 1.0                      # frac of a tiny negative number rounds to 1.0 -- the reference's rand() can return exactly 1
    assert np.all((fr >= 0) & (fr <= 1))


def test_rand_stream_properties(rng):
    """rand() = frac(sin(seed++ + Time) * 43758.5453123) (kernel.glsl:39-40): values in [0, 1], mean 1/2, the fp32 product
    leaves at most 8 fraction bits so exact zeros occur (the degenerate-axis rays of DESIGN.md section 4)."""
    import ctypes as C
    out = np.zeros(200000, np.float32)
    ol.lib().tbo_rand_stream(0.172363, 0.0, out.size, out.ctypes.data_as(C.c_void_p))
    assert out.min() >= 0.0 and out.max() <= 1.0 and abs(float(out.mean()) - 0.5) < 5e-3
    assert (out == 0.0).sum() > 0
    s = np.empty(2000, np.float32); v = np.float32(0.172363)
    for i in range(2000): s[i] = v; v = np.float32(v + np.float32(1.0))          # seed++ in binary32, one rounding per call
    assert np.array_equal(out[:2000], (lambda p: p - np.floor(p))(ol.math_fn(SIN, s) * np.float32(43758.5453123)))


@pytest.mark.gpu
def test_device_evaluates_every_function_to_the_host_bits(gpu_tb, rng):
    """gfx950 == x86-64 for every function of the header, operands incl. NaN, +-0, +-inf, denormals (tb_device_math, C ABI)."""
    edge = f32([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38, 1e9, 2.5e9, 6.2831855, 0.70710678, 8388609.0])
    cases = {
        SIN: (rng.uniform(-5000, 5000, N), None), COS: (rng.uniform(-5000, 5000, N), None), ACOS: (rng.uniform(-1.01, 1.01, N), None), ASIN: (rng.uniform(-1.01, 1.01, N), None),
        ATAN2: (rng.normal(0, 3, N), rng.normal(0, 3, N)), EXP: (rng.uniform(-100, 100, N), None), LOG: (np.exp(rng.uniform(-100, 88, N)), None),
        POW: (rng.uniform(-0.2, 4, N), rng.uniform(-20, 20, N)), SQRT: (np.exp(rng.uniform(-100, 88, N)), None), EXP2: (rng.uniform(-160, 135, N), None),
        LOG2: (np.exp(rng.uniform(-104, 88, N)), None), MIN: (rng.normal(0, 1, N), rng.normal(0, 1, N)), MAX: (rng.normal(0, 1, N), rng.normal(0, 1, N)),
        FRAC: (rng.uniform(-1e5, 1e5, N), None), FLOOR: (rng.uniform(-1e5, 1e5, N), None), RCP: (np.exp(rng.uniform(-100, 88, N)) * rng.choice([-1, 1], N), None),
    }
    for fn, (a, b) in cases.items():
        a = np.concatenate([f32(a), edge, edge])
        b2 = None if b is None else np.concatenate([f32(b), edge, edge[::-1]])
        dev = gpu_tb.DeviceMath(fn, a, b2)
        host = ol.math_fn(fn, a, b2)
        same = (dev.view(np.uint32) == host.view(np.uint32)) | (np.isnan(dev) & np.isnan(host))
        assert same.all(), (fn, a[~same][:4], None if b2 is None else b2[~same][:4], dev[~same][:4], host[~same][:4])
