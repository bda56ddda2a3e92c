/* Is rcp + Newton the SAME BITS as the IEEE division 1.0f / x for every binary32 x?  Exhaustive: all 2^32 bit patterns on the device, against the
 * compiler's own correctly rounded division (v_div_scale / v_rcp / fma ... / v_div_fmas / v_div_fixup; -ffp-contract=off).
 *   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 scripts/microbench/rcp_exact.hip -o scripts/microbench/rcp_exact && scripts/microbench/rcp_exact
 * Variant 1: r = rcp(x); e = fma(-x, r, 1); r = fma(e, r, r).   Variant 2: the same step twice.
 * Prints, per variant, the number of mismatching inputs in all, among |x| in [2^-126, 2^126] and among |x| in [2^-100, 2^100], and the first few. */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float rcp1(float x) { float r = __builtin_amdgcn_rcpf(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ float rcp2(float x) { float r = rcp1(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }

__global__ void check(unsigned long long* out, uint32_t* first)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad[6] = {0, 0, 0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float ref = 1.0f / x;
        const uint32_t ex = (bits >> 23) & 0xffu;
        const bool normal = ex >= 1 && ex <= 253, mid = ex >= 27 && ex <= 227; /* 2^-126 .. 2^126 | 2^-100 .. 2^100 */
        for (int v = 0; v < 2; v++) {
            const float got = v == 0 ? rcp1(x) : rcp2(x);
            const bool same = __float_as_uint(got) == __float_as_uint(ref) || (got != got && ref != ref);
            if (!same) { bad[3 * v]++; if (normal) bad[3 * v + 1]++; if (mid) { bad[3 * v + 2]++; const unsigned long long k = atomicAdd(&out[6 + v], 1ull); if (k < 8) first[8 * v + k] = bits; } }
        }
    }
    for (int k = 0; k < 6; k++) if (bad[k]) atomicAdd(&out[k], bad[k]);
}

int main()
{
    unsigned long long* out; uint32_t* first;
    hipMalloc(&out, 8 * 8); hipMalloc(&first, 16 * 4); hipMemset(out, 0, 64); hipMemset(first, 0, 64);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, out, first);
    unsigned long long h[8]; uint32_t f[16];
    if (hipMemcpy(h, out, 64, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(f, first, 64, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    for (int v = 0; v < 2; v++) {
        printf("variant %d (%d Newton step%s): mismatches of 2^32 inputs: %llu in all, %llu with |x| in [2^-126, 2^126], %llu with |x| in [2^-100, 2^100]", v + 1, v + 1, v ? "s" : "", h[3 * v], h[3 * v + 1], h[3 * v + 2]);
        for (int k = 0; k < 8 && k < (int)h[6 + v]; k++) { float x; memcpy(&x, &f[8 * v + k], 4); printf("%s 0x%08x (%g)", k ? "," : "; first:", f[8 * v + k], x); }
        printf("\n");
    }
    return 0;
}
