// valu_issue.hip -- how many cycles does a SIMD of gfx950 need to issue one wave64 VALU instruction?  (measurement, not product code)
//   hipcc -O3 --offload-arch=gfx950 -o valu_issue valu_issue.hip && ./valu_issue
// Settles the peak that bench.py's "valu" roofline divides by: each wave runs UNROLL independent
// accumulator chains of one instruction kind, long enough that loop overhead disappears; waves-per-SIMD 1, 2, 4 show whether a
// second wave fills slots the first cannot.  Cycles are s_memtime ticks of wave 0 of each SIMD, converted with the measured
// ratio to the 100 MHz s_memrealtime counter.  Kinds: v_fma_f32, v_pk_fma_f32 (two fp32 fmas per lane), v_mul_f32, v_max_f32,
// v_cndmask_b32 (select), v_add_u32, and a fma whose chain is DEPENDENT (latency, not issue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int UNROLL = 16, ITERS = 2048;

template <int KIND>
__global__ __launch_bounds__(1024) void issue_kernel(float* out, unsigned long long* cycles, unsigned long long* realtime)
{
    float a = (float)threadIdx.x * 1e-3f + 1.0f, b = 0.999f;
    float acc[UNROLL];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc2[UNROLL], a2 = {a, a}, b2 = {b, b};
    uint32_t iacc[UNROLL];
    for (int i = 0; i < UNROLL; i++) { acc[i] = (float)i; acc2[i] = f2{(float)i, (float)i}; iacc[i] = (uint32_t)i; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
            if (KIND == 2) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(b));
            if (KIND == 3) asm volatile("v_max_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
            if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a));
            if (KIND == 5) asm volatile("v_add_u32 %0, %1, %0" : "+v"(iacc[i]) : "v"(iacc[0]));
            if (KIND == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));   /* one dependent chain */
            if (KIND == 7) asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0; for (int i = 0; i < UNROLL; i++) s += acc[i] + acc2[i].x + acc2[i].y + (float)iacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; realtime[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r1 - r0; }
}

template <int KIND>
void run(const char* name, int wavesPerSimd)
{
    int dev = 0, cus = 0; CHECK(hipGetDevice(&dev)); CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int block = 256 * wavesPerSimd, waves = block / 64, grid = cus;   /* one workgroup per CU, wavesPerSimd waves on each of its 4 SIMDs */
    float* out; unsigned long long *cyc, *rt;
    CHECK(hipMalloc(&out, (size_t)grid * block * 4)); CHECK(hipMalloc(&cyc, (size_t)grid * waves * 8)); CHECK(hipMalloc(&rt, (size_t)grid * waves * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    issue_kernel<KIND><<<grid, block>>>(out, cyc, rt);   /* warm */
    CHECK(hipEventRecord(e0));
    issue_kernel<KIND><<<grid, block>>>(out, cyc, rt);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hc((size_t)grid * waves), hr((size_t)grid * waves);
    CHECK(hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hr.data(), rt, hr.size() * 8, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end()); std::sort(hr.begin(), hr.end());
    const double medCyc = (double)hc[hc.size() / 2], medRt = (double)hr[hr.size() / 2];
    const double insts = (double)UNROLL * ITERS;                       /* wave-instructions per wave */
    const double memtimeMHz = medCyc / (medRt / 100.0);               /* s_memrealtime ticks at 100 MHz */
    /* per SIMD: wavesPerSimd waves share it; cycles per wave-instruction ISSUED BY THE SIMD = wave time / (insts * wavesPerSimd) */
    printf("%-22s waves/SIMD %d: %7.3f s_memtime ticks per wave-instr per SIMD  (s_memtime runs at %.0f MHz; kernel %.3f ms => %.3f ns per wave-instr per SIMD)\n",
           name, wavesPerSimd, medCyc / (insts * wavesPerSimd), memtimeMHz, ms, ms * 1e6 / (insts * wavesPerSimd));
    CHECK(hipFree(out)); CHECK(hipFree(cyc)); CHECK(hipFree(rt));
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_mul_f32", w); run<3>("v_max_f32", w);
        run<7>("v_min3_f32", w); run<4>("v_cndmask_b32", w); run<5>("v_add_u32", w); run<6>("v_fma_f32 dependent", w);
    }
    printf("ns per wave-instr per SIMD x shader clock (GHz) = cycles; at 2.4 GHz: 2 cycles = 0.833 ns, 4 cycles = 1.667 ns\n");
    return 0;
}
