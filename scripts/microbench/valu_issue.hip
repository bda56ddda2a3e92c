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

/* round 3: does a VALU instruction cost less when part of the wave is masked off?  (it would pay to pack the live rays of a walk into
 * the low lanes if it did)  LANES = how many lanes of each wave run the loop: 64, the low 32, the low 16, every 4th (16 lanes spread). */
template <int LANES>
__global__ __launch_bounds__(1024) void masked_kernel(float* out, unsigned long long* cycles, unsigned long long* realtime)
{
    float a = (float)threadIdx.x * 1e-3f + 1.0f, b = 0.999f;
    float acc[UNROLL], acd[UNROLL];
    for (int i = 0; i < UNROLL; i++) { acc[i] = (float)i; acd[i] = (float)i * 0.5f; }
    const unsigned lane = threadIdx.x & 63u;
    const bool on = LANES == 64 ? true : (LANES == 32 ? lane < 32 : (LANES == 16 ? lane < 16 : (lane & 3u) == 0));
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    if (on)
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < UNROLL; i++) asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_max_f32 %1, %2, %1" : "+v"(acc[i]), "+v"(acd[i]) : "v"(a), "v"(b));
        }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0; for (int i = 0; i < UNROLL; i++) s += acc[i] + acd[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; realtime[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r1 - r0; }
}

template <int KIND>
__global__ __launch_bounds__(1024) void issue_kernel(float* out, unsigned long long* cycles, unsigned long long* realtime)
{
    float a = (float)threadIdx.x * 1e-3f + 1.0f, b = 0.999f;
    float acc[UNROLL];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc2[UNROLL], a2 = {a, a}, b2 = {b, b};
    uint32_t iacc[UNROLL];
    double dacc[UNROLL], da = (double)a, db = 0.999;
    for (int i = 0; i < UNROLL; i++) dacc[i] = (double)i + 0.5;
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
            if (KIND == 8) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(acc[i]) : "v"(a));              /* mask in an SGPR pair */
            if (KIND == 9) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a) : "vcc");   /* compare + select pair (2 instrs) */
            if (KIND == 10) asm volatile("v_cmp_lt_f32 vcc, %1, %0" : : "v"(acc[i]), "v"(a) : "vcc");
            if (KIND == 11) asm volatile("v_cmp_lt_f32_e64 s[10:11], %1, %0" : : "v"(acc[i]), "v"(a) : "s10", "s11");
            if (KIND == 12) asm volatile("v_lshl_add_u32 %0, %1, 4, %0" : "+v"(iacc[i]) : "v"(iacc[0]));
            if (KIND == 13) asm volatile("v_max_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(a), "v"(b));                       /* no read of the destination */
            if (KIND == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(acc[i]) : "v"(a));
            if (KIND == 15) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(acc[i]) : "v"(a));
            if (KIND == 16) asm volatile("s_and_b64 s[10:11], s[10:11], exec" : : : "s10", "s11", "scc");                 /* SALU issue */
            if (KIND == 18) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[i]) : "v"(da), "v"(db));
            if (KIND == 19) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(dacc[i]) : "v"(db));
            if (KIND == 20) asm volatile("v_rndne_f64 %0, %0" : "+v"(dacc[i]));
            if (KIND == 21) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dacc[i]) : "v"(acc[i]));
            if (KIND == 22) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(acc[i]) : "v"(dacc[i]));
            if (KIND == 23) asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[i]));
            if (KIND == 24) asm volatile("v_sqrt_f32 %0, %0" : "+v"(acc[i]));
            if (KIND == 25) asm volatile("v_div_scale_f32 %0, vcc, %1, %1, %0" : "+v"(acc[i]) : "v"(a) : "vcc");
            if (KIND == 26) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 27) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 28) asm volatile("v_floor_f32 %0, %0" : "+v"(acc[i]));
            if (KIND == 29) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
            if (KIND == 30) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc2[i]) : "v"(a2));
            if (KIND == 31) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 32) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(iacc[i]) : "v"(dacc[i]));
            if (KIND == 33) asm volatile("v_floor_f64 %0, %0" : "+v"(dacc[i]));
            /* round 3: what a compact (16-bit) BVH node costs to unpack */
            if (KIND == 34) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(acc[i]) : "v"(iacc[i]));
            if (KIND == 35) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(acc[i]) : "v"(iacc[i]));
            if (KIND == 36) asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(acc[i]) : "v"(iacc[i]));
            if (KIND == 37) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(iacc[i]), "v"(b));   /* f16 (high half) x f32 + f32 */
            if (KIND == 38) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(iacc[i]) : "v"(iacc[0]), "v"(a), "v"(b));
            if (KIND == 39) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(iacc[i]) : "v"(iacc[0]), "v"(a));
            if (KIND == 40) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(acc[i]) : "v"(iacc[i]));
            if (KIND == 41) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc2[i]) : "v"(a2));
            if (KIND == 42) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 43) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(iacc[i]) : "v"(a), "v"(b));
            if (KIND == 44) asm volatile("v_lshl_or_b32 %0, %1, 7, %0" : "+v"(iacc[i]) : "v"(iacc[0]));
            if (KIND == 45) asm volatile("v_bfe_u32 %0, %1, 16, 16" : "=v"(iacc[i]) : "v"(iacc[0]));
            if (KIND == 46) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 47) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(acc[i]) : "v"(iacc[i]));
            if (KIND == 48) asm volatile("v_min_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 49) asm volatile("v_max_f32_e64 %0, %1, %2" : "=v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 17) asm volatile("v_fma_f32 %0, %1, %2, %0\n\ts_and_b64 s[10:11], s[10:11], exec" : "+v"(acc[i]) : "v"(a), "v"(b) : "s10", "s11", "scc");   /* VALU + SALU pair */
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0; for (int i = 0; i < UNROLL; i++) s += acc[i] + acc2[i].x + acc2[i].y + (float)iacc[i] + (float)dacc[i];
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

template <int LANES>
void run_masked(const char* name, int wavesPerSimd)
{
    int dev = 0, cus = 0; CHECK(hipGetDevice(&dev)); CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int block = 256 * wavesPerSimd, waves = block / 64, grid = cus;
    float* out; unsigned long long *cyc, *rt;
    CHECK(hipMalloc(&out, (size_t)grid * block * 4)); CHECK(hipMalloc(&cyc, (size_t)grid * waves * 8)); CHECK(hipMalloc(&rt, (size_t)grid * waves * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    masked_kernel<LANES><<<grid, block>>>(out, cyc, rt);
    CHECK(hipEventRecord(e0));
    masked_kernel<LANES><<<grid, block>>>(out, cyc, rt);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = 2.0 * UNROLL * ITERS;
    printf("%-34s waves/SIMD %d: kernel %.3f ms => %.3f ns per wave-instr per SIMD\n", name, wavesPerSimd, ms, ms * 1e6 / (insts * wavesPerSimd));
    CHECK(hipFree(out)); CHECK(hipFree(cyc)); CHECK(hipFree(rt));
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 4}) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_mul_f32", w); run<3>("v_max_f32", w);
        run<7>("v_min3_f32", w); run<4>("v_cndmask_b32 vcc", w); run<8>("v_cndmask_b32 sgpr", w); run<9>("v_cmp+v_cndmask (2)", w); run<10>("v_cmp_lt_f32 vcc", w);
        run<11>("v_cmp_lt_f32 sgpr", w); run<12>("v_lshl_add_u32", w); run<13>("v_max_f32 (no dst read)", w); run<14>("v_mov_b32", w); run<15>("v_mov_b32 dpp quad", w);
        run<18>("v_fma_f64", w); run<19>("v_mul_f64", w); run<20>("v_rndne_f64", w); run<33>("v_floor_f64", w); run<21>("v_cvt_f64_f32", w); run<22>("v_cvt_f32_f64", w); run<32>("v_cvt_i32_f64", w);
        run<23>("v_rcp_f32", w); run<24>("v_sqrt_f32", w); run<25>("v_div_scale_f32", w); run<26>("v_div_fmas_f32", w); run<27>("v_div_fixup_f32", w); run<28>("v_floor_f32", w);
        run<29>("v_add_f32", w); run<30>("v_pk_mul_f32", w); run<31>("v_med3_f32", w);
        run<34>("v_cvt_f32_u32", w); run<35>("v_cvt_f32_u32 sdwa WORD_1", w); run<36>("v_cvt_f32_ubyte2", w); run<37>("v_fma_mix_f32 (f16 hi)", w); run<38>("v_perm_b32", w);
        run<39>("v_and_or_b32", w); run<40>("v_cvt_f32_f16", w); run<47>("v_cvt_f32_f16 sdwa WORD_1", w); run<41>("v_pk_add_f32", w); run<42>("v_max3_f32", w); run<43>("v_pk_fma_f16", w);
        run<44>("v_lshl_or_b32", w); run<45>("v_bfe_u32", w); run<46>("v_dot2_f32_f16", w); run<48>("v_min_f32 3-operand", w); run<49>("v_max_f32_e64", w);
        run<16>("s_and_b64", w); run<17>("v_fma_f32 + s_and_b64 (2)", w); run<5>("v_add_u32", w); run<6>("v_fma_f32 dependent", w);
    }
    for (int w : {1, 4}) { run_masked<64>("fma+max, all 64 lanes", w); run_masked<32>("fma+max, low 32 lanes", w); run_masked<16>("fma+max, low 16 lanes", w); run_masked<4>("fma+max, every 4th lane", w); }
    printf("ns per wave-instr per SIMD x shader clock (GHz) = cycles; at 2.4 GHz: 2 cycles = 0.833 ns, 4 cycles = 1.667 ns\n");
    return 0;
}
