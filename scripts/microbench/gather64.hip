// gather64.hip -- how fast can a CU chase pointers through 64-B BVH nodes?  (measurement, not product code)
//   hipcc -O3 --offload-arch=gfx950 -o gather64 gather64.hip && ./gather64
// Every lane walks its own dependent chain node -> node (the next index is derived from the bytes just loaded), like
// one ray per lane descending a BVH with incoherent neighbours.  Modes:
//   A  4 x global_load_dwordx4 of the lane's own 64-B node            (what traverse() does today)
//   B  2 x dwordx4 (32-B nodes)                                       (what a half-size node would cost)
//   D  1 x dwordx4 (16-B nodes)
//   C  quad-cooperative: the 4 lanes of a quad fetch each other's nodes as 4 contiguous 16-B pieces (one 64-B
//      request per quad per instruction instead of four 16-B ones), then transpose through LDS
//   P  mode A, but the NEXT node of the chain is prefetched one step ahead (two independent chains per lane) -- MLP
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint4 a) { return a.x ^ (a.y * 0x9e3779b1u) ^ (a.z >> 3) ^ a.w; }

template <int PIECES>
__global__ __launch_bounds__(256) void walk_own(const uint4* __restrict__ nodes, uint32_t mask, uint32_t steps, uint32_t* out)
{
    const uint32_t salt = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    uint32_t ref = salt & mask;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        uint32_t h = 0;
#pragma unroll
        for (int p = 0; p < PIECES; p++) h += mix(nodes[(size_t)ref * PIECES + p]);
        acc += h; ref = (h + salt + s * 40503u) & mask; /* salted: plain h & mask walks would merge (random functional graph) */
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

/* mode A with only every `every`-th lane of the wave walking (the others idle): does a divergent load cost per
 * instruction or per active lane? */
template <int PIECES = 4>
__global__ __launch_bounds__(256) void walk_sparse(const uint4* __restrict__ nodes, uint32_t mask, uint32_t steps, uint32_t* out, uint32_t every, uint32_t contiguous)
{
    const uint32_t salt = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    uint32_t ref = salt & mask;
    uint32_t acc = 0;
    const uint32_t lane = threadIdx.x & 63u;
    const bool active = contiguous ? lane < 64u / every : (lane % every) == 0;
    if (active)
        for (uint32_t s = 0; s < steps; s++) {
            uint32_t h = 0;
#pragma unroll
            for (int p = 0; p < PIECES; p++) h += mix(nodes[(size_t)ref * PIECES + p]);
            acc += h; ref = (h + salt + s * 40503u) & mask;
        }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void walk_two_chains(const uint4* __restrict__ nodes, uint32_t mask, uint32_t steps, uint32_t* out)
{
    const uint32_t salt = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    uint32_t r0 = salt & mask, r1 = (r0 * 40503u + 17u) & mask;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; s += 2) {
        uint32_t h0 = 0, h1 = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) { h0 += mix(nodes[(size_t)r0 * 4 + p]); h1 += mix(nodes[(size_t)r1 * 4 + p]); }
        acc += h0 ^ h1; r0 = (h0 + salt + s * 40503u) & mask; r1 = (h1 + salt * 3u + s * 40503u) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void walk_quad(const uint4* __restrict__ nodes, uint32_t mask, uint32_t steps, uint32_t* out)
{
    __shared__ uint4 xpose[256 * 4];
    const uint32_t lane = threadIdx.x, q = lane & ~3u, piece = lane & 3u;
    const uint32_t salt = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    uint32_t ref = salt & mask;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        uint4 r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t other = __shfl(ref, (int)((lane & 63u & ~3u) + k), 64); /* ref of quad-mate k */
            r[k] = nodes[(size_t)other * 4 + piece];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) xpose[(q + k) * 4 + piece] = r[k]; /* slot of lane q+k, piece `piece` */
        /* same wave writes and reads its own 64 slots: no barrier needed beyond LDS ordering within the wave */
        __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) */
        uint32_t h = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) h += mix(xpose[lane * 4 + p]);
        acc += h; ref = (h + salt + s * 40503u) & mask; /* salted: plain h & mask walks would merge (random functional graph) */
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

/* Sparse waves, cooperative fetch: only every `every`-th lane owns a chain (like a wave whose other rays have finished), but ALL
 * lanes load: the n owners publish their node index in LDS by rank, LPR lanes per owner fetch the node's 4 / LPR 16-B pieces each
 * (one or two full-width load instructions instead of four sparse ones), the pieces go through LDS, the owner reads its 64 B back. */
template <int LPR>
__global__ __launch_bounds__(256) void walk_sparse_coop(const uint4* __restrict__ nodes, uint32_t mask, uint32_t steps, uint32_t* out, uint32_t every)
{
    constexpr int PPL = 4 / LPR;
    __shared__ uint32_t refs[4][64];
    __shared__ uint4 stage[4][64 / LPR * 4];
    const uint32_t salt = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    uint32_t ref = salt & mask;
    uint32_t acc = 0;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const bool active = (lane % every) == 0;
    const uint32_t q = lane / LPR, j = lane % LPR;
    for (uint32_t s = 0; s < steps; s++) {
        const unsigned long long m = __ballot(active && ref != 0xffffffffu); /* recomputed every step, as a walk would have to */
        const uint32_t n = (uint32_t)__popcll(m);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (active) refs[w][rank] = ref;
        __builtin_amdgcn_wave_barrier();
        uint4 c[PPL];
        if (q < n) {
            const uint32_t r = refs[w][q];
#pragma unroll
            for (int p = 0; p < PPL; p++) c[p] = nodes[(size_t)r * 4 + j * PPL + p];
#pragma unroll
            for (int p = 0; p < PPL; p++) stage[w][q * 4 + ((j * PPL + p) ^ ((q >> 2) & 3u))] = c[p];
        }
        __builtin_amdgcn_wave_barrier();
        if (active) {
            uint32_t h = 0;
#pragma unroll
            for (int p = 0; p < 4; p++) h += mix(stage[w][rank * 4 + (p ^ ((rank >> 2) & 3u))]);
            acc += h; ref = (h + salt + s * 40503u) & mask;
        }
        __builtin_amdgcn_wave_barrier();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
    printf("device %s  CUs %d  clock %.2f GHz\n", prop.name, cus, ghz);
    const uint32_t steps = 2000;
    uint32_t* out; CHECK(hipMalloc(&out, 256u * 64 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (uint32_t logN : {6u, 14u, 20u}) {
        const uint32_t N = 1u << logN;
        std::vector<uint32_t> host((size_t)N * 16);
        uint64_t x = 88172645463325252ull;
        for (auto& v : host) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x; }
        uint4* nodes; CHECK(hipMalloc(&nodes, (size_t)N * 64)); CHECK(hipMemcpy(nodes, host.data(), (size_t)N * 64, hipMemcpyHostToDevice));
        for (int blocksPerCu : {4}) {
            const int grid = cus * blocksPerCu;
            auto run = [&](const char* name, auto launch, double bytesPerVisit) {
                launch(); CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double visits = (double)grid * 256 * steps;
                printf("table %7.2f MB  %d waves/SIMD  %-18s %8.1f Gvisits/s  %6.0f clk per wave-visit per CU  %7.0f GB/s\n", N * 64.0 / 1e6, blocksPerCu, name,
                       visits / ms * 1e-6, ghz * 1e9 * (ms * 1e-3) / (visits / 64 / cus), visits * bytesPerVisit / ms * 1e-6);
            };
            run("A own 4x16B", [&] { hipLaunchKernelGGL(walk_own<4>, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out); }, 64);
            run("B own 2x16B", [&] { hipLaunchKernelGGL(walk_own<2>, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out); }, 32);
            run("D own 1x16B", [&] { hipLaunchKernelGGL(walk_own<1>, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out); }, 16);
            run("C quad+LDS", [&] { hipLaunchKernelGGL(walk_quad, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out); }, 64);
            for (uint32_t every : {2u, 4u, 8u}) for (uint32_t contig : {0u, 1u}) {
                char nm[32]; snprintf(nm, sizeof nm, "A 1/%u %s", every, contig ? "contig" : "strided");
                run(nm, [&] { hipLaunchKernelGGL(walk_sparse<4>, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out, every, contig); }, 64.0 / every);
            }
            /* round 3: the half-size (layout C) and quarter-size node at the lane counts the real walk runs at */
            for (uint32_t every : {2u, 4u, 8u}) {
                char nm[32]; snprintf(nm, sizeof nm, "B 1/%u strided", every);
                run(nm, [&] { hipLaunchKernelGGL(walk_sparse<2>, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out, every, 0u); }, 32.0 / every);
                snprintf(nm, sizeof nm, "D 1/%u strided", every);
                run(nm, [&] { hipLaunchKernelGGL(walk_sparse<1>, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out, every, 0u); }, 16.0 / every);
            }
            run("S 1/2 coop 2/ray", [&] { hipLaunchKernelGGL(walk_sparse_coop<2>, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out, 2u); }, 32);
            run("S 1/4 coop 4/ray", [&] { hipLaunchKernelGGL(walk_sparse_coop<4>, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out, 4u); }, 16);
            run("S 1/8 coop 4/ray", [&] { hipLaunchKernelGGL(walk_sparse_coop<4>, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out, 8u); }, 8);
            run("S 1/4 coop 2/ray", [&] { hipLaunchKernelGGL(walk_sparse_coop<2>, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out, 4u); }, 16);
            run("P 2 chains", [&] { hipLaunchKernelGGL(walk_two_chains, dim3(grid), dim3(256), 0, 0, nodes, N - 1, steps, out); }, 64);
        }
        CHECK(hipFree(nodes));
    }
    return 0;
}
