// l2_stale.hip -- does a kernel ever read what an EARLIER kernel left in its XCD's L2 instead of what the kernel before it wrote?
// (measurement, not product code)     hipcc -O2 --offload-arch=gfx950 -o l2_stale l2_stale.hip && ./l2_stale
//
// Round 3 saw it in the product: hit records written by pt_primary and read by pt_persistent on the SAME stream
// came back, about once in 300 small renders, as the launch before last had left them -- although a kernel boundary is supposed to
// write the producer's L2 back and invalidate the consumer's.  This program isolates the pattern:
//   writer  W_k : block b writes slice b of a buffer with (epoch k, word index) -- plain 16-B stores, as the product's
//   reader  R_k : block b reads slice (b + 3) mod blocks -- another XCD's slice (block b runs on XCD b mod 8) -- and counts every word
//                 that is not epoch k's; stale words are classified by how many epochs old they are
// over `rounds` epochs, in the launch shapes the product uses:
//   one-stream      W_k, R_k back to back on one stream, one buffer
//   two-streams     round k uses stream k & 1 and buffer k & 1 (the product's two side streams / two buffers), nothing waits for anything
//                   but its own stream: R_k overlaps W_(k+1) on the other stream
//   host-sync       two-streams with hipStreamSynchronize between W_k and R_k
//   persistent      two-streams, and R_k is a RESIDENT grid (2 blocks per CU) that loops over the slices, like pt_persistent
//   sc1-reader      two-streams, the reader loads with agent scope (global_load ... sc1): bypasses L1, served by L2
//   sys-reader      two-streams, the reader loads with system scope (sc0 sc1)
// buffer sizes from L2-resident (256 KB) to beyond the Infinity Cache (512 MB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void writer(uint4* buf, uint32_t wordsPerSlice, uint32_t slices, uint32_t epoch)
{
    for (uint32_t s = blockIdx.x; s < slices; s += gridDim.x)
        for (uint32_t i = threadIdx.x; i < wordsPerSlice; i += blockDim.x) {
            const uint32_t w = s * wordsPerSlice + i;
            buf[w] = uint4{epoch, w, epoch ^ w, 0x5a5a5a5au};
        }
}

template <int SCOPE> /* 0 plain, 1 agent (sc1), 2 system (sc0 sc1) */
__global__ void reader(const uint4* buf, uint32_t wordsPerSlice, uint32_t slices, uint32_t epoch, unsigned long long* stats)
{
    unsigned long long bad[4] = {0, 0, 0, 0}; /* 1 epoch old, 2 epochs old, older, torn */
    for (uint32_t s0 = blockIdx.x; s0 < slices; s0 += gridDim.x) {
        const uint32_t s = (s0 + 3u) % slices;
        for (uint32_t i = threadIdx.x; i < wordsPerSlice; i += blockDim.x) {
            const uint32_t w = s * wordsPerSlice + i;
            uint4 v;
            if (SCOPE == 0) v = buf[w];
            else {
                const unsigned long long* p = (const unsigned long long*)(buf + w);
                const int scope = SCOPE == 1 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM;
                const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, scope), b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, scope);
                v = uint4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
            }
            if (v.x == epoch && v.y == w && v.z == (epoch ^ w)) continue;
            if (v.y != w || v.z != (v.x ^ w)) bad[3]++;
            else if (v.x + 1 == epoch) bad[0]++;
            else if (v.x + 2 == epoch) bad[1]++;
            else bad[2]++;
        }
    }
    for (int k = 0; k < 4; k++) if (bad[k]) atomicAdd(&stats[k], bad[k]);
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    int cus = 0; CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipStream_t st[2]; for (auto& s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long* stats; CHECK(hipMalloc(&stats, 64));
    const char* modes[] = {"one-stream", "two-streams", "host-sync", "persistent", "sc1-reader", "sys-reader"};
    const size_t sizes[] = {256u << 10, 3u << 20, 24u << 20, 192u << 20, 512u << 20};
    printf("# %d rounds per cell; stale words: 1 epoch old / 2 epochs old / older / torn, of words read\n", rounds);
    for (size_t bytes : sizes) {
        uint4* buf[2]; for (auto& b : buf) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 0, bytes)); }
        const uint32_t words = (uint32_t)(bytes / 16), slices = 2048, wps = words / slices;
        for (int mode = 0; mode < 6; mode++) {
            const int r = bytes > (64u << 20) ? rounds / 10 : rounds;
            CHECK(hipMemset(stats, 0, 64)); CHECK(hipMemset(buf[0], 0, bytes)); CHECK(hipMemset(buf[1], 0, bytes)); CHECK(hipDeviceSynchronize());
            for (int k = 1; k <= r; k++) {
                const int par = mode == 0 ? 0 : (k & 1);
                hipStream_t s = st[par]; uint4* b = buf[par];
                const uint32_t epoch = (uint32_t)(k + (mode << 20));
                hipLaunchKernelGGL(writer, dim3(slices), dim3(256), 0, s, b, wps, slices, epoch);
                if (mode == 2) CHECK(hipStreamSynchronize(s));
                const dim3 grid(mode == 3 ? 2 * cus : slices);
                if (mode == 4) hipLaunchKernelGGL(reader<1>, grid, dim3(256), 0, s, b, wps, slices, epoch, stats);
                else if (mode == 5) hipLaunchKernelGGL(reader<2>, grid, dim3(256), 0, s, b, wps, slices, epoch, stats);
                else hipLaunchKernelGGL(reader<0>, grid, dim3(256), 0, s, b, wps, slices, epoch, stats);
            }
            CHECK(hipDeviceSynchronize());
            unsigned long long h[4]; CHECK(hipMemcpy(h, stats, 32, hipMemcpyDeviceToHost));
            printf("%8.2f MB  %-12s  %llu / %llu / %llu / %llu of %.3g words\n", bytes / 1048576.0, modes[mode], h[0], h[1], h[2], h[3], (double)wps * slices * r);
            fflush(stdout);
        }
        for (auto& b : buf) CHECK(hipFree(b));
    }
    return 0;
}
