// scratch_grow.hip -- does a kernel whose dispatch makes the runtime GROW a queue's scratch, launched right behind another kernel on the
// same stream, ever read what that kernel wrote as it was before?  (measurement, not product code)
//   hipcc -O2 --offload-arch=gfx950 -o scratch_grow scratch_grow.hip && for i in $(seq 200); do ./scratch_grow; done | sort | uniq -c
// Round 3 saw "one wrong 16x16 region in the first render of 1 process in ~3 000" and blamed a dispatch that had to wait for bigger
// scratch behind a kernel that had just written its input (context.cpp warms every kernel once for that reason).  Every process has fresh
// queues, so the experiment is one process per sample: writer (no scratch) -> reader with 1 KB of scratch per lane (first scratch on the
// queue) -> writer -> reader with 8 KB per lane (growth) -> writer -> reader with 32 KB per lane (growth again), all back to back on one
// non-blocking stream, the reader checking every word of another XCD's slice.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void writer(uint4* buf, uint32_t wordsPerSlice, uint32_t epoch)
{
    for (uint32_t i = threadIdx.x; i < wordsPerSlice; i += blockDim.x) { const uint32_t w = blockIdx.x * wordsPerSlice + i; buf[w] = uint4{epoch, w, epoch ^ w, 0x5a5a5a5au}; }
}

template <int WORDS> /* private array of WORDS dwords per lane, indexed at run time: lives in scratch */
__global__ void reader(const uint4* buf, uint32_t wordsPerSlice, uint32_t epoch, uint32_t twist, unsigned long long* bad)
{
    volatile uint32_t local[WORDS];
    for (int k = 0; k < WORDS; k += 64) local[(k + twist) % WORDS] = (uint32_t)k ^ twist;
    unsigned long long n = 0;
    const uint32_t s = (blockIdx.x + 3u) % gridDim.x;
    for (uint32_t i = threadIdx.x; i < wordsPerSlice; i += blockDim.x) {
        const uint32_t w = s * wordsPerSlice + i; const uint4 v = buf[w];
        if (!(v.x == epoch && v.y == w && v.z == (epoch ^ w))) n++;
        local[(i + twist) % WORDS] += v.x;
    }
    if (local[twist % WORDS] == 0xdeadbeefu) n += 1ull << 40; /* keeps the array alive */
    if (n) atomicAdd(bad, n);
}

int main()
{
    const uint32_t slices = 2048, wps = 64; /* 2 MB: stays in the L2s */
    uint4* buf; unsigned long long* bad; unsigned long long h[3];
    CHECK(hipMalloc(&buf, (size_t)slices * wps * 16)); CHECK(hipMalloc(&bad, 24)); CHECK(hipMemset(bad, 0, 24)); CHECK(hipMemset(buf, 0, (size_t)slices * wps * 16));
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(writer, dim3(slices), dim3(256), 0, s, buf, wps, 1u);
    hipLaunchKernelGGL(reader<256>, dim3(slices), dim3(256), 0, s, buf, wps, 1u, 7u, bad);
    hipLaunchKernelGGL(writer, dim3(slices), dim3(256), 0, s, buf, wps, 2u);
    hipLaunchKernelGGL(reader<2048>, dim3(slices), dim3(256), 0, s, buf, wps, 2u, 11u, bad + 1);
    hipLaunchKernelGGL(writer, dim3(slices), dim3(256), 0, s, buf, wps, 3u);
    hipLaunchKernelGGL(reader<8192>, dim3(slices), dim3(256), 0, s, buf, wps, 3u, 13u, bad + 2);
    CHECK(hipStreamSynchronize(s)); CHECK(hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost));
    printf("stale words behind a scratch dispatch of 1 KB / 8 KB / 32 KB per lane: %llu / %llu / %llu\n", h[0], h[1], h[2]);
    return 0;
}
