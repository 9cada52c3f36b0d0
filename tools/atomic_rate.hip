// Rate of 64-bit integer global atomics (no return) as the one-launch iterations use them: nadd adds spread over naddr distinct
// addresses from a grid of 256-thread workgroups, consecutive lanes on consecutive addresses (the column sums of a tile: 128 x ns
// values per tile).  usage: atomic_rate [naddr] [adds_per_addr] [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256) k(unsigned long long *acc, long long naddr, long long chunks_per_wg, long long nchunks) {
    // a chunk = 1024 consecutive addresses (128 columns x 8 signals); workgroup w takes chunks w, w + G, ... -- like tiles of a launch
    for (long long c = blockIdx.x, it = 0; it < chunks_per_wg; c += gridDim.x, ++it) {
        const long long base = (c % nchunks) * 1024;
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(acc + (base + r * 256 + threadIdx.x) % naddr, (unsigned long long)(c + r + 1));
    }
}
int main(int argc, char **argv) {
    const long long naddr = argc > 1 ? atoll(argv[1]) : 262144, per = argc > 2 ? atoll(argv[2]) : 128;
    const int G = argc > 3 ? atoi(argv[3]) : 256;
    unsigned long long *acc; hipMalloc(&acc, naddr * 8); hipMemset(acc, 0, naddr * 8);
    const long long nchunks = naddr / 1024, total_chunks = nchunks * per, cpw = total_chunks / G;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a); hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, acc, naddr, cpw, nchunks); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double adds = (double)cpw * G * 1024;
        printf("naddr %lld, %lld adds per address, %d workgroups: %.1f M atomics in %.1f us = %.1f G atomics/s (%.1f GB/s of 8-byte operands)\n", naddr, per, G, adds * 1e-6, ms * 1e3, adds / ms * 1e-6, adds * 8 / ms * 1e-6);
    }
    return 0;
}
