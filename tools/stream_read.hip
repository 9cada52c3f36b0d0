// How fast can a buffer of B bytes be re-read by back-to-back launches?  (Is the 156.5 MB packed inverse of cfg3 served by the
// 256 MiB Infinity Cache between two ADMM iterations, and what is the ceiling of a plain streaming read at that size?)
// Tile-shaped access like the mat-vec: workgroup b reads a contiguous chunk of `chunk` bytes with 16-B loads per lane.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/stream_read.hip -o /tmp/stream_read && /tmp/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void __launch_bounds__(256) read_chunks(const u32x4 *__restrict__ src, int64_t chunk16, int64_t nchunks, unsigned *sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const u32x4 *p = src + c * chunk16;
        for (int64_t e = threadIdx.x; e < chunk16; e += 256 * UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = e + u * 256 < chunk16 ? __builtin_nontemporal_load(p + e + u * 256) : acc;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
template <int UNROLL>
__global__ void __launch_bounds__(256) read_chunks_plain(const u32x4 *__restrict__ src, int64_t chunk16, int64_t nchunks, unsigned *sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const u32x4 *p = src + c * chunk16;
        for (int64_t e = threadIdx.x; e < chunk16; e += 256 * UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = e + u * 256 < chunk16 ? p[e + u * 256] : acc;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
    const int64_t chunk = 74240;                       // one fixed-point tile
    unsigned *sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (double mb : {38.0, 100.0, 156.5, 200.0, 268.5, 600.0}) {
        const int64_t nchunks = (int64_t)(mb * 1e6 / chunk);
        const int64_t bytes = nchunks * chunk;
        u32x4 *buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
        for (int variant = 0; variant < 2; ++variant)
            for (int grid : {256, 512, 768, 1024, 2048, (int)nchunks}) {
                const int reps = 100;
                for (int pass = 0; pass < 2; ++pass) {
                    if (pass) hipEventRecord(e0, 0);
                    for (int r = 0; r < (pass ? reps : 5); ++r) {
                        if (variant == 0) hipLaunchKernelGGL(read_chunks_plain<4>, dim3(grid), dim3(256), 0, 0, buf, chunk / 16, nchunks, sink);
                        else hipLaunchKernelGGL(read_chunks<4>, dim3(grid), dim3(256), 0, 0, buf, chunk / 16, nchunks, sink);
                    }
                    if (pass) hipEventRecord(e1, 0);
                }
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("%7.1f MB  %s  grid %5d: %7.2f us per pass  %6.2f TB/s\n", bytes / 1e6, variant ? "nt   " : "plain", grid, ms * 1e3 / reps,
                       bytes / (ms * 1e-3 / reps) / 1e12);
            }
        hipFree(buf);
    }
    return 0;
}
