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


// the mat-vec's shape: ONE workgroup per 74 240-byte tile (grid = nchunks), every thread issues its 18 16-byte loads (+2 small) up front.
// PATTERN 0: as the packed tiles are laid out now -- a wave instruction covers 4 rows x 256 B at 512-B stride;
// PATTERN 1: lane order -- a wave instruction covers 1 KiB contiguous.
template <int PATTERN>
__global__ void __launch_bounds__(256, 3) read_tiles(const u32x4 *__restrict__ src, int64_t chunk16, unsigned *sink) {
    const u32x4 *p = src + (int64_t)blockIdx.x * chunk16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    u32x4 v[18];
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        if (PATTERN == 0) {
            const int base = ((wave * 32 + g + 4 * rg) * 128 + 4 * c) / 4;      // in 16-byte units: row (wave*32 + 4 rg + g), column 4c
            v[2 * rg] = p[base];
            v[2 * rg + 1] = p[base + 16];
        } else {
            const int base = ((wave * 8 + rg) * 2) * 64 + lane;
            v[2 * rg] = p[base];
            v[2 * rg + 1] = p[base + 64];
        }
    }
    v[16] = p[4096 + (wave * 64 + lane) * 2];
    v[17] = p[4096 + (wave * 64 + lane) * 2 + 1];
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 18; ++u) acc ^= v[u];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// ... and with the mat-vec's arithmetic on the loaded words (MATH 1: the 36-bit decode of admm.hip -- alignbit, bfe, lshl_or, add_f64 --
// and two FMAs per element, then the row / column reductions; MATH 2: one instruction less per element: the assembled double is used as
// it is, biased, and the bias is taken off the row / column sums), to see what the arithmetic costs on top of the stream.
__device__ __forceinline__ double opaque_d(double v) { asm volatile("" : "+v"(v)); return v; }
template <int MATH>
__global__ void __launch_bounds__(256, 3) math_tiles(const u32x4 *__restrict__ src, int64_t chunk16, const double *__restrict__ rhs, double *__restrict__ out,
                                                      unsigned long long *__restrict__ acc = nullptr, const long long *__restrict__ acc_prev = nullptr) {
    __shared__ double sI[128], sJ[128], sT[4][128];
    const u32x4 *p = src + (int64_t)blockIdx.x * chunk16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    u32x4 ha[8], hb[8];
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        const int base = ((wave * 32 + g + 4 * rg) * 128 + 4 * c) / 4;
        ha[rg] = p[base];
        hb[rg] = p[base + 16];
    }
    const u32x4 n0 = p[4096 + (wave * 64 + lane) * 2], n1 = p[4096 + (wave * 64 + lane) * 2 + 1];
    if (MATH == 3) {
        // the right-hand side rebuilt from the previous launch's fixed-point sums + two state vectors (a stand-in for prox and dual update)
        const int blk = threadIdx.x < 128 ? (blockIdx.x % 61) : (blockIdx.x % 59), e = blk * 128 + (threadIdx.x & 127);
        const double xv = (double)acc_prev[e] * 0x1p-40 + rhs[e], uv = rhs[8192 + e];
        const double vv = xv + uv, zv = vv > 0.1 ? vv - 0.1 : (vv < -0.1 ? vv + 0.1 : 0.0);
        const double r = (2.0 * zv - vv) * 20.0;
        if (threadIdx.x < 128) sI[threadIdx.x] = r; else sJ[threadIdx.x - 128] = r;
    } else {
        if (threadIdx.x < 128) sI[threadIdx.x] = rhs[(blockIdx.x % 61) * 128 + threadIdx.x]; else sJ[threadIdx.x - 128] = rhs[(blockIdx.x % 59) * 128 + threadIdx.x - 128];
    }
    __syncthreads();
    double rj[8], tc[8], v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) tc[k] = 0.0;
    const unsigned nw[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
    double sumr = 0;
    if (MATH == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) sumr += rj[k];
    }
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        const double ri = sI[wave * 32 + 4 * rg + g];
        const unsigned hh[8] = {ha[rg].x, ha[rg].y, ha[rg].z, ha[rg].w, hb[rg].x, hb[rg].y, hb[rg].z, hb[rg].w};
        double a0 = 0, a1 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double m;
            if (MATH == 1 || MATH == 3) {
                const unsigned top = __builtin_amdgcn_alignbit(0x04330000u, hh[k], 28);
                unsigned lo = (nw[rg] >> (4 * k)) & 15u;
                asm("v_lshl_or_b32 %0, %1, 4, %0" : "+v"(lo) : "v"(hh[k]));
                m = __hiloint2double((int)top, (int)lo) - (0x1p52 + 0x1p35);
            } else {
                // 2^36 + q: exponent 0x423, mantissa = q << 16:  hi dword = 0x42300000 | (q >> 16) = alignbit(0x423, hh, 12),  lo dword = (hh << 20) | (nib << 16)
                const unsigned top = __builtin_amdgcn_alignbit(0x423u, hh[k], 12);
                unsigned lo = ((nw[rg] >> (4 * k)) & 15u) << 16;       // (a layout with the nibble pre-shifted would make this one v_and)
                asm("v_lshl_or_b32 %0, %1, 20, %0" : "+v"(lo) : "v"(hh[k]));
                m = __hiloint2double((int)top, (int)lo);
            }
            tc[k] = opaque_d(fma(m, ri, tc[k]));
            if (k & 1) a1 = fma(m, rj[k], a1); else a0 = fma(m, rj[k], a0);
        }
        v[rg] = (a0 + a1) - (MATH == 2 ? (0x1p36 + 0x1p35) * sumr : 0.0);
    }
#pragma unroll
    for (int m = 8, cnt = 4; m >= 2; m >>= 1, cnt >>= 1) {
        const bool up = (c & m) != 0;
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            const double lo_ = opaque_d(v[k]), hi_ = opaque_d(v[k + cnt]);
            v[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
        }
    }
    v[0] += __shfl_xor(v[0], 1, 64);
    double *o1 = out + (int64_t)blockIdx.x * 256, *o2 = o1 + 128;
    if ((c & 1) == 0) {
        const int rg = ((c & 8) ? 4 : 0) + ((c & 4) ? 2 : 0) + ((c & 2) ? 1 : 0);
        if (MATH == 3) atomicAdd(acc + (blockIdx.x % 61) * 128 + wave * 32 + 4 * rg + g, (unsigned long long)(long long)rint(v[0] * 0x1p40));
        else o1[wave * 32 + 4 * rg + g] = v[0];
    }
#pragma unroll
    for (int m = 32, cnt = 4; m >= 16; m >>= 1, cnt >>= 1) {
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            const double lo_ = opaque_d(tc[k]), hi_ = opaque_d(tc[k + cnt]);
            tc[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
        }
    }
    const int col = ((lane & 32) ? 64 : 0) + 4 * c + ((lane & 16) ? 2 : 0);
    sT[wave][col] = tc[0]; sT[wave][col + 1] = tc[1];
    __syncthreads();
    if (threadIdx.x < 128) {
        const double r2 = ((sT[0][threadIdx.x] + sT[1][threadIdx.x]) + sT[2][threadIdx.x]) + sT[3][threadIdx.x];
        if (MATH == 3) atomicAdd(acc + (blockIdx.x % 59) * 128 + threadIdx.x, (unsigned long long)(long long)rint(r2 * 0x1p40));
        else o2[threadIdx.x] = r2;
    }
}

// ... and a PERSISTENT variant of the last one: G resident workgroups walk their tiles for `passes` iterations inside one launch; after
// a pass every workgroup requests the first tile of its next pass (the matrix does not change), then a grid-wide barrier on a counter
// (agent-scope atomics, bounded spin) lets the accumulators settle.  Does keeping the tile stream running across the iteration
// boundary beat back-to-back launches (whose ramp and end are exposed)?
struct TileRegs { u32x4 ha[8], hb[8], n0, n1; };
__device__ __forceinline__ void tile_issue(TileRegs &r, const u32x4 *p, int wave, int lane, int g, int c) {
    r.n0 = p[4096 + (wave * 64 + lane) * 2]; r.n1 = p[4096 + (wave * 64 + lane) * 2 + 1];
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        const int base = ((wave * 32 + g + 4 * rg) * 128 + 4 * c) / 4;
        r.ha[rg] = p[base]; r.hb[rg] = p[base + 16];
    }
}
__global__ void __launch_bounds__(256, 3) persist_tiles(const u32x4 *__restrict__ src, int64_t chunk16, int nchunks, const double *__restrict__ rhs,
                                                        unsigned long long *acc /* [2][8192] */, unsigned *counter, int passes, unsigned *fail) {
    __shared__ double sI[128], sJ[128], sT[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const int G = gridDim.x;
    TileRegs tr;
    tile_issue(tr, src + (int64_t)blockIdx.x * chunk16, wave, lane, g, c);
    for (int pass = 0; pass < passes; ++pass) {
        unsigned long long *acc_cur = acc + (pass & 1) * 8192;
        const long long *acc_prev = reinterpret_cast<const long long *>(acc + ((pass + 1) & 1) * 8192);
        for (int t = blockIdx.x; t < nchunks; t += G) {
            {   // right-hand side rebuilt from the previous pass's sums (agent-scope loads: other XCDs' atomics)
                const int blk = threadIdx.x < 128 ? (t % 61) : (t % 59), e = blk * 128 + (threadIdx.x & 127);
                const long long a = __hip_atomic_load(acc_prev + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double xv = (double)a * 0x1p-40 + rhs[e], uv = rhs[8192 + e];
                const double vv = xv + uv, zv = vv > 0.1 ? vv - 0.1 : (vv < -0.1 ? vv + 0.1 : 0.0);
                const double r = (2.0 * zv - vv) * 20.0;
                __syncthreads();
                if (threadIdx.x < 128) sI[threadIdx.x] = r; else sJ[threadIdx.x - 128] = r;
                __syncthreads();
            }
            double rj[8], tc[8], v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) tc[k] = 0.0;
            const unsigned nw[8] = {tr.n0.x, tr.n0.y, tr.n0.z, tr.n0.w, tr.n1.x, tr.n1.y, tr.n1.z, tr.n1.w};
#pragma unroll
            for (int rg = 0; rg < 8; ++rg) {
                const double ri = sI[wave * 32 + 4 * rg + g];
                const unsigned hh[8] = {tr.ha[rg].x, tr.ha[rg].y, tr.ha[rg].z, tr.ha[rg].w, tr.hb[rg].x, tr.hb[rg].y, tr.hb[rg].z, tr.hb[rg].w};
                double a0 = 0, a1 = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned top = __builtin_amdgcn_alignbit(0x04330000u, hh[k], 28);
                    unsigned lo = (nw[rg] >> (4 * k)) & 15u;
                    asm("v_lshl_or_b32 %0, %1, 4, %0" : "+v"(lo) : "v"(hh[k]));
                    const double m = __hiloint2double((int)top, (int)lo) - (0x1p52 + 0x1p35);
                    tc[k] = opaque_d(fma(m, ri, tc[k]));
                    if (k & 1) a1 = fma(m, rj[k], a1); else a0 = fma(m, rj[k], a0);
                }
                v[rg] = a0 + a1;
            }
            // the next tile of this pass, or the first tile of the next pass: requested before the reductions
            const int tn = t + G < nchunks ? t + G : blockIdx.x;
            tile_issue(tr, src + (int64_t)tn * chunk16, wave, lane, g, c);
#pragma unroll
            for (int m = 8, cnt = 4; m >= 2; m >>= 1, cnt >>= 1) {
                const bool up = (c & m) != 0;
#pragma unroll
                for (int k = 0; k < cnt; ++k) {
                    const double lo_ = opaque_d(v[k]), hi_ = opaque_d(v[k + cnt]);
                    v[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
                }
            }
            v[0] += __shfl_xor(v[0], 1, 64);
            if ((c & 1) == 0) {
                const int rg = ((c & 8) ? 4 : 0) + ((c & 4) ? 2 : 0) + ((c & 2) ? 1 : 0);
                atomicAdd(acc_cur + (t % 61) * 128 + wave * 32 + 4 * rg + g, (unsigned long long)(long long)rint(v[0] * 0x1p40));
            }
#pragma unroll
            for (int m = 32, cnt = 4; m >= 16; m >>= 1, cnt >>= 1) {
                const bool up = (lane & m) != 0;
#pragma unroll
                for (int k = 0; k < cnt; ++k) {
                    const double lo_ = opaque_d(tc[k]), hi_ = opaque_d(tc[k + cnt]);
                    tc[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
                }
            }
            const int col = ((lane & 32) ? 64 : 0) + 4 * c + ((lane & 16) ? 2 : 0);
            sT[wave][col] = tc[0]; sT[wave][col + 1] = tc[1];
            __syncthreads();
            if (threadIdx.x < 128) {
                const double r2 = ((sT[0][threadIdx.x] + sT[1][threadIdx.x]) + sT[2][threadIdx.x]) + sT[3][threadIdx.x];
                atomicAdd(acc_cur + (t % 59) * 128 + threadIdx.x, (unsigned long long)(long long)rint(r2 * 0x1p40));
            }
        }
        // grid barrier: my adds are performed (release), one arrival per workgroup, bounded spin
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)G * (unsigned)(pass + 1);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 2000000 || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
}

int main() {
    const int64_t chunk = 74240;                       // one fixed-point tile
    unsigned *sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (double mb : {38.0, 156.5, 268.5, 764.0}) {
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
        for (int pattern = 0; pattern < 2; ++pattern) {
            const int reps = 100;
            for (int pass = 0; pass < 2; ++pass) {
                if (pass) hipEventRecord(e0, 0);
                for (int r = 0; r < (pass ? reps : 5); ++r) {
                    if (pattern == 0) hipLaunchKernelGGL(read_tiles<0>, dim3((unsigned)nchunks), dim3(256), 0, 0, buf, chunk / 16, sink);
                    else hipLaunchKernelGGL(read_tiles<1>, dim3((unsigned)nchunks), dim3(256), 0, 0, buf, chunk / 16, sink);
                }
                if (pass) hipEventRecord(e1, 0);
            }
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%7.1f MB  tile-shaped, one workgroup per tile, %s: %7.2f us per pass  %6.2f TB/s\n", bytes / 1e6,
                   pattern ? "lane-order layout (1 KiB per wave instruction)" : "row layout (4 x 256 B per wave instruction) ", ms * 1e3 / reps,
                   bytes / (ms * 1e-3 / reps) / 1e12);
        }
        {
            double *rhs, *out; hipMalloc(&rhs, 8192 * 8); hipMemset(rhs, 0, 8192 * 8); hipMalloc(&out, nchunks * 256 * 8);
            unsigned long long *acc; hipMalloc(&acc, 2 * 8192 * 8); hipMemset(acc, 0, 2 * 8192 * 8);
            double *rhs3; hipMalloc(&rhs3, 2 * 8192 * 8); hipMemset(rhs3, 0, 2 * 8192 * 8);
            for (int math = 1; math <= 3; ++math) {
                const int reps = 100;
                for (int pass = 0; pass < 2; ++pass) {
                    if (pass) hipEventRecord(e0, 0);
                    for (int r = 0; r < (pass ? reps : 5); ++r) {
                        if (math == 1) hipLaunchKernelGGL(math_tiles<1>, dim3((unsigned)nchunks), dim3(256), 0, 0, buf, chunk / 16, rhs, out);
                        else if (math == 2) hipLaunchKernelGGL(math_tiles<2>, dim3((unsigned)nchunks), dim3(256), 0, 0, buf, chunk / 16, rhs, out);
                        else hipLaunchKernelGGL(math_tiles<3>, dim3((unsigned)nchunks), dim3(256), 0, 0, buf, chunk / 16, rhs3, out, acc + (r & 1) * 8192, reinterpret_cast<const long long *>(acc + ((r + 1) & 1) * 8192));
                    }
                    if (pass) hipEventRecord(e1, 0);
                }
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("%7.1f MB  tile-shaped + the mat-vec's arithmetic (%s): %7.2f us per pass  %6.2f TB/s\n", bytes / 1e6,
                       math == 1 ? "decode as admm.hip: 4 + 2 instructions per element" : math == 2 ? "biased decode: 3 + 2 instructions per element      " : "as the first, partial sums by 64-bit fixed-point global atomics, rhs rebuilt in the prologue", ms * 1e3 / reps,
                       bytes / (ms * 1e-3 / reps) / 1e12);
            }
            if (mb > 150 && mb < 160) {   // the persistent variant at the cfg3 size only
                unsigned *ctr; hipMalloc(&ctr, 8);
                for (int G : {520, 640, 693, 768}) {
                    const int passes = 200;
                    hipMemset(ctr, 0, 8); hipMemset(acc, 0, 2 * 8192 * 8);
                    hipLaunchKernelGGL(persist_tiles, dim3(G), dim3(256), 0, 0, buf, chunk / 16, (int)nchunks, rhs3, acc, ctr, 3, ctr + 1);   // warm
                    hipMemset(ctr, 0, 8);
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(persist_tiles, dim3(G), dim3(256), 0, 0, buf, chunk / 16, (int)nchunks, rhs3, acc, ctr, passes, ctr + 1);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    unsigned h[2]; hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost);
                    printf("%7.1f MB  persistent, %d workgroups, %d passes in one launch, grid barrier per pass: %7.2f us per pass  %6.2f TB/s  (barrier failed: %u)\n",
                           bytes / 1e6, G, passes, ms * 1e3 / passes, bytes / (ms * 1e-3 / passes) / 1e12, h[1]);
                }
                hipFree(ctr);
            }
            hipFree(rhs); hipFree(out);
        }
        hipFree(buf);
    }
    return 0;
}
