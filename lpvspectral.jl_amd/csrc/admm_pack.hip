// admm_pack.hip -- storage of the inverse the ADMM mat-vec streams: the tile-packed lower triangle of M = (G + I/mu)^-1 in doubles, floats
// (_f32 handles), 6-byte float-head elements, or the mixed storage (36-bit fixed-point tiles where a tile's entries are small against max|M|,
// decided per tile when packing: DESIGN.md 4.1), for single problems and window batches; the float <-> double conversions of the _f32 entry points.
#include "lpvs_internal.h"
#include "admm_device.h"
#include "admm_host.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include <cmath>
#include <cstdlib>
#include <string>
#include <type_traits>

namespace lpvs {

namespace {

// single-precision copy of the packed tiles (streamed by the _f32 problems)
__global__ void __launch_bounds__(256)
pack_tiles_f32_kernel(const double *__restrict__ M, int64_t np, float *__restrict__ Mp) {
    int I, J;
    tile_index(blockIdx.x, I, J);
    const double2 *src = reinterpret_cast<const double2 *>(M + (int64_t)I * TS * np + (int64_t)J * TS);
    float2 *dst = reinterpret_cast<float2 *>(Mp + (int64_t)blockIdx.x * TS * TS);
    for (int e = threadIdx.x; e < TS * TS / 2; e += 256) {
        const int r = e / (TS / 2), c = e % (TS / 2);
        const double2 v = src[(int64_t)r * (np / 2) + c];
        dst[e] = make_float2((float)v.x, (float)v.y);
    }
}

__global__ void __launch_bounds__(256) cvt_f32_f64_kernel(const float *__restrict__ src, double *__restrict__ dst, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) dst[i] = (double)src[i];
}
__global__ void __launch_bounds__(256) cvt_f64_f32_kernel(const double *__restrict__ src, float *__restrict__ dst, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) dst[i] = (float)src[i];
}

__global__ void __launch_bounds__(256)
pack_tiles_kernel(const double *__restrict__ Mall, int64_t np, double *__restrict__ Mpall) {
    int I, J;
    tile_index(blockIdx.x, I, J);
    const double *M = Mall + (int64_t)blockIdx.y * np * np;                       // blockIdx.y = problem of a batch
    double *Mp = Mpall + (int64_t)blockIdx.y * gridDim.x * TS * TS;
    const double2 *src = reinterpret_cast<const double2 *>(M + (int64_t)I * TS * np + (int64_t)J * TS);
    double2 *dst = reinterpret_cast<double2 *>(Mp + (int64_t)blockIdx.x * TS * TS);
    for (int e = threadIdx.x; e < TS * TS / 2; e += 256) {
        const int r = e / (TS / 2), c = e % (TS / 2);
        dst[e] = src[(int64_t)r * (np / 2) + c];
    }
}

__global__ void __launch_bounds__(256)
pack_tiles_split_kernel(const double *__restrict__ M, int64_t np, unsigned char *__restrict__ Mp) {
    int I, J;
    tile_index(blockIdx.x, I, J);
    M += (int64_t)blockIdx.y * np * np;                                  // blockIdx.y = problem of a batch
    Mp += (size_t)blockIdx.y * gridDim.x * kSplitTileBytes;
    const double *src = M + (int64_t)I * TS * np + (int64_t)J * TS;
    float *head = reinterpret_cast<float *>(Mp + (size_t)blockIdx.x * kSplitTileBytes);
    unsigned short *tail = reinterpret_cast<unsigned short *>(Mp + (size_t)blockIdx.x * kSplitTileBytes + (size_t)TS * TS * 4);
    for (int e = threadIdx.x; e < TS * TS; e += 256) {
        const int r = e >> 7, col = e & 127;
        const double m = src[(int64_t)r * np + col];
        unsigned long long B = (unsigned long long)__double_as_longlong(m);
        B = (B + (1ull << 12)) & ~((1ull << 13) - 1);                    // round to nearest at bit 13 (carries run into the exponent)
        float h = (float)__longlong_as_double((long long)(B & ~((1ull << 29) - 1)));   // exact: 23 mantissa bits left
        unsigned int q = (unsigned int)(B >> 13) & 0xffffu;
        if (!(fabs(m) >= 0x1p-120) || !(fabs(m) < 0x1p127)) { h = (float)m; q = 0; }   // outside the float range (never for an inverse): plain float
        head[e] = h;
        tail[r * TS + 8 * ((col & 63) >> 2) + 4 * (col >> 6) + (col & 3)] = (unsigned short)q;
    }
}

// max|M| of a symmetric positive definite matrix sits on its diagonal (|m_ij| <= sqrt(m_ii m_jj)): np loads instead of a pass over np^2
// entries (0.2 ms at n = 8192).  Every matrix packed here is the inverse of G + shift I, G a Gram matrix.
__global__ void __launch_bounds__(256)
absmax_kernel(const double *__restrict__ M, int64_t np, unsigned long long *__restrict__ out) {
    M += (int64_t)blockIdx.y * np * np;              // blockIdx.y = matrix of a batch
    out += blockIdx.y;
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < np; i += (int64_t)gridDim.x * 256) m = fmax(m, fabs(M[i * np + i]));
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) m = fmax(m, __shfl_xor(m, w, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));   // positive doubles order like integers
}

// mixed packing: tile blockIdx.x of matrix blockIdx.y -> float-head format or fixed point; types[matrix][tile] says which
__global__ void __launch_bounds__(256)
pack_tiles_mixed_kernel(const double *__restrict__ M, int64_t np, unsigned char *__restrict__ Mp, unsigned char *__restrict__ types,
                        const unsigned long long *__restrict__ absmax_bits, double step_scale, int diag_float /* diagonal tiles always in the float-head format */,
                        double *__restrict__ abs1 /* [tile][128] sums of |m| over the tile's rows, or nullptr */, double *__restrict__ abs2 /* ... over its columns */,
                        int fix_drop_bits /* 0; experiments: low bits of the 36 set to zero */) {
    int I, J;
    tile_index(blockIdx.x, I, J);
    M += (int64_t)blockIdx.y * np * np;              // blockIdx.y = matrix of a batch
    Mp += (size_t)blockIdx.y * gridDim.x * kSplitTileBytes;
    types += (size_t)blockIdx.y * gridDim.x;
    absmax_bits += blockIdx.y;
    const double *src = M + (int64_t)I * TS * np + (int64_t)J * TS;
    unsigned char *slot = Mp + (size_t)blockIdx.x * kSplitTileBytes;
    __shared__ float rowstep[TS];
    __shared__ int bad;
    __shared__ double colabs[4][TS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) bad = (diag_float && I == J) ? 1 : 0;
    __syncthreads();
    {
        const double limit = __longlong_as_double((long long)*absmax_bits) * step_scale;     // largest admissible step
        double c0 = 0.0, c1 = 0.0;                                    // this lane's two columns over the wave's rows
        for (int r = wave; r < TS; r += 4) {
            const double a0 = fabs(src[(int64_t)r * np + lane]), a1 = fabs(src[(int64_t)r * np + 64 + lane]);
            if (abs1 != nullptr) {
                c0 += a0; c1 += a1;
                const double rs = wave_sum(a0 + a1);
                if (lane == 0) abs1[(size_t)blockIdx.x * TS + r] = rs;
            }
            const double e0 = (I == J && lane == r) ? 0.0 : a0;          // (a diagonal tile: without its diagonal)
            const double e1 = (I == J && 64 + lane == r) ? 0.0 : a1;
            double m = fmax(e0, e1);
#pragma unroll
            for (int w = 32; w >= 1; w >>= 1) m = fmax(m, __shfl_xor(m, w, 64));
            if (lane == 0) {
                int e = 0;
                (void)frexp(m, &e);                                   // m < 2^e
                const double st = ldexp(1.0, e - 35);
                rowstep[r] = m > 0.0 ? (float)st : 0x1p-100f;
                if (m > 0.0 && (!(st <= limit) || e - 35 < -120)) atomicOr(&bad, 1);
            }
        }
        if (abs1 != nullptr) { colabs[wave][lane] = c0; colabs[wave][64 + lane] = c1; }
    }
    __syncthreads();
    if (abs2 != nullptr && threadIdx.x < TS)
        abs2[(size_t)blockIdx.x * TS + threadIdx.x] = ((colabs[0][threadIdx.x] + colabs[1][threadIdx.x]) + colabs[2][threadIdx.x]) + colabs[3][threadIdx.x];
    const bool fixed = !bad;
    if (threadIdx.x == 0) types[blockIdx.x] = fixed ? (I == J ? 2 : 1) : 0;
    if (!fixed) {
        float *head = reinterpret_cast<float *>(slot);
        unsigned short *tail = reinterpret_cast<unsigned short *>(slot + (size_t)TS * TS * 4);
        for (int e = threadIdx.x; e < TS * TS; e += 256) {
            const int r = e >> 7, col = e & 127;
            const double m = src[(int64_t)r * np + col];
            unsigned long long B = (unsigned long long)__double_as_longlong(m);
            B = (B + (1ull << 12)) & ~((1ull << 13) - 1);                    // round to nearest at bit 13 (carries run into the exponent)
            float h = (float)__longlong_as_double((long long)(B & ~((1ull << 29) - 1)));   // exact: 23 mantissa bits left
            unsigned int q = (unsigned int)(B >> 13) & 0xffffu;
            if (!(fabs(m) >= 0x1p-120) || !(fabs(m) < 0x1p127)) { h = (float)m; q = 0; }   // outside the float range (never for an inverse): plain float
            head[e] = h;
            tail[r * TS + 8 * ((col & 63) >> 2) + 4 * (col >> 6) + (col & 3)] = (unsigned short)q;
        }
        return;
    }
    const double drop = (double)(1 << fix_drop_bits), drop_inv = 1.0 / drop;
    unsigned int *hi = reinterpret_cast<unsigned int *>(slot);
    unsigned int *nib = reinterpret_cast<unsigned int *>(slot + kFixHeadBytes);
    float *steps = reinterpret_cast<float *>(slot + kFixHeadBytes + kFixNibBytes);
    // thread = (wave w, lane (g, c)): the eight row groups of its rows, eight columns each
    const int g = lane >> 4, c = lane & 15;
    for (int rg = 0; rg < 8; ++rg) {
        const int r = wave * 32 + 4 * rg + g;
        const double inv = 1.0 / (double)rowstep[r];                 // power of two: exact
        unsigned int word = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int col = k < 4 ? 4 * c + k : 64 + 4 * c + (k - 4);
            double qd = (I == J && col == r) ? 0.0 : rint(src[(int64_t)r * np + col] * inv * drop_inv) * drop;   // (drop > 1: LPVS_FIX_BITS experiments, fewer significant bits in the same format)
            qd = fmin(fmax(qd, -0x1p35 + 1.0), 0x1p35 - 1.0);
            const unsigned long long q = (unsigned long long)((long long)qd + (1ll << 35));   // biased: 0 < q < 2^36
            hi[r * TS + col] = (unsigned int)(q >> 4);
            word |= (unsigned int)(q & 15) << (4 * k);
        }
        nib[(wave * 64 + lane) * 8 + rg] = word;
        if (c == 0) steps[(wave * 4 + g) * 8 + rg] = rowstep[r];
    }
    if (I == J && threadIdx.x < TS)
        reinterpret_cast<double *>(slot + kFixHeadBytes + kFixNibBytes + TS * 4)[threadIdx.x] = src[(int64_t)threadIdx.x * np + threadIdx.x];
}

}  // namespace

int32_t launch_pack_tiles_batch(const double *M, int64_t np, int nbatch, double *Mp, hipStream_t s) {
    const int nblk = (int)(np / TS);
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned)(nblk * (nblk + 1) / 2), (unsigned)nbatch), dim3(256), 0, s, M, np, Mp);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

size_t symv_part_doubles(int64_t np, int64_t ns) {
    const int64_t nblk = np / TS;
    return ((size_t)(nblk * (nblk + 1) / 2) * TS * 2 + 2 * (size_t)nblk + 2) * (size_t)ns + 16;   // part1, part2, block norms (two parities), tickets
}
size_t symv_packed_doubles(int64_t np) {
    const int64_t nblk = np / TS;
    return (size_t)(nblk * (nblk + 1) / 2) * TS * TS;
}

int32_t launch_pack_tiles_f32(const double *M, int64_t np, float *Mp, hipStream_t s) {
    const int nblk = (int)(np / TS);
    hipLaunchKernelGGL(pack_tiles_f32_kernel, dim3((unsigned)(nblk * (nblk + 1) / 2)), dim3(256), 0, s, M, np, Mp);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_cvt_f32_f64(const float *src, double *dst, int64_t count, hipStream_t s) {
    if (count <= 0) return LPVS_OK;
    hipLaunchKernelGGL(cvt_f32_f64_kernel, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, s, src, dst, count);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_cvt_f64_f32(const double *src, float *dst, int64_t count, hipStream_t s) {
    if (count <= 0) return LPVS_OK;
    hipLaunchKernelGGL(cvt_f64_f32_kernel, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, s, src, dst, count);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_pack_tiles_split_batch(const double *M, int64_t np, int nbatch, unsigned char *Mp, hipStream_t s) {
    const int nblk = (int)(np / TS);
    hipLaunchKernelGGL(pack_tiles_split_kernel, dim3((unsigned)(nblk * (nblk + 1) / 2), (unsigned)nbatch), dim3(256), 0, s, M, np, Mp);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

// mixed packing (single matrix): types = ntiles bytes after the tile slots; absmax = 8 bytes of device scratch
__global__ void __launch_bounds__(256)
absmax_vec_kernel(const double *__restrict__ v, int64_t n, unsigned long long *__restrict__ out) {   // max |v_i|, i < n: one workgroup
    double m = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) m = fmax(m, fabs(v[i]));
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) m = fmax(m, __shfl_xor(m, w, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}
// abs_part (symv_part_doubles(np) doubles of scratch: the handle's tile-partial buffer) != nullptr: the tiles' absolute row / column sums are
// gathered into the largest absolute row sum over rows < n_valid, left behind max|M| in absmax[1] (bit pattern of a non-negative double)
int32_t launch_pack_tiles_mixed(const double *M, int64_t np, unsigned char *Mp, unsigned char *types, unsigned long long *absmax, hipStream_t s,
                                bool diag_float, double *abs_part, int64_t n_valid, double *rows_scratch, int fix_bits) {
    return launch_pack_tiles_mixed_batch(M, np, 1, Mp, types, absmax, s, diag_float, abs_part, n_valid, rows_scratch, fix_bits);
}

// ... of nbatch matrices: types = [nbatch][ntiles] bytes, absmax = nbatch * 8 bytes of device scratch
int32_t launch_pack_tiles_mixed_batch(const double *M, int64_t np, int nbatch, unsigned char *Mp, unsigned char *types, unsigned long long *absmax,
                                      hipStream_t s, bool diag_float, double *abs_part, int64_t n_valid, double *rows_scratch, int fix_bits) {
    const int nblk = (int)(np / TS);
    LPVS_HIP(hipMemsetAsync(absmax, 0, sizeof(unsigned long long) * ((size_t)nbatch + (abs_part ? 1 : 0)), s));
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<int64_t>(16, ceil_div(np, 256)), (unsigned)nbatch), dim3(256), 0, s, M, np, absmax);
    const double step_scale = 0x1p-44 * std::sqrt(8192.0 / (double)np);
    hipLaunchKernelGGL(pack_tiles_mixed_kernel, dim3((unsigned)(nblk * (nblk + 1) / 2), (unsigned)nbatch), dim3(256), 0, s, M, np, Mp, types, absmax, step_scale, diag_float ? 1 : 0,
                       abs_part, abs_part ? abs_part + (size_t)(nblk * (nblk + 1) / 2) * TS : nullptr,
                       fix_bits >= 20 && fix_bits < 36 ? 36 - fix_bits : 0);
    if (abs_part != nullptr && rows_scratch != nullptr && nbatch == 1) {   // R = the largest absolute row sum over the valid rows, from the tiles' |m| sums (the quantum bound of the one-launch iteration)
        const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2);
        double *rows = rows_scratch;                                       // np doubles
        launch_gather_tile_partials(abs_part, abs_part + (size_t)ntiles * TS, nblk, ntiles, np, rows, s);   // (admm.hip: the gather of the two-launch iteration)
        hipLaunchKernelGGL(absmax_vec_kernel, dim3(1), dim3(256), 0, s, rows, n_valid, absmax + 1);
    }
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_pack_tiles_split(const double *M, int64_t np, unsigned char *Mp, hipStream_t s) {
    const int nblk = (int)(np / TS);
    hipLaunchKernelGGL(pack_tiles_split_kernel, dim3((unsigned)(nblk * (nblk + 1) / 2)), dim3(256), 0, s, M, np, Mp);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_pack_tiles(const double *M, int64_t np, double *Mp, hipStream_t s) {
    const int nblk = (int)(np / TS);
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned)(nblk * (nblk + 1) / 2)), dim3(256), 0, s, M, np, Mp);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

}  // namespace lpvs
