// linalg.hip -- dense SPD inverse on device: M = (G + I/mu)^-1, the factor-once replacement of
// the reference's per-iteration CG solve (ProximalOperators LeastSquares/Quadratic iterative=true,
// call sites src/lasso.jl:51,98,121,151).
//
// Blocked symmetric sweep (Gauss-Jordan without pivoting, valid for SPD): for every 64-wide pivot
// block k:  P = A_kk^-1;  B = A[:,k] (block k zeroed);  C = B P;  A -= C B' (all blocks != k);
// A[:,k] = C, A[k,:] = C', A_kk = -P.  After all sweeps A = -A0^-1.  The rank-64 trailing update
// is an f64 MFMA contraction over the lower triangle (n^3 flop in total).  Only the lower triangle is
// kept current during the sweeps (row panels are read transposed from it); one pass at the end
// negates and mirrors it, so the result is exactly symmetric.
#include "lpvs_internal.h"

namespace lpvs {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int NB = 64;

// P = inv(A_kk) by Gauss-Jordan without pivoting (SPD); status |= 1 on a non-positive pivot.
// 256 threads, each owning a 4x4 sub-block in registers; per pivot only the pivot row and column travel
// through LDS (two barriers), instead of the whole 64x64 block.
__global__ void __launch_bounds__(256)
diag_inverse_kernel(const double *__restrict__ Aall, int64_t np, int k, double *__restrict__ Pall, int *status,
                    int64_t strideA, int64_t strideW) {
    __shared__ double colp[NB], rowp[NB];
    const double *A = Aall + (int64_t)blockIdx.x * strideA;     // blockIdx.x = problem of a batch
    double *P = Pall + (int64_t)blockIdx.x * strideW;
    const double *blk = A + (int64_t)k * NB * np + (int64_t)k * NB;
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;     // owns rows 4ti..4ti+3, cols 4tj..4tj+3
    double v[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {                            // only the lower triangle of A is current
            const int i = 4 * ti + a, j = 4 * tj + b;
            v[a][b] = j <= i ? blk[(int64_t)i * np + j] : blk[(int64_t)j * np + i];
        }
    for (int pb = 0; pb < NB / 4; ++pb) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int p = 4 * pb + pp;
            if (tj == pb) {
#pragma unroll
                for (int a = 0; a < 4; ++a) colp[4 * ti + a] = v[a][pp];
            }
            if (ti == pb) {
#pragma unroll
                for (int b = 0; b < 4; ++b) rowp[4 * tj + b] = v[pp][b];
            }
            __syncthreads();
            const double d = colp[p];
            if (threadIdx.x == 0 && !(d > 0)) atomicOr(status + blockIdx.x, 1);
            const double inv = 1.0 / d;
            double cp[4], rp[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { cp[a] = colp[4 * ti + a]; rp[a] = rowp[4 * tj + a] * inv; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int i = 4 * ti + a, j = 4 * tj + b;
                    double nv;
                    if (i == p && j == p) nv = inv;
                    else if (i == p) nv = rp[b];
                    else if (j == p) nv = -cp[a] * inv;
                    else nv = v[a][b] - cp[a] * rp[b];
                    v[a][b] = nv;
                }
            __syncthreads();
        }
    }
    // symmetrise (P is symmetric up to rounding): exchange with the transposed owner through global memory
    __shared__ double T[NB][NB + 1];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) T[4 * ti + a][4 * tj + b] = v[a][b];
    __syncthreads();
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        P[e] = 0.5 * (T[i][j] + T[j][i]);
    }
}

// One workgroup per 64-row block i: B_i = A[i,k] (taken from the lower triangle: A[k,i]' for i < k; zero for
// i == k), C_i = B_i P; panels are stored operand-major: the 16 rows x 4 k values one MFMA operand needs are 64
// contiguous doubles (element (r,c) at ((r/16)*16 + c/4)*64 + (c%4)*16 + r%16), so a wave's operand load is
// one coalesced 512-byte read.  Writes back C_i into the lower triangle (A[i,k] or A[k,i]') and A[k,k] = -P.
__global__ void __launch_bounds__(256)
panel_kernel(double *__restrict__ Aall, int64_t np, int k, const double *__restrict__ Pall,
             double *__restrict__ Bpall, double *__restrict__ Cpall, int64_t strideA, int64_t strideW) {
    __shared__ double sB[NB][NB + 1], sP[NB][NB + 1];
    double *A = Aall + (int64_t)blockIdx.y * strideA;
    const double *P = Pall + (int64_t)blockIdx.y * strideW;
    double *Bp = Bpall + (int64_t)blockIdx.y * strideW, *Cp = Cpall + (int64_t)blockIdx.y * strideW;
    const int i = blockIdx.x;
    const int64_t r0 = (int64_t)i * NB, k0 = (int64_t)k * NB;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;       // c fastest: coalesced reads of either orientation
        double v = 0.0;
        if (i > k) v = A[(r0 + r) * np + k0 + c];
        if (i > k) sB[r][c] = v;
        if (i < k) sB[c][r] = A[(k0 + r) * np + r0 + c];   // B_i[c][r] = A[k0+r][r0+c]
        if (i == k) sB[r][c] = 0.0;
        sP[r][c] = P[e];
    }
    __syncthreads();
    double res[NB * NB / 256];
#pragma unroll
    for (int u = 0; u < NB * NB / 256; ++u) {
        const int e = threadIdx.x + 256 * u;
        const int r = e % NB, c = e / NB;  // r fastest: coalesced panel stores
        double s = 0;
        for (int q = 0; q < NB; ++q) s = fma(sB[r][q], sP[q][c], s);
        const int64_t po = (((r0 + r) >> 4) * 16 + (c >> 2)) * 64 + (c & 3) * 16 + ((r0 + r) & 15);
        Bp[po] = sB[r][c];
        Cp[po] = s;
        res[u] = (i == k) ? -sP[r][c] : s;
    }
    __syncthreads();                       // every product has consumed sB before it is overwritten
#pragma unroll
    for (int u = 0; u < NB * NB / 256; ++u) {
        const int e = threadIdx.x + 256 * u;
        sB[e % NB][e / NB] = res[u];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;       // c fastest: coalesced stores
        if (i >= k) A[(r0 + r) * np + k0 + c] = sB[r][c];
        else A[(k0 + r) * np + r0 + c] = sB[c][r];
    }
}

// A[i,j] -= C_i B_j' for 64-blocks i >= j, i != k, j != k; result mirrored to A[j,i].
// Workgroup = 4 waves on a 128x128 tile of the lower triangle; wave (wi,wj) owns one 64x64 block.
__global__ void __launch_bounds__(256)
sweep_update_kernel(double *__restrict__ Aall, int64_t np, int k, const double *__restrict__ Bpall,
                    const double *__restrict__ Cpall, int64_t strideA, int64_t strideW) {
    double *A = Aall + (int64_t)blockIdx.y * strideA;
    const double *Bp = Bpall + (int64_t)blockIdx.y * strideW, *Cp = Cpall + (int64_t)blockIdx.y * strideW;
    // linear lower-triangle tile index -> (ti, tj), ti >= tj
    const int t = blockIdx.x;
    int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bi = ti * 2 + (wave >> 1), bj = tj * 2 + (wave & 1);
    if (bi < bj || bi == k || bj == k) return;
    const int li = lane & 15, lk = lane >> 4;
    // operand q (16 rows) of k-step kk sits at ((block*4 + q)*16 + kk)*64 + lane
    const double *cbase = Cp + ((int64_t)bi * 4 * 16) * 64 + lane;
    const double *bbase = Bp + ((int64_t)bj * 4 * 16) * 64 + lane;
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    double nA[4], nB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { nA[q] = cbase[(q * 16 + 0) * 64]; nB[q] = bbase[(q * 16 + 0) * 64]; }
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
        double opA[4], opB[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { opA[q] = nA[q]; opB[q] = nB[q]; }
        if (kk + 1 < NB / 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { nA[q] = cbase[(q * 16 + kk + 1) * 64]; nB[q] = bbase[(q * 16 + kk + 1) * 64]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[i], opB[j], acc[i][j], 0, 0, 0);
    }
    const int64_t r0 = (int64_t)bi * NB, c0 = (int64_t)bj * NB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = r0 + i * 16 + lk + 4 * r, col = c0 + j * 16 + li;
                if (bi == bj && col > row) continue;  // only the lower triangle is maintained
                A[row * np + col] -= acc[i][j][r];
            }
}

// diag(M) += shift on the valid part; the pad block becomes the identity
__global__ void __launch_bounds__(256) add_diag_kernel(double *__restrict__ Mall, int64_t np, int64_t n, double shift) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double *M = Mall + (int64_t)blockIdx.y * np * np;
    if (i < np) M[i * np + i] = i < n ? M[i * np + i] + shift : 1.0;
}

// A <- -A on the lower triangle, mirrored into the upper one (32x32 LDS-transposed tiles, both sides coalesced)
__global__ void __launch_bounds__(256) negate_mirror_kernel(double *__restrict__ Aall, int64_t np) {
    __shared__ double tile[32][33];
    double *A = Aall + (int64_t)blockIdx.z * np * np;
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)bi * 32, c0 = (int64_t)bj * 32;
    for (int r = ty; r < 32; r += 8) {
        const double v = -A[(r0 + r) * np + c0 + tx];
        tile[r][tx] = v;
        if (bi != bj || tx <= r) A[(r0 + r) * np + c0 + tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8)   // upper block (bj, bi): element (c0+r, r0+tx) = tile[tx][r]
        if (bi != bj || tx > r) A[(c0 + r) * np + r0 + tx] = tile[tx][r];
}

// C = A * B, all np x np, A and B symmetric (so row-major == column-major); diagnostics only.
__global__ void __launch_bounds__(256)
symm_matmul_kernel(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C, int64_t np) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.y * 2 + (wave >> 1)) * 64, c0 = ((int64_t)blockIdx.x * 2 + (wave & 1)) * 64;
    const int li = lane & 15, lk = lane >> 4;
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int64_t kk = 0; kk < np; kk += 4) {
        double opA[4], opB[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            opA[q] = A[(kk + lk) * np + r0 + q * 16 + li];  // A[r][k] = A[k][r]
            opB[q] = B[(kk + lk) * np + c0 + q * 16 + li];  // B[k][c]
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[i], opB[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(r0 + i * 16 + lk + 4 * r) * np + c0 + j * 16 + li] = acc[i][j][r];
}

}  // namespace

size_t spd_inverse_work_bytes(int64_t np) { return sizeof(double) * (size_t)(2 * np * NB + NB * NB); }

// nbatch independent matrices A + q*np*np; work holds nbatch * spd_inverse_work_bytes(np); status_dev nbatch ints
int32_t spd_inverse_inplace_batch(double *A, int64_t np, int nbatch, double *work, int *status_dev, hipStream_t s) {
    if (np % 128 != 0) { set_error("spd_inverse: np=%lld not a multiple of 128", (long long)np); return LPVS_ESTATE; }
    const int64_t strideW = (int64_t)(2 * np * NB + NB * NB), strideA = np * np;
    double *Bp = work, *Cp = work + np * NB, *P = work + 2 * np * NB;
    const int nblk = (int)(np / NB);
    const int nt = (int)(np / 128);
    const unsigned ntiles = (unsigned)(nt * (nt + 1) / 2);
    LPVS_HIP(hipMemsetAsync(status_dev, 0, sizeof(int) * (size_t)nbatch, s));
    for (int k = 0; k < nblk; ++k) {
        hipLaunchKernelGGL(diag_inverse_kernel, dim3((unsigned)nbatch), dim3(256), 0, s, A, np, k, P, status_dev, strideA, strideW);
        hipLaunchKernelGGL(panel_kernel, dim3((unsigned)nblk, (unsigned)nbatch), dim3(256), 0, s, A, np, k, P, Bp, Cp, strideA, strideW);
        hipLaunchKernelGGL(sweep_update_kernel, dim3(ntiles, (unsigned)nbatch), dim3(256), 0, s, A, np, k, Bp, Cp, strideA, strideW);
    }
    LPVS_HIP(hipGetLastError());
    hipLaunchKernelGGL(negate_mirror_kernel, dim3((unsigned)(np / 32), (unsigned)(np / 32), (unsigned)nbatch), dim3(256), 0, s, A, np);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t spd_inverse_inplace(double *A, int64_t np, double *work, int *status_dev, hipStream_t s) {
    return spd_inverse_inplace_batch(A, np, 1, work, status_dev, s);
}

int32_t launch_add_diag_batch(double *M, int64_t np, int64_t n, double shift, int nbatch, hipStream_t s) {
    hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)ceil_div(np, 256), (unsigned)nbatch), dim3(256), 0, s, M, np, n, shift);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_add_diag(double *M, int64_t np, int64_t n, double shift, hipStream_t s) {
    hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)ceil_div(np, 256)), dim3(256), 0, s, M, np, n, shift);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_symm_matmul(const double *A, const double *B, double *C, int64_t np, hipStream_t s) {
    if (np % 128 != 0) return LPVS_ESTATE;
    hipLaunchKernelGGL(symm_matmul_kernel, dim3((unsigned)(np / 128), (unsigned)(np / 128)), dim3(256), 0, s, A, B, C, np);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

}  // namespace lpvs
