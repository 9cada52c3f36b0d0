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

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

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


// ---- two-level sweep (one large matrix) ---------------------------------------------------------------------
// The 64-wide sweep above streams the whole lower triangle once per 64 pivots (128 read-modify-write passes over
// 268 MB at np = 8192: HBM-bound, ~17 TFLOP/s).  For large matrices the same sweep is applied with 128-wide pivot
// blocks (LPVS_KW=256: 256-wide, pivot block inverted by the 64-wide sweep): the pivot block is inverted by one
// workgroup in registers, the panel C = B P is one small MFMA GEMM, and the trailing update A -= C B' has depth 128
// (half the passes over A, MFMA-bound, LDS-DMA staged like gram.hip).
constexpr int KW = 256;                      // largest outer pivot width (sizes the panel buffers)
constexpr int RU_TM = 128, RU_TN = 128;      // trailing-update tile: rows x cols
constexpr int RU_BK = 16;                    // pivots per LDS stage
constexpr int RU_WN = RU_TN / 64;            // waves along the columns
constexpr int RU_THREADS = 64 * 2 * RU_WN;   // 2 (rows) x RU_WN (cols) waves, 64x64 outputs each; 64 KiB of LDS -> two workgroups per CU

__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// P[r][c] = A[k0+r][k0+c] (only the lower triangle is meaningful, which is all the 64-wide sweep reads)
__global__ void __launch_bounds__(256)
pivot_extract_kernel(const double *__restrict__ A, int64_t np, int64_t k0, int kw, double *__restrict__ P) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < kw; c += 256) P[(int64_t)r * kw + c] = A[(k0 + r) * np + k0 + c];
}


// P = inv(A_kk) for a W x W pivot block (W = 256 or 128) in ONE workgroup: the symmetric sweep operator applied W
// times to the lower triangle held in registers.  The block is cut into 8x8 sub-blocks; thread t owns sub-block
// (ti, tj), ti >= tj, of the lower triangle (G(G+1)/2 threads, G = W/8).  Per pivot p the symmetric column
// w[i] = A[i][p] is published through LDS by the owners of block column / block row p/8 (two barriers), then
//     A[i][j] -= w[i] w[j] / d   (i, j != p),   A[i][p] = w[i] / d,   A[p][p] = -1/d.
// After W sweeps the registers hold -inv(A_kk); it is negated and mirrored on the way out.  status |= 1 on a
// non-positive pivot (the remaining Schur complement of an SPD matrix stays positive definite).
template <int W>
__global__ void __launch_bounds__(((W / 8) * (W / 8 + 1) / 2 + 63) / 64 * 64)
pivot_inverse_kernel(const double *__restrict__ A, int64_t np, int64_t k0, double *__restrict__ P, int *status) {
    constexpr int G = W / 8, NBLK = G * (G + 1) / 2;
    // the published pivot column is double-buffered by pivot parity: ONE barrier per pivot (a thread that is done reading pivot p's
    // column publishes pivot p+1's into the other buffer; nobody overwrites buffer p & 1 before the barrier of pivot p+1, which a
    // thread reaches only after its reads of pivot p)
    __shared__ double wbuf[2][W];
    const int t = threadIdx.x;
    const bool live = t < NBLK;
    int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    int tj = t - ti * (ti + 1) / 2;
    if (!live) { ti = G - 1; tj = 0; }        // idle lanes of the last wave: harmless duplicates that never publish or store
    const double *blk = A + (k0 + 8 * ti) * np + k0 + 8 * tj;
    double v[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) v[a][b] = blk[(int64_t)a * np + b];
    if (ti == tj) {   // diagonal sub-blocks: only the lower triangle of A is current
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = a + 1; b < 8; ++b) v[a][b] = v[b][a];
    }
    for (int pb = 0; pb < G; ++pb) {
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int p = 8 * pb + pp;
            double *w = wbuf[pp & 1];            // (8 pivots per pb: the parity of p is the parity of pp)
            if (live && tj == pb) {            // block column pb: rows 8ti .. 8ti+7 of column p
#pragma unroll
                for (int a = 0; a < 8; ++a) w[8 * ti + a] = v[a][pp];
            } else if (live && ti == pb) {     // block row pb (tj < pb): columns 8tj .. 8tj+7 of row p
#pragma unroll
                for (int b = 0; b < 8; ++b) w[8 * tj + b] = v[pp][b];
            }
            __syncthreads();
            const double d = w[p];
            if (t == 0 && !(d > 0)) atomicOr(status, 1);
            const double inv = 1.0 / d;
            double rj[8], wi[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) { rj[b] = w[8 * tj + b] * inv; wi[b] = w[8 * ti + b]; }
#pragma unroll
            for (int a = 0; a < 8; ++a) {
#pragma unroll
                for (int b = 0; b < 8; ++b) v[a][b] = fma(-wi[a], rj[b], v[a][b]);
            }
            if (ti == pb) {                    // row p of the block
#pragma unroll
                for (int b = 0; b < 8; ++b) v[pp][b] = rj[b];
            }
            if (tj == pb) {                    // column p of the block
#pragma unroll
                for (int a = 0; a < 8; ++a) v[a][pp] = wi[a] * inv;
            }
            if (ti == pb && tj == pb) v[pp][pp] = -inv;
        }
    }
    if (!live) return;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int i = 8 * ti + a, j = 8 * tj + b;
            if (j > i) continue;
            const double o = -v[a][b];
            P[(int64_t)i * W + j] = o;
            P[(int64_t)j * W + i] = o;
        }
}

// ---- the same inverse on the matrix cores: 16 x 16 sub-blocks, the lower triangle's 36 blocks as accumulators in registers -------
// The register kernel above spends ~1.1 us per pivot (two LDS round trips and 64 dependent FMAs per thread and pivot: 147 us for
// 128 pivots once the trailing update keeps every other CU's matrix pipe busy).  Here the SAME blocked sweep that the large matrix
// uses is applied inside the pivot block with 16-wide sub-blocks: per sub-step k
//   publish the column panel B = S[:, k] and the pivot sub-block D = S_kk (LDS, 17-double rows: conflict-free operand reads),
//   P = D^-1 by ONE wave (16 symmetric sweeps through lane shuffles: lane (r, q) holds D[r][4q .. 4q+3]),
//   C = B P on the owners of the panel's blocks (4 MFMAs each; block (k, j), j < k, takes the transposed product P B_j'),
//   S_ij -= C_i B_j' on the owners of the other blocks (4 MFMAs each),  S_kk = -P.
// Wave w owns blocks b = w, w + 4, ... (b = i (i + 1) / 2 + j): nine f64x4 accumulators that never leave its registers.  ~140
// matrix instructions and one 16 x 16 inverse per sub-step instead of 16 x (2 barriers + 64 FMAs).  After 8 sub-steps the blocks
// hold -inv(A_kk); negated and mirrored on the way out (exactly symmetric).  status |= 1 on a non-positive pivot.
__global__ void __launch_bounds__(256)
pivot_inverse_mfma_kernel(const double *__restrict__ A, int64_t np, int64_t k0, double *__restrict__ P, int *status) {
    constexpr int W = 128, NB16 = W / 16, PS = 17, NOWN = 9;
    __shared__ double Bp[W * PS], Cn[W * PS], Db[16 * PS], Pb[16 * PS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    f64x4 acc[NOWN];
    int bi[NOWN], bj[NOWN];
#pragma unroll
    for (int n = 0; n < NOWN; ++n) {
        const int b = wave + 4 * n;
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= b) ++i;                       // (scalar: b is wave-uniform)
        bi[n] = i; bj[n] = b - i * (i + 1) / 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = 16 * bi[n] + lk + 4 * r, col = 16 * bj[n] + li;
            if (col > row) { const int t = row; row = col; col = t; }   // diagonal blocks: only the lower triangle of A is current
            acc[n][r] = A[(k0 + row) * np + k0 + col];
        }
    }
    for (int k = 0; k < NB16; ++k) {
        // ---- publish the column panel B[R][q] = S[R][16k + q] (R outside the pivot rows) and the pivot sub-block
#pragma unroll
        for (int n = 0; n < NOWN; ++n) {
            if (bi[n] == k && bj[n] == k) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Db[(lk + 4 * r) * PS + li] = acc[n][r];
            } else if (bj[n] == k) {                                  // block (i, k), i > k: rows of the panel as they stand
#pragma unroll
                for (int r = 0; r < 4; ++r) Bp[(16 * bi[n] + lk + 4 * r) * PS + li] = acc[n][r];
            } else if (bi[n] == k) {                                  // block (k, j), j < k: S[16k + q][16j + c] = B[16j + c][q]
#pragma unroll
                for (int r = 0; r < 4; ++r) Bp[(16 * bj[n] + li) * PS + lk + 4 * r] = acc[n][r];
            }
        }
        __syncthreads();
        if (wave == 0) {
            // P = D^-1: the symmetric sweep operator applied 16 times, lane (r = lane & 15, qg = lane >> 4) holds D[r][4qg .. 4qg+3]
            const int r = li, qg = lk;
            double e[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) e[m] = Db[r * PS + 4 * qg + m];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int ps = p & 3, pq = p >> 2;
                double prow[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) prow[m] = __shfl(e[m], p + 16 * qg, 64);   // D[p][4qg + m]
                const double pcol = __shfl(e[ps], r + 16 * pq, 64);                     // D[r][p]
                const double d = __shfl(e[ps], p + 16 * pq, 64);                        // D[p][p]
                if (lane == 0 && !(d > 0)) atomicOr(status, 1);
                const double inv = 1.0 / d, ci = pcol * inv;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int c = 4 * qg + m;
                    double nv = fma(-ci, prow[m], e[m]);
                    if (r == p) nv = prow[m] * inv;
                    if (c == p) nv = ci;
                    if (r == p && c == p) nv = -inv;
                    e[m] = nv;
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) Pb[r * PS + 4 * qg + m] = -e[m];                // the sweeps leave -D^-1
        }
        __syncthreads();
        // ---- the panel's blocks: C = B P (and S_kk = -P); the negated panel Cn[R][q] = -C[R][q] goes to LDS for the update
#pragma unroll
        for (int n = 0; n < NOWN; ++n) {
            if (bi[n] == k && bj[n] == k) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[n][r] = -Pb[(lk + 4 * r) * PS + li];
            } else if (bj[n] == k) {                                  // (i, k): C_i[r][c] = sum_q B[16i + r][q] P[q][c]
                f64x4 c = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(Bp[(16 * bi[n] + li) * PS + 4 * kk + lk], Pb[(4 * kk + lk) * PS + li], c, 0, 0, 0);
                acc[n] = c;
#pragma unroll
                for (int r = 0; r < 4; ++r) Cn[(16 * bi[n] + lk + 4 * r) * PS + li] = -c[r];
            } else if (bi[n] == k) {                                  // (k, j): new[r][c] = C[16j + c][r] = sum_q P[r][q] B[16j + c][q]
                f64x4 c = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(Pb[li * PS + 4 * kk + lk], Bp[(16 * bj[n] + li) * PS + 4 * kk + lk], c, 0, 0, 0);
                acc[n] = c;
#pragma unroll
                for (int r = 0; r < 4; ++r) Cn[(16 * bj[n] + li) * PS + lk + 4 * r] = -c[r];
            }
        }
        __syncthreads();
        // ---- every other block: S_ij += (-C_i) B_j'
#pragma unroll
        for (int n = 0; n < NOWN; ++n) {
            if (bi[n] != k && bj[n] != k) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(Cn[(16 * bi[n] + li) * PS + 4 * kk + lk], Bp[(16 * bj[n] + li) * PS + 4 * kk + lk], acc[n], 0, 0, 0);
            }
        }
        __syncthreads();                                              // the panels are rewritten by the next sub-step
    }
#pragma unroll
    for (int n = 0; n < NOWN; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * bi[n] + lk + 4 * r, col = 16 * bj[n] + li;
            if (col > row) continue;
            const double o = -acc[n][r];
            P[(int64_t)row * W + col] = o;
            P[(int64_t)col * W + row] = o;
        }
}

// k-major panel Bk[c][R] = B[R][c]: B = A[:, pivot columns] read from the lower triangle, pivot rows and pad columns zero
__global__ void __launch_bounds__(256)
panel_gather_kernel(const double *__restrict__ A, int64_t np, int64_t ldp, int64_t k0, int kw, double *__restrict__ Bk) {
    __shared__ double tile[64][65];
    const int64_t R0 = (int64_t)blockIdx.x * 64;
    if (R0 >= np || (R0 >= k0 && R0 < k0 + kw)) {
        for (int e = threadIdx.x; e < kw * 64; e += 256) Bk[(int64_t)(e >> 6) * ldp + R0 + (e & 63)] = 0.0;
        return;
    }
    if (R0 < k0) {   // B[R][c] = A[k0+c][R]: rows of A are rows of the panel
        for (int e = threadIdx.x; e < kw * 64; e += 256) {
            const int c = e >> 6, r = e & 63;
            Bk[(int64_t)c * ldp + R0 + r] = A[(k0 + c) * np + R0 + r];
        }
        return;
    }
    for (int c0 = 0; c0 < kw; c0 += 64) {   // B[R][c] = A[R][k0+c]: transpose through LDS
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            tile[r][c] = A[(R0 + r) * np + k0 + c0 + c];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
            const int c = e >> 6, r = e & 63;
            Bk[(int64_t)(c0 + c) * ldp + R0 + r] = tile[r][c];
        }
        __syncthreads();
    }
}

// C' = P Bk (k-major, P symmetric): the panel gets Ck = -C', A gets A[:,k] = C (lower-triangle positions) and
// A_kk = -P.  A wave owns 64 pivot columns x 32 matrix rows.
__global__ void __launch_bounds__(256)
panel_gemm_kernel(double *__restrict__ A, int64_t np, int64_t ldp, int64_t k0, int kw, const double *__restrict__ P,
                  const double *__restrict__ Bk, double *__restrict__ Ck) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = kw / 64;                 // 64-wide pivot-column blocks: 4 or 2
    const int rsub = wave / cb, cblk = wave - rsub * cb;
    const int64_t R0 = ((int64_t)blockIdx.x * (4 / cb) + rsub) * 32;
    if (R0 >= ldp) return;
    const int li = lane & 15, lk = lane >> 4;
    f64x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const double *pa = P + (int64_t)lk * kw + cblk * 64 + li;      // P[c][q] = P[q][c]
    const double *pb = Bk + (int64_t)lk * ldp + R0 + li;
    double nA[4], nB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) nA[i] = pa[16 * i];
#pragma unroll
    for (int j = 0; j < 2; ++j) nB[j] = pb[16 * j];
    const int nk = kw / 4;
    for (int kk = 0; kk < nk; ++kk) {
        double opA[4], opB[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) opA[i] = nA[i];
#pragma unroll
        for (int j = 0; j < 2; ++j) opB[j] = nB[j];
        if (kk + 1 < nk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) nA[i] = pa[(int64_t)(4 * (kk + 1)) * kw + 16 * i];
#pragma unroll
            for (int j = 0; j < 2; ++j) nB[j] = pb[(int64_t)(4 * (kk + 1)) * ldp + 16 * j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[i], opB[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = cblk * 64 + 16 * i + lk + 4 * r;
                const int64_t R = R0 + 16 * j + li;
                const double v = acc[i][j][r];
                Ck[(int64_t)c * ldp + R] = -v;   // the trailing update accumulates A + (-C) B' on top of A
                if (R < np) {
                    if (R < k0) A[(k0 + c) * np + R] = v;
                    else if (R >= k0 + kw) A[R * np + k0 + c] = v;
                    else A[R * np + k0 + c] = -P[(R - k0) * kw + c];
                }
            }
}

// A[Ri][Rj] += sum_c Ck[c][Ri] Bk[c][Rj] (Ck = -C') on the lower triangle, skipping pivot rows / columns.  The
// accumulators start from the tile of A itself (its loads fly while the first stage lands), so the epilogue is
// stores only and the read-modify-write latency is not exposed.
// which = 0: every tile; 1: only tiles inside the next pivot band [n0, n0+nw); 2: every tile outside it;
//         3: only tiles inside the SECOND next band [n1, n1+nw1) and outside the next one; 4: every tile outside both.
__global__ void __launch_bounds__(RU_THREADS, 2)
rank_update_kernel(double *__restrict__ A, int64_t np, int64_t ldp, int64_t k0, int kw, const double *__restrict__ Ck,
                   const double *__restrict__ Bk, const int2 *__restrict__ tiles, int ntiles, int which, int64_t n0, int nw, int band_tile,
                   int64_t n1, int nw1) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave / RU_WN, wb = wave % RU_WN;
    // XCD-aware: the 8 XCDs are dealt consecutive blocks round-robin; give each a contiguous run of the tile list
    int2 tt;
    if (band_tile >= 0) {   // band launch: the tiles of the band's tile rows / tile columns, enumerated directly
        const int nr = (int)(np / RU_TM);          // grid = nr per 128-wide slice of the band
        const int bsl = blockIdx.x / nr, e = blockIdx.x - bsl * nr, bt = band_tile + bsl;
        const int bwidth = which == 3 ? nw1 : nw;                   // width of the band being enumerated
        if (bsl == 0 && bwidth > RU_TM && e == band_tile + 1) return;   // tile (band_tile+1, band_tile) belongs to the second slice
        tt = e <= bt ? make_int2(bt, e) : make_int2(e, bt);
    } else {
        const int bq = ntiles / 8, br = ntiles % 8, xcd = blockIdx.x % 8, bm = blockIdx.x / 8;
        const int item = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bm;
        tt = tiles[item];
    }
    const int64_t a0 = (int64_t)tt.x * RU_TM, b0 = (int64_t)tt.y * RU_TN;
    auto in_band = [&](int64_t lo, int64_t start, int64_t width) { return lo >= start && lo < start + width; };
    auto dead = [&](int wa_, int wb_) {     // the wave's 64x64 block has nothing to update
        const int64_t r = a0 + wa_ * 64, c = b0 + wb_ * 64;
        if (c > r + 63 || c >= np) return true;
        if (in_band(r, k0, kw) || in_band(c, k0, kw)) return true;
        if (which != 0) {
            const bool next = in_band(r, n0, nw) || in_band(c, n0, nw);
            if (which <= 2) { if ((which == 1) != next) return true; }
            else {
                if (next) return true;
                const bool second = nw1 > 0 && (in_band(r, n1, nw1) || in_band(c, n1, nw1));
                if ((which == 3) != second) return true;
            }
        }
        return false;
    };
    bool any = false;
#pragma unroll
    for (int q = 0; q < 2 * RU_WN; ++q) any = any || !dead(q / RU_WN, q % RU_WN);
    if (!any) return;
    const bool skip_wave = dead(wa, wb);

    constexpr int ROW = RU_TM + RU_TN, STAGE = RU_BK * ROW;
    double *buf0 = lds, *buf1 = lds + STAGE;
    auto stage_load = [&](double *buf, int s0) {
        constexpr int PARTS = 1 + RU_TN / 128;
        for (int p = wave; p < RU_BK * PARTS; p += RU_THREADS / 64) {   // piece = (pivot, part): 128 panel values
            const int k = p / PARTS, part = p - k * PARTS;
            const double *src = part == 0 ? Ck + (int64_t)(s0 + k) * ldp + a0 : Bk + (int64_t)(s0 + k) * ldp + b0 + (part - 1) * 128;
            glds16(src + 2 * lane, buf + p * 128);
        }
    };
    const int li = lane & 15, lk = lane >> 4;
    int offA[4], offB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { offA[t] = wa * 64 + t * 16 + li; offB[t] = RU_TM + wb * 64 + t * 16 + li; }
    const int nstages = kw / RU_BK;
    stage_load(buf0, 0);
    const int64_t r_lo = a0 + wa * 64, c_lo = b0 + wb * 64;
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = r_lo + i * 16 + lk + 4 * r, col = c_lo + j * 16 + li;
                acc[i][j][r] = (!skip_wave && col <= row) ? A[row * np + col] : 0.0;
            }
    __syncthreads();
    double rA[4], rB[4];
    auto fetch = [&](const double *img, int kk) {
        const double *row = img + (kk * 4 + lk) * ROW;
#pragma unroll
        for (int t = 0; t < 4; ++t) { rA[t] = row[offA[t]]; rB[t] = row[offB[t]]; }
    };
    for (int s = 0; s < nstages; ++s) {
        double *cur = (s & 1) ? buf1 : buf0, *nxt = (s & 1) ? buf0 : buf1;
        if (s + 1 < nstages) stage_load(nxt, (s + 1) * RU_BK);
        if (!skip_wave) {
            fetch(cur, 0);
#pragma unroll
            for (int kk = 0; kk < RU_BK / 4; ++kk) {
                double opA[4], opB[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { opA[t] = rA[t]; opB[t] = rB[t]; }
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < RU_BK / 4) fetch(cur, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[i], opB[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    if (skip_wave) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = r_lo + i * 16 + lk + 4 * r, col = c_lo + j * 16 + li;
                if (col <= row) A[row * np + col] = acc[i][j][r];
            }
}

// ---- the tail of a pivot chain in ONE launch (kw = 128): gather + panel GEMM + write-back -------------------------------------
// One workgroup = 16 matrix rows R0 .. R0+15 against all 128 pivot columns:
//   S[q][r] = B[R0+r][q] (B = A[:, pivot columns] read from the lower triangle; zero in the pivot rows) through 16 KB of LDS,
//   Bk[q][R] = S (k-major panel for the trailing update),  C' = P S on the matrix cores (wave w: pivot columns 32w .. 32w+31),
//   Ck = -C',  A[:, k] = C (lower-triangle positions),  A_kk = -P.
// 16 KB of LDS and < 128 registers: the kernel is resident NEXT TO two trailing-update workgroups of a CU (2 x 64 KB of LDS, 2 x ~190
// registers per SIMD lane) instead of waiting for one of them to retire -- the separate gather (33 KB of LDS) and GEMM kernels took
// 68 + 66 us of a step's pivot chain that way.  Rows above the pivot band are read as rows of A (coalesced as they stand); rows below
// it are 1 KB row segments, transposed on the way into the LDS image (XOR swizzle: conflict-free both ways).
__global__ void __launch_bounds__(256, 5)          // <= 96 registers: one of its waves fits a SIMD next to two trailing-update waves
panel_fused_kernel(double *__restrict__ A, int64_t np, int64_t ldp, int64_t k0, const double *__restrict__ P,
                   double *__restrict__ Bk, double *__restrict__ Ck) {
    constexpr int KWF = 128, RS = 16;
    __shared__ __attribute__((aligned(16))) double S[KWF * RS];       // S[q][r ^ (q & 15)]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t R0 = (int64_t)blockIdx.x * RS;
    const bool pad = R0 >= np, band = !pad && R0 >= k0 && R0 < k0 + KWF, low = !pad && R0 < k0;
    if (pad || band) {
        for (int e = tid; e < KWF * RS; e += 256) {                   // no contribution: zero panels
            const int q = e >> 4, r = e & 15;
            Bk[(int64_t)q * ldp + R0 + r] = 0.0; Ck[(int64_t)q * ldp + R0 + r] = 0.0;
        }
        if (band)
            for (int e = tid; e < KWF * RS; e += 256) {               // A_kk = -P
                const int r = e >> 7, c = e & 127;
                A[(R0 + r) * np + k0 + c] = -P[(R0 + r - k0) * KWF + c];
            }
        return;
    }
    if (low) {                                                        // B[R][q] = A[k0+q][R]: rows of A are rows of the panel
#pragma unroll
        for (int i = 0; i < KWF * RS / 256; ++i) {
            const int e = tid + 256 * i, q = e >> 4, r = e & 15;
            S[q * RS + (r ^ (q & 15))] = A[(k0 + q) * np + R0 + r];
        }
    } else {                                                          // B[R][q] = A[R][k0+q]: 1 KB per row, transposed into S
#pragma unroll
        for (int i = 0; i < KWF * RS / 256; ++i) {
            const int e = tid + 256 * i, r = e >> 7, q = e & 127;
            S[q * RS + (r ^ (q & 15))] = A[(R0 + r) * np + k0 + q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KWF * RS / 256; ++i) {                        // the k-major panel of B
        const int e = tid + 256 * i, q = e >> 4, r = e & 15;
        Bk[(int64_t)q * ldp + R0 + r] = S[q * RS + (r ^ (q & 15))];
    }
    // C'[c][r] = sum_q P[c][q] S[q][r]:  A operand = P[q][c] (P symmetric: lanes walk c), B operand = S[q][r]
    const int li = lane & 15, lk = lane >> 4;
    f64x4 acc[2] = {(f64x4){0.0, 0.0, 0.0, 0.0}, (f64x4){0.0, 0.0, 0.0, 0.0}};
    const double *pa = P + (int64_t)lk * KWF + 32 * wave + li;
    constexpr int CH = 4;                                             // k-steps per chunk: two chunks in flight (P comes from L2)
    double opA[2][CH][2];
#pragma unroll
    for (int k = 0; k < CH; ++k) { opA[0][k][0] = pa[(int64_t)(4 * k) * KWF]; opA[0][k][1] = pa[(int64_t)(4 * k) * KWF + 16]; }
#pragma unroll
    for (int ch = 0; ch < 32 / CH; ++ch) {
        if (ch + 1 < 32 / CH) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                opA[(ch + 1) & 1][k][0] = pa[(int64_t)(4 * (CH * (ch + 1) + k)) * KWF];
                opA[(ch + 1) & 1][k][1] = pa[(int64_t)(4 * (CH * (ch + 1) + k)) * KWF + 16];
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int q = 4 * (CH * ch + k) + lk;
            const double b = S[q * RS + (li ^ (q & 15))];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[ch & 1][k][0], b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[ch & 1][k][1], b, acc[1], 0, 0, 0);
        }
    }
    // D: col = lane & 15 = r, row = lk + 4 reg = pivot column within the 16-block
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Ck[(int64_t)(32 * wave + 16 * b + lk + 4 * rg) * ldp + R0 + li] = -acc[b][rg];   // the update adds A + (-C) B'
    if (low) {                                                        // A[k0+c][R] = C[R][c]: coalesced as it stands
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) A[(k0 + 32 * wave + 16 * b + lk + 4 * rg) * np + R0 + li] = acc[b][rg];
        return;
    }
    __syncthreads();                                                  // every wave is done reading S
    // A[R][k0+c] = C[R][c]: through LDS as T[r][c ^ (2 r)] so that a row leaves as one 1 KB segment
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int c = 32 * wave + 16 * b + lk + 4 * rg;
            S[li * KWF + (c ^ (2 * li))] = acc[b][rg];
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KWF * RS / 256; ++i) {
        const int e = tid + 256 * i, r = e >> 7, c = e & 127;
        A[(R0 + r) * np + k0 + c] = S[r * KWF + (c ^ (2 * r))];
    }
}

// ---- trailing update with SEVERAL pivot panels in one pass (groups of 128-wide steps) ---------------------------------------------
// A tile receives  A += sum_i Ck_i' Bk_i  (Ck = -C') for a RANGE of the group's panels in ONE read-modify-write.  Fitted on this
// kernel's own timings (one panel per pass: 56.4 us per workgroup, two: 93.8 us): a workgroup costs 19 us + 0.29 us per pivot -- the
// accumulator load / store, the first stage and the lock-step of the two workgroups of a CU weigh as much as 65 pivots of matrix
// instructions -- so the passes over A are made as deep as the dependencies allow: kMaxGroup = 4 panels (512 pivots) per pass.
// Which panels a tile takes follows from where it lies.  With jr / jc = the group-local index of the pivot band that holds the
// tile's row / column block (-1: none; jc <= jr on the lower triangle), every panel i outside {jr, jc} is applied EXACTLY ONCE:
//   band launch j (tiles of band j, before chain j reads them):  i in [jc + 1 if 0 <= jc < j else 0,  j)
//   priority / rest launches (after the group's last chain):     i in (max(jr, jc), mg)          (all of them when jr = jc = -1)
// -- a tile inside band jc was overwritten with C by chain jc: panels before jc went into that chain's input, panels after it
// update the stored C (the Gauss-Jordan sweep keeps updating the columns of earlier pivots).
constexpr int kMaxGroup = 4;
struct RuGroup {
    const double *Ck[kMaxGroup], *Bk[kMaxGroup];    // panel i of the group (k-major, ldp apart per pivot)
    int kb, mg;                                      // first pivot block of the group, panels in it
};
// select = 0: the tiles of the list, except those inside the blocks [skip0, skip0 + nskip) (the next group's bands);
//          1: the tiles of the bands [e0, e0 + nslices) enumerated directly; band >= 0: a band launch for that group-local band
// BK = pivots per LDS stage, WGS = workgroups per CU the register budget is set for (BK = 16: 64 KB of LDS, two; BK = 8: 32 KB, three)
template <int BK, int WGS>
__global__ void __launch_bounds__(RU_THREADS, WGS)
rank_updatem_kernel(double *__restrict__ A, int64_t np, int64_t ldp, RuGroup g, const int2 *__restrict__ tiles, int ntiles,
                    int select, int e0, int nslices, int band, int skip0, int nskip) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave / RU_WN, wb = wave % RU_WN;
    int2 tt;
    if (select == 1) {
        const int nr = (int)(np / RU_TM);
        const int bsl = blockIdx.x / nr, e = blockIdx.x - bsl * nr, bt = e0 + bsl;
        if (e > bt && e < e0 + nslices) return;                      // tile (e, bt) with e a later slice: enumerated there as (e, bt)
        tt = e <= bt ? make_int2(bt, e) : make_int2(e, bt);
    } else {
        const int bq = ntiles / 8, br = ntiles % 8, xcd = blockIdx.x % 8, bm = blockIdx.x / 8;
        const int item = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bm;
        tt = tiles[item];
        if ((tt.x >= skip0 && tt.x < skip0 + nskip) || (tt.y >= skip0 && tt.y < skip0 + nskip)) return;
    }
    const int jr = (tt.x >= g.kb && tt.x < g.kb + g.mg) ? tt.x - g.kb : -1, jc = (tt.y >= g.kb && tt.y < g.kb + g.mg) ? tt.y - g.kb : -1;
    int lo, hi;
    if (band >= 0) { lo = (jc >= 0 && jc < band) ? jc + 1 : 0; hi = band; }
    else { lo = (jr > jc ? jr : jc) + 1; hi = g.mg; }
    if (lo >= hi) return;
    const int64_t a0 = (int64_t)tt.x * RU_TM, b0 = (int64_t)tt.y * RU_TN;
    const int64_t r_lo = a0 + wa * 64, c_lo = b0 + wb * 64;
    const bool skip_wave = c_lo > r_lo + 63 || c_lo >= np;          // above the diagonal: nothing to maintain
    constexpr int SPP = 128 / BK;                                 // stages per panel
    const int nstages = (hi - lo) * SPP;

    constexpr int ROW = RU_TM + RU_TN, STAGE = BK * ROW;
    double *buf0 = lds, *buf1 = lds + STAGE;
    auto stage_load = [&](double *buf, int st) {
        const int pi = lo + st / SPP, s0 = (st % SPP) * BK;
        const double *Ck = nullptr, *Bk = nullptr;                   // (selected without indexing the kernel-argument arrays dynamically)
#pragma unroll
        for (int i = 0; i < kMaxGroup; ++i) if (i == pi) { Ck = g.Ck[i]; Bk = g.Bk[i]; }
        for (int p = wave; p < BK * 2; p += RU_THREADS / 64) {     // piece = (pivot, part): 128 panel values
            const int k = p >> 1, part = p & 1;
            const double *src = part == 0 ? Ck + (int64_t)(s0 + k) * ldp + a0 : Bk + (int64_t)(s0 + k) * ldp + b0;
            glds16(src + 2 * lane, buf + p * 128);
        }
    };
    const int li = lane & 15, lk = lane >> 4;
    int offA[4], offB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { offA[t] = wa * 64 + t * 16 + li; offB[t] = RU_TM + wb * 64 + t * 16 + li; }
    stage_load(buf0, 0);
    f64x4 acc[4][4];
    // accumulator addresses = a wave-uniform pointer per (i, r, j) + ONE 32-bit lane offset (saddr + voffset form: no 64-bit vector
    // arithmetic per element); a wave strictly below the diagonal takes every element, so its 64 loads / stores carry no predicate
    const unsigned voffb = (unsigned)((lk * np + li) * 8);          // byte offset of the lane inside a 4-row x 16-column patch (< 2^32: np < 2^26)
    const bool below = c_lo + 63 <= r_lo;                            // (wave-uniform) col <= row for every element of the wave's 64 x 64 block
    if (!skip_wave && below) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double *sp = A + (r_lo + i * 16 + 4 * r) * np + c_lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j][r] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(sp + j * 16) + voffb);
            }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = r_lo + i * 16 + lk + 4 * r, col = c_lo + j * 16 + li;
                    acc[i][j][r] = (!skip_wave && col <= row) ? A[row * np + col] : 0.0;
                }
    }
    __syncthreads();
    double rA[4], rB[4];
    auto fetch = [&](const double *img, int kk) {
        const double *row = img + (kk * 4 + lk) * ROW;
#pragma unroll
        for (int t = 0; t < 4; ++t) { rA[t] = row[offA[t]]; rB[t] = row[offB[t]]; }
    };
#pragma unroll 1
    for (int s = 0; s < nstages; ++s) {
        double *cur = (s & 1) ? buf1 : buf0, *nxt = (s & 1) ? buf0 : buf1;
        if (s + 1 < nstages) stage_load(nxt, s + 1);
        if (!skip_wave) {
            fetch(cur, 0);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double opA[4], opB[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { opA[t] = rA[t]; opB[t] = rB[t]; }
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < BK / 4) fetch(cur, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[i], opB[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    if (skip_wave) return;
    if (below) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double *sp = A + (r_lo + i * 16 + 4 * r) * np + c_lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<double *>(reinterpret_cast<char *>(sp + j * 16) + voffb) = acc[i][j][r];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = r_lo + i * 16 + lk + 4 * r, col = c_lo + j * 16 + li;
                if (col <= row) A[row * np + col] = acc[i][j][r];
            }
}

// ---- band launches on 64 x 64 tiles ------------------------------------------------------------------------------------------------
// A band launch (the tiles of one or a few 128-wide bands) has fewer tiles than the chip has workgroup slots: it lasts as long as ONE
// tile at the pass's depth (19 us + 0.29 us per pivot on 128 x 128 tiles), and it sits on the pivot chains' critical path.  The same
// update on 64 x 64 tiles -- four times the workgroups, a quarter of the matrix instructions each (a wave owns 32 x 32: four
// accumulators), half the LDS stage (a pivot's 64 + 64 panel values are ONE 1-KB LDS-DMA piece: lanes 0-31 fetch Ck, lanes 32-63 Bk) --
// applies the same panels to the same elements in the same order per element (bit-identical results).
// Tile rules as rank_updatem_kernel with select = 1.
template <int BK>
__global__ void __launch_bounds__(256, 4)
rank_updateb_kernel(double *__restrict__ A, int64_t np, int64_t ldp, RuGroup g, int e0, int nslices, int band) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;
    const int sub = blockIdx.x & 3, bidx = blockIdx.x >> 2, sx = sub >> 1, sy = sub & 1;
    const int nr = (int)(np / 128);
    const int bsl = bidx / nr, e = bidx - bsl * nr, bt = e0 + bsl;
    if (e > bt && e < e0 + nslices) return;                          // tile (e, bt) with e a later slice: enumerated there as (e, bt)
    const int2 tt = e <= bt ? make_int2(bt, e) : make_int2(e, bt);
    if (tt.x == tt.y && sy > sx) return;                             // quarter above the diagonal
    const int jr = (tt.x >= g.kb && tt.x < g.kb + g.mg) ? tt.x - g.kb : -1, jc = (tt.y >= g.kb && tt.y < g.kb + g.mg) ? tt.y - g.kb : -1;
    int lo, hi;
    if (band >= 0) { lo = (jc >= 0 && jc < band) ? jc + 1 : 0; hi = band; }
    else { lo = (jr > jc ? jr : jc) + 1; hi = g.mg; }
    if (lo >= hi) return;
    const int64_t a0 = (int64_t)tt.x * 128 + sx * 64, b0 = (int64_t)tt.y * 128 + sy * 64;
    const int64_t r_lo = a0 + wa * 32, c_lo = b0 + wb * 32;
    const bool skip_wave = c_lo > r_lo + 31;                         // above the diagonal: nothing to maintain
    constexpr int SPP = 128 / BK;                                    // stages per panel
    const int nstages = (hi - lo) * SPP;
    constexpr int ROW = 128, STAGE = BK * ROW;
    double *buf0 = lds, *buf1 = lds + STAGE;
    auto stage_load = [&](double *buf, int st) {
        const int pi = lo + st / SPP, s0 = (st % SPP) * BK;
        const double *Ck = nullptr, *Bk = nullptr;
#pragma unroll
        for (int i = 0; i < kMaxGroup; ++i) if (i == pi) { Ck = g.Ck[i]; Bk = g.Bk[i]; }
        for (int k = wave; k < BK; k += 4) {                         // piece = pivot: [64 of Ck | 64 of Bk]
            const double *src = lane < 32 ? Ck + (int64_t)(s0 + k) * ldp + a0 + 2 * lane : Bk + (int64_t)(s0 + k) * ldp + b0 + 2 * (lane - 32);
            glds16(src, buf + k * ROW);
        }
    };
    const int li = lane & 15, lk = lane >> 4;
    int offA[2], offB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { offA[t] = wa * 32 + t * 16 + li; offB[t] = 64 + wb * 32 + t * 16 + li; }
    stage_load(buf0, 0);
    f64x4 acc[2][2];
    const unsigned voffb = (unsigned)((lk * np + li) * 8);
    const bool below = c_lo + 31 <= r_lo;                            // (wave-uniform) col <= row for every element of the wave's 32 x 32 block
    if (!skip_wave && below) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double *sp = A + (r_lo + i * 16 + 4 * r) * np + c_lo;
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j][r] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(sp + j * 16) + voffb);
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = r_lo + i * 16 + lk + 4 * r, col = c_lo + j * 16 + li;
                    acc[i][j][r] = (!skip_wave && col <= row) ? A[row * np + col] : 0.0;
                }
    }
    __syncthreads();
    double rA[2], rB[2];
    auto fetch = [&](const double *img, int kk) {
        const double *row = img + (kk * 4 + lk) * ROW;
#pragma unroll
        for (int t = 0; t < 2; ++t) { rA[t] = row[offA[t]]; rB[t] = row[offB[t]]; }
    };
#pragma unroll 1
    for (int s = 0; s < nstages; ++s) {
        double *cur = (s & 1) ? buf1 : buf0, *nxt = (s & 1) ? buf0 : buf1;
        if (s + 1 < nstages) stage_load(nxt, s + 1);
        if (!skip_wave) {
            fetch(cur, 0);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double opA[2], opB[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) { opA[t] = rA[t]; opB[t] = rB[t]; }
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < BK / 4) fetch(cur, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[i], opB[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    if (skip_wave) return;
    if (below) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double *sp = A + (r_lo + i * 16 + 4 * r) * np + c_lo;
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<double *>(reinterpret_cast<char *>(sp + j * 16) + voffb) = acc[i][j][r];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = r_lo + i * 16 + lk + 4 * r, col = c_lo + j * 16 + li;
                if (col <= row) A[row * np + col] = acc[i][j][r];
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

constexpr int64_t kTwoLevelMinNp = 1024;

static size_t sweep64_work_doubles(int64_t np) { return (size_t)(2 * np * NB + NB * NB); }
static size_t ru_tile_capacity(int64_t np) { const int64_t nr = np / RU_TM, nc = ceil_div(np, RU_TN); return (size_t)(nr * nc); }

size_t spd_inverse_work_bytes(int64_t np) {
    size_t d = sweep64_work_doubles(np);
    if (np >= kTwoLevelMinNp) {
        const int64_t ldp = round_up(np, RU_TN);
        const size_t two = (size_t)(8 * KW * ldp + KW * KW) + sweep64_work_doubles(KW) + ru_tile_capacity(np);   // 8 panel slots (groups of four steps, two groups in flight)   // int2 = one double
        if (two > d) d = two;
    }
    return sizeof(double) * d;
}

// the 64-wide sweep: A <- -A^-1 on the lower triangle (status is not cleared here)
static void sweep64(double *A, int64_t np, int nbatch, double *work, int *status_dev, hipStream_t s) {
    const int64_t strideW = (int64_t)sweep64_work_doubles(np), strideA = np * np;
    double *Bp = work, *Cp = work + np * NB, *P = work + 2 * np * NB;
    const int nblk = (int)(np / NB);
    const int nt = (int)(np / 128);
    const unsigned ntiles = (unsigned)(nt * (nt + 1) / 2);
    for (int k = 0; k < nblk; ++k) {
        hipLaunchKernelGGL(diag_inverse_kernel, dim3((unsigned)nbatch), dim3(256), 0, s, A, np, k, P, status_dev, strideA, strideW);
        hipLaunchKernelGGL(panel_kernel, dim3((unsigned)nblk, (unsigned)nbatch), dim3(256), 0, s, A, np, k, P, Bp, Cp, strideA, strideW);
        hipLaunchKernelGGL(sweep_update_kernel, dim3(ntiles, (unsigned)nbatch), dim3(256), 0, s, A, np, k, Bp, Cp, strideA, strideW);
    }
}

// lower-triangle tiles of the trailing update, in bands of four tile rows, column-major inside a band: consecutive
// tiles (which the XCD-aware launch order keeps on one L2) share their B columns and a 512-row slice of C
static const std::vector<int2> &ru_tiles(int64_t np) {
    static std::mutex mu;
    static std::map<int64_t, std::vector<int2>> cache;
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(np);
    if (it != cache.end()) return it->second;
    std::vector<int2> t;
    const int nr = (int)(np / RU_TM), nc = (int)ceil_div(np, RU_TN);
    for (int band = 0; band < nr; band += 4)
        for (int tj = 0; tj < nc; ++tj)
            for (int ti = band; ti < band + 4 && ti < nr; ++ti)
                if ((int64_t)tj * RU_TN <= (int64_t)ti * RU_TM + RU_TM - 1) t.push_back(make_int2(ti, tj));
    return cache.emplace(np, std::move(t)).first->second;
}

int32_t SweepAux::ensure() {
    if (side) return LPVS_OK;
    int lo = 0, hi = 0;   // numerically lower = higher priority
    LPVS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    LPVS_HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
    LPVS_HIP(hipEventCreateWithFlags(&panel, hipEventDisableTiming));
    LPVS_HIP(hipEventCreateWithFlags(&rest, hipEventDisableTiming));
    LPVS_HIP(hipEventCreateWithFlags(&band, hipEventDisableTiming));
    LPVS_HIP(hipEventCreateWithFlags(&second, hipEventDisableTiming));
    LPVS_HIP(hipEventCreateWithFlags(&bulkdone, hipEventDisableTiming));
    // The pair schedule's trailing updates fill every CU with two workgroups of ~208 registers per lane: the one-workgroup pivot
    // inverse (304 registers per lane) then waits for a workgroup to retire, and once resident shares its SIMDs with a stream of f64
    // matrix instructions that its own vector instructions cannot overlap (measured: 235 us instead of 80).  The updates therefore run
    // on a stream whose CU mask leaves a few CUs (LPVS_OPT_RESERVE_CUS, default 8 of 256) to the side stream's chains.
    const int o_res = option_in_effect(LPVS_OPT_RESERVE_CUS);   // (the creating thread's default, else LPVS_RESERVE_CUS)
    const int reserve = o_res == LPVS_RESERVE_NONE ? 0 : (o_res > 0 ? o_res : 8);
    int dev = 0, cus = 0;
    if (reserve > 0 && hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        cus > 4 * reserve) {
        std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
        for (int c = reserve; c < cus; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
        if (hipExtStreamCreateWithCUMask(&bulk, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); bulk = nullptr; }
    }
    return LPVS_OK;
}
SweepAux::~SweepAux() {
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
    if (bulk) { (void)hipStreamSynchronize(bulk); (void)hipStreamDestroy(bulk); }
    if (bulkdone) (void)hipEventDestroy(bulkdone);
    if (panel) (void)hipEventDestroy(panel);
    if (rest) (void)hipEventDestroy(rest);
    if (band) (void)hipEventDestroy(band);
    if (second) (void)hipEventDestroy(second);
}

// Step k of the outer sweep:  chain_k = { P = inv(A_kk); Bk = A[:,k]; Ck = -(Bk' P)'; A[:,k] = C; A_kk = -P },
// then the trailing update U_k.  With look-ahead U_k is split: the tiles that intersect the NEXT pivot band run first
// on the side stream, followed by chain_{k+1} (into the other panel buffer), while the rest of U_k runs on the main
// stream; U_{k+1} starts when both are done.  The pivot chain (latency-bound, one workgroup) thus hides under the
// MFMA-bound bulk update.
static int32_t spd_inverse_two_level(double *A, int64_t np, double *work, int *status_dev, hipStream_t s, SweepAux *aux) {
    const int64_t ldp = round_up(np, RU_TN);
    double *panelbuf[3] = {work, work + 2 * KW * ldp, work + 4 * KW * ldp};   // {Bk, Ck} x 3 (depth-2 look-ahead rotates three)
    double *P = work + 8 * KW * ldp, *inner = P + KW * KW;
    int2 *tiles = reinterpret_cast<int2 *>(inner + sweep64_work_doubles(KW));
    const std::vector<int2> &ht = ru_tiles(np);   // persistent host copy: the async upload may outlive this call
    LPVS_HIP(hipMemcpyAsync(tiles, ht.data(), sizeof(int2) * ht.size(), hipMemcpyHostToDevice, s));
    const size_t lds = sizeof(double) * 2 * RU_BK * (RU_TM + RU_TN);
    const bool single_wg_pivot = [] { const char *e = experiment_env("LPVS_PIVOT"); return !(e && std::string(e) == "sweep64"); }();
    const bool lookahead_on = [] { const char *e = experiment_env("LPVS_LOOKAHEAD"); return !(e && e[0] == '0'); }();
    LPVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rank_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // 128-wide pivot blocks (one-workgroup inverse) up to np ~ 12k; beyond, the bulk update is long enough to hide the
    // 256-wide chain (pivot block by the 64-wide sweep) and the deeper update runs closer to the MFMA peak
    const int kw_env = [] { const char *e = experiment_env("LPVS_KW"); return e ? atoi(e) : 0; }();
    // (the group schedule below runs 128-wide steps: 39.9 ms against 47.5 with 256-wide steps at np = 12288, 89.5 / 93.5 at 16384, equal at 32768)
    const bool steps_scheme = [] { const char *e = experiment_env("LPVS_FACTOR_SCHEME"); return e && std::string(e) == "steps"; }();
    const int kw_outer = kw_env == 128 || kw_env == 256 ? kw_env : ((steps_scheme && np >= 12288) ? 256 : 128);
    const bool la = lookahead_on && aux != nullptr && np > kw_outer;
    if (la) LPVS_TRY(aux->ensure());

    auto width = [&](int64_t k0) { return (int)(np - k0 < kw_outer ? np - k0 : kw_outer); };   // with 256: 128 for a ragged last block
    auto chain = [&](int64_t k0, double *Bk, double *Ck, hipStream_t st) {
        const int kw = width(k0);
        if (single_wg_pivot && kw == 128) {
            hipLaunchKernelGGL(pivot_inverse_kernel<128>, dim3(1), dim3(192), 0, st, A, np, k0, P, status_dev);
        } else {
            hipLaunchKernelGGL(pivot_extract_kernel, dim3((unsigned)kw), dim3(256), 0, st, A, np, k0, kw, P);
            sweep64(P, kw, 1, inner, status_dev, st);
            hipLaunchKernelGGL(negate_mirror_kernel, dim3((unsigned)(kw / 32), (unsigned)(kw / 32), 1u), dim3(256), 0, st, P, (int64_t)kw);
        }
        hipLaunchKernelGGL(panel_gather_kernel, dim3((unsigned)(ldp / 64)), dim3(256), 0, st, A, np, ldp, k0, kw, Bk);
        hipLaunchKernelGGL(panel_gemm_kernel, dim3((unsigned)(ldp / (32 * (4 / (kw / 64))))), dim3(256), 0, st, A, np, ldp, k0, kw, P, Bk, Ck);
    };
    auto update = [&](int64_t k0, const double *Bk, const double *Ck, int which, hipStream_t st) {
        const int kw = width(k0);
        const int64_t n0 = k0 + kw;
        const int nw = n0 < np ? width(n0) : 0;
        const int64_t n1 = n0 + nw;
        const int nw1 = (nw > 0 && n1 < np) ? width(n1) : 0;
        // a 128-wide band is one tile row + one tile column: launch just those (32k early-exit workgroups cost 0.3 ms at np = 32768)
        const bool direct1 = which == 1 && nw % RU_TM == 0 && nw > 0 && RU_TM == RU_TN;
        const bool direct3 = which == 3 && nw1 % RU_TM == 0 && nw1 > 0 && RU_TM == RU_TN;
        const unsigned grid = direct1 ? (unsigned)(np / RU_TM * (nw / RU_TM)) : (direct3 ? (unsigned)(np / RU_TM * (nw1 / RU_TM)) : (unsigned)ht.size());
        const int band_tile = direct1 ? (int)(n0 / RU_TM) : (direct3 ? (int)(n1 / RU_TM) : -1);
        hipLaunchKernelGGL(rank_update_kernel, dim3(grid), dim3(RU_THREADS), lds, st, A, np, ldp, k0,
                           kw, Ck, Bk, tiles, (int)ht.size(), which, n0, nw, band_tile, n1, nw1);
    };

    // ---- groups of up to four 128-wide steps: the trailing update takes the group's panels in ONE pass (rank_updatem_kernel) --------
    // Per group (pivot blocks kb .. kb + mg - 1; panel kb is ready when the group starts; the next group starts at block nb0):
    //   side:  for j = 1 .. mg-1:  band j += panels [.., j)  ->  chain kb+j        ->  [event panels]
    //          [wait: previous group's rest]  the NEXT group's bands += this group's panels (priority)  ->  chain nb0  (the next group's first panel)
    //   main:  [wait panels]  every tile outside the next group's bands += this group's panels                 ->  [event rest]
    // The side stream (high priority) thus owns everything the pivot chains wait for -- its priority launch shares the chip with the
    // main stream's pass instead of running ahead of it --, and the main stream is one deep pass per group.  Band / priority tiles are
    // disjoint from the rest launch's; the priority tiles were last written by the previous group's rest launch (hence the wait).
    // Panels live in 2 x kMaxGroup slots (slot = pivot block mod 8): chain nb0 and the next group's chains write the other half
    // while this group's launches still read theirs; a half is rewritten two groups later, behind the events above.
    const int group_env = [] { const char *e = experiment_env("LPVS_FACTOR_GROUP"); return e ? atoi(e) : 0; }();   // 1 .. 4 panels per pass (diagnostic)
    // Measured at np = 8192 (tools/factor_ab3.sh): 1 panel per pass 16.2 ms, 2: 14.0, 3: 15.0, 4: 15.4 (round 2's schedule: 17.8).  Deeper
    // passes amortise the per-tile overhead further, but the side stream's share of the matrix work grows with the group (bands of
    // depth 128 .. 128 (mg - 1) and a priority launch of mg bands: 15 % of the flops at mg = 2, 30 % at mg = 4) and its workgroups
    // queue behind 170-us workgroups of the main pass: from mg = 3 the side stream is the critical path again.
    // From np = 12288 the main pass is long enough to cover a four-step side chain: 84.5 ms with groups of four (and the 8-pivot
    // stages below) against 89.6 with pairs at np = 16384 (tools/factor_ab4.sh).
    const int mgmax = group_env >= 1 && group_env <= kMaxGroup ? group_env : (np >= 12288 ? 4 : 2);
    const bool groups_on = [] { const char *e = experiment_env("LPVS_FACTOR_SCHEME"); return !(e && std::string(e) == "steps"); }();
    const bool fused_chain = [] { const char *e = experiment_env("LPVS_CHAIN"); return !(e && std::string(e) == "split"); }();
    const bool mfma_pivot = [] { const char *e = experiment_env("LPVS_PIVOT"); return !(e && std::string(e) == "regs"); }();   // regs: the register kernel
    if (la && groups_on && single_wg_pivot && kw_outer == 128 && np >= 2048 && np % 128 == 0 && ldp == np) {
        hipStream_t side = aux->side;
        // LPVS_RU_STAGE=8: 8-pivot LDS stages (32 KB per workgroup) and a register budget for three workgroups per CU
        // (np = 8192: 14.65 ms against 13.3 with 16-pivot stages and two workgroups per CU; np = 16384: 84.5 against 86.5 -- default from 12288)
        const int stage_env = [] { const char *e = experiment_env("LPVS_RU_STAGE"); return e ? atoi(e) : 0; }();
        const bool bk8 = stage_env == 8 || (stage_env != 16 && np >= 12288);
        const size_t ldsm = bk8 ? lds / 2 : lds;
        auto ru_kernel = bk8 ? rank_updatem_kernel<8, 3> : rank_updatem_kernel<16, 2>;
        LPVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ru_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm));
        const int nblocks = (int)(np / 128);
        auto slotB = [&](int blk) { return work + (int64_t)(blk & (2 * kMaxGroup - 1)) * (2 * 128 * ldp); };
        auto slotC = [&](int blk) { return slotB(blk) + 128 * ldp; };
        // The one-workgroup pivot inverse asks for the rest of a CU's LDS (unused): it can then only be placed on a CU that holds no other
        // workgroup -- one of those the deep passes' CU mask leaves out -- and nothing joins it there (np = 8192: 13.5 -> 13.2 ms; with three
        // update workgroups per CU, from np = 12288, it waits too long for an empty CU: 84.5 -> 86.3 ms at 16384).  LPVS_PIVOT_ALONE=0/1.
        const int alone_env = [] { const char *e = experiment_env("LPVS_PIVOT_ALONE"); return e ? atoi(e) : -1; }();
        const bool pivot_alone = alone_env >= 0 ? alone_env != 0 : np < 12288;
        const size_t pivot_pad = (pivot_alone && aux->bulk) ? (size_t)(160 * 1024 - 40 * 1024) : 0;
        if (pivot_pad) LPVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pivot_inverse_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pivot_pad));
        auto chain128 = [&](int blk, hipStream_t st) {
            const int64_t k0 = (int64_t)blk * 128;
            if (mfma_pivot) hipLaunchKernelGGL(pivot_inverse_mfma_kernel, dim3(1), dim3(256), pivot_pad, st, A, np, k0, P, status_dev);
            else hipLaunchKernelGGL(pivot_inverse_kernel<128>, dim3(1), dim3(192), 0, st, A, np, k0, P, status_dev);
            if (fused_chain) {
                hipLaunchKernelGGL(panel_fused_kernel, dim3((unsigned)(ldp / 16)), dim3(256), 0, st, A, np, ldp, k0, (const double *)P, slotB(blk), slotC(blk));
            } else {
                hipLaunchKernelGGL(panel_gather_kernel, dim3((unsigned)(ldp / 64)), dim3(256), 0, st, A, np, ldp, k0, 128, slotB(blk));
                hipLaunchKernelGGL(panel_gemm_kernel, dim3((unsigned)(ldp / 64)), dim3(256), 0, st, A, np, ldp, k0, 128, P, slotB(blk), slotC(blk));
            }
        };
        auto group_of = [&](int kb, int mg) {
            RuGroup g{};
            for (int i = 0; i < kMaxGroup; ++i) { g.Ck[i] = slotC(kb + (i < mg ? i : 0)); g.Bk[i] = slotB(kb + (i < mg ? i : 0)); }
            g.kb = kb; g.mg = mg;
            return g;
        };
        const int nr = (int)(np / RU_TM);
        // band launches on 64 x 64 tiles (rank_updateb_kernel) below np = 8192: 4.31 -> 3.98 ms at 4096; from 8192 the deep pass is the
        // critical path either way (side chain 445 -> 390 us per pair against a 393-us deep pass; 12.9-13.1 ms with 128 x 128 band tiles,
        // 13.1-13.3 with 64 x 64).  LPVS_BAND_TILE=64|128 forces one.
        const int band_env = [] { const char *e = experiment_env("LPVS_BAND_TILE"); return e ? atoi(e) : 0; }();
        const bool band64 = band_env == 64 || (band_env != 128 && np < 8192);
        auto launch_bands = [&](hipStream_t st, const RuGroup &g, int e0, int nsl, int band) {   // the tiles of bands e0 .. e0 + nsl - 1
            if (band64)
                hipLaunchKernelGGL(rank_updateb_kernel<16>, dim3((unsigned)(4 * nr * nsl)), dim3(256), 2 * 16 * 128 * sizeof(double), st, A, np, ldp, g, e0, nsl, band);
            else
                hipLaunchKernelGGL(ru_kernel, dim3((unsigned)(nr * nsl)), dim3(RU_THREADS), ldsm, st, A, np, ldp, g, tiles, (int)ht.size(), 1, e0, nsl, band, 0, 0);
        };
        auto launch_rest = [&](hipStream_t st, const RuGroup &g, int skip0, int nskip) {
            hipLaunchKernelGGL(ru_kernel, dim3((unsigned)ht.size()), dim3(RU_THREADS), ldsm, st, A, np, ldp, g, tiles, (int)ht.size(), 0, 0, 0, -1, skip0, nskip);
        };
        hipStream_t mb = aux->bulk ? aux->bulk : s;             // the deep passes' stream (CU-masked: see SweepAux::ensure)
        LPVS_HIP(hipEventRecord(aux->rest, s));                 // A is ready on the caller's stream
        LPVS_HIP(hipStreamWaitEvent(side, aux->rest, 0));
        if (mb != s) LPVS_HIP(hipStreamWaitEvent(mb, aux->rest, 0));
        chain128(0, side);
        bool rest_pending = false;                               // a rest launch of an earlier group is recorded in aux->second
        for (int kb = 0; kb < nblocks;) {
            const int mg = nblocks - kb < mgmax ? nblocks - kb : mgmax, nb0 = kb + mg;
            const int mgn = nblocks - nb0 < mgmax ? nblocks - nb0 : mgmax;       // bands of the next group (0: this is the last)
            const RuGroup g = group_of(kb, mg);
            for (int j = 1; j < mg; ++j) {
                launch_bands(side, g, kb + j, 1, j);             // band j += the group's earlier panels, then its chain
                chain128(kb + j, side);
            }
            LPVS_HIP(hipEventRecord(aux->panel, side));         // the group's panels are complete
            LPVS_HIP(hipStreamWaitEvent(mb, aux->panel, 0));
            launch_rest(mb, g, nb0, mgn);
            if (mgn > 0) {
                if (rest_pending) LPVS_HIP(hipStreamWaitEvent(side, aux->second, 0));   // the previous group's rest wrote the tiles of these bands
                launch_bands(side, g, nb0, mgn, -1);             // priority: the next group's pivot bands
                chain128(nb0, side);
            }
            LPVS_HIP(hipEventRecord(aux->second, mb));          // this group's rest
            rest_pending = true;
            LPVS_HIP(hipGetLastError());
            kb = nb0;
        }
        if (mb != s) {
            LPVS_HIP(hipEventRecord(aux->bulkdone, mb));
            LPVS_HIP(hipStreamWaitEvent(s, aux->bulkdone, 0));
        }
        LPVS_HIP(hipEventRecord(aux->rest, side));              // neither helper stream has anything pending when the caller goes on
        LPVS_HIP(hipStreamWaitEvent(s, aux->rest, 0));
        return LPVS_OK;
    }

    if (!la) {
        for (int64_t k0 = 0; k0 < np; k0 += kw_outer) {
            chain(k0, panelbuf[0], panelbuf[0] + KW * ldp, s);
            update(k0, panelbuf[0], panelbuf[0] + KW * ldp, 0, s);
            LPVS_HIP(hipGetLastError());
        }
        return LPVS_OK;
    }
    hipStream_t side = aux->side;
    LPVS_HIP(hipEventRecord(aux->rest, s));                 // A is ready on the main stream
    LPVS_HIP(hipStreamWaitEvent(side, aux->rest, 0));
    chain(0, panelbuf[0], panelbuf[0] + KW * ldp, side);
    LPVS_HIP(hipEventRecord(aux->panel, side));
    // measured at np = 8192: 18.4 -> 17.8 ms; neutral at 16384, slightly slower at 4096 (5.6 -> 5.8 ms): used from np = 6144
    const bool depth2 = np >= 6144 && [] { const char *e = experiment_env("LPVS_LOOKAHEAD"); return !(e && e[0] == '1'); }();   // LPVS_LOOKAHEAD=1: depth one
    if (depth2) {
        // Depth-2 look-ahead.  With depth one the main stream alternates  band_k | bulk_k  (the band of step k, which the next
        // pivot chain waits for, runs alone on the chip and two event hand-overs sit between consecutive bulks: 57 of 286 us
        // per step at np = 8192; profiles/r02_factor_timeline.txt).  Here the side stream owns everything the pivot chains need
        // and never blocks the bulk (the second-band launch costs the main stream 38 us, so the step shrinks to 278 us only):
        //   side:  band_k (tile row / column k+1, needs panel_k and second_{k-1})  ->  chain_{k+1}  ->  panel_{k+1}
        //   main:  bulk_k = { second band (tile row / column k+2) -> event second_k ;  everything else }   (needs panel_k)
        // band_k's tiles received the updates of steps < k as the "second band" of step k-1; the side stream may run a whole
        // step ahead, so the panels rotate through three buffers (chain_{k+1} writes while bulk_{k-1} may still read its own).
        int cur = 0;
        for (int64_t k0 = 0; k0 < np; k0 += kw_outer, cur = (cur + 1) % 3) {
            double *Bk = panelbuf[cur], *Ck = Bk + KW * ldp;
            const int64_t n0 = k0 + width(k0);
            const bool next = n0 < np;
            LPVS_HIP(hipStreamWaitEvent(s, aux->panel, 0));     // panel k (captured before the side stream re-records the event)
            if (next) {
                if (k0 > 0) LPVS_HIP(hipStreamWaitEvent(side, aux->second, 0));   // second band of step k-1 = this step's band tiles
                update(k0, Bk, Ck, 1, side);
                double *Bn = panelbuf[(cur + 1) % 3];
                chain(n0, Bn, Bn + KW * ldp, side);
                LPVS_HIP(hipEventRecord(aux->panel, side));
                if (n0 + width(n0) < np) update(k0, Bk, Ck, 3, s);
                LPVS_HIP(hipEventRecord(aux->second, s));
                update(k0, Bk, Ck, 4, s);
            } else {
                update(k0, Bk, Ck, 0, s);
            }
            LPVS_HIP(hipGetLastError());
        }
        LPVS_HIP(hipEventRecord(aux->rest, side));              // the side stream has nothing pending when the caller goes on
        LPVS_HIP(hipStreamWaitEvent(s, aux->rest, 0));
        return LPVS_OK;
    }
    int cur = 0;
    for (int64_t k0 = 0; k0 < np; k0 += kw_outer, cur ^= 1) {
        double *Bk = panelbuf[cur], *Ck = Bk + KW * ldp;
        const int64_t n0 = k0 + width(k0);
        const bool next = n0 < np;
        LPVS_HIP(hipStreamWaitEvent(s, aux->panel, 0));     // panel k (captured before the side stream re-records the event)
        if (next) {
            LPVS_HIP(hipStreamWaitEvent(side, aux->rest, 0));   // bulk of U_{k-1}
            update(k0, Bk, Ck, 1, side);
            LPVS_HIP(hipEventRecord(aux->band, side));
            chain(n0, panelbuf[cur ^ 1], panelbuf[cur ^ 1] + KW * ldp, side);
            LPVS_HIP(hipEventRecord(aux->panel, side));
            // the bulk starts when the band is done: the one-workgroup pivot inverse (which needs most of a CU's
            // registers) is then dispatched onto an empty chip instead of starving behind the bulk's workgroups
            LPVS_HIP(hipStreamWaitEvent(s, aux->band, 0));
        }
        update(k0, Bk, Ck, next ? 2 : 0, s);
        LPVS_HIP(hipEventRecord(aux->rest, s));
        LPVS_HIP(hipGetLastError());
    }
    return LPVS_OK;
}

// nbatch independent matrices A + q*np*np; work holds nbatch * spd_inverse_work_bytes(np); status_dev nbatch ints
static int32_t spd_inverse_impl(double *A, int64_t np, int nbatch, double *work, int *status_dev, hipStream_t s, SweepAux *aux) {
    if (np % 128 != 0) { set_error("spd_inverse: np=%lld not a multiple of 128", (long long)np); return LPVS_ESTATE; }
    LPVS_HIP(hipMemsetAsync(status_dev, 0, sizeof(int) * (size_t)nbatch, s));
    const bool two_level_on = [] { const char *e = experiment_env("LPVS_FACTOR"); return !(e && std::string(e) == "sweep64"); }();
    if (nbatch == 1 && np >= kTwoLevelMinNp && two_level_on) {
        LPVS_TRY(spd_inverse_two_level(A, np, work, status_dev, s, aux));
    } else {
        sweep64(A, np, nbatch, work, status_dev, s);
        LPVS_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(negate_mirror_kernel, dim3((unsigned)(np / 32), (unsigned)(np / 32), (unsigned)nbatch), dim3(256), 0, s, A, np);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t spd_inverse_inplace_batch(double *A, int64_t np, int nbatch, double *work, int *status_dev, hipStream_t s) {
    return spd_inverse_impl(A, np, nbatch, work, status_dev, s, nullptr);
}

int32_t spd_inverse_inplace(double *A, int64_t np, double *work, int *status_dev, hipStream_t s, SweepAux *aux) {
    return spd_inverse_impl(A, np, 1, work, status_dev, s, aux);
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
