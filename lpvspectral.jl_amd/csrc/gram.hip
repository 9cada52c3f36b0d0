// gram.hip -- G = Phi' Phi (and b = Phi' y) as tall-skinny f64 MFMA contractions on gfx950.
//
// The N x n regressor Phi is never read from HBM for the solve.  Two operand generators feed
// one MFMA core (v_mfma_f64_16x16x4_f64):
//   KR    (LPV, src/lasso.jl:35-50): Phi[k][f*2nb+c] = T[k][f][c>=nb] * K[k][c mod nb] -- the
//         Khatri-Rao structure is expanded IN REGISTERS: per 4-sample k-step a lane reads one
//         trig value and one activation from LDS and multiplies them into its MFMA operand, so
//         a 128x256 output tile stages only (24 freqs x 16 B + 9 x 8 B) per sample instead of
//         384 x 8 B.
//   KRS   (LPV, symmetric-pair form): K_j K_j' is symmetric in (j,j'), so
//         G[(f,c,j),(f',c',j')] = sum_k T[k][p] * (T[k][p'] * KK[k][{j,j'}]),  p=(f,c), KK = nb(nb+1)/2 pair
//         products.  Rows of the contraction are the 2Nf trig columns, columns are (p', pair) -> the same
//         MFMA core computes G3[p][p'*P+pair] for p' <= p with 2nb/(nb+1) (1.78x at nb=8) fewer flops than
//         the n x n lower triangle; an expansion kernel scatters G3 into G.  Rounding differs from KR only
//         in the grouping of each 4-factor product (<= 3 ulp per term).
//   PANEL (Fourier, src/lsfft.jl:26-49): Phi[k][c] = P[k][c], a k-major panel staged as is,
//         with the optional row weight of A' diag(W) A (src/lasso.jl:119) applied to the A operand.
// Staging is LDS-DMA (global_load_lds_dwordx4, dense lane-linear LDS images, two buffers, one
// barrier per 32-sample stage); the k-loop is software-pipelined by hand.  Work items are (lower-triangle 128x256 tile) x (sample chunk);
// each writes its partial tile to a slab and a second kernel sums the chunks in fixed order, so
// the result is bit-reproducible (no float atomics).
//
// Roofline: MFMA-bound.  Algorithmic work N*n*(n+1) flop; the kernel issues
// tiles*128*256*2*N flop (97 % useful at n = 8192: diagonal tiles are computed whole).
#include "lpvs_internal.h"

#include <cstdlib>
#include <vector>

namespace lpvs {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128;     // tile extent along a (rows of G)
constexpr int TN = 256;     // tile extent along b (cols of G)
constexpr int BK_ALIGN = 64;  // sample chunks are multiples of this (every stage depth divides it)
constexpr int NTHREADS = 512;  // 8 waves: 2 (a) x 4 (b), 64x64 outputs each

struct GramArgs {
    // common
    int64_t n;             // valid columns
    int64_t rows_per_chunk;  // multiple of BK
    int ksplit;
    int ntiles;
    const int2 *tiles;     // (ti, tj) of every lower-triangle tile
    double *slab;          // [tile][chunk][TM][TN]
    // KR
    const double2 *T;      // [Npad][Nf]
    const double *K;       // KR: [Npad][ldk], K[.][nb..ldk) == 0;  KRS: pair products KK[Npad][ldk], ldk = P
    int Nf, nb, ldk;
    int npair;             // KRS: nb(nb+1)/2 activation pairs
    int64_t nq;            // KRS: valid columns 2Nf*P (rows are the 2Nf trig columns)
    // PANEL
    const double *P;       // [Npad][ld]
    const double *W;       // [Npad] or nullptr (shared by every problem of a batch)
    int64_t ld;
    // batch of independent PANEL problems (windows): problem q uses P + q*batch_stride_P, slab + q*items*TM*TN
    int nbatch;            // 0 or 1: single problem
    int64_t batch_stride_P;
};

__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
    // 64 lanes x 16 B -> LDS at (wave-uniform base) + lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int MODE, int BK>  // MODE 0 = KR, 1 = PANEL, 2 = KRS; BK = samples per stage
__global__ void __launch_bounds__(NTHREADS, 2) gram_kernel(const GramArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: LDS-DMA bases derive from it
    const int wa = wave >> 2, wb = wave & 3;  // wave's 64x64 block inside the 128x256 tile
    // XCD-aware order: blocks are dealt round-robin to the 8 XCDs, so blocks b, b+8, b+16, ... share an L2.
    // Give each XCD a contiguous run of the (chunk, tile)-sorted work list: the tiles of one row panel
    // over one sample chunk then stream the same trig rows through the same L2 at about the same time.
    const int per_problem = a.ntiles * a.ksplit;
    const int total = per_problem * (a.nbatch > 1 ? a.nbatch : 1);
    const int bq = total / 8, br = total % 8, xcd = blockIdx.x % 8, bm = blockIdx.x / 8;   // bijective for any total
    const int item_all = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bm;
    const int prob = item_all / per_problem, item = item_all - prob * per_problem;
    const int chunk = item / a.ntiles, tile = item - chunk * a.ntiles;
    const int2 tt = a.tiles[tile];
    const int64_t a0 = (int64_t)tt.x * TM, b0 = (int64_t)tt.y * TN;
    const int64_t r_begin = (int64_t)chunk * a.rows_per_chunk;
    const int nstages = (int)(a.rows_per_chunk / BK);

    // ---- per-tile LDS image geometry -----------------------------------------------------
    // KR:    stage image = [BK][nfTot] double2 trig values, then [BK][ldk] activations
    // PANEL: stage image = [BK][TM+TN] panel values, then [BK] weights
    int nfI = 0, nfJ = 0, fI0 = 0, fJ0 = 0, nfTot = 0;
    int img_doubles;      // doubles per stage image (first part)
    int aux_doubles;      // doubles of the second part
    if (MODE == 0) {
        const int g2 = 2 * a.nb;
        const int64_t alast = (a0 + TM - 1 < a.n - 1 ? a0 + TM - 1 : a.n - 1);
        const int64_t blast = (b0 + TN - 1 < a.n - 1 ? b0 + TN - 1 : a.n - 1);
        fI0 = (int)(a0 / g2); nfI = (int)(alast / g2) - fI0 + 1;
        fJ0 = (int)(b0 / g2); nfJ = (int)(blast / g2) - fJ0 + 1;
        nfTot = nfI + nfJ;
        img_doubles = BK * nfTot * 2;
        aux_doubles = BK * a.ldk;
    } else if (MODE == 2) {
        const int64_t np2 = 2 * (int64_t)a.Nf;
        const int64_t alast = (a0 + TM - 1 < np2 - 1 ? a0 + TM - 1 : np2 - 1);
        const int64_t blast = (b0 + TN - 1 < a.nq - 1 ? b0 + TN - 1 : a.nq - 1);
        fI0 = (int)(a0 / 2); nfI = (int)(alast / 2) - fI0 + 1;
        fJ0 = (int)((b0 / a.npair) / 2); nfJ = (int)((blast / a.npair) / 2) - fJ0 + 1;
        nfTot = nfI + nfJ;
        img_doubles = BK * nfTot * 2;
        aux_doubles = BK * a.ldk;
    } else {
        img_doubles = BK * (TM + TN);
        aux_doubles = BK;
    }
    // both parts rounded up to whole 1 KiB DMA pieces so that buffers stay 16-B aligned
    const int img_pad = (img_doubles + 127) & ~127, aux_pad = (aux_doubles + 127) & ~127;
    const int stage_doubles = img_pad + aux_pad;
    double *buf0 = lds, *buf1 = lds + stage_doubles;

    // ---- per-lane operand addressing (fixed for the whole K loop) --------------------------
    // MFMA operand lane map: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]
    const int li = lane & 15, lk = lane >> 4;
    int offA[4], offB[4];      // offset (doubles) of the lane's value inside a k-row of the image
    int offKA[4], offKB[4];    // KR: offset of the activation inside a k-row of the K image
    int rowT, rowK;            // k-row strides (doubles)
    if (MODE == 0) {
        const int g2 = 2 * a.nb;
        rowT = nfTot * 2; rowK = a.ldk;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            {
                const int64_t col = a0 + wa * 64 + t * 16 + li;
                const bool ok = col < a.n;
                const int f = ok ? (int)(col / g2) : fI0;
                const int c = ok ? (int)(col - (int64_t)f * g2) : 0;
                const int cs = c >= a.nb;
                offA[t] = (f - fI0) * 2 + cs;
                offKA[t] = ok ? c - cs * a.nb : a.nb;  // K[.][nb] == 0 masks columns >= n
            }
            {
                const int64_t col = b0 + wb * 64 + t * 16 + li;
                const bool ok = col < a.n;
                const int f = ok ? (int)(col / g2) : fJ0;
                const int c = ok ? (int)(col - (int64_t)f * g2) : 0;
                const int cs = c >= a.nb;
                offB[t] = (nfI + f - fJ0) * 2 + cs;
                offKB[t] = ok ? c - cs * a.nb : a.nb;
            }
        }
    } else if (MODE == 2) {
        // rows/columns beyond the valid range are clamped, not masked: those G3 entries are never read
        rowT = nfTot * 2; rowK = a.ldk;
        const int64_t np2 = 2 * (int64_t)a.Nf;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int64_t pr = a0 + wa * 64 + t * 16 + li;
            if (pr > np2 - 1) pr = np2 - 1;
            offA[t] = (int)(pr - 2 * (int64_t)fI0);           // (f - fI0)*2 + c
            offKA[t] = 0;                                      // A operand is the trig value alone
            int64_t q = b0 + wb * 64 + t * 16 + li;
            if (q > a.nq - 1) q = a.nq - 1;
            const int64_t pc = q / a.npair;
            offB[t] = nfI * 2 + (int)(pc - 2 * (int64_t)fJ0);
            offKB[t] = (int)(q - pc * a.npair);                // pair index
        }
    } else {
        rowT = TM + TN; rowK = 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            offA[t] = wa * 64 + t * 16 + li;
            offB[t] = TM + wb * 64 + t * 16 + li;
            offKA[t] = offKB[t] = 0;
        }
    }

    // ---- stage loader: every wave copies whole 1 KiB pieces, lane-linear -------------------
    auto stage_load = [&](double *buf, int64_t r0) {
        if (MODE != 1) {
            const int total = BK * nfTot;  // double2 elements
            for (int p = wave; p * 64 < total; p += NTHREADS / 64) {
                const int e = p * 64 + lane;
                if (e < total) {
                    const int k = e / nfTot, q = e - k * nfTot;
                    const int f = q < nfI ? fI0 + q : fJ0 + (q - nfI);
                    glds16(a.T + (r0 + k) * a.Nf + f, buf + p * 128);
                }
            }
            const int totalk = BK * a.ldk / 2;  // 16-B elements; rows r0..r0+BK are contiguous
            const double *ksrc = a.K + r0 * a.ldk;
            for (int p = wave; p * 64 < totalk; p += NTHREADS / 64) {
                const int e = p * 64 + lane;
                if (e < totalk) glds16(ksrc + 2 * e, buf + img_pad + p * 128);
            }
        } else {
            // piece p = (k, part): part 0 -> I columns [a0,a0+128), parts 1,2 -> J columns
            for (int p = wave; p < BK * 3; p += NTHREADS / 64) {
                const int k = p / 3, part = p - k * 3;
                const int64_t col = part == 0 ? a0 : b0 + (part - 1) * 128;
                glds16(a.P + (int64_t)prob * a.batch_stride_P + (r0 + k) * a.ld + col + 2 * lane, buf + p * 128);
            }
            if (a.W != nullptr && wave == 0 && lane < BK / 2)
                glds16(a.W + r0 + 2 * lane, buf + img_pad);
        }
    };

    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    stage_load(buf0, r_begin);
    __syncthreads();  // drains the DMA (vmcnt(0)) and publishes the image

    // A wave whose 64x64 block lies entirely above the diagonal produces only entries nobody reads (the
    // reduce / expand kernels take the lower triangle): it keeps staging and hitting the barriers but issues no
    // MFMAs, which leaves the matrix pipe of its SIMD to the partner wave (band tiles finish ~1.8x sooner).
    bool skip_wave;
    if (MODE == 2) skip_wave = (b0 + wb * 64) / a.npair > a0 + wa * 64 + 63;       // first p' of the block > its last row p
    else skip_wave = b0 + wb * 64 > a0 + wa * 64 + 63;                             // first column > last row
    const bool weighted = (MODE == 1) && a.W != nullptr;
    // KR with 2*nb == 16: every 16-wide MFMA tile is one frequency, so all eight tiles of a lane use the
    // same activation K[k][lane & 7] -> one LDS read per k-step instead of eight.
    const bool one_k = (MODE == 0) && (2 * a.nb == 16) && (a.n % 16 == 0);
    (void)offKA;

    // raw operand fetch for k-step kk of the image `img`: the two LDS values each operand is built from
    double rT[8], rK[8];
    auto fetch = [&](const double *img, const double *aux, int kk) {
        const int row = kk * 4 + lk;
        if (MODE == 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                rT[t] = img[row * rowT + offA[t]];
                rT[4 + t] = img[row * rowT + offB[t]];
                rK[4 + t] = aux[row * rowK + offKB[t]];
            }
        } else if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                rT[t] = img[row * rowT + offA[t]];
                rT[4 + t] = img[row * rowT + offB[t]];
            }
            if (one_k) {
                rK[0] = aux[row * rowK + offKA[0]];
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    rK[t] = aux[row * rowK + offKA[t]];
                    rK[4 + t] = aux[row * rowK + offKB[t]];
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                rT[t] = img[row * rowT + offA[t]];
                rT[4 + t] = img[row * rowT + offB[t]];
            }
            if (weighted) rK[0] = aux[row];
        }
    };

    for (int s = 0; s < nstages; ++s) {
        double *cur = (s & 1) ? buf1 : buf0;
        double *nxt = (s & 1) ? buf0 : buf1;
        if (s + 1 < nstages) stage_load(nxt, r_begin + (int64_t)(s + 1) * BK);

        const double *img = cur, *aux = cur + img_pad;
        if (!skip_wave) {
            // software pipeline: the LDS reads of k-step kk+1 are issued before the 16 MFMAs of k-step kk and
            // land while the matrix pipe works (sched_barrier pins that order)
            fetch(img, aux, 0);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double opA[4], opB[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (MODE == 2) {
                        opA[t] = rT[t];
                        opB[t] = rT[4 + t] * rK[4 + t];
                    } else if (MODE == 0) {
                        opA[t] = rT[t] * (one_k ? rK[0] : rK[t]);
                        opB[t] = rT[4 + t] * (one_k ? rK[0] : rK[4 + t]);
                    } else {
                        opA[t] = weighted ? rT[t] * rK[0] : rT[t];
                        opB[t] = rT[4 + t];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < BK / 4) fetch(img, aux, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[i], opB[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // next image landed (vmcnt(0)) and everyone is done with `cur`
    }

    // ---- epilogue: partial tile -> slab[tile][chunk][TM][TN] -------------------------------
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
    if (skip_wave) return;
    double *out = a.slab + (((int64_t)prob * a.ntiles + tile) * a.ksplit + chunk) * (TM * TN);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = wa * 64 + i * 16 + lk + 4 * r;
                const int jl = wb * 64 + j * 16 + li;
                out[il * TN + jl] = acc[i][j][r];
            }
}

// G[a][b] = G[b][a] = sum_chunk slab[tile(a,b)][chunk][a%128][b%256], a >= b, fixed chunk order.
__global__ void __launch_bounds__(256)
gram_reduce_kernel(const double *__restrict__ slab, const int2 *__restrict__ tiles, int ksplit, int64_t n,
                   double *__restrict__ Gall, int64_t ldg) {
    // blockIdx.x = tile, blockIdx.y = 1/32 slice of the tile (few tiles x many chunks must still fill the chip),
    // blockIdx.z = problem of a batch
    const int tile = blockIdx.x;
    const int2 tt = tiles[tile];
    const int64_t a0 = (int64_t)tt.x * TM, b0 = (int64_t)tt.y * TN;
    const double *base = slab + ((int64_t)blockIdx.z * gridDim.x + tile) * ksplit * (TM * TN);
    double *G = Gall + (int64_t)blockIdx.z * ldg * ldg;
    constexpr int SLICE = TM * TN / 32;
    for (int e = blockIdx.y * SLICE + threadIdx.x; e < (int)(blockIdx.y + 1) * SLICE; e += 256) {
        const int il = e / TN, jl = e - il * TN;
        const int64_t ga = a0 + il, gb = b0 + jl;
        if (ga >= n || gb > ga) continue;
        double s = base[e];
        for (int c = 1; c < ksplit; ++c) s += base[(int64_t)c * (TM * TN) + e];
        G[ga * ldg + gb] = s;
        G[gb * ldg + ga] = s;
    }
}

// KRS: G3[p][q] = sum_chunk slab[tile][chunk][p%128][q%256] for the valid part of every tile (coalesced)
__global__ void __launch_bounds__(256)
gram_reduce3_kernel(const double *__restrict__ slab, const int2 *__restrict__ tiles, int ksplit, int64_t np2, int64_t nq,
                    double *__restrict__ G3) {
    const int tile = blockIdx.x;
    const int2 tt = tiles[tile];
    const int64_t a0 = (int64_t)tt.x * TM, b0 = (int64_t)tt.y * TN;
    const double *base = slab + (int64_t)tile * ksplit * (TM * TN);
    for (int e = threadIdx.x; e < TM * TN; e += 256) {
        const int il = e / TN, jl = e - il * TN;
        const int64_t gp = a0 + il, gq = b0 + jl;
        if (gp >= np2 || gq >= nq) continue;
        double s = base[e];
        for (int c = 1; c < ksplit; ++c) s += base[(int64_t)c * (TM * TN) + e];
        G3[gp * nq + gq] = s;
    }
}

// KRS: G[a][b] = G[b][a] = G3[p(a)][p(b)*P + pair(j(a), j(b))] for a >= b, a = f*2nb + c*nb + j, p = 2f + c
__global__ void __launch_bounds__(256)
gram_expand_kernel(const double *__restrict__ G3, int64_t nq, int nb, int P, int64_t n, double *__restrict__ G, int64_t ldg) {
    const int64_t ga = blockIdx.y;
    const int64_t gb = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gb > ga || ga >= n) return;
    const int64_t pa = ga / nb, pb = gb / nb;
    int ja = (int)(ga - pa * nb), jb = (int)(gb - pb * nb);
    if (ja > jb) { const int t = ja; ja = jb; jb = t; }
    const int pair = ja * nb - ja * (ja - 1) / 2 + (jb - ja);
    const double v = G3[pa * nq + pb * P + pair];
    G[ga * ldg + gb] = v;
    G[gb * ldg + ga] = v;
}

// KK[n][pair(j,j')] = K[n][j] * K[n][j'], j <= j'
__global__ void __launch_bounds__(256)
pair_table_kernel(const double *__restrict__ K, int64_t ldk, int nb, int64_t Npad, double *__restrict__ KK, int P) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= Npad * P) return;
    const int64_t r = idx / P;
    int pair = (int)(idx - r * P), j = 0;
    while (pair >= nb - j) { pair -= nb - j; ++j; }
    KK[idx] = K[r * ldk + j] * K[r * ldk + j + pair];
}

// ---- right-hand side b = Phi' (W .* y) -----------------------------------------------------
constexpr int RHS_ROWS = 512;   // samples per partial (many short, independent chains: the kernel is latency-bound otherwise)

template <int MODE>
__global__ void __launch_bounds__(256)
rhs_kernel(const double2 *__restrict__ T, int Nf, const double *__restrict__ K, int ldk, int nb,
           const double *__restrict__ P, int64_t ld, const double *__restrict__ W,
           const double *__restrict__ y, int64_t N, int64_t ncol, double *__restrict__ part) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * RHS_ROWS;
    const int64_t r1 = r0 + RHS_ROWS < N ? r0 + RHS_ROWS : N;
    if (col >= ncol) return;
    double s = 0;
    if (MODE == 0) {
        const int g2 = 2 * nb;
        const int f = (int)(col / g2), c = (int)(col - (int64_t)f * g2);
        const int cs = c >= nb, j = c - cs * nb;
        const double *Td = reinterpret_cast<const double *>(T);
        double s1 = 0, s2 = 0, s3 = 0;
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {   // four independent chains, loads of a group issued together
            const double a0 = Td[((r + 0) * Nf + f) * 2 + cs] * K[(r + 0) * ldk + j], a1 = Td[((r + 1) * Nf + f) * 2 + cs] * K[(r + 1) * ldk + j];
            const double a2 = Td[((r + 2) * Nf + f) * 2 + cs] * K[(r + 2) * ldk + j], a3 = Td[((r + 3) * Nf + f) * 2 + cs] * K[(r + 3) * ldk + j];
            s = fma(a0, y[r], s); s1 = fma(a1, y[r + 1], s1); s2 = fma(a2, y[r + 2], s2); s3 = fma(a3, y[r + 3], s3);
        }
        for (; r < r1; ++r) s = fma(Td[(r * Nf + f) * 2 + cs] * K[r * ldk + j], y[r], s);
        s = (s + s1) + (s2 + s3);
    } else {
        double s1 = 0, s2 = 0, s3 = 0;
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {
            const double p0 = P[(r + 0) * ld + col], p1 = P[(r + 1) * ld + col], p2 = P[(r + 2) * ld + col], p3 = P[(r + 3) * ld + col];
            const double y0 = W ? W[r] * y[r] : y[r], y1 = W ? W[r + 1] * y[r + 1] : y[r + 1];
            const double y2 = W ? W[r + 2] * y[r + 2] : y[r + 2], y3 = W ? W[r + 3] * y[r + 3] : y[r + 3];
            s = fma(p0, y0, s); s1 = fma(p1, y1, s1); s2 = fma(p2, y2, s2); s3 = fma(p3, y3, s3);
        }
        for (; r < r1; ++r) s = fma(P[r * ld + col], W ? W[r] * y[r] : y[r], s);
        s = (s + s1) + (s2 + s3);
    }
    part[(int64_t)blockIdx.y * ncol + col] = s;
}

// batch of windows: problem q = blockIdx.z reads P + q*strideP and y + yoff[q]; partials [q][part][ncol]
__global__ void __launch_bounds__(256)
rhs_panel_batch_kernel(const double *__restrict__ Pall, int64_t strideP, int64_t ld, const double *__restrict__ W,
                       const double *__restrict__ yall, const int64_t *__restrict__ yoff, int64_t N, int64_t ncol,
                       double *__restrict__ partall) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * RHS_ROWS;
    const int64_t r1 = r0 + RHS_ROWS < N ? r0 + RHS_ROWS : N;
    if (col >= ncol) return;
    const double *P = Pall + (int64_t)blockIdx.z * strideP;
    const double *y = yall + yoff[blockIdx.z];
    double s = 0, s1 = 0, s2 = 0, s3 = 0;
    int64_t r = r0;
    for (; r + 4 <= r1; r += 4) {
        const double p0 = P[(r + 0) * ld + col], p1 = P[(r + 1) * ld + col], p2 = P[(r + 2) * ld + col], p3 = P[(r + 3) * ld + col];
        const double y0 = W ? W[r] * y[r] : y[r], y1 = W ? W[r + 1] * y[r + 1] : y[r + 1];
        const double y2 = W ? W[r + 2] * y[r + 2] : y[r + 2], y3 = W ? W[r + 3] * y[r + 3] : y[r + 3];
        s = fma(p0, y0, s); s1 = fma(p1, y1, s1); s2 = fma(p2, y2, s2); s3 = fma(p3, y3, s3);
    }
    for (; r < r1; ++r) s = fma(P[r * ld + col], W ? W[r] * y[r] : y[r], s);
    s = (s + s1) + (s2 + s3);
    partall[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * ncol + col] = s;
}

__global__ void __launch_bounds__(256)
rhs_reduce_batch_kernel(const double *__restrict__ part, int nparts, int64_t ncol, double *__restrict__ b, int64_t ldb) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncol) return;
    double s = 0;
    for (int p = 0; p < nparts; ++p) s += part[((int64_t)blockIdx.y * nparts + p) * ncol + col];
    b[(int64_t)blockIdx.y * ldb + col] = s;
}

__global__ void __launch_bounds__(256)
rhs_reduce_kernel(const double *__restrict__ part, int nparts, int64_t ncol, double *__restrict__ b) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncol) return;
    double s = 0;
    for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * ncol + col];
    b[col] = s;
}

struct TileList {
    std::vector<int2> host;
};
TileList make_tiles(int64_t n) {
    TileList tl;
    const int64_t nti = ceil_div(n, TM);
    for (int64_t ti = 0; ti < nti; ++ti)
        for (int64_t tj = 0; tj <= ti / 2; ++tj) tl.host.push_back(make_int2((int)ti, (int)tj));
    return tl;
}

// KRS tiles: rows p in [0,2Nf), columns q = p'*P + pair; tile needed when its first column's p' <= its last row
TileList make_tiles_pairs(int64_t np2, int64_t P) {
    TileList tl;
    const int64_t nti = ceil_div(np2, TM), ntj = ceil_div(np2 * P, TN);
    for (int64_t ti = 0; ti < nti; ++ti) {
        const int64_t plast = (ti * TM + TM - 1 < np2 - 1) ? ti * TM + TM - 1 : np2 - 1;
        for (int64_t tj = 0; tj < ntj; ++tj)
            if ((tj * TN) / P <= plast) tl.host.push_back(make_int2((int)ti, (int)tj));
    }
    return tl;
}

static void choose_split(GramPlan &pl, int64_t N, int64_t nbatch = 1) {
    // split the samples so that (tiles x chunks [x problems of a batch]) fills the 256 CUs in whole rounds
    const int64_t nstage = ceil_div(N, BK_ALIGN);
    const int64_t max_split = nstage / 8 > 0 ? nstage / 8 : 1;  // >= 512 samples per chunk
    int64_t want = ceil_div(256 * 16, pl.tiles * nbatch);
    if (want > max_split) want = max_split;
    int64_t best = 1; double best_eff = -1;
    for (int64_t ks = want / 2 > 0 ? want / 2 : 1; ks <= want * 2 && ks <= max_split; ++ks) {
        const int64_t items = pl.tiles * ks * nbatch;
        const double eff = (double)items / (double)(ceil_div(items, 256) * 256);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = ks; }
    }
    pl.ksplit = best;
    pl.rows_per_chunk = round_up(ceil_div(N, pl.ksplit), BK_ALIGN);
    pl.slab_bytes = sizeof(double) * (size_t)pl.tiles * (size_t)pl.ksplit * TM * TN;
}

}  // namespace

GramPlan make_gram_plan_pairs(int64_t Nf, int64_t nb, int64_t N) {
    GramPlan pl;
    pl.n = 2 * Nf * nb; pl.N = N; pl.pairs = nb * (nb + 1) / 2; pl.np2 = 2 * Nf;
    pl.tiles = (int64_t)make_tiles_pairs(pl.np2, pl.pairs).host.size();
    choose_split(pl, N);
    return pl;
}

GramPlan make_gram_plan(int64_t n, int64_t N, int64_t nbatch) {
    GramPlan pl;
    pl.n = n; pl.N = N;
    pl.tiles = (int64_t)make_tiles(n).host.size();
    choose_split(pl, N, nbatch < 1 ? 1 : nbatch);
    return pl;
}


// Device copy of the tile list, cached per (n, pairs) on the calling thread.
static int32_t get_tiles(int64_t n, int64_t pairs, hipStream_t s, const int2 **out, int *count) {
    thread_local int64_t cached_n = -1, cached_pairs = -1;
    thread_local DevBuf cached;
    thread_local int cached_count = 0;
    thread_local int cached_dev = -1;
    int dev = 0;
    LPVS_HIP(hipGetDevice(&dev));
    if (cached_n != n || cached_pairs != pairs || cached_dev != dev) {
        TileList tl = pairs > 0 ? make_tiles_pairs(n, pairs) : make_tiles(n);
        cached.release();
        LPVS_TRY(cached.alloc(sizeof(int2) * tl.host.size()));
        LPVS_HIP(hipMemcpyAsync(cached.p, tl.host.data(), sizeof(int2) * tl.host.size(), hipMemcpyHostToDevice, s));
        LPVS_HIP(hipStreamSynchronize(s));
        cached_n = n; cached_pairs = pairs; cached_count = (int)tl.host.size(); cached_dev = dev;
    }
    *out = cached.as<int2>();
    *count = cached_count;
    return LPVS_OK;
}

static size_t gram_lds_bytes(int mode, int BK, int64_t nb, int64_t ldk) {
    int img, aux;
    if (mode == 0) {
        const int g2 = (int)(2 * nb);
        const int nfI = (TM + g2 - 2) / g2 + 1, nfJ = (TN + g2 - 2) / g2 + 1;  // upper bounds
        img = BK * (nfI + nfJ) * 2;
        aux = (int)(BK * ldk);
    } else if (mode == 2) {
        const int P = (int)(nb * (nb + 1) / 2);
        const int nfI = TM / 2 + 1, nfJ = ((TN + P - 2) / P + 1) / 2 + 2;     // upper bounds
        img = BK * (nfI + nfJ) * 2;
        aux = (int)(BK * ldk);
    } else {
        img = BK * (TM + TN);
        aux = BK;
    }
    const int img_pad = (img + 127) & ~127, aux_pad = (aux + 127) & ~127;
    return sizeof(double) * 2 * (size_t)(img_pad + aux_pad);
}

template <int MODE, int BK>
static int32_t launch_gram_t(const GramArgs &a, unsigned grid, size_t lds, hipStream_t s) {
    LPVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gram_kernel<MODE, BK>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gram_kernel<MODE, BK>), dim3(grid), dim3(NTHREADS), lds, s, a);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

constexpr size_t kLdsBudget = 160 * 1024;

int32_t launch_gram_kr(const GramPlan &pl, const double2 *T, int64_t Nf, const double *K, int64_t ldk,
                       int64_t nb, double *slab, hipStream_t s) {
    GramArgs a{};
    a.n = pl.n; a.rows_per_chunk = pl.rows_per_chunk; a.ksplit = (int)pl.ksplit;
    LPVS_TRY(get_tiles(pl.n, 0, s, &a.tiles, &a.ntiles));
    a.slab = slab; a.T = T; a.K = K; a.Nf = (int)Nf; a.nb = (int)nb; a.ldk = (int)ldk;
    const unsigned grid = (unsigned)(pl.tiles * pl.ksplit);
    // deepest stage whose two LDS images fit (few basis functions -> many frequencies per tile)
    if (gram_lds_bytes(0, 32, nb, ldk) <= kLdsBudget) return launch_gram_t<0, 32>(a, grid, gram_lds_bytes(0, 32, nb, ldk), s);
    if (gram_lds_bytes(0, 16, nb, ldk) <= kLdsBudget) return launch_gram_t<0, 16>(a, grid, gram_lds_bytes(0, 16, nb, ldk), s);
    if (gram_lds_bytes(0, 8, nb, ldk) <= kLdsBudget) return launch_gram_t<0, 8>(a, grid, gram_lds_bytes(0, 8, nb, ldk), s);
    set_error("gram_kr: LDS image exceeds 160 KiB even at 8 samples per stage (nb=%lld)", (long long)nb);
    return LPVS_EUNSUPPORTED;
}

int32_t launch_gram_krs(const GramPlan &pl, const double2 *T, int64_t Nf, const double *KK, int64_t nb, double *slab,
                        hipStream_t s) {
    GramArgs a{};
    a.n = pl.n; a.rows_per_chunk = pl.rows_per_chunk; a.ksplit = (int)pl.ksplit;
    LPVS_TRY(get_tiles(pl.np2, pl.pairs, s, &a.tiles, &a.ntiles));
    a.slab = slab; a.T = T; a.K = KK; a.Nf = (int)Nf; a.nb = (int)nb; a.ldk = (int)pl.pairs; a.npair = (int)pl.pairs;
    a.nq = pl.np2 * pl.pairs;
    const unsigned grid = (unsigned)(pl.tiles * pl.ksplit);
    if (gram_lds_bytes(2, 32, nb, pl.pairs) <= kLdsBudget) return launch_gram_t<2, 32>(a, grid, gram_lds_bytes(2, 32, nb, pl.pairs), s);
    if (gram_lds_bytes(2, 16, nb, pl.pairs) <= kLdsBudget) return launch_gram_t<2, 16>(a, grid, gram_lds_bytes(2, 16, nb, pl.pairs), s);
    set_error("gram_krs: LDS image exceeds 160 KiB (nb=%lld)", (long long)nb);
    return LPVS_EUNSUPPORTED;
}

bool gram_krs_fits(int64_t nb) {
    const int64_t P = nb * (nb + 1) / 2;
    return nb >= 2 && gram_lds_bytes(2, 16, nb, P) <= kLdsBudget;
}

int32_t launch_pair_table(const double *K, int64_t ldk, int64_t nb, int64_t Npad, double *KK, hipStream_t s) {
    const int P = (int)(nb * (nb + 1) / 2);
    hipLaunchKernelGGL(pair_table_kernel, dim3((unsigned)ceil_div(Npad * P, 256)), dim3(256), 0, s, K, ldk, (int)nb, Npad, KK, P);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

// slab -> G3 (2Nf x 2Nf*P) -> G (n x n, ldg), both symmetric halves written
int32_t launch_gram_reduce_krs(const GramPlan &pl, const double *slab, int64_t nb, double *G3, double *G, int64_t ldg,
                               hipStream_t s) {
    const int2 *tiles; int nt;
    LPVS_TRY(get_tiles(pl.np2, pl.pairs, s, &tiles, &nt));
    const int64_t nq = pl.np2 * pl.pairs;
    hipLaunchKernelGGL(gram_reduce3_kernel, dim3((unsigned)nt), dim3(256), 0, s, slab, tiles, (int)pl.ksplit, pl.np2, nq, G3);
    LPVS_HIP(hipGetLastError());
    dim3 grid((unsigned)ceil_div(pl.n, 256), (unsigned)pl.n);
    hipLaunchKernelGGL(gram_expand_kernel, grid, dim3(256), 0, s, G3, nq, (int)nb, (int)pl.pairs, pl.n, G, ldg);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_gram_panel(const GramPlan &pl, const double *P, int64_t ld, const double *W, double *slab,
                          hipStream_t s) {
    GramArgs a{};
    a.n = pl.n; a.rows_per_chunk = pl.rows_per_chunk; a.ksplit = (int)pl.ksplit;
    LPVS_TRY(get_tiles(pl.n, 0, s, &a.tiles, &a.ntiles));
    a.slab = slab; a.P = P; a.W = W; a.ld = ld;
    return launch_gram_t<1, 16>(a, (unsigned)(pl.tiles * pl.ksplit), gram_lds_bytes(1, 16, 0, 0), s);  // 2 x 48 KiB
}

// batch of nbatch independent panel problems of identical shape (windows of ls_windowpsd)
int32_t launch_gram_panel_batch(const GramPlan &pl, int nbatch, const double *P, int64_t batch_stride_P, int64_t ld,
                                const double *W, double *slab, hipStream_t s) {
    GramArgs a{};
    a.n = pl.n; a.rows_per_chunk = pl.rows_per_chunk; a.ksplit = (int)pl.ksplit;
    LPVS_TRY(get_tiles(pl.n, 0, s, &a.tiles, &a.ntiles));
    a.slab = slab; a.P = P; a.W = W; a.ld = ld; a.nbatch = nbatch; a.batch_stride_P = batch_stride_P;
    return launch_gram_t<1, 16>(a, (unsigned)(pl.tiles * pl.ksplit * nbatch), gram_lds_bytes(1, 16, 0, 0), s);
}

int32_t launch_gram_reduce_batch(const GramPlan &pl, int nbatch, const double *slab, double *G, int64_t ldg, hipStream_t s) {
    const int2 *tiles; int nt;
    LPVS_TRY(get_tiles(pl.n, 0, s, &tiles, &nt));
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)nt, 32, (unsigned)nbatch), dim3(256), 0, s, slab, tiles, (int)pl.ksplit, pl.n, G, ldg);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_gram_reduce(const GramPlan &pl, const double *slab, double *G, int64_t ldg, hipStream_t s) {
    const int2 *tiles; int nt;
    LPVS_TRY(get_tiles(pl.n, 0, s, &tiles, &nt));
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)nt, 32, 1), dim3(256), 0, s, slab, tiles, (int)pl.ksplit, pl.n, G, ldg);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

size_t rhs_scratch_bytes(int64_t N, int64_t n) { return sizeof(double) * (size_t)ceil_div(N, RHS_ROWS) * (size_t)n; }

int32_t launch_rhs_kr(const double2 *T, int64_t Nf, const double *K, int64_t ldk, int64_t nb, const double *y,
                      int64_t N, double *b, double *scratch, size_t scratch_bytes, hipStream_t s) {
    const int64_t n = 2 * Nf * nb;
    const int64_t parts = ceil_div(N, RHS_ROWS);
    if (scratch_bytes < rhs_scratch_bytes(N, n)) { set_error("rhs scratch too small"); return LPVS_ESTATE; }
    dim3 grid((unsigned)ceil_div(n, 256), (unsigned)parts);
    hipLaunchKernelGGL(rhs_kernel<0>, grid, dim3(256), 0, s, T, (int)Nf, K, (int)ldk, (int)nb, (const double *)nullptr,
                       (int64_t)0, (const double *)nullptr, y, N, n, scratch);
    LPVS_HIP(hipGetLastError());
    hipLaunchKernelGGL(rhs_reduce_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, scratch, (int)parts, n, b);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_rhs_panel_batch(int nbatch, const double *P, int64_t strideP, int64_t ld, int64_t ncol, const double *W,
                               const double *y, const int64_t *yoff_dev, int64_t N, double *b, int64_t ldb, double *scratch,
                               size_t scratch_bytes, hipStream_t s) {
    const int64_t parts = ceil_div(N, RHS_ROWS);
    if (scratch_bytes < rhs_scratch_bytes(N, ncol) * (size_t)nbatch) { set_error("rhs scratch too small"); return LPVS_ESTATE; }
    dim3 grid((unsigned)ceil_div(ncol, 256), (unsigned)parts, (unsigned)nbatch);
    hipLaunchKernelGGL(rhs_panel_batch_kernel, grid, dim3(256), 0, s, P, strideP, ld, W, y, yoff_dev, N, ncol, scratch);
    LPVS_HIP(hipGetLastError());
    hipLaunchKernelGGL(rhs_reduce_batch_kernel, dim3((unsigned)ceil_div(ncol, 256), (unsigned)nbatch), dim3(256), 0, s, scratch,
                       (int)parts, ncol, b, ldb);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_rhs_panel(const double *P, int64_t ld, int64_t ncol, const double *W, const double *y, int64_t N,
                         double *b, double *scratch, size_t scratch_bytes, hipStream_t s) {
    const int64_t parts = ceil_div(N, RHS_ROWS);
    if (scratch_bytes < rhs_scratch_bytes(N, ncol)) { set_error("rhs scratch too small"); return LPVS_ESTATE; }
    dim3 grid((unsigned)ceil_div(ncol, 256), (unsigned)parts);
    hipLaunchKernelGGL(rhs_kernel<1>, grid, dim3(256), 0, s, (const double2 *)nullptr, 0, (const double *)nullptr, 0, 0, P,
                       ld, W, y, N, ncol, scratch);
    LPVS_HIP(hipGetLastError());
    hipLaunchKernelGGL(rhs_reduce_kernel, dim3((unsigned)ceil_div(ncol, 256)), dim3(256), 0, s, scratch, (int)parts, ncol, b);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

}  // namespace lpvs
