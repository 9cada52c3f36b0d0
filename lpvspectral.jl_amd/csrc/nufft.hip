// Slot sums of the structured Gram by a type-1 non-uniform FFT (hand-written HIP, gfx950).
//
// nudft.hip evaluates   tab[j][q] = sum_n c_q[n] {cos, sin}(j D x_n),   j < nslots,  q < nq (+ the x-weighted twins)
// directly: 4 N nslots nq FMAs (cfg3: 1.6e11, 9.8 ms).  When the slot frequencies are the integer multiples j*D of ONE step
// (the merged slot layout of api.hip) these are the Fourier coefficients of a non-uniform point set at theta_n = D x_n mod 2 pi,
// and the standard three steps get them in O(N w + nf nslots) per weight vector:
//   1. spread:  every sample is convolved onto a fine periodic grid of nf >= 3.9 nslots points (nf a power of two) with the
//      "exponential of semicircle" kernel  phi(z) = exp(beta (sqrt(1 - z^2) - 1)),  w = 16 grid points wide, beta = 2.30 w
//      (Barnett, Magland, af Klinteberg, SIAM J. Sci. Comput. 41 (2019)): 2e-14 of sum|c| in a coefficient at this oversampling,
//      aliasing and the double rounding of steps 2 and 3 together (w = 14: 3e-14, w = 12: 2e-12; tools/nufft_accuracy.py restates
//      the experiment in numpy against long-double direct sums);
//   2. the nslots wanted modes of the grid by a pruned direct DFT (nf * nslots complex MACs per grid: 0.3 GFLOP per grid at cfg3,
//      not worth an FFT);
//   3. division by the kernel's Fourier transform phihat(j) (Gauss-Legendre quadrature on the host, long double).
// Work at cfg3 (N = 2^20, 1032 slots, 72 weight vectors): 1.2e9 fixed-point accumulations + 3e8 complex MACs instead of 1.6e11 FMAs.
//
// DETERMINISM.  Spreading is a scatter; floating-point atomics would make the result depend on the order the hardware
// happens to serve them in.  The grid is therefore accumulated in 64-bit FIXED POINT (LDS ds_add_u64, then integer sums
// over the sample chunks): integer addition is associative, so the result does not depend on any order.  A weight vector's
// quantum is q = N max|c| 2^-62 (no overflow for any input); an addend is rounded to it once, by the FMA that also adds the
// 1.5 * 2^52 "magic" constant whose mantissa then holds the integer: error <= q/2 = max|c| 2^-43 per addend, ~3e-16 (max|c| /
// mean|c|) of sum|c| in a coefficient -- below the kernel's own error.
//
// Phase: theta_n is formed as in nudft.hip -- D in double-double, the product D x_n with an FMA-exact low part, reduced by 2 pi
// in double-double -- so it is the phase of the REAL progression step.
#include "lpvs_internal.h"

#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace lpvs {
namespace {

constexpr int NU_W = 16;                             // kernel width (grid points)
constexpr double NU_BETA = 2.30 * NU_W;
// grids per spreading workgroup: GPW x nf x 8 B of LDS -- 4 for nf <= 4096, 2 for nf = 8192 (the cfg5 slot count)
static int nu_gpw(int nf) { return nf <= 4096 ? 4 : 2; }
constexpr double NU_MAGIC = 6755399441055744.0;      // 1.5 * 2^52
constexpr int NU_CHUNK = 32768;                      // samples per spreading workgroup

// sample -> first grid cell of its support and the 16 kernel values
__global__ void __launch_bounds__(256)
nufft_coord_kernel(const double *__restrict__ x, int64_t N, double D_hi, double D_lo, int nf, int *__restrict__ cell0, double *__restrict__ taps) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    constexpr double I2PI_HI = 0.15915494309189535, I2PI_LO = -9.8393384885635288e-18;   // 1 / (2 pi) in double-double
    const double xv = x[i];
    const double th = D_hi * xv, tl = fma(D_hi, xv, -th) + D_lo * xv;                     // D x (double-double)
    const double uh = th * I2PI_HI, ul = fma(th, I2PI_HI, -uh) + (th * I2PI_LO + tl * I2PI_HI);   // turns
    double r = (uh - rint(uh)) + ul;
    r -= floor(r);                                   // [0, 1)
    double p = r * (double)nf;
    if (p >= (double)nf) p -= (double)nf;
    const int g0 = (int)ceil(p - 0.5 * NU_W);        // grid points g0 .. g0 + 15 lie within |g - p| <= 8
    cell0[i] = (g0 + nf) & (nf - 1);
#pragma unroll
    for (int k = 0; k < NU_W; ++k) {
        const double z = ((double)(g0 + k) - p) * (2.0 / NU_W);
        const double s = 1.0 - z * z;
        taps[i * NU_W + k] = s > 0.0 ? exp(NU_BETA * (sqrt(s) - 1.0)) : exp(-NU_BETA);
    }
}

// max |Wt[n][q]| per column, one coalesced pass (order-independent: integer max on the bit patterns of non-negative doubles)
__global__ void __launch_bounds__(256)
nufft_colmax_kernel(const double *__restrict__ Wt, const double *__restrict__ y, int64_t ldw, int nq, int64_t N, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long m[256];
    for (int q = threadIdx.x; q < 256; q += 256) m[q] = 0;
    __syncthreads();
    const int64_t total = N * ldw;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int q = (int)(e % ldw);
        if (q < nq) {
            const double v = fabs(y ? Wt[e] * y[e / ldw] : Wt[e]);
            // (a NaN weight counts as +Inf: the caller then takes the direct sums, which propagate it as the reference does)
            if (v > 0.0 || v != v) atomicMax(&m[q & 255], v != v ? 0x7ff0000000000000ull : (unsigned long long)__double_as_longlong(v));
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nq && q < 256; q += 256) if (m[q]) atomicMax(out + q, m[q]);
}

// grid gamma = 2 q + twin (twin: weights x_n c_q[n]);  invq[gamma] = 1 / quantum.  Workgroup (group of GPW grids = GPW/2 weight
// columns, chunk of samples).
template <int GPW>
__global__ void __launch_bounds__(256)
nufft_spread_kernel(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ Wt, int64_t ldw, int nq, int64_t N, int nf,
                    const int *__restrict__ cell0, const double *__restrict__ taps, const double *__restrict__ invq,
                    long long *__restrict__ partial, int ngroups, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) long long G[];          // [GPW][nf]
    // consecutive workgroup ids go round the eight XCDs: all column groups of a chunk are dealt to ONE XCD, so the chunk's kernel
    // values (4 MB) are fetched into one L2 instead of eight (PMC before: 2.2 GB fetched per launch at cfg3, 16x the table)
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int gg = jj % ngroups, ch = xcd + 8 * (jj / ngroups);
    if (ch >= nchunks) return;
    for (int e = threadIdx.x; e < GPW * nf; e += 256) G[e] = 0;
    __syncthreads();
    constexpr int NC = GPW / 2;                      // weight columns of the group (plain + twin grid each)
    const int q0 = gg * NC;
    double iq[GPW];
#pragma unroll
    for (int c = 0; c < NC; ++c) { iq[2 * c] = q0 + c < nq ? invq[2 * (q0 + c)] : 0.0; iq[2 * c + 1] = q0 + c < nq ? invq[2 * (q0 + c) + 1] : 0.0; }
    const int64_t n0 = (int64_t)ch * NU_CHUNK, n1 = n0 + NU_CHUNK < N ? n0 + NU_CHUNK : N;
    const long long magic_bits = __double_as_longlong(NU_MAGIC);
    for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
        const int c0 = cell0[n];
        const double xv = x[n], yv = y ? y[n] : 1.0;
        double cw[GPW];                              // weights in quanta
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double wv = q0 + c < nq ? Wt[n * ldw + q0 + c] * yv : 0.0;
            cw[2 * c] = wv * iq[2 * c];
            cw[2 * c + 1] = (xv * wv) * iq[2 * c + 1];
        }
        const double2 *tp = reinterpret_cast<const double2 *>(taps + n * NU_W);
#pragma unroll
        for (int k2 = 0; k2 < NU_W / 2; ++k2) {
            const double2 t = tp[k2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double tv = h ? t.y : t.x;
                const int cell = (c0 + 2 * k2 + h) & (nf - 1);
#pragma unroll
                for (int gi = 0; gi < GPW; ++gi) {
                    const long long a = __double_as_longlong(fma(cw[gi], tv, NU_MAGIC)) - magic_bits;   // round(cw * tap), exactly
                    atomicAdd(reinterpret_cast<unsigned long long *>(&G[gi * nf + cell]), (unsigned long long)a);
                }
            }
        }
    }
    __syncthreads();
    long long *out = partial + ((int64_t)ch * ngroups + gg) * GPW * nf;
    for (int e = threadIdx.x; e < GPW * nf; e += 256) out[e] = G[e];
}

// integer sum of the chunk partials -> grid[gamma][g] as doubles (gamma = 2 q + twin)
__global__ void __launch_bounds__(256)
nufft_reduce_kernel(const long long *__restrict__ partial, int ngroups, int nchunks, int nf, int gpw, double *__restrict__ grid) {
    const int gamma = blockIdx.y, q = gamma >> 1, twin = gamma & 1;
    const int gg = q / (gpw / 2), gi = (q % (gpw / 2)) * 2 + twin;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= nf) return;
    long long s = 0;
    for (int ch = 0; ch < nchunks; ++ch) s += partial[(((int64_t)ch * ngroups + gg) * gpw + gi) * nf + g];
    grid[(int64_t)gamma * nf + g] = (double)s;
}

// grid gamma: pruned DFT for the modes of this workgroup's block, deconvolution, scaling
//   tab[(j * nq + q) * 4 + 2 twin + {0, 1}] = {Re, Im} sum_g grid[g] e^{+2 pi i j g / nf} * quantum / phihat[j]
// HALF: the twiddle table holds g < nf / 2 only (e^{2 pi i (g + nf/2) / nf} = -e^{2 pi i g / nf}, the sign goes onto the grid value):
// 16 nf instead of 24 nf bytes of LDS, which is what lets nf = 8192 fit.
template <bool HALF>
__global__ void __launch_bounds__(256)
nufft_modes_kernel(const double *__restrict__ grid, int nf, int nq, int mode0, int nslots,
                   const double *__restrict__ scale /* [2 nq][nslots] = quantum / phihat(mode0 + slot) */, double *__restrict__ tab,
                   int64_t scale_stride /* nslots; 0: one row for every grid */, int64_t grid_bstride, int64_t tab_bstride /* blockIdx.z = batch entry */) {
    grid += (int64_t)blockIdx.z * grid_bstride;
    tab += (int64_t)blockIdx.z * tab_bstride;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Gd = lds;                                // [nf]
    double2 *tw = reinterpret_cast<double2 *>(lds + nf);   // [nf] or [nf / 2]  e^{2 pi i g / nf}
    const int gamma = blockIdx.x, q = gamma >> 1, twin = gamma & 1;
    const int ntw = HALF ? nf >> 1 : nf;
    for (int g = threadIdx.x; g < nf; g += 256) {
        Gd[g] = grid[(int64_t)gamma * nf + g];
        if (g < ntw) {
            double sn, cs;
            sincospi(2.0 * (double)g / (double)nf, &sn, &cs);
            tw[g] = make_double2(cs, sn);
        }
    }
    __syncthreads();
    const int j = blockIdx.y * 256 + threadIdx.x;    // output slot j = mode mode0 + j
    if (j >= nslots) return;
    double ac0 = 0, as0 = 0, ac1 = 0, as1 = 0;       // two interleaved chains (even / odd g), added at the end: fixed order
    int idx = 0;
    const int mask = nf - 1, mj = (mode0 + j) & mask;
    const int hmask = mask >> 1, hbit = nf >> 1;
    for (int g = 0; g < nf; g += 2) {
        const double2 t0 = tw[HALF ? idx & hmask : idx];
        const bool n0 = HALF && (idx & hbit);
        idx = (idx + mj) & mask;
        const double2 t1 = tw[HALF ? idx & hmask : idx];
        const bool n1 = HALF && (idx & hbit);
        idx = (idx + mj) & mask;
        double v0 = Gd[g], v1 = Gd[g + 1];
        if (HALF) { v0 = n0 ? -v0 : v0; v1 = n1 ? -v1 : v1; }
        ac0 = fma(v0, t0.x, ac0); as0 = fma(v0, t0.y, as0);
        ac1 = fma(v1, t1.x, ac1); as1 = fma(v1, t1.y, as1);
    }
    const double sc = scale[(int64_t)gamma * scale_stride + j];
    double *o = tab + ((int64_t)j * nq + q) * 4 + 2 * twin;
    o[0] = (ac0 + ac1) * sc;
    o[1] = (as0 + as1) * sc;
}


// ---- windows of one long signal (the batched-window engine): ONE weight vector per window and signal, so a workgroup owns a
// window outright: its maxima (-> quanta), the spreading of up to two weight columns (A: the window function alone -- the Gram --
// or a signal; B: a signal) with their x-weighted twins into four LDS grids, kernel values computed on the fly (nothing is shared
// between weight vectors that would pay for a table), and the scaled grids to global.  A window's result depends on its own
// samples only: window shards reproduce the whole run bit for bit.
__global__ void __launch_bounds__(256)
nufft_window_kernel(const double *__restrict__ x, const double *__restrict__ W, const double *__restrict__ yA, const double *__restrict__ yB, int hasB,
                    const int64_t *__restrict__ offs, int64_t n, double D_hi, double D_lo, int nf, double *__restrict__ gridA,
                    double *__restrict__ gridB, int64_t grid_bstride) {
    extern __shared__ __attribute__((aligned(16))) long long G[];          // [4][nf]: A, x A, B, x B
    __shared__ unsigned long long smax[4];
    const int win = blockIdx.x;
    const int64_t r0 = offs[win];
    const int ng = hasB ? 4 : 2;
    for (int e = threadIdx.x; e < ng * nf; e += 256) G[e] = 0;
    if (threadIdx.x < 4) smax[threadIdx.x] = 0;
    __syncthreads();
    // maxima of |c| and |x c| (integer max on the bit patterns of non-negative doubles: order-independent)
    {
        double m[4] = {0, 0, 0, 0};
        for (int64_t i = threadIdx.x; i < n; i += 256) {
            const double xv = fabs(x[r0 + i]), wv = W ? W[i] : 1.0;
            const double cA = fabs(yA ? wv * yA[r0 + i] : wv), cB = hasB ? fabs(wv * yB[r0 + i]) : 0.0;
            m[0] = fmax(m[0], cA); m[1] = fmax(m[1], xv * cA); m[2] = fmax(m[2], cB); m[3] = fmax(m[3], xv * cB);
            if (!(xv * cA < 1.0 / 0.0)) m[0] = m[1] = 1.0 / 0.0;      // NaN or Inf in x, W or the signal: the window's sums are NaN, as the
            if (!(xv * cB < 1.0 / 0.0)) m[2] = m[3] = 1.0 / 0.0;      // direct sums (and the reference) would have them
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (m[g] > 0.0) atomicMax(&smax[g], (unsigned long long)__double_as_longlong(m[g]));
    }
    __syncthreads();
    double quantum[4], iq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const double bound = __longlong_as_double((long long)smax[g]) * (double)n;   // >= the sum of |weights| falling into any cell
        const bool finite = bound < 1.0 / 0.0;
        int e = 0;
        (void)frexp(bound > 0.0 && finite ? bound : 1.0, &e);                        // bound < 2^e
        quantum[g] = finite ? ldexp(1.0, e - 62) : __longlong_as_double(0x7ff8000000000000ll);   // power of two: the scalings are exact
        iq[g] = bound > 0.0 && finite ? 1.0 / ldexp(1.0, e - 62) : 0.0;
    }
    constexpr double I2PI_HI = 0.15915494309189535, I2PI_LO = -9.8393384885635288e-18;   // 1 / (2 pi) in double-double
    const long long magic_bits = __double_as_longlong(NU_MAGIC);
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double xv = x[r0 + i], wv = W ? W[i] : 1.0;
        const double cA = yA ? wv * yA[r0 + i] : wv, cB = hasB ? wv * yB[r0 + i] : 0.0;
        const double cw[4] = {cA * iq[0], (xv * cA) * iq[1], cB * iq[2], (xv * cB) * iq[3]};
        const double th = D_hi * xv, tl = fma(D_hi, xv, -th) + D_lo * xv;                     // D x (double-double), as nufft_coord_kernel
        const double uh = th * I2PI_HI, ul = fma(th, I2PI_HI, -uh) + (th * I2PI_LO + tl * I2PI_HI);
        double r = (uh - rint(uh)) + ul;
        r -= floor(r);
        double p = r * (double)nf;
        if (p >= (double)nf) p -= (double)nf;
        const int g0 = (int)ceil(p - 0.5 * NU_W);
#pragma unroll 4
        for (int k = 0; k < NU_W; ++k) {
            const double z = ((double)(g0 + k) - p) * (2.0 / NU_W);
            const double sq = 1.0 - z * z;
            const double tv = sq > 0.0 ? exp(NU_BETA * (sqrt(sq) - 1.0)) : exp(-NU_BETA);
            const int cell = (g0 + k + nf) & (nf - 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g >= 2 && !hasB) break;          // (uniform)
                const long long a = __double_as_longlong(fma(cw[g], tv, NU_MAGIC)) - magic_bits;   // round(cw * tap), exactly
                atomicAdd(reinterpret_cast<unsigned long long *>(&G[g * nf + cell]), (unsigned long long)a);
            }
        }
    }
    __syncthreads();
    double *oA = gridA + (int64_t)win * grid_bstride, *oB = hasB ? gridB + (int64_t)win * grid_bstride : nullptr;
    for (int e = threadIdx.x; e < 2 * nf; e += 256) {
        const int twin = e >= nf;
        oA[e] = (double)G[e] * quantum[twin];
        if (hasB) oB[e] = (double)G[2 * nf + e] * quantum[2 + twin];
    }
}

}  // namespace

// phihat(j) = int_{-w/2}^{w/2} phi(2 t / w) cos(2 pi j t / nf) dt  (grid units), j < nmodes: 96-point Gauss-Legendre in long double;
// depends on the grid size only -- computed once per process and (nf, nmodes)
static std::vector<long double> nufft_phihat(int nf, int nmodes) {
    static const std::vector<std::pair<long double, long double>> gl = [] {
        const int n = 96;                                                      // nodes / weights on [-1, 1] by Newton on P_n
        std::vector<std::pair<long double, long double>> r((size_t)n);
        const long double pi = 3.141592653589793238462643383279502884L;
        for (int i = 0; i < n; ++i) {
            long double z = cosl(pi * (i + 0.75L) / (n + 0.5L)), pp = 0;
            for (int it = 0; it < 100; ++it) {
                long double p1 = 1, p2 = 0;
                for (int k = 1; k <= n; ++k) { const long double p3 = p2; p2 = p1; p1 = ((2 * k - 1) * z * p2 - (k - 1) * p3) / k; }
                pp = n * (z * p1 - p2) / (z * z - 1);
                const long double dz = p1 / pp;
                z -= dz;
                if (fabsl(dz) < 1e-19L) break;
            }
            r[(size_t)i] = {z, 2 / ((1 - z * z) * pp * pp)};
        }
        return r;
    }();
    static std::mutex phat_mu;
    static std::map<std::pair<int, int>, std::vector<long double>> phat_cache;
    std::lock_guard<std::mutex> lk(phat_mu);
    auto it = phat_cache.find({nf, nmodes});
    if (it == phat_cache.end()) {
        std::vector<long double> v((size_t)nmodes);
        const long double pi = 3.141592653589793238462643383279502884L, half = 0.5L * NU_W;
        for (int j = 0; j < nmodes; ++j) {
            long double acc = 0;
            for (const auto &nw : gl) {
                const long double z = nw.first, t = z * half;
                acc += nw.second * half * expl((long double)NU_BETA * (sqrtl(1 - z * z) - 1)) * cosl(2 * pi * (long double)j * t / (long double)nf);
            }
            v[(size_t)j] = acc;
        }
        it = phat_cache.emplace(std::make_pair(nf, nmodes), std::move(v)).first;
    }
    return it->second;
}

// fine grid for nslots modes: the smallest power of two >= 3.9 nslots (oversampling ~2 of the two-sided mode range)
int nufft_grid_size(int64_t nslots) {
    int nf = 256;
    while ((double)nf < 3.9 * (double)nslots) nf *= 2;
    return nf;
}
bool nufft_applicable(int64_t N, int64_t nslots, int64_t nq) {
    const int nf = nufft_grid_size(nslots);
    return nf <= 8192 && N >= 4096 && nq >= 1 && nq <= 256;   // (at least two grids of nf 64-bit cells have to fit the LDS)
}
size_t nufft_work_bytes(int64_t N, int64_t nslots, int64_t nq) {
    const int nf = nufft_grid_size(nslots), gpw = nu_gpw(nf);
    const int64_t ngroups = (nq + gpw / 2 - 1) / (gpw / 2), nchunks = (N + NU_CHUNK - 1) / NU_CHUNK;
    size_t b = 0;
    b += sizeof(double) * (size_t)N * NU_W;                                   // taps
    b += ((sizeof(int) * (size_t)N + 255) / 256) * 256;                       // cell0
    b += sizeof(long long) * (size_t)nchunks * (size_t)ngroups * (size_t)gpw * (size_t)nf;   // chunk partial grids
    b += sizeof(double) * (size_t)(2 * nq) * (size_t)(nslots + 1);            // scale table + 1 / quantum
    b += sizeof(double) * (size_t)(2 * nq) * (size_t)nf;                      // reduced grids
    b += sizeof(unsigned long long) * (size_t)nq + 256;                       // column maxima
    return b;
}

// tab[nslots][nq][4] = sum_n y_n Wt[n][q] {cos, sin, x cos, x sin}((mode0 + j) D x_n),  D = D_hi + D_lo;  xam = max |x|; y may be
// nullptr (= 1).  nf = fine grid size (>= nufft_grid_size(mode0 + nslots)); reuse_coords: the sample coordinates at the head of
// `work` are those of an earlier call with the same x, N, D and nf.
int32_t launch_nufft_tab(const double *x, const double *y, int64_t N, double xam, const double *Wt, int64_t ldw, int nq, double D_hi, double D_lo,
                         int mode0, int nslots, int nf, bool reuse_coords, void *work, double *tab, hipStream_t s) {
    const int gpw = nu_gpw(nf);
    const int ngroups = (nq + gpw / 2 - 1) / (gpw / 2), nchunks = (int)((N + NU_CHUNK - 1) / NU_CHUNK);
    unsigned char *wp = static_cast<unsigned char *>(work);
    double *taps = reinterpret_cast<double *>(wp); wp += sizeof(double) * (size_t)N * NU_W;
    int *cell0 = reinterpret_cast<int *>(wp); wp += ((sizeof(int) * (size_t)N + 255) / 256) * 256;
    long long *partial = reinterpret_cast<long long *>(wp); wp += sizeof(long long) * (size_t)nchunks * (size_t)ngroups * (size_t)gpw * (size_t)nf;
    double *scale = reinterpret_cast<double *>(wp); wp += sizeof(double) * (size_t)(2 * nq) * (size_t)nslots;
    double *invq = reinterpret_cast<double *>(wp); wp += sizeof(double) * (size_t)(2 * nq);
    double *grid = reinterpret_cast<double *>(wp); wp += sizeof(double) * (size_t)(2 * nq) * (size_t)nf;
    unsigned long long *colmax = reinterpret_cast<unsigned long long *>(wp);

    LPVS_HIP(hipMemsetAsync(colmax, 0, sizeof(unsigned long long) * (size_t)nq, s));
    if (!reuse_coords) hipLaunchKernelGGL(nufft_coord_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, x, N, D_hi, D_lo, nf, cell0, taps);
    if (nq > 256) { set_error("nufft: more than 256 weight vectors"); return LPVS_EUNSUPPORTED; }
    hipLaunchKernelGGL(nufft_colmax_kernel, dim3(1024), dim3(256), 0, s, Wt, y, ldw, nq, N, colmax);
    LPVS_HIP(hipGetLastError());
    // quanta and the per-mode scale on the host (nq maxima come back: one small synchronising copy)
    std::vector<unsigned long long> hm((size_t)nq);
    LPVS_HIP(hipMemcpyAsync(hm.data(), colmax, sizeof(unsigned long long) * (size_t)nq, hipMemcpyDeviceToHost, s));
    LPVS_HIP(hipStreamSynchronize(s));
    for (int q = 0; q < nq; ++q)                     // non-finite weights (or abscissae): no fixed-point grid can hold them
        if (!std::isfinite(__builtin_bit_cast(double, hm[(size_t)q])) || !std::isfinite(xam)) return kNufftNonFinite;
    std::vector<double> hscale((size_t)(2 * nq) * (size_t)nslots), hinvq((size_t)(2 * nq));
    const std::vector<long double> phat = nufft_phihat(nf, mode0 + nslots);
    for (int q = 0; q < nq; ++q) {
        const double cmax = __builtin_bit_cast(double, hm[(size_t)q]);
        for (int twin = 0; twin < 2; ++twin) {
            const double bound = (twin ? xam : 1.0) * cmax * (double)N;        // >= sum of |weights|
            int e = 0;
            (void)std::frexp(bound > 0 ? bound : 1.0, &e);                     // bound < 2^e
            const double quantum = std::ldexp(1.0, e - 62);                    // power of two: the scalings are exact
            hinvq[(size_t)(2 * q + twin)] = bound > 0 ? 1.0 / quantum : 0.0;
            for (int j = 0; j < nslots; ++j) hscale[(size_t)(2 * q + twin) * (size_t)nslots + (size_t)j] = (double)((long double)quantum / phat[(size_t)(mode0 + j)]);
        }
    }
    LPVS_TRY(copy_to_device(scale, hscale.data(), sizeof(double) * hscale.size(), s));
    LPVS_TRY(copy_to_device(invq, hinvq.data(), sizeof(double) * hinvq.size(), s));
    const size_t lds_spread = sizeof(long long) * (size_t)gpw * (size_t)nf;
    if (gpw == 4) {
        LPVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&nufft_spread_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_spread));
        hipLaunchKernelGGL(nufft_spread_kernel<4>, dim3((unsigned)(8 * ngroups * ceil_div(nchunks, 8))), dim3(256), lds_spread, s, x, y, Wt, ldw, nq, N, nf, cell0,
                           taps, invq, partial, ngroups, nchunks);
    } else {
        LPVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&nufft_spread_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_spread));
        hipLaunchKernelGGL(nufft_spread_kernel<2>, dim3((unsigned)(8 * ngroups * ceil_div(nchunks, 8))), dim3(256), lds_spread, s, x, y, Wt, ldw, nq, N, nf, cell0,
                           taps, invq, partial, ngroups, nchunks);
    }
    hipLaunchKernelGGL(nufft_reduce_kernel, dim3((unsigned)ceil_div(nf, 256), (unsigned)(2 * nq)), dim3(256), 0, s, partial, ngroups, nchunks, nf, gpw, grid);
    if (nf <= 4096) {
        const size_t lds_modes = sizeof(double) * 3 * (size_t)nf;
        LPVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&nufft_modes_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_modes));
        hipLaunchKernelGGL(nufft_modes_kernel<false>, dim3((unsigned)(2 * nq), (unsigned)ceil_div(nslots, 256)), dim3(256), lds_modes, s, grid, nf, nq, mode0,
                           nslots, scale, tab, (int64_t)nslots, (int64_t)0, (int64_t)0);
    } else {
        const size_t lds_modes = sizeof(double) * 2 * (size_t)nf;
        LPVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&nufft_modes_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_modes));
        hipLaunchKernelGGL(nufft_modes_kernel<true>, dim3((unsigned)(2 * nq), (unsigned)ceil_div(nslots, 256)), dim3(256), lds_modes, s, grid, nf, nq, mode0,
                           nslots, scale, tab, (int64_t)nslots, (int64_t)0, (int64_t)0);
    }
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}


// ---- batched windows
bool nufft_windows_applicable(int64_t n, int64_t nslots) {
    return n >= 2048 && nslots >= 64 && nufft_grid_size(nslots) <= 4096;   // (four grids of nf 64-bit cells in LDS)
}
// 1 / phihat(mode0 + j), j < nslots: the per-mode scale shared by every window (their quanta are folded into the grids)
std::vector<double> nufft_window_scale(int nf, int mode0, int nslots) {
    const std::vector<long double> phat = nufft_phihat(nf, mode0 + nslots);
    std::vector<double> r((size_t)nslots);
    for (int j = 0; j < nslots; ++j) r[(size_t)j] = (double)(1.0L / phat[(size_t)(mode0 + j)]);
    return r;
}
// Spread window q = samples offs[q] .. offs[q] + n of (x, yA, yB) with the window weights W[0..n) (nullptr = 1) onto fine grids:
// gridA / gridB + q * grid_bstride hold [2][nf] doubles (plain, x-weighted).  yA == nullptr: unit signal (the Gram's weights).
int32_t launch_nufft_window_spread(const double *x, const double *W, const double *yA, const double *yB, bool hasB, const int64_t *offs_dev, int nwin,
                                   int64_t n, double D_hi, double D_lo, int nf, double *gridA, double *gridB, int64_t grid_bstride, hipStream_t s) {
    const size_t lds = sizeof(long long) * 4 * (size_t)nf;
    LPVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&nufft_window_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(nufft_window_kernel, dim3((unsigned)nwin), dim3(256), lds, s, x, W, yA, yB, hasB ? 1 : 0, offs_dev, n, D_hi, D_lo, nf, gridA, gridB,
                       grid_bstride);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}
// tab + q * tab_bstride: [nslots][4] = {Re, Im, x-weighted Re, x-weighted Im} of modes mode0 .. of window q's grids
int32_t launch_nufft_window_modes(const double *grids, int64_t grid_bstride, int nwin, int nf, int mode0, int nslots, const double *scale_dev, double *tab,
                                  int64_t tab_bstride, hipStream_t s) {
    const size_t lds_modes = sizeof(double) * 3 * (size_t)nf;
    LPVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&nufft_modes_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_modes));
    hipLaunchKernelGGL(nufft_modes_kernel<false>, dim3(2, (unsigned)ceil_div(nslots, 256), (unsigned)nwin), dim3(256), lds_modes, s, grids, nf, 1, mode0, nslots,
                       scale_dev, tab, (int64_t)0, grid_bstride, tab_bstride);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

}  // namespace lpvs
