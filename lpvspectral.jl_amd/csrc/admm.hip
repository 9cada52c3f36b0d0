// admm.hip -- the scaled-form ADMM loop of src/lasso.jl:136-171 on the resident Gram.
//
//   x <- prox_{mu f}(z - u)   = M (b + (z-u)/mu),  M = (G + I/mu)^-1     (src/lasso.jl:150-151)
//   z <- prox_{mu g}(x + u)                                               (src/lasso.jl:152-153)
//   u <- u + (x - z);  stop when ||x - z||_2 < tol                        (src/lasso.jl:154-157,164)
//
// One iteration = one symmetric mat-vec (HBM / Infinity-Cache bound: np^2 * 8 B streamed with
// 16-B loads, 4 rows per wave, 64-lane shuffle reductions) + one O(n) kernel that fuses the prox,
// the dual update, the ||x-z|| reduction, the next right-hand side and the device-side convergence
// flag.  Kernels of iterations after convergence see the flag and exit, so a chunk of iterations
// can be enqueued without a host round trip and still stop at exactly the reference's iteration.
#include "lpvs_internal.h"

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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// x = M rhs.  Wave handles RPW rows; lane covers columns 2*lane + 128*t.
template <int RPW>
__global__ void __launch_bounds__(256)
symv_kernel(const double *__restrict__ M, int64_t np, const double *__restrict__ rhs_all, double *__restrict__ x_all,
            const AdmmStatus *status) {
    const int sg = blockIdx.y;                                   // signal (right-hand side) of a shared-regressor batch
    if (status != nullptr && status[sg].converged) return;
    const double *rhs = rhs_all + (int64_t)sg * np;
    double *x = x_all + (int64_t)sg * np;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RPW;
    if (row0 >= np) return;
    const double2 *r2 = reinterpret_cast<const double2 *>(rhs);
    const double2 *m2[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) m2[r] = reinterpret_cast<const double2 *>(M + (row0 + r) * np);
    double acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = 0.0;
    const int64_t nv = np / 2;  // double2 elements per row (np % 128 == 0)
#pragma unroll 4
    for (int64_t j = lane; j < nv; j += 64) {
        const double2 v = r2[j];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const double2 m = m2[r][j];
            acc[r] = fma(m.x, v.x, acc[r]);
            acc[r] = fma(m.y, v.y, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const double s = wave_sum(acc[r]);
        if (lane == 0) x[row0 + r] = s;
    }
}

__global__ void __launch_bounds__(256)
admm_init_kernel(AdmmParams p) {
    const int sg = blockIdx.y;
    const int64_t o = (int64_t)sg * p.np;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.np; i += (int64_t)gridDim.x * 256) {
        const bool ok = i < p.n;
        const double xi = ok ? p.x[o + i] : 0.0;
        p.x[o + i] = xi;
        p.z[o + i] = xi;                                    // z = copy(x)      src/lasso.jl:146
        p.u[o + i] = 0.0;                                   // u = zeros        src/lasso.jl:147
        p.rhs[o + i] = ok ? (p.xb ? (xi - 0.0) / p.mu : p.b[o + i] + (xi - 0.0) / p.mu) : 0.0;  // b + (z-u)/mu (offset form: (z-u)/mu)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.status[sg].iters = 0; p.status[sg].converged = 0; p.status[sg].nxz = 0.0; p.status[sg].pad = 0;
    }
}

// re-entry from saved iterates (x, z, u already copied in): next right-hand side and iteration count
__global__ void __launch_bounds__(256)
admm_restate_kernel(AdmmParams p, long long iters) {
    const int sg = blockIdx.y;
    const int64_t o = (int64_t)sg * p.np;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.np; i += (int64_t)gridDim.x * 256) {
        const bool ok = i < p.n;
        if (!ok) { p.x[o + i] = 0.0; p.z[o + i] = 0.0; p.u[o + i] = 0.0; }
        const double v = (p.z[o + i] - p.u[o + i]) / p.mu;
        p.rhs[o + i] = ok ? (p.xb ? v : p.b[o + i] + v) : 0.0;   // b + (z-u)/mu, as the update kernels write it
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.status[sg].iters = iters; p.status[sg].converged = 0; p.status[sg].nxz = 0.0; p.status[sg].pad = 0;
    }
}

// ---- prox_g + dual update + residual norm + next rhs: ONE workgroup of 1024 threads ----------
__device__ double block_sum_1024(double v, double *sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < 16; ++w) t += sh[w];  // fixed order -> reproducible
    return t;
}

__device__ unsigned long long abs_key(double v) { return (unsigned long long)__double_as_longlong(fabs(v)); }

// Threshold key of the r-th largest |v| (radix select, 8 bits per pass, most significant first) and the number of
// equal-key elements to keep (lowest indices first).  All 1024 threads participate.  Histogram updates are
// aggregated per wave (the leading bytes of |v| fall into a handful of bins, which would serialise plain LDS
// atomics), and the bin holding the r-th largest is found by a parallel suffix scan of the 256 counts.
__device__ void topr_threshold(const double *v, int64_t n, int64_t r, unsigned long long *thr,
                               long long *keep_equal, long long *equal_count, unsigned int *hist /*[256]*/,
                               long long *shll /*[3]*/) {
    unsigned long long prefix = 0, mask = 0;
    long long remaining = r;  // how many of the candidates (matching prefix) we still need
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 256; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int pass = 7; pass >= 0; --pass) {
        const int shift = pass * 8;
        for (int64_t i0 = 0; i0 < n; i0 += 1024) {
            const int64_t i = i0 + threadIdx.x;
            const unsigned long long k = i < n ? abs_key(v[i]) : 0ull;
            bool todo = i < n && (k & mask) == prefix;
            const unsigned int bin = (unsigned int)((k >> shift) & 255);
            unsigned long long pending = __ballot(todo);
            while (pending) {                                   // one LDS atomic per distinct bin of the wave
                const int leader = __ffsll((long long)pending) - 1;
                const unsigned int lb = __shfl(bin, leader, 64);
                const unsigned long long same = __ballot(todo && bin == lb);
                if (lane == leader) atomicAdd(&hist[lb], (unsigned int)__popcll(same));
                if (todo && bin == lb) todo = false;
                pending &= ~same;
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {   // wave 0: suffix sums over the 256 bins, 4 bins per lane, no workgroup barrier inside
            unsigned int c[4], tot = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { c[q] = hist[4 * lane + q]; tot += c[q]; hist[4 * lane + q] = 0; }   // cleared for the next pass
            unsigned int sfx = tot;                                                                          // inclusive suffix over lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int t = __shfl_down(sfx, o, 64);
                if (lane + o < 64) sfx += t;
            }
            unsigned int above = sfx - tot;                     // counts in bins of higher lanes
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const unsigned int S = above + c[q];            // S[b] for b = 4*lane+q, above = S[b+1]
                if ((long long)S >= remaining && (long long)above < remaining) { shll[0] = 4 * lane + q; shll[1] = remaining - above; shll[2] = c[q]; }
                above = S;
            }
        }
        __syncthreads();
        prefix |= ((unsigned long long)shll[0]) << shift;
        mask |= 255ull << shift;
        remaining = shll[1];
        const long long in_bin = shll[2];   // (shll is rewritten only after the next pass's barrier)
        if (in_bin == 1 && pass > 0) {
            // a single candidate is left (the usual case after two or three bytes): its key is the threshold
            __syncthreads();                // everyone has read shll before it is reused
            for (int64_t i = threadIdx.x; i < n; i += 1024) {
                const unsigned long long k = abs_key(v[i]);
                if ((k & mask) == prefix) shll[0] = (long long)k;     // exactly one writer
            }
            __syncthreads();
            prefix = (unsigned long long)shll[0];
            __syncthreads();
            break;
        }
    }
    *thr = prefix;
    *keep_equal = remaining;
    *equal_count = shll[2];   // elements whose key equals the threshold (count of the last pass's bin)
}

// Two-level top-r for n <= 32768: a radix select on monotone 32-bit float keys held in LDS decides everything except
// the few elements whose float key equals the threshold's; those candidates are ranked exactly (64-bit key, lowest
// index first on ties) and the winners are marked in an LDS bitmap.  Returns false (nothing written) when there
// are more than 1024 candidates -- the caller then falls back to the 64-bit select.
//   keep(i) = key32[i] > thr32  ||  bit i of `mark`
__device__ bool topr_float_keys_radix(const double *__restrict__ X, const double *__restrict__ U, int64_t n, long long r,
                                      unsigned int *key32 /*[n] LDS*/, unsigned int *mark /*[1024] LDS words*/,
                                      unsigned int *hist /*[256]*/, long long *shll /*[3]*/, int *scan /*[1024] LDS*/,
                                      unsigned int *thr_out) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = threadIdx.x; i < n; i += 1024) key32[i] = __float_as_uint((float)fabs(X[i] + U[i]));   // monotone in |v|
    for (int i = threadIdx.x; i < 256; i += 1024) hist[i] = 0;
    for (int i = threadIdx.x; i < 1024; i += 1024) mark[i] = 0;
    __syncthreads();
    unsigned int prefix = 0, mask = 0;
    long long remaining = r;
    for (int pass = 3; pass >= 0; --pass) {
        const int shift = pass * 8;
        for (int64_t i0 = 0; i0 < n; i0 += 1024) {
            const int64_t i = i0 + threadIdx.x;
            const unsigned int k = i < n ? key32[i] : 0u;
            bool todo = i < n && (k & mask) == prefix;
            const unsigned int bin = (k >> shift) & 255;
            unsigned long long pending = __ballot(todo);
            while (pending) {
                const int leader = __ffsll((long long)pending) - 1;
                const unsigned int lb = __shfl(bin, leader, 64);
                const unsigned long long same = __ballot(todo && bin == lb);
                if (lane == leader) atomicAdd(&hist[lb], (unsigned int)__popcll(same));
                if (todo && bin == lb) todo = false;
                pending &= ~same;
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            unsigned int c[4], tot = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { c[q] = hist[4 * lane + q]; tot += c[q]; hist[4 * lane + q] = 0; }
            unsigned int sfx = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int t = __shfl_down(sfx, o, 64);
                if (lane + o < 64) sfx += t;
            }
            unsigned int above = sfx - tot;
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const unsigned int S = above + c[q];
                if ((long long)S >= remaining && (long long)above < remaining) { shll[0] = 4 * lane + q; shll[1] = remaining - above; shll[2] = c[q]; }
                above = S;
            }
        }
        __syncthreads();
        prefix |= ((unsigned int)shll[0]) << shift;
        mask |= 255u << shift;
        remaining = shll[1];
        __syncthreads();
    }
    const long long need = remaining, ncand = shll[2];   // keep `need` of the `ncand` elements with key32 == prefix
    *thr_out = prefix;
    if (ncand > 1023) return false;                       // (uniform) candidate list lives in scan[1..1023]
    __syncthreads();
    if (threadIdx.x == 0) scan[0] = 0;
    __syncthreads();
    int *cidx = scan + 1;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
        if (key32[i] == prefix) cidx[atomicAdd(&scan[0], 1)] = (int)i;
    __syncthreads();
    // exact rank among the candidates: 64-bit key descending, index ascending
    if ((long long)threadIdx.x < ncand) {
        const int me = cidx[threadIdx.x];
        const unsigned long long km = abs_key(X[me] + U[me]);
        int rank = 0;
        for (int c = 0; c < (int)ncand; ++c) {
            const int o = cidx[c];
            const unsigned long long ko = abs_key(X[o] + U[o]);
            rank += (ko > km) || (ko == km && o < me);
        }
        if (rank < need) atomicOr(&mark[me >> 5], 1u << (me & 31));
    }
    __syncthreads();
    return true;
}

// The same selection in ONE histogram pass (what cfg5 runs every iteration: n = 32768, r = 32): the float keys go to LDS and, on the
// way, into a histogram of their top 11 bits (exponent + 3 mantissa bits: plain LDS atomics -- the keys of a wave spread over tens of
// bins; the four byte-wise passes above cost a ballot loop per distinct bin each); a suffix scan of the 2048 counts finds the bin of the
// r-th largest; the (few) elements of that bin are ranked exactly on their 64-bit keys held in LDS (lowest index first on ties).
//   keep(i) = key32[i] > *thr_out  ||  bit i of `mark`          (*thr_out = the bin's largest float key)
// More than 1023 elements in the threshold bin: the byte-wise select above decides (and may hand on to the 64-bit one).
__device__ bool topr_float_keys(const double *__restrict__ X, const double *__restrict__ U, int64_t n, long long r,
                                unsigned int *key32 /*[n] LDS*/, unsigned int *mark /*[1024] LDS words*/,
                                unsigned int *hist /*[256]*/, unsigned int *hist11 /*[2048]*/, unsigned long long *ckey /*[1024]*/,
                                long long *shll /*[3]*/, int *scan /*[1024] LDS*/, unsigned int *thr_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2048; i += 1024) hist11[i] = 0;
    mark[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < n; c0 += 8192) {          // eight elements per thread in flight
        double xv[8], uv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t i = c0 + threadIdx.x + 1024 * k;
            xv[k] = i < n ? X[i] : 0.0; uv[k] = i < n ? U[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t i = c0 + threadIdx.x + 1024 * k;
            if (i < n) {
                const unsigned int key = __float_as_uint((float)fabs(xv[k] + uv[k]));   // monotone in |v|
                key32[i] = key;
                atomicAdd(&hist11[key >> 20], 1u);
            }
        }
    }
    __syncthreads();
    {   // suffix sums over the 2048 bins: thread t owns bins 2t, 2t + 1
        const unsigned int c0 = hist11[2 * threadIdx.x], c1 = hist11[2 * threadIdx.x + 1], tot = c0 + c1;
        unsigned int sfx = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t = __shfl_down(sfx, o, 64);
            if (lane + o < 64) sfx += t;
        }
        if (lane == 0) scan[wave] = (int)sfx;           // the wave's total
        __syncthreads();
        unsigned int above = sfx - tot;                  // counts in higher bins of this wave ...
        for (int w2 = wave + 1; w2 < 16; ++w2) above += (unsigned int)scan[w2];   // ... and of the higher waves
        const unsigned int S1 = above + c1, S0 = S1 + c0;
        if ((long long)S1 >= r && (long long)above < r) { shll[0] = 2 * threadIdx.x + 1; shll[1] = r - above; shll[2] = c1; }
        if ((long long)S0 >= r && (long long)S1 < r) { shll[0] = 2 * threadIdx.x; shll[1] = r - S1; shll[2] = c0; }
        __syncthreads();
    }
    const unsigned int bin = (unsigned int)shll[0];
    const long long need = shll[1], ncand = shll[2];      // keep `need` of the `ncand` elements of the bin
    if (ncand > 1023) {                                   // (uniform)
        __syncthreads();
        return topr_float_keys_radix(X, U, n, r, key32, mark, hist, shll, scan, thr_out);
    }
    *thr_out = (bin << 20) | 0xFFFFFu;
    if (threadIdx.x == 0) scan[0] = 0;
    __syncthreads();
    int *cidx = scan + 1;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
        if ((key32[i] >> 20) == bin) cidx[atomicAdd(&scan[0], 1)] = (int)i;
    __syncthreads();
    if ((long long)threadIdx.x < ncand) { const int me = cidx[threadIdx.x]; ckey[threadIdx.x] = abs_key(X[me] + U[me]); }
    __syncthreads();
    if ((long long)threadIdx.x < ncand) {                 // exact rank among the candidates: 64-bit key descending, index ascending
        const int me = cidx[threadIdx.x];
        const unsigned long long km = ckey[threadIdx.x];
        int rank = 0;
        for (int c = 0; c < (int)ncand; ++c) {
            const int o = cidx[c];
            const unsigned long long ko = ckey[c];
            rank += (ko > km) || (ko == km && o < me);
        }
        if (rank < need) atomicOr(&mark[me >> 5], 1u << (me & 31));
    }
    __syncthreads();
    return true;
}

// Elements are processed in coalesced passes of 1024 threads x EPT elements: all loads of a pass are
// issued before any store (no aliasing-serialised round trips); group norms go through LDS.
constexpr int EPT = 8;                 // elements per thread per pass
constexpr int PASS = 1024 * EPT;       // 8192 elements per pass

__global__ void __launch_bounds__(1024)
admm_prox_kernel(AdmmParams p) {
    __shared__ double sh[16];
    __shared__ unsigned int hist[256];
    __shared__ long long shll[3];
    __shared__ int scan[1024];
    __shared__ double lds2[2 * PASS];  // 128 KiB: group prox uses it as sq / gscale, IndBallL0 as float keys or v cache
    __shared__ unsigned int mark[1024];
    __shared__ unsigned int hist11[2048];
    __shared__ unsigned long long ckey[1024];
    double *sq = lds2;                 // v^2 of the current pass (group prox)
    double *gscale = lds2 + PASS;      // per-group scale of the current pass
    const int sg = blockIdx.x;         // one workgroup per signal
    AdmmStatus *status = p.status + sg;
    if (status->converged) return;
    const int64_t n = p.n, so = (int64_t)sg * p.np;
    const double mu = p.mu;
    const double *__restrict__ X = p.x + so;
    const double *__restrict__ B = p.b + so;
    double *__restrict__ Z = p.z + so;
    double *__restrict__ U = p.u + so;
    double *__restrict__ R = p.rhs + so;
    double ss = 0;  // sum (x-z)^2 over this thread's elements

    const bool offset_form = p.xb != nullptr;   // x = xb + M (z-u)/mu: the right-hand side carries no b
    auto finish = [&](int64_t i, double xi, double ui, double bi, double zi) {
        const double d = xi - zi;              // tmp = x - z            src/lasso.jl:154
        const double un = ui + d;              // u += tmp               src/lasso.jl:155
        Z[i] = zi; U[i] = un;
        R[i] = offset_form ? (zi - un) / mu : bi + (zi - un) / mu;   // next x-update: b + (z-u)/mu
        ss = fma(d, d, ss);
    };

    const int kind = p.prox_kind;
    double thr_l1 = mu * p.prox_param, thr_l0 = sqrt(2.0 * mu * p.prox_param);
    unsigned long long ball_thr = 0; long long ball_keep_eq = 0, ball_eq_count = 0, ball_r = (long long)p.prox_param;
    double *vbuf = n <= PASS ? sq : p.scratch + 2 * so;   // v = x + u is scanned 8 times: keep it in LDS when it fits
    unsigned int *key32 = reinterpret_cast<unsigned int *>(lds2);
    unsigned int thr32 = 0;
    bool ball_fast = false;                                // two-level selection with float keys in LDS succeeded
    if (kind == LPVS_PROX_BALL_L0 && ball_r > 0 && ball_r < n) {
        if (n <= 4 * PASS) ball_fast = topr_float_keys(X, U, n, ball_r, key32, mark, hist, hist11, ckey, shll, scan, &thr32);
        if (!ball_fast) {
            __syncthreads();
            for (int64_t i = threadIdx.x; i < n; i += 1024) vbuf[i] = X[i] + U[i];
            __syncthreads();
            topr_threshold(vbuf, n, ball_r, &ball_thr, &ball_keep_eq, &ball_eq_count, hist, shll);
        }
    }
    const bool ball_ties = ball_keep_eq < ball_eq_count;   // only then do equal keys have to be ranked by index
    long long eq_seen = 0;  // equal-key elements at lower indices (uniform across threads)

    if (kind == LPVS_PROX_GROUP_L2) {
        const int64_t gl = p.group_len, ng = n / gl;
        const double lm = p.prox_param * mu;
        const int64_t gpp = PASS / gl;                         // whole groups per pass
        for (int64_t g0 = 0; g0 < ng; g0 += gpp) {
            const int64_t gcount = (ng - g0 < gpp) ? ng - g0 : gpp;
            const int64_t c0 = g0 * gl, cnt = gcount * gl;
            double xv[EPT], uv[EPT], bv[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int64_t e = threadIdx.x + 1024 * k;
                const bool ok = e < cnt;
                xv[k] = ok ? X[c0 + e] : 0.0; uv[k] = ok ? U[c0 + e] : 0.0; bv[k] = ok && !offset_form ? B[c0 + e] : 0.0;   // (offset form: the right-hand side carries no b)
            }
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int64_t e = threadIdx.x + 1024 * k;
                const double v = xv[k] + uv[k];
                if (e < cnt) sq[e] = v * v;
            }
            __syncthreads();
            for (int64_t g = threadIdx.x; g < gcount; g += 1024) {
                double s2 = 0;
                for (int64_t q = 0; q < gl; ++q) s2 += sq[g * gl + q];   // sequential, as norm() on a short slice
                double scale = 1.0 - lm / sqrt(s2);                    // s2 == 0 -> -inf -> 0
                if (!(scale > 0)) scale = 0.0;
                gscale[g] = scale;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int64_t e = threadIdx.x + 1024 * k;
                if (e < cnt) finish(c0 + e, xv[k], uv[k], bv[k], gscale[e / gl] * (xv[k] + uv[k]));
            }
            __syncthreads();
        }
        for (int64_t i = ng * gl + threadIdx.x; i < n; i += 1024)  // entries outside every slice: prox! leaves z
            finish(i, X[i], U[i], B[i], Z[i]);
    } else {
        for (int64_t c0 = 0; c0 < n; c0 += PASS) {
            double xv[EPT], uv[EPT], bv[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int64_t i = c0 + threadIdx.x + 1024 * k;
                const bool ok = i < n;
                xv[k] = ok ? X[i] : 0.0; uv[k] = ok ? U[i] : 0.0; bv[k] = ok && !offset_form ? B[i] : 0.0;   // (offset form: the right-hand side carries no b)
            }
            if (kind == LPVS_PROX_L1 || kind == LPVS_PROX_L0 || ball_r <= 0 || ball_r >= n) {
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    const int64_t i = c0 + threadIdx.x + 1024 * k;
                    if (i >= n) continue;
                    const double v = xv[k] + uv[k];
                    double zi;
                    if (kind == LPVS_PROX_L1) zi = v + (v <= -thr_l1 ? thr_l1 : (v >= thr_l1 ? -thr_l1 : -v));
                    else if (kind == LPVS_PROX_L0) zi = fabs(v) > thr_l0 ? v : 0.0;
                    else zi = ball_r >= n ? v : 0.0;
                    finish(i, xv[k], uv[k], bv[k], zi);
                }
            } else if (ball_fast) {   // IndBallL0, two-level selection: float key above the threshold, or a marked candidate
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    const int64_t i = c0 + threadIdx.x + 1024 * k;
                    if (i >= n) continue;
                    const double v = xv[k] + uv[k];
                    const bool keep = key32[i] > thr32 || ((mark[i >> 5] >> (i & 31)) & 1u);
                    finish(i, xv[k], uv[k], bv[k], keep ? v : 0.0);
                }
            } else if (!ball_ties) {  // IndBallL0, every key equal to the threshold is kept: no ranking needed
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    const int64_t i = c0 + threadIdx.x + 1024 * k;
                    if (i >= n) continue;
                    const double v = xv[k] + uv[k];
                    finish(i, xv[k], uv[k], bv[k], abs_key(v) >= ball_thr ? v : 0.0);
                }
            } else {  // IndBallL0: keys > threshold, plus the first keep_eq equal keys in index order
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    const int64_t i = c0 + threadIdx.x + 1024 * k;
                    const bool in = i < n;
                    const double v = xv[k] + uv[k];
                    const unsigned long long key = in ? abs_key(v) : 0ull;
                    const int iseq = in && key == ball_thr;
                    scan[threadIdx.x] = iseq;
                    __syncthreads();
                    for (int o = 1; o < 1024; o <<= 1) {   // inclusive Hillis-Steele scan
                        const int t = (int)threadIdx.x >= o ? scan[threadIdx.x - o] : 0;
                        __syncthreads();
                        scan[threadIdx.x] += t;
                        __syncthreads();
                    }
                    const long long rank_eq = eq_seen + scan[threadIdx.x];
                    const long long tot = scan[1023];
                    if (in) finish(i, xv[k], uv[k], bv[k], (key > ball_thr || (iseq && rank_eq <= ball_keep_eq)) ? v : 0.0);
                    eq_seen += tot;
                    __syncthreads();
                }
            }
        }
    }

    const double tot = block_sum_1024(ss, sh);
    if (threadIdx.x == 0) {
        const double nxz = sqrt(tot);          // norm(tmp)               src/lasso.jl:157
        status->iters += 1;
        status->nxz = nxz;
        if (nxz < p.tol) status->converged = 1;  //                      src/lasso.jl:164
    }
}

// ---- symmetric mat-vec on the lower-triangle 128x128 tiles of M (np^2*4 bytes instead of np^2*8) -------
// M is re-stored tile-packed: Mp[t][128][128], t = I(I+1)/2 + J, I >= J, so a workgroup streams one
// contiguous 128 KiB tile.  Per tile:  part1[t][i] = sum_j T[i][j] r[J*128+j]      (rows)
//                                      part2[t][j] = sum_i T[i][j] r[I*128+i]      (transpose, I != J)
// Each wave holds 32 rows (16 B per lane per row, all 32 loads in flight); the 32 row sums are reduced
// across the 64 lanes by a halving butterfly (32 shuffles per wave instead of 32*6).
constexpr int TS = 128;

__device__ __forceinline__ void tile_index(int t, int &I, int &J) {
    I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    while (I * (I + 1) / 2 > t) --I;
    J = t - I * (I + 1) / 2;
}

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

template <typename T> struct Pair;
template <> struct Pair<double> { typedef double2 type; };
template <> struct Pair<float> { typedef float2 type; };

// T = storage type of the packed matrix (double, or float for the _f32 problems); products and sums are double
template <typename T>
__global__ void __launch_bounds__(256)
symv_tile_kernel(const T *__restrict__ Mp, const double *__restrict__ rhs_all, int64_t np, int ns, int ntiles,
                 double *__restrict__ part1_all, double *__restrict__ part2_all, const AdmmStatus *status) {
    typedef typename Pair<T>::type T2;
    if (status != nullptr) {   // nothing to do once every signal has converged
        bool all = true;
        for (int q = 0; q < ns; ++q) all = all && status[q].converged;
        if (all) return;
    }
    __shared__ double sI[TS], sJ[TS], sT[4][TS];
    const int t = blockIdx.x;
    int I, J;
    tile_index(t, I, J);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T2 *base = reinterpret_cast<const T2 *>(Mp + (int64_t)t * TS * TS + wave * 32 * TS) + lane;
    T2 m[32];   // the tile is read once and applied to every right-hand side
#pragma unroll
    for (int r = 0; r < 32; ++r) m[r] = base[r * (TS / 2)];
    for (int sg = 0; sg < ns; ++sg) {
        if (status != nullptr && status[sg].converged) continue;
        const double *rhs = rhs_all + (int64_t)sg * np;
        double *part1 = part1_all + (int64_t)sg * ntiles * TS, *part2 = part2_all + (int64_t)sg * ntiles * TS;
        __syncthreads();   // previous signal's readers are done with sI / sJ / sT
        if (threadIdx.x < TS) sI[threadIdx.x] = rhs[(int64_t)I * TS + threadIdx.x];
        else sJ[threadIdx.x - TS] = rhs[(int64_t)J * TS + threadIdx.x - TS];
        __syncthreads();
        const double rj0 = sJ[2 * lane], rj1 = sJ[2 * lane + 1];
        double t0 = 0, t1 = 0, v[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const double ri = sI[wave * 32 + r];
            const double mx = (double)m[r].x, my = (double)m[r].y;
            t0 = fma(mx, ri, t0);
            t1 = fma(my, ri, t1);
            v[r] = fma(mx, rj0, my * rj1);
        }
        // halving butterfly: after the step with mask w a lane keeps the rows whose bit matches its own
#pragma unroll
        for (int w = 32, cnt = 16; w >= 2; w >>= 1, cnt >>= 1) {
            const bool hi = (lane & w) != 0;
#pragma unroll
            for (int k = 0; k < cnt; ++k) {
                const double send = hi ? v[k] : v[k + cnt];
                const double keep = hi ? v[k + cnt] : v[k];
                v[k] = keep + __shfl_xor(send, w, 64);
            }
        }
        v[0] += __shfl_xor(v[0], 1, 64);
        if ((lane & 1) == 0) {
            const int row = ((lane & 32) ? 16 : 0) + ((lane & 16) ? 8 : 0) + ((lane & 8) ? 4 : 0) + ((lane & 4) ? 2 : 0) + ((lane & 2) ? 1 : 0);
            part1[(int64_t)t * TS + wave * 32 + row] = v[0];
        }
        if (I != J) {
            sT[wave][2 * lane] = t0; sT[wave][2 * lane + 1] = t1;
            __syncthreads();
            if (threadIdx.x < TS)
                part2[(int64_t)t * TS + threadIdx.x] = ((sT[0][threadIdx.x] + sT[1][threadIdx.x]) + sT[2][threadIdx.x]) + sT[3][threadIdx.x];
        }
    }
}


// ---- 6-byte ("split") storage of the packed inverse ----------------------------------------------------------------
// The mat-vec is HBM-bound on the bytes of M, and M = (G + I/mu)^-1 comes out of the block sweep with a normwise error of
// ~1e-12 (|M H - I|_max = 2e-13 at n = 8192, tools/factor_check.py): the trailing 13 bits of its doubles carry no
// information.  An element is stored as the 48 leading bits of its double, rounded to nearest at bit 13, in two parts:
//   head = those bits down to bit 29 as a FLOAT (sign, exponent, 23 mantissa bits: exactly the double with its low 29 bits
//          cleared, which is always a float for |M| in [2^-120, 2^127]; smaller magnitudes are flushed to 0),
//   tail = the next 16 mantissa bits (bits 28..13 of the double) as an unsigned short.
// 40 significant bits, relative error <= 2^-40 = 9.1e-13 per element -- below the accuracy M has anyway -- in 6 bytes
// instead of 8: 25 % fewer bytes per iteration.  Decoding is exact and costs three VALU instructions per element:
// v_cvt_f64_f32 (whose low dword has only its top 3 bits set), extract the tail, v_lshl_or_b32 into that low dword.
//   tile layout (98304 B): head[128][128] float, then tail[128][128] uint16 with the columns of a row permuted so that the
//   8 tails a lane needs are one 16-byte load: position 8c + 4h + k holds column 64h + 4c + k  (c < 16, h < 2, k < 4).
// Lane (g = lane >> 4, c = lane & 15) of wave w owns rows 32w + 4rg + g (rg < 8) and columns {4c+k, 64+4c+k}: per row group
// two float4 and one uint4 load (every instruction covers whole 128-byte lines): 24 loads = 384 bytes in flight per lane,
// two workgroups per CU (three would need <= 168 registers and spill: measured 38.3 us against 30.4 us per launch at
// np = 8192, i.e. 6.7 TB/s of 6-byte elements; the 8-byte kernel: 41.6 us, 6.56 TB/s).  Single right-hand side only
// (multi-signal handles keep doubles for the matrix-core tile product).
//
// ACCURACY.  A reduced-precision copy of M must not multiply the large constant vector b: the rounding of the small
// eigenvalues of M (the large ones of G) would be amplified by cond(G + I/mu) -- measured 5.3e-9 rel-L2 in z after 2000
// iterations at the cfg3 size (9e-9 at n = 32768), above the 1e-9 parity bound.  So the x-update runs in OFFSET FORM
// (AdmmParams::xb): xb = M b once from the full-precision inverse, and per iteration x = xb + M~ (z-u)/mu.  Near the
// solution (z-u)/mu = x/mu - subgradient, so |dM (z-u)/mu| <= 2^-40 |M| |x| / mu <= 2^-40 |x|: no amplification.
// Measured with the offset form: 1.2e-10 rel-L2 in z against the 8-byte storage at cfg3 (2000 iterations), 6e-11 against
// an exact-solve CPU run of the same ADMM at n = 2176 (300 iterations; the 8-byte storage: 8e-13), identical supports and stopping iterations.
constexpr size_t kSplitTileBytes = (size_t)TS * TS * 6;

__device__ __forceinline__ double split_decode(float head, unsigned int tail16) {
    const double d = (double)head;
    return __hiloint2double(__double2hiint(d), (int)((tail16 << 13) | (unsigned int)__double2loint(d)));
}
// the same with the shift-or as ONE instruction (the compiler otherwise masks after shifting and ors separately): on gfx950
// every vector instruction of a wave that shares a SIMD with fp64 MFMAs costs the matrix pipe ~7 cycles
__device__ __forceinline__ double split_decode_lo(float head, unsigned int pair) {   // tail = low half of `pair`
    const double d = (double)head;
    unsigned int lo = (unsigned int)__double2loint(d), t = pair & 0xffffu;
    asm("v_lshl_or_b32 %0, %1, 13, %0" : "+v"(lo) : "v"(t));
    return __hiloint2double(__double2hiint(d), (int)lo);
}
__device__ __forceinline__ double split_decode_hi(float head, unsigned int pair) {   // tail = high half of `pair`
    const double d = (double)head;
    unsigned int lo = (unsigned int)__double2loint(d), t = pair >> 16;
    asm("v_lshl_or_b32 %0, %1, 13, %0" : "+v"(lo) : "v"(t));
    return __hiloint2double(__double2hiint(d), (int)lo);
}
// an SSA value the optimiser cannot look through: keeps `up ? a[k] : a[k+cnt]` from becoming a dynamically indexed array
// access (which the backend then lowers to an 8-way select chain per value)
__device__ __forceinline__ double opaque(double v) { asm volatile("" : "+v"(v)); return v; }

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

// ---- 36-bit fixed-point tiles (mixed storage of the single-signal packed inverse) -------------------------------------
// M = (G + I/mu)^-1 of the LPV / Fourier problems is strongly diagonally dominant: at cfg3 the largest entry of an
// off-diagonal tile is 2^-9.4 .. 2^-10.7 of the diagonal's.  The error of the product M~ v is then dominated by the rounding
// of the LARGE entries (diagonal tiles, 2^-41 relative); the small entries' 40 significant bits are ~10 bits more absolute
// precision than is ever felt.  A tile whose rows are all small is therefore stored as 36-bit fixed point against a per-row
// power-of-two step:   element = q * step[row],  q = 16 * hi32 + nibble  (two's complement, |q| < 2^35),
//   stored biased, q + 2^35 = 16 * hi + nibble with hi an unsigned dword;
//   tile slot (same 98304-byte stride): hi[128][128] uint32 (65536 B), nibbles (8192 B), step[128] float (512 B) = 74240 B.
// Measured on cfg3's inverse (tools/quant_study.py): |dM v| / |x| = 3.9e-13 with these tiles against 3.2e-13 with 40-bit
// elements everywhere (32-bit fixed point: 3.8e-12).  Eligibility is decided per tile when packing: every row's step must be
// <= 2^-44 * max|M| * sqrt(8192 / np) (the fixed-point errors of a row add up over ~np entries); diagonal tiles and tiles
// that fail keep the 6-byte float-head format, so a matrix without this structure loses nothing.  A DIAGONAL tile whose entries
// off the main diagonal pass the same test is stored as fixed point too, with its 128 diagonal entries apart in doubles (1024 B
// after the steps; their fixed-point value is 0) -- the nearly diagonal inverses of the Fourier windows.  tile type: 0 = float
// head + 16-bit tail, 1 = fixed point, 2 = fixed point + double diagonal.  Decoding: v_bfe_u32, v_alignbit_b32, v_lshl_or_b32, one v_add_f64 (exact integer in a double);
// the row step multiplies the row sum once and the row's right-hand-side value once (for the transposed product).
// Layouts follow the lane ownership of the split kernel (lane (g, c) of wave w: rows 32w + 4rg + g, columns 4c+k, 64+4c+k):
//   nibbles: dword (w*64 + lane)*8 + rg holds the row group's 8 nibbles, nibble k at bits 4k (k < 4: column 4c+k, else 64+4c+k-4)
//   steps:   float (w*4 + g)*8 + rg = step of row 32w + 4rg + g
constexpr size_t kFixHeadBytes = (size_t)TS * TS * 4, kFixNibBytes = (size_t)TS * TS / 2;

// The common tail of the single-signal tile products: v[rg] = the lane's partial row sums of its 8 row groups (rows 32w + 4rg + g),
// tc[k] = its partial column sums of its 8 columns; row sums by a halving butterfly over the 16 column lanes, column sums over the
// wave's four row lanes and then over the four waves through LDS.  All 256 threads call it.
__device__ __forceinline__ void tile_reduce_store(double (&v)[8], double (&tc)[8], double (*sT)[TS], bool offdiag,
                                                  double *__restrict__ part1, double *__restrict__ part2,
                                                  const double *__restrict__ diag = nullptr, const double *sI = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 8, cnt = 4; m >= 2; m >>= 1, cnt >>= 1) {
        const bool up = (c & m) != 0;
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            const double lo_ = opaque(v[k]), hi_ = opaque(v[k + cnt]);
            v[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
        }
    }
    v[0] += __shfl_xor(v[0], 1, 64);
    if ((c & 1) == 0) {
        const int rg = ((c & 8) ? 4 : 0) + ((c & 4) ? 2 : 0) + ((c & 2) ? 1 : 0);
        const int row = wave * 32 + 4 * rg + g;
        part1[row] = diag != nullptr ? fma(diag[row], sI[row], v[0]) : v[0];   // (a diagonal tile whose diagonal is kept apart in doubles)
    }
    if (offdiag) {
#pragma unroll
        for (int m = 32, cnt = 4; m >= 16; m >>= 1, cnt >>= 1) {
            const bool up = (lane & m) != 0;
#pragma unroll
            for (int k = 0; k < cnt; ++k) {
                const double lo_ = opaque(tc[k]), hi_ = opaque(tc[k + cnt]);
                tc[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
            }
        }
        const int col = ((lane & 32) ? 64 : 0) + 4 * c + ((lane & 16) ? 2 : 0);
        sT[wave][col] = tc[0]; sT[wave][col + 1] = tc[1];
        __syncthreads();
        if (threadIdx.x < TS)
            part2[threadIdx.x] = ((sT[0][threadIdx.x] + sT[1][threadIdx.x]) + sT[2][threadIdx.x]) + sT[3][threadIdx.x];
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

struct FixRaw { int4 ha[8], hb[8]; uint4 nq[2]; float4 st[2]; };

// q = 16 * hi + nib (biased by 2^35) -> the double q - 2^35, exactly: the bits of 2^52 + q are assembled with two integer
// instructions (v_alignbit_b32 puts the top four bits of q under the exponent, v_lshl_or_b32 forms the low dword), then one
// subtraction.  (Integer conversions would be three double-rate instructions more per element.)
__device__ __forceinline__ double fix_decode(unsigned int hi, unsigned int nib) {
    const unsigned int top = __builtin_amdgcn_alignbit(0x04330000u, hi, 28);   // (0x04330000 << 4) | (hi >> 28) = 0x43300000 | q[35:32]
    unsigned int lo = nib;
    asm("v_lshl_or_b32 %0, %1, 4, %0" : "+v"(lo) : "v"(hi));
    return __hiloint2double((int)top, (int)lo) - (0x1p52 + 0x1p35);
}

// one 16-byte load; NT: non-temporal (streams larger than the 256 MiB Infinity Cache read ~14 % faster that way -- tools/stream_read.hip:
// 600 MB at 7.0 instead of 6.1 TB/s -- while a stream that fits it, the single-problem inverse of cfg3, gains nothing)
template <bool NT, typename V>
__device__ __forceinline__ V load16(const void *p) {
    typedef unsigned int u32x4n __attribute__((ext_vector_type(4)));
    static_assert(sizeof(V) == 16, "16-byte vectors only");
    const u32x4n r = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4n *>(p)) : *reinterpret_cast<const u32x4n *>(p);
    return __builtin_bit_cast(V, r);
}

// fmode (wave-uniform): 0 = the 36-bit element (heads + nibbles); 1 = its 32 leading bits only -- the nibbles are not read (4 B per element
// instead of 4.5) and count as zero; 2 = the NIBBLES only (the heads are the bias, so an element decodes to its nibble x step): the part
// mode 1 leaves out, for the stale nibble product of handles that iterate on 32-bit reads (launch_nibble_refresh)
template <bool NT = false>
__device__ __forceinline__ void fix_load(const unsigned char *tile, int wave, int lane, FixRaw &w, int fmode = 0) {
    const int g = lane >> 4, c = lane & 15;
    const bool fix32 = fmode == 1;
    // nibbles and steps FIRST: loads return in order, and the first row group's products need them -- requested last, they kept
    // every product waiting for the tile's last byte (all of a tile's arithmetic then sat at the end of its load)
    const uint4 *nq = reinterpret_cast<const uint4 *>(tile + kFixHeadBytes) + (wave * 64 + lane) * 2;
    if (fix32) { w.nq[0] = make_uint4(0, 0, 0, 0); w.nq[1] = w.nq[0]; }
    else { w.nq[0] = load16<NT, uint4>(nq); w.nq[1] = load16<NT, uint4>(nq + 1); }
    const float4 *st = reinterpret_cast<const float4 *>(tile + kFixHeadBytes + kFixNibBytes) + (wave * 4 + g) * 2;
    w.st[0] = load16<NT, float4>(st); w.st[1] = load16<NT, float4>(st + 1);
    const int *head = reinterpret_cast<const int *>(tile) + (wave * 32 + g) * TS + 4 * c;
    if (fmode == 2) {
        const int bias = (int)0x80000000u;           // 16 * 2^31 = 2^35: the element decodes to nibble * step
#pragma unroll
        for (int rg = 0; rg < 8; ++rg) { w.ha[rg] = make_int4(bias, bias, bias, bias); w.hb[rg] = w.ha[rg]; }
        return;
    }
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        w.ha[rg] = load16<NT, int4>(head + rg * 4 * TS);
        w.hb[rg] = load16<NT, int4>(head + rg * 4 * TS + 64);
    }
}

// the product of split_tile_product for a fixed-point tile (always off the diagonal)
__device__ __forceinline__ void fix_tile_product(const FixRaw &w, const double *sI, const double *sJ, double (*sT)[TS],
                                                 double *__restrict__ part1, double *__restrict__ part2, const double *__restrict__ diag = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    double rj[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
    double tc[8], v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tc[k] = 0.0;
    const float stv[8] = {w.st[0].x, w.st[0].y, w.st[0].z, w.st[0].w, w.st[1].x, w.st[1].y, w.st[1].z, w.st[1].w};
    const unsigned int nw[8] = {w.nq[0].x, w.nq[0].y, w.nq[0].z, w.nq[0].w, w.nq[1].x, w.nq[1].y, w.nq[1].z, w.nq[1].w};
    // the row's step folded into its right-hand-side value; requested one row group ahead, so that the column products can
    // be issued together with the row products (otherwise the compiler parks the eight decoded elements -- and spills)
    double ri = (double)stv[0] * sI[wave * 32 + g];
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        const double step = (double)stv[rg];
        const double ri_next = rg + 1 < 8 ? (double)stv[rg + 1] * sI[wave * 32 + 4 * (rg + 1) + g] : 0.0;
        const int hh[8] = {w.ha[rg].x, w.ha[rg].y, w.ha[rg].z, w.ha[rg].w, w.hb[rg].x, w.hb[rg].y, w.hb[rg].z, w.hb[rg].w};
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const double m0 = fix_decode((unsigned int)hh[k], (nw[rg] >> (4 * k)) & 15u);             // exact 36-bit integers
            const double m1 = fix_decode((unsigned int)hh[k + 1], (nw[rg] >> (4 * k + 4)) & 15u);
            tc[k] = opaque(fma(m0, ri, tc[k]));          // (pinned: left to itself the compiler parks all 64 decoded elements of the
            tc[k + 1] = opaque(fma(m1, ri, tc[k + 1]));  //  lane and issues the column products after the loop -- and spills)
            a0 = fma(m0, rj[k], a0);
            a1 = fma(m1, rj[k + 1], a1);
        }
        v[rg] = step * (a0 + a1);
        ri = ri_next;
        __builtin_amdgcn_sched_barrier(0);               // (row group by row group, as the bytes arrive)
    }
    tile_reduce_store(v, tc, sT, diag == nullptr, part1, part2, diag, sI);
}

// raw registers of one lane's share of a split tile (8 row groups: two float4 heads, one uint4 of tails)
struct SplitRaw { float4 ha[8], hb[8]; uint4 lq[8]; };

__device__ __forceinline__ void split_load(const unsigned char *tile, int wave, int g, int c, SplitRaw &w) {
    const float *head = reinterpret_cast<const float *>(tile) + (wave * 32 + g) * TS + 4 * c;
    const unsigned short *tail = reinterpret_cast<const unsigned short *>(tile + (size_t)TS * TS * 4) + (wave * 32 + g) * TS + 8 * c;
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        w.ha[rg] = *reinterpret_cast<const float4 *>(head + rg * 4 * TS);
        w.hb[rg] = *reinterpret_cast<const float4 *>(head + rg * 4 * TS + 64);
        w.lq[rg] = *reinterpret_cast<const uint4 *>(tail + rg * 4 * TS);
    }
}

// one right-hand side against the tile held in `w`: sI / sJ hold the right-hand side's blocks I and J (already visible),
// part1 / part2 point at this tile's 128 partials; sT is scratch.  All 256 threads of the workgroup call it.
__device__ __forceinline__ void split_tile_product(const SplitRaw &w, const double *sI, const double *sJ, double (*sT)[TS], bool offdiag,
                                                   double *__restrict__ part1, double *__restrict__ part2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    double rj[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
    double tc[8], v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tc[k] = 0.0;
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        const double ri = sI[wave * 32 + 4 * rg + g];
        const float hh[8] = {w.ha[rg].x, w.ha[rg].y, w.ha[rg].z, w.ha[rg].w, w.hb[rg].x, w.hb[rg].y, w.hb[rg].z, w.hb[rg].w};
        const unsigned int qq[8] = {w.lq[rg].x & 0xffffu, w.lq[rg].x >> 16, w.lq[rg].y & 0xffffu, w.lq[rg].y >> 16,
                                    w.lq[rg].z & 0xffffu, w.lq[rg].z >> 16, w.lq[rg].w & 0xffffu, w.lq[rg].w >> 16};
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const double m0 = split_decode(hh[k], qq[k]), m1 = split_decode(hh[k + 1], qq[k + 1]);
            tc[k] = fma(m0, ri, tc[k]);                  // (pinning these as in fix_tile_product frees 40 registers and a third
            tc[k + 1] = fma(m1, ri, tc[k + 1]);          //  workgroup per CU, but measured 31.9 us against 30.6)
            a0 = fma(m0, rj[k], a0);
            a1 = fma(m1, rj[k + 1], a1);
        }
        v[rg] = a0 + a1;
    }
    // row sums: halving butterfly over the 16 column lanes (after the step with mask m a lane keeps the row groups whose
    // bit matches its own), then the last pair
#pragma unroll
    for (int m = 8, cnt = 4; m >= 2; m >>= 1, cnt >>= 1) {
        const bool up = (c & m) != 0;
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            const double lo_ = opaque(v[k]), hi_ = opaque(v[k + cnt]);
            v[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
        }
    }
    v[0] += __shfl_xor(v[0], 1, 64);
    if ((c & 1) == 0) {
        const int rg = ((c & 8) ? 4 : 0) + ((c & 4) ? 2 : 0) + ((c & 2) ? 1 : 0);
        part1[wave * 32 + 4 * rg + g] = v[0];
    }
    if (offdiag) {
        // column sums: over the wave's four row lanes g (masks 32, 16), then over the four waves through LDS
#pragma unroll
        for (int m = 32, cnt = 4; m >= 16; m >>= 1, cnt >>= 1) {
            const bool up = (lane & m) != 0;
#pragma unroll
            for (int k = 0; k < cnt; ++k) {
                const double lo_ = opaque(tc[k]), hi_ = opaque(tc[k + cnt]);
                tc[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
            }
        }
        const int col = ((lane & 32) ? 64 : 0) + 4 * c + ((lane & 16) ? 2 : 0);
        sT[wave][col] = tc[0]; sT[wave][col + 1] = tc[1];
        __syncthreads();
        if (threadIdx.x < TS)
            part2[threadIdx.x] = ((sT[0][threadIdx.x] + sT[1][threadIdx.x]) + sT[2][threadIdx.x]) + sT[3][threadIdx.x];
    }
}

__global__ void __launch_bounds__(256, 2)
symv_tile_split_kernel(const unsigned char *__restrict__ Mp, const double *__restrict__ rhs, int64_t np, int ntiles,
                       double *__restrict__ part1, double *__restrict__ part2, const AdmmStatus *status) {
    if (status != nullptr && status[0].converged) return;
    __shared__ double sI[TS], sJ[TS], sT[4][TS];
    const int t = blockIdx.x;
    int I, J;
    tile_index(t, I, J);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    SplitRaw w;
    split_load(Mp + (size_t)t * kSplitTileBytes, wave, lane >> 4, lane & 15, w);
    if (threadIdx.x < TS) sI[threadIdx.x] = rhs[(int64_t)I * TS + threadIdx.x];
    else sJ[threadIdx.x - TS] = rhs[(int64_t)J * TS + threadIdx.x - TS];
    __syncthreads();
    split_tile_product(w, sI, sJ, sT, I != J, part1 + (int64_t)t * TS, part2 + (int64_t)t * TS);
}

// Single-precision copy of M (the _f32 handles), one right-hand side: the lane ownership and reductions of the split kernel on
// plain float tiles (64 KB: three workgroups per CU).  The generic symv_tile_kernel<float> ran at 4.5 TB/s inside the iteration
// (30 us per launch at n = 8192).
__global__ void __launch_bounds__(256, 3)
symv_tile_f32_kernel(const float *__restrict__ Mp, const double *__restrict__ rhs, int64_t np, int ntiles,
                     double *__restrict__ part1_all, double *__restrict__ part2_all, const AdmmStatus *status) {
    if (status != nullptr && status[0].converged) return;
    __shared__ double sI[TS], sJ[TS], sT[4][TS];
    const int t = blockIdx.x;
    int I, J;
    tile_index(t, I, J);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const float *head = Mp + (size_t)t * TS * TS + (wave * 32 + g) * TS + 4 * c;
    float4 ha[8], hb[8];
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        ha[rg] = *reinterpret_cast<const float4 *>(head + rg * 4 * TS);
        hb[rg] = *reinterpret_cast<const float4 *>(head + rg * 4 * TS + 64);
    }
    if (threadIdx.x < TS) sI[threadIdx.x] = rhs[(int64_t)I * TS + threadIdx.x];
    else sJ[threadIdx.x - TS] = rhs[(int64_t)J * TS + threadIdx.x - TS];
    __syncthreads();
    double rj[8], tc[8], v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) tc[k] = 0.0;
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        const double ri = sI[wave * 32 + 4 * rg + g];
        const float hh[8] = {ha[rg].x, ha[rg].y, ha[rg].z, ha[rg].w, hb[rg].x, hb[rg].y, hb[rg].z, hb[rg].w};
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const double m0 = (double)hh[k], m1 = (double)hh[k + 1];
            tc[k] = opaque(fma(m0, ri, tc[k]));
            tc[k + 1] = opaque(fma(m1, ri, tc[k + 1]));
            a0 = fma(m0, rj[k], a0);
            a1 = fma(m1, rj[k + 1], a1);
        }
        v[rg] = a0 + a1;
    }
    tile_reduce_store(v, tc, sT, I != J, part1_all + (int64_t)t * TS, part2_all + (int64_t)t * TS);
}

// The mixed storage's kernel: fixed-point tiles hold 74 KB instead of 96, so THREE workgroups per CU are needed to keep as many
// bytes in flight (two: 27.5 us at cfg3 = 5.7 TB/s).  The fixed-point path fits the 168 registers that allows; the few float-head
// tiles (the diagonal ones) are processed in two halves of 64 rows to fit as well.
__global__ void __launch_bounds__(256, 3)
symv_tile_mixed_kernel(const unsigned char *__restrict__ Mp, const unsigned char *__restrict__ types, const double *__restrict__ rhs, int64_t np,
                       int ntiles, double *__restrict__ part1_all, double *__restrict__ part2_all, const AdmmStatus *status, int fmode) {
    if (status != nullptr && status[0].converged) return;
    __shared__ double sI[TS], sJ[TS], sT[4][TS];
    const int t = blockIdx.x;
    int I, J;
    tile_index(t, I, J);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *tile = Mp + (size_t)t * kSplitTileBytes;
    double *part1 = part1_all + (int64_t)t * TS, *part2 = part2_all + (int64_t)t * TS;
    if (types[t] != 0) {                             // (uniform) 36- / 32-bit fixed point; type 2: a diagonal tile, its diagonal apart in doubles
        FixRaw f;
        fix_load(tile, wave, lane, f, fmode);
        if (threadIdx.x < TS) sI[threadIdx.x] = rhs[(int64_t)I * TS + threadIdx.x];
        else sJ[threadIdx.x - TS] = rhs[(int64_t)J * TS + threadIdx.x - TS];
        __syncthreads();
        fix_tile_product(f, sI, sJ, sT, part1, part2, (types[t] == 2 && fmode != 2) ? reinterpret_cast<const double *>(tile + kFixHeadBytes + kFixNibBytes + TS * 4) : nullptr);
        return;
    }
    if (fmode == 2) {                                // (uniform) a float-head tile has no nibbles: zero partials
        if (threadIdx.x < TS) { part1[threadIdx.x] = 0.0; part2[threadIdx.x] = 0.0; }
        return;
    }
    // float head + 16-bit tail, two halves of four row groups
    const int c = lane & 15, g = lane >> 4;
    if (threadIdx.x < TS) sI[threadIdx.x] = rhs[(int64_t)I * TS + threadIdx.x];
    else sJ[threadIdx.x - TS] = rhs[(int64_t)J * TS + threadIdx.x - TS];
    __syncthreads();
    double rj[8], tc[8], v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) tc[k] = 0.0;
    const float *head = reinterpret_cast<const float *>(tile) + (wave * 32 + g) * TS + 4 * c;
    const unsigned short *tail = reinterpret_cast<const unsigned short *>(tile + (size_t)TS * TS * 4) + (wave * 32 + g) * TS + 8 * c;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float4 ha[4], hb[4];
        uint4 lq[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int rg = 4 * half + r4;
            ha[r4] = *reinterpret_cast<const float4 *>(head + rg * 4 * TS);
            hb[r4] = *reinterpret_cast<const float4 *>(head + rg * 4 * TS + 64);
            lq[r4] = *reinterpret_cast<const uint4 *>(tail + rg * 4 * TS);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int rg = 4 * half + r4;
            const double ri = sI[wave * 32 + 4 * rg + g];
            const float hh[8] = {ha[r4].x, ha[r4].y, ha[r4].z, ha[r4].w, hb[r4].x, hb[r4].y, hb[r4].z, hb[r4].w};
            const unsigned int qq[8] = {lq[r4].x & 0xffffu, lq[r4].x >> 16, lq[r4].y & 0xffffu, lq[r4].y >> 16,
                                        lq[r4].z & 0xffffu, lq[r4].z >> 16, lq[r4].w & 0xffffu, lq[r4].w >> 16};
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const double m0 = split_decode(hh[k], qq[k]), m1 = split_decode(hh[k + 1], qq[k + 1]);
                tc[k] = opaque(fma(m0, ri, tc[k]));
                tc[k + 1] = opaque(fma(m1, ri, tc[k + 1]));
                a0 = fma(m0, rj[k], a0);
                a1 = fma(m1, rj[k + 1], a1);
            }
            v[rg] = a0 + a1;
        }
    }
    tile_reduce_store(v, tc, sT, I != J, part1, part2);
}

// mixed storage for a batch of problems that each own their matrix (windows): blockIdx.y = matrix, serving nrhs right-hand sides
// (one right-hand side per matrix only: a loop over right-hand sides around the tile held in registers does not fit the 168
// registers of three workgroups per CU -- batches with several signals per window keep the uniform 6-byte kernel)
template <bool NT>
__global__ void __launch_bounds__(256, 3)
symv_tile_mixed_batch_kernel(const unsigned char *__restrict__ Mp_all, const unsigned char *__restrict__ types_all, size_t mp_stride,
                             const double *__restrict__ rhs_all, int64_t np, int ntiles, double *__restrict__ part1_all,
                             double *__restrict__ part2_all, const AdmmStatus *status) {
    constexpr int nrhs = 1;
    const int mat = blockIdx.y;
    {
        bool all = status != nullptr;
        for (int r = 0; r < nrhs && all; ++r) all = status[mat * nrhs + r].converged != 0;
        if (all) return;
    }
    __shared__ double sI[TS], sJ[TS], sT[4][TS];
    const int t = blockIdx.x;
    int I, J;
    tile_index(t, I, J);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *tile = Mp_all + (size_t)mat * mp_stride + (size_t)t * kSplitTileBytes;
    auto stage_rhs = [&](int sg, int rr) {            // right-hand side sg's blocks I and J -> LDS
        const double *rhs = rhs_all + (int64_t)sg * np;
        if (rr > 0) __syncthreads();   // previous right-hand side's readers are done with sI / sJ / sT
        if (threadIdx.x < TS) sI[threadIdx.x] = rhs[(int64_t)I * TS + threadIdx.x];
        else sJ[threadIdx.x - TS] = rhs[(int64_t)J * TS + threadIdx.x - TS];
        __syncthreads();
    };
    const unsigned char ttype = types_all[(size_t)mat * ntiles + t];
    if (ttype != 0) {                                // (uniform) 36-bit fixed point; type 2: a diagonal tile, its diagonal apart in doubles
        FixRaw f;
        fix_load<NT>(tile, wave, lane, f);
        stage_rhs(mat, 0);
        fix_tile_product(f, sI, sJ, sT, part1_all + ((int64_t)mat * ntiles + t) * TS, part2_all + ((int64_t)mat * ntiles + t) * TS,
                         ttype == 2 ? reinterpret_cast<const double *>(tile + kFixHeadBytes + kFixNibBytes + TS * 4) : nullptr);
        return;
    }
    // float head + 16-bit tail, two halves of four row groups
    const int c = lane & 15, g = lane >> 4;
    const float *head = reinterpret_cast<const float *>(tile) + (wave * 32 + g) * TS + 4 * c;
    const unsigned short *tail = reinterpret_cast<const unsigned short *>(tile + (size_t)TS * TS * 4) + (wave * 32 + g) * TS + 8 * c;
    for (int rr = 0; rr < nrhs; ++rr) {
        const int sg = mat * nrhs + rr;
        if (status != nullptr && status[sg].converged) continue;   // uniform
        stage_rhs(sg, rr);
        double rj[8], tc[8], v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) tc[k] = 0.0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 ha[4], hb[4];
            uint4 lq[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rg = 4 * half + r4;
                ha[r4] = load16<NT, float4>(head + rg * 4 * TS);
                hb[r4] = load16<NT, float4>(head + rg * 4 * TS + 64);
                lq[r4] = load16<NT, uint4>(tail + rg * 4 * TS);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rg = 4 * half + r4;
                const double ri = sI[wave * 32 + 4 * rg + g];
                const float hh[8] = {ha[r4].x, ha[r4].y, ha[r4].z, ha[r4].w, hb[r4].x, hb[r4].y, hb[r4].z, hb[r4].w};
                const unsigned int qq[8] = {lq[r4].x & 0xffffu, lq[r4].x >> 16, lq[r4].y & 0xffffu, lq[r4].y >> 16,
                                            lq[r4].z & 0xffffu, lq[r4].z >> 16, lq[r4].w & 0xffffu, lq[r4].w >> 16};
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const double m0 = split_decode(hh[k], qq[k]), m1 = split_decode(hh[k + 1], qq[k + 1]);
                    tc[k] = opaque(fma(m0, ri, tc[k]));
                    tc[k + 1] = opaque(fma(m1, ri, tc[k + 1]));
                    a0 = fma(m0, rj[k], a0);
                    a1 = fma(m1, rj[k + 1], a1);
                }
                v[rg] = a0 + a1;
            }
        }
        tile_reduce_store(v, tc, sT, I != J, part1_all + ((int64_t)sg * ntiles + t) * TS, part2_all + ((int64_t)sg * ntiles + t) * TS);
    }
}

// the same for a batch of problems that each own their matrix (windows): blockIdx.y = matrix, serving nrhs right-hand sides
__global__ void __launch_bounds__(256, 2)
symv_tile_split_batch_kernel(const unsigned char *__restrict__ Mp_all, size_t mp_stride, const double *__restrict__ rhs_all, int64_t np,
                             int ntiles, int nrhs, double *__restrict__ part1_all, double *__restrict__ part2_all, const AdmmStatus *status) {
    const int mat = blockIdx.y;
    {
        bool all = status != nullptr;
        for (int r = 0; r < nrhs && all; ++r) all = status[mat * nrhs + r].converged != 0;
        if (all) return;
    }
    __shared__ double sI[TS], sJ[TS], sT[4][TS];
    const int t = blockIdx.x;
    int I, J;
    tile_index(t, I, J);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    SplitRaw w;
    split_load(Mp_all + (size_t)mat * mp_stride + (size_t)t * kSplitTileBytes, wave, lane >> 4, lane & 15, w);
    for (int rr = 0; rr < nrhs; ++rr) {
        const int sg = mat * nrhs + rr;
        if (status != nullptr && status[sg].converged) continue;   // uniform
        const double *rhs = rhs_all + (int64_t)sg * np;
        if (rr > 0) __syncthreads();   // previous right-hand side's readers are done with sI / sJ / sT
        if (threadIdx.x < TS) sI[threadIdx.x] = rhs[(int64_t)I * TS + threadIdx.x];
        else sJ[threadIdx.x - TS] = rhs[(int64_t)J * TS + threadIdx.x - TS];
        __syncthreads();
        split_tile_product(w, sI, sJ, sT, I != J, part1_all + ((int64_t)sg * ntiles + t) * TS, part2_all + ((int64_t)sg * ntiles + t) * TS);
    }
}


// Several right-hand sides sharing M (multichannel problems): same tile product, but the right-hand sides of up to NSB
// signals are staged together and the column sums of all of them are reduced together -- two barriers per block of
// signals instead of three per signal (the barriers, not the arithmetic, kept the memory pipe idle between tiles).
template <typename T, int NSB>
__global__ void __launch_bounds__(256)
symv_tile_multi_kernel(const T *__restrict__ Mp, const double *__restrict__ rhs_all, int64_t np, int ns, int ntiles,
                       double *__restrict__ part1_all, double *__restrict__ part2_all, const AdmmStatus *status) {
    typedef typename Pair<T>::type T2;
    if (status != nullptr) {
        bool all = true;
        for (int q = 0; q < ns; ++q) all = all && status[q].converged;
        if (all) return;
    }
    __shared__ double sI[NSB][TS], sJ[NSB][TS], sT[NSB][4][TS];
    const int t = blockIdx.x;
    int I, J;
    tile_index(t, I, J);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T2 *base = reinterpret_cast<const T2 *>(Mp + (int64_t)t * TS * TS + wave * 32 * TS) + lane;
    T2 m[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) m[r] = base[r * (TS / 2)];
    for (int s0 = 0; s0 < ns; s0 += NSB) {
        const int nsb = ns - s0 < NSB ? ns - s0 : NSB;
        if (s0 > 0) __syncthreads();   // previous block's readers are done with the staging arrays
        for (int e = threadIdx.x; e < nsb * 2 * TS; e += 256) {
            const int q = e / (2 * TS), i = e - q * 2 * TS;
            const double *rhs = rhs_all + (int64_t)(s0 + q) * np;
            if (i < TS) sI[q][i] = rhs[(int64_t)I * TS + i];
            else sJ[q][i - TS] = rhs[(int64_t)J * TS + i - TS];
        }
        __syncthreads();
        for (int q = 0; q < nsb; ++q) {
            const int sg = s0 + q;
            if (status != nullptr && status[sg].converged) continue;   // uniform
            double *part1 = part1_all + (int64_t)sg * ntiles * TS;
            const double rj0 = sJ[q][2 * lane], rj1 = sJ[q][2 * lane + 1];
            double t0 = 0, t1 = 0, v[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const double ri = sI[q][wave * 32 + r];
                const double mx = (double)m[r].x, my = (double)m[r].y;
                t0 = fma(mx, ri, t0);
                t1 = fma(my, ri, t1);
                v[r] = fma(mx, rj0, my * rj1);
            }
#pragma unroll
            for (int w = 32, cnt = 16; w >= 2; w >>= 1, cnt >>= 1) {
                const bool hi = (lane & w) != 0;
#pragma unroll
                for (int k = 0; k < cnt; ++k) {
                    const double send = hi ? v[k] : v[k + cnt];
                    const double keep = hi ? v[k + cnt] : v[k];
                    v[k] = keep + __shfl_xor(send, w, 64);
                }
            }
            v[0] += __shfl_xor(v[0], 1, 64);
            if ((lane & 1) == 0) {
                const int row = ((lane & 32) ? 16 : 0) + ((lane & 16) ? 8 : 0) + ((lane & 8) ? 4 : 0) + ((lane & 4) ? 2 : 0) + ((lane & 2) ? 1 : 0);
                part1[(int64_t)t * TS + wave * 32 + row] = v[0];
            }
            sT[q][wave][2 * lane] = t0; sT[q][wave][2 * lane + 1] = t1;
        }
        if (I != J) {
            __syncthreads();
            for (int e = threadIdx.x; e < nsb * TS; e += 256) {
                const int q = e / TS, i = e - q * TS;
                if (status != nullptr && status[s0 + q].converged) continue;
                part2_all[(int64_t)(s0 + q) * ntiles * TS + (int64_t)t * TS + i] = ((sT[q][0][i] + sT[q][1][i]) + sT[q][2][i]) + sT[q][3][i];
            }
        }
    }
}


// ---- multi-signal tile product on the matrix cores ------------------------------------------------------------------
// With ns right-hand sides sharing M the per-signal cross-lane row reductions of symv_tile_multi_kernel, not the memory
// pipe, bound the kernel (2.9 TB/s at ns = 8).  Here a tile is two small GEMMs on v_mfma_f64_16x16x4_f64:
//     P1[i][s] = sum_c T[i][c] R_J[c][s]      (A operand = T, 16 rows x 4 cols;  B = R_J)
//     P2[s][c] = sum_i R_I[i][s] T[i][c]      (A = R_I', B = T, 4 rows x 16 cols)
// The two products need T in transposed operand layouts, so the tile goes through LDS in four 32-row stages (LDS-DMA,
// one 1 KiB row per instruction, rows padded by 16 B) and is read from there in either layout: waves 0-1 form P1 of the
// stage's two 16-row blocks, waves 2-3 accumulate P2 over the stages (four 16-column blocks each).  43 KB of LDS per
// workgroup: three workgroups per CU keep the memory pipe busy while others multiply (double-buffering the stages
// inside a workgroup at two workgroups per CU was slower: 1.36 vs 1.24 ms at ns = 8, n = 32768).  Signals are processed eight
// (ns <= 8: the MFMA's 16-wide signal dimension is half used) or sixteen at a time (no padding: the tile product costs the
// same matrix-pipe time for twice the signals).  Partials have the layout of the scalar kernels, so the update kernels
// are shared.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int MT_RS = TS + 2;                        // padded row stride of the staged rows (doubles)
constexpr int MT_ROWS = 32;                          // tile rows per stage
// NS = signals per pass: 8 (half of the MFMA's 16-wide signal dimension is padding) or 16 (none); LDS 43 / 53 KB
template <int NS> constexpr size_t symv_mfma_lds() { return sizeof(double) * ((size_t)MT_ROWS * MT_RS + (size_t)MT_ROWS * NS + (size_t)TS * NS); }

__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int NS>
__global__ void __launch_bounds__(256, 3)
symv_tile_mfma_kernel(const double *__restrict__ Mp, const double *__restrict__ rhs_all, int64_t np, int ns, int ntiles,
                      double *__restrict__ part1_all, double *__restrict__ part2_all, const AdmmStatus *status) {
    if (status != nullptr) {
        bool all = true;
        for (int q = 0; q < ns; ++q) all = all && status[q].converged;
        if (all) return;
    }
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *stg = lds;                               // [32][MT_RS]   the current 32 rows of the tile
    double *ri = stg + MT_ROWS * MT_RS;              // [32][NS]      right-hand sides of row block I, rows of the current stage
    double *rj = ri + MT_ROWS * NS;                  // [128][NS]     right-hand sides of row block J
    const int t = blockIdx.x;
    int I, J;
    tile_index(t, I, J);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const double *src = Mp + (int64_t)t * TS * TS;
    for (int s0 = 0; s0 < ns; s0 += NS) {
        const int nsb = ns - s0 < NS ? ns - s0 : NS;
        f64x4 acc2[4];                               // waves 2-3: P2 blocks, columns 64*(wave-2) + 16*u .., over all stages
#pragma unroll
        for (int u = 0; u < 4; ++u) acc2[u] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int q = 0; q < TS / MT_ROWS; ++q) {
            __syncthreads();                         // everyone is done with the previous contents of the LDS images
#pragma unroll
            for (int r = 0; r < MT_ROWS / 4; ++r) {  // wave w brings rows 8w..8w+7 of the stage: one 1 KiB row per instruction
                const int row = wave * (MT_ROWS / 4) + r;
                glds16(src + (int64_t)(MT_ROWS * q + row) * TS + 2 * lane, stg + row * MT_RS);
            }
            for (int e = tid; e < MT_ROWS * NS; e += 256) {   // R_I rows of this stage: 32 x NS values (zero beyond the pass's signals)
                const int sq = e / MT_ROWS, i = e - sq * MT_ROWS;
                ri[i * NS + sq] = sq < nsb ? rhs_all[(int64_t)(s0 + sq) * np + (int64_t)I * TS + MT_ROWS * q + i] : 0.0;
            }
            if (q == 0)
                for (int e = tid; e < TS * NS; e += 256) {
                    const int sq = e / TS, i = e - sq * TS;
                    rj[i * NS + sq] = sq < nsb ? rhs_all[(int64_t)(s0 + sq) * np + (int64_t)J * TS + i] : 0.0;
                }
            __syncthreads();                         // DMA landed (vmcnt(0)), staging visible
            if (wave < 2) {
                // P1: rows 16*wave .. of this stage, all 128 columns.  A[i = li][k = lk], B[k = lk][j = s = li]
                f64x4 a0 = (f64x4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
                const double *arow = stg + (16 * wave + li) * MT_RS;
#pragma unroll 8
                for (int kk = 0; kk < 32; kk += 2) {
                    const int c = 4 * kk + lk;
                    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[c], li < NS ? rj[c * NS + li] : 0.0, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[c + 4], li < NS ? rj[(c + 4) * NS + li] : 0.0, a1, 0, 0, 0);
                }
                // D: col = lane&15 = s, row = lk + 4*reg
                if (li < nsb && !(status != nullptr && status[s0 + li].converged)) {
                    double *p1 = part1_all + (int64_t)(s0 + li) * ntiles * TS + (int64_t)t * TS + MT_ROWS * q + 16 * wave + lk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) p1[4 * r] = a0[r] + a1[r];
                }
            } else {
                // P2: columns 64*(wave-2) + 16*u .., the 32 rows of this stage.  A[s = li][k = lk], B[k = lk][j = c = li]
#pragma unroll
                for (int kk = 0; kk < MT_ROWS / 4; ++kk) {
                    const int i = 4 * kk + lk;
                    const double a = li < NS ? ri[i * NS + li] : 0.0;
                    const double *brow = stg + i * MT_RS + 64 * (wave - 2) + li;
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc2[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, brow[16 * u], acc2[u], 0, 0, 0);
                }
            }
        }
        if (I != J && wave >= 2) {   // P2: D row = s = lk + 4*reg, col = li
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < NS / 4; ++r) {
                    const int sgl = lk + 4 * r;
                    if (sgl < nsb && !(status != nullptr && status[s0 + sgl].converged))
                        part2_all[(int64_t)(s0 + sgl) * ntiles * TS + (int64_t)t * TS + 64 * (wave - 2) + 16 * u + li] = acc2[u][r];
                }
        }
    }
}

// ---- the multi-signal tile product, wave-specialised and streamed (6-byte or 8-byte packed copy) --------------------------
// What bounds symv_tile_mfma_kernel (measured at n = 32768, ns = 8: 1.02 ms per launch, 4.2 TB/s of 8-byte tiles):
//   * the 16.8 M v_mfma_f64_16x16x4_f64 of a launch need 0.52 ms of the matrix pipe (its 66 TFLOP/s issue ceiling);
//   * on gfx950 the fp64 MFMA and the vector ALU exclude each other: every VALU instruction issued by ANY wave of the SIMD
//     costs the matrix pipe ~7 cycles (measured by adding dummy v_add_u32 to a co-resident wave: +7.1 cycles of MFMA time
//     each; scalar instructions are free).  Address arithmetic, selects for the padded signal lanes, LDS-DMA bookkeeping and
//     the decode of 6-byte elements all bill the matrix pipe;
//   * waves that alternate "stage a tile slice" / "multiply" between barriers leave either pipe idle half of the time, and
//     co-resident workgroups fall into step instead of filling each other's gaps.
// Here ONE 512-thread workgroup per CU is persistent (tiles t = blockIdx.x, += gridDim.x) and split into roles:
//   waves 4-7, LOADERS: global -> registers (a ring of D 32-row stages in flight per CU: ~100 KB) -> decode -> the LDS
//     image of the NEXT stage (double-buffered); all addresses are a scalar base plus a loop-invariant lane offset, so the
//     only vector instructions left are the three per element of the 6-byte decode;
//   waves 0-3, MFMA (one per SIMD): waves 0-1 form P1 of the stage's two 16-row blocks, waves 2-3 accumulate P2 over the
//     tile's four stages; operands come from LDS with immediate offsets (no selects: the right-hand-side images always hold
//     16 signal columns; columns beyond ns hold signal ns-1 again and their results are not stored).
// One barrier per stage.  Signals are processed 16 per pass; the last pass of ns > 16 re-covers the last 16 signals.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (a native vector: stays in registers where HIP's uint4 struct did not)
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int WS_NS = 16;                            // signal columns of the LDS images = the MFMA's N
constexpr size_t symv_ws_lds() { return sizeof(double) * 2 * ((size_t)MT_ROWS * MT_RS + (size_t)MT_ROWS * WS_NS + (size_t)TS * WS_NS); }

struct StreamVisit { int t, pass, I, J, k, end, u; };  // one (tile, signal pass): four 32-row stages; RUNS: of segment k = [.., end); PANEL: of unit u = row I of panel k, columns .. end

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_f64(double v, __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    const u32x2 w = {(unsigned int)__double2loint(v), (unsigned int)__double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, voffset, soffset, 0);
}

// Q4 (ns <= 8): the products run on v_mfma_f64_4x4x4_4b_f64 instead -- four independent 4x4x4 blocks per instruction (lane layout
// probed by tools/mfma_4x4x4_layout.hip: A[blk][i][k] in lane 16k + 4blk + i, B[blk][k][j] in lane 16k + 4blk + j, D[blk][i][j] in lane
// 16i + 4blk + j).  With the four blocks on four row quads (P1) or four column quads (P2) the TILE operand has exactly the lane layout
// of the 16x16x4 form, and the small operand is a 4 x 4 block of right-hand sides repeated in every block: signals 0-3 and 4-7 are two
// instructions on the same tile registers -- 8 signal columns cost 2 x 16 cycles where the 16-column instruction costs 64 with half of
// it padding.  The right-hand-side images hold 8 signals per row, ordered (s & 3) * 2 + (s >> 2): a lane's two quads are one 16-byte read.
// RUNS: the row-major triangle is cut into nseg SEGMENTS of consecutive tiles, segment k = [k ntiles / nseg, (k+1) ntiles / nseg)
// (about 8 tiles), workgroup g walks the segments g, g + G, ... -- at any moment the workgroups still stream one contiguous stretch of
// M between them, as with single tiles -- and the P1 waves keep the sums of a row block in registers over a RUN of tiles of the same
// row I within a segment: one record (id I + k: at most nseg + nblk of them) per run instead of one per tile -- most of one half of the
// partials (270 MB of 539 MB per launch at cfg5) is never written and never read back.  The consumers find row block I's records as
// [I + k(I,0), I + k(I,I)] (part1_range below).
// FIX (with SPLIT): the mixed storage -- tiles whose format byte is non-zero are 36-bit fixed point (74 240 of the slot's 98 304 bytes: the
// same 64 KB of leading dwords, then 8 KB of nibbles and a float step per row), the others (all diagonal tiles among them) float-head.
// A visit's format byte is requested (scalar load) a whole visit before its first stage is fetched.  Both formats' loads are issued for
// every stage, through buffer descriptors of size zero for the format the tile does not have (those loads are dropped: no load under a
// branch); a fixed-point element decodes in four vector instructions (bit-field extract, two integer instructions that assemble
// 2^52 + q, one FMA with the row's step).
// PANEL (with RUNS and Q4; round 5): the triangle is walked in COLUMN PANELS of kPanelC tile columns, a panel from its diagonal down, row by
// row -- unit (k, I) = the tiles (I, kC .. min(kC + C - 1, I)) of row I in panel k, unit index u = k nblk - C k (k - 1) / 2 + I - k C.  A
// workgroup walks a contiguous range of units [u0, u1) (the host cuts the unit list into gridDim.x ranges of equal tile counts: panel_plan).
//   P1 (row sums): one record per unit, id u -- the run logic of RUNS with the unit as the run;
//   P2 (column sums): the C column blocks of the panel keep their sums in registers of the two P2 waves across
//   ALL rows a workgroup walks in the panel, and are written out once per (workgroup, panel) as records f C + c, f = the running
//   flush index (ptab gives the workgroup's first; the flushes of panel k are the contiguous range [F0[k], F0[k+1]), which is
//   how the consumers find them).  At n = 32768: 8 224 P1 records + ~320 x 4 P2 records per signal instead of 4 370 + 32 640 -- the
//   270 MB of per-tile column sums of a launch (and their re-read by the reduction) become 10 MB.
constexpr int kPanelC = 4;
constexpr size_t kPanelLds = 0;                                       // (the column sums live in registers)
#if defined(LPVS_TIMELINE) && LPVS_TIMELINE == 3
// Debug build only (make timeline3 -> liblpvspectral_timeline3.so; tools/ws_timeline.py): where a launch of the wave-specialised kernel spends its
// time, BY ROLE.  One wave of each role (wave 0: P1, wave 2: P2, wave 4: loader) keeps the 100-MHz wall clock (s_memrealtime) at entry and end and
// the SUM of the time it spent waiting at the stage barriers (stamp before / after every barrier); the loader also the sum of the time inside `put`
// (waiting for the stage's bytes + decoding them into the LDS image).  The role that waits least at the barriers is the one the others wait for.
// 32 words per workgroup: role r at [8 r ..]: {entry, -, end, barrier-wait ticks, barriers, put ticks (loader)}; [24] XCC_ID, [25] HW_ID.
__device__ unsigned long long *g_lpvs_tl_ws = nullptr;
extern "C" int32_t lpvs_debug_set_timeline_ws(unsigned long long *dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lpvs_tl_ws), &dev_buf, sizeof(dev_buf)) == hipSuccess ? LPVS_OK : LPVS_EDEVICE;
}
// (32-bit tick arithmetic in VECTOR registers: the kernel has no scalar register to spare -- 102 used, any more spill --, and six vector
// instructions per stage cost the matrix pipe ~40 of a stage's ~2800 cycles)
__device__ __forceinline__ unsigned int ws_tl_now() {
    unsigned int v;
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"((unsigned int)t));
    return v;
}
#define WS_TL_DECL unsigned int tl_pre = ws_tl_now(), tl_post = tl_pre, tl_wait, tl_put, tl_t = tl_pre, tl_nbar; \
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0" : "=v"(tl_wait), "=v"(tl_put), "=v"(tl_nbar)); (void)tl_t; (void)tl_put; \
    if (g_lpvs_tl_ws != nullptr && lane == 0 && (wave == 0 || wave == 2 || wave == 4)) g_lpvs_tl_ws[(size_t)blockIdx.x * 32 + 8 * (wave >> 1)] = __builtin_amdgcn_s_memrealtime();
#define WS_BARRIER() do { tl_wait += tl_post - tl_pre; tl_pre = ws_tl_now(); __syncthreads(); tl_post = ws_tl_now(); ++tl_nbar; } while (0)
#define WS_TL_PUT_BEGIN() do { tl_t = ws_tl_now(); } while (0)
#define WS_TL_PUT_END() do { tl_put += ws_tl_now() - tl_t; } while (0)
#define WS_TL_FINISH(role) do { if (g_lpvs_tl_ws != nullptr && lane == 0) { unsigned long long *r_ = g_lpvs_tl_ws + (size_t)blockIdx.x * 32 + 8 * (role); \
        r_[2] = __builtin_amdgcn_s_memrealtime(); r_[3] = tl_wait + (tl_post - tl_pre); r_[4] = tl_nbar; r_[5] = tl_put; \
        if ((role) == 0) { r_[24] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20); r_[25] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4); } } } while (0)
#else
#define WS_TL_DECL
#define WS_BARRIER() __syncthreads()
#define WS_TL_PUT_BEGIN() do { } while (0)
#define WS_TL_PUT_END() do { } while (0)
#define WS_TL_FINISH(role) do { } while (0)
#endif
template <bool SPLIT, bool Q4, bool RUNS, bool FIX, bool PANEL = false>
__global__ void __launch_bounds__(512, 1)
symv_tile_mfma_ws_kernel(const unsigned char *__restrict__ Mp, const double *__restrict__ rhs_all, int64_t np, int ns, int ntiles,
                         double *__restrict__ part1_all, double *__restrict__ part2_all, const AdmmStatus *status, int nseg,
                         const unsigned char *__restrict__ types /* FIX: per-tile formats */, const int *__restrict__ ptab = nullptr /* PANEL: panel_plan's table */) {
    static_assert(!PANEL || (RUNS && Q4), "the panel walk keeps P1 runs and needs the 8-signal LDS images (room for the column sums)");
    if (status != nullptr) {
        bool all = true;
        for (int q = 0; q < ns; ++q) all = all && status[q].converged;
        if (all) return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NS = Q4 ? 8 : WS_NS, NQ = TS / MT_ROWS;
    static_assert(NQ == 4, "a tile is four stages: ring slot = stage, LDS parity = stage & 1");
    constexpr unsigned kStgB = MT_ROWS * MT_RS * 8, kRiB = MT_ROWS * NS * 8, kRjB = TS * NS * 8;   // bytes of one image
    unsigned char *stg = lds_raw;                    // [2][32][MT_RS]   32 rows of a tile, by stage parity
    unsigned char *ri = stg + 2 * kStgB;             // [2][32][16]      right-hand sides of row block I, rows of the stage
    unsigned char *rj = ri + 2 * kRiB;               // [2][128][16]     right-hand sides of row block J, by visit parity
    constexpr size_t kTileBytes = SPLIT ? kSplitTileBytes : (size_t)TS * TS * 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ltid = tid & 255;                      // index within the role's four waves
    const int li = lane & 15, lk = lane >> 4;
    const int G = gridDim.x;
    // Step q of a tile is its stage (q + q0) & 3 with q0 = blockIdx.x & 3, so that workgroups advancing in near lockstep do
    // not all ask for the same quarter of their (power-of-two sized) tiles at the same time.
    const int q0 = blockIdx.x & (NQ - 1);
    const int npass = (ns + NS - 1) / NS, nvalid = ns < NS ? ns : NS, s0max = ns > NS ? ns - NS : 0;
    constexpr int tend_unused = 0; (void)tend_unused;
    const int tend = ntiles;
    auto seg_begin = [&](int k) -> int { return (int)(((unsigned long long)(unsigned)k * (unsigned)ntiles) / (unsigned)nseg); };
    if ((int)blockIdx.x >= (RUNS ? nseg : ntiles)) return;
    const int nblk = (int)(np / TS);
    int pu1 = 0, pf0 = 0;                            // PANEL: end of this workgroup's unit range, its first flush index
    auto next = [&](StreamVisit v) -> StreamVisit {  // scalar only; t >= ntiles after the workgroup's last visit
        if (++v.pass == npass) {
            v.pass = 0;
            if constexpr (PANEL) {
                if (++v.J <= v.end) ++v.t;
                else if (++v.u < pu1) {              // the next unit: the next row of the panel, or the first row of the next panel
                    if (++v.I == nblk) { ++v.k; v.I = v.k * kPanelC; }
                    v.J = v.k * kPanelC;
                    v.end = v.J + kPanelC - 1 < v.I ? v.J + kPanelC - 1 : v.I;
                    v.t = v.I * (v.I + 1) / 2 + v.J;
                } else v.t = ntiles;
            } else if constexpr (RUNS) {
                if (++v.t < v.end) {
                    if (++v.J > v.I) { v.J = 0; ++v.I; }
                } else if ((v.k += G) < nseg) {      // the workgroup's next segment
                    v.t = __builtin_amdgcn_readfirstlane(seg_begin(v.k)); v.end = __builtin_amdgcn_readfirstlane(seg_begin(v.k + 1));
                    tile_index(v.t, v.I, v.J);
                    v.I = __builtin_amdgcn_readfirstlane(v.I); v.J = __builtin_amdgcn_readfirstlane(v.J);
                } else v.t = ntiles;
            } else {
                v.t += G;
                v.J += G;                            // t = I(I+1)/2 + J, 0 <= J <= I
                while (v.J > v.I) { v.J -= v.I + 1; ++v.I; }
            }
        }
        return v;
    };
    auto s0_of = [&](const StreamVisit &v) -> int { const int s0 = v.pass * NS; return s0 < s0max ? s0 : s0max; };
    StreamVisit cv{RUNS ? seg_begin(blockIdx.x) : (int)blockIdx.x, 0, 0, 0, (int)blockIdx.x, RUNS ? seg_begin(blockIdx.x + 1) : 0, 0};   // the visit being multiplied
    if constexpr (PANEL) {
        const int *row = ptab + 5 * blockIdx.x;      // {u0, u1, k0, I0, f0}: scalar loads
        cv.u = __builtin_amdgcn_readfirstlane(row[0]); pu1 = __builtin_amdgcn_readfirstlane(row[1]);
        cv.k = __builtin_amdgcn_readfirstlane(row[2]); cv.I = __builtin_amdgcn_readfirstlane(row[3]); pf0 = __builtin_amdgcn_readfirstlane(row[4]);
        if (cv.u >= pu1) return;                     // (more workgroups than units: uniform)
        cv.J = cv.k * kPanelC;
        cv.end = cv.J + kPanelC - 1 < cv.I ? cv.J + kPanelC - 1 : cv.I;
        cv.t = cv.I * (cv.I + 1) / 2 + cv.J;
    } else {
        cv.t = __builtin_amdgcn_readfirstlane(cv.t); cv.end = __builtin_amdgcn_readfirstlane(cv.end);
        tile_index(cv.t, cv.I, cv.J);
        cv.I = __builtin_amdgcn_readfirstlane(cv.I); cv.J = __builtin_amdgcn_readfirstlane(cv.J);
    }
    int tp = 0;                                      // parity of the visit count
    WS_TL_DECL

    if (wave >= 4) {
        // ---- loader waves.  Thread (r = ltid >> 4, c = ltid & 15) owns rows r and r + 16 of a stage:
        //   split:  two 16-byte pieces of heads (columns 4c.., 64+4c..) and one of tails per row (pack_tiles_split_kernel's layout)
        //   double: four 16-byte pieces per row (columns 32j + 2c, 2c+1)
        const int r = ltid >> 4, c = ltid & 15;
        static_assert(!FIX || SPLIT, "fixed-point tiles live in the 6-byte slots");
        constexpr int NRAW = SPLIT ? (FIX ? 7 : 6) : 8, RI = MT_ROWS * NS / 256, RJ = TS * NS / 256;
        // fixed-point tiles: the dwords of this thread's two rows' nibbles and steps (pack_tiles_mixed_kernel's layout: row 32 w + 4 rg + g,
        // lane (g, c); this thread's rows are w = stage, rg = r >> 2 and (r >> 2) + 4, g = r & 3)
        const int lo_nib = (int)kFixHeadBytes + ((((r & 3) * 16 + c) * 8) + (r >> 2)) * 4;
        const int lo_stp = (int)(kFixHeadBytes + kFixNibBytes) + (((r & 3) * 8) + (r >> 2)) * 4;
        u32x4 raw[NQ][NRAW];                          // ring slot = step of the tile
        double pri[NQ][RI], prj[RJ];                 // (the J slice travels with step 0)
        // loop-invariant lane offsets (bytes)
        const int lo_a = SPLIT ? r * (TS * 4) + c * 16 : r * (TS * 8) + c * 16;   // heads (split) / doubles
        const int lo_t = r * (TS * 2) + c * 16;                                    // tails (split)
        int go_ri[RI], go_rj[RJ];
        unsigned wo_ri[RI], wo_rj[RJ];
#pragma unroll
        for (int k = 0; k < RI; ++k) {
            const int e = ltid + 256 * k, sq = e >> 5, i = e & 31;
            go_ri[k] = (int)(((int64_t)(sq < ns ? sq : ns - 1) * np + i) * 8);
            wo_ri[k] = (i * NS + (Q4 ? (sq & 3) * 2 + (sq >> 2) : sq)) * 8;
        }
#pragma unroll
        for (int k = 0; k < RJ; ++k) {
            const int e = ltid + 256 * k, sq = e >> 7, i = e & 127;
            go_rj[k] = (int)(((int64_t)(sq < ns ? sq : ns - 1) * np + i) * 8);
            wo_rj[k] = (i * NS + (Q4 ? (sq & 3) * 2 + (sq >> 2) : sq)) * 8;
        }
        const unsigned wo_stg = SPLIT ? (r * MT_RS + 4 * c) * 8 : (r * MT_RS + 2 * c) * 8;
        // Every load of a step is unconditional (past the last visit an earlier one is requested again): a load under a
        // branch, or registers that differ between two paths into the loop, make the compiler wait for the prefetch right
        // where it was issued.  Buffer loads: scalar descriptor + scalar offset + loop-invariant lane offset.
        auto type_of = [&](const StreamVisit &v) -> unsigned {   // (scalar load of the dword that holds the tile's format byte)
            if constexpr (!FIX) return 0u;
            const int t = v.t < tend ? v.t : tend - 1;
            return (reinterpret_cast<const unsigned int *>(types)[t >> 2] >> (8 * (t & 3))) & 255u;
        };
        auto fetch = [&](auto qc, const StreamVisit &v, unsigned ty) {
            constexpr int Q = decltype(qc)::value;
            const int qp = (Q + q0) & (NQ - 1);
            const __amdgpu_buffer_rsrc_t tile = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(Mp + (size_t)v.t * kTileBytes), 0, (int)kTileBytes, 0x00020000);
            if constexpr (SPLIT) {
                const int oh = qp * (MT_ROWS * TS * 4), ot = TS * TS * 4 + qp * (MT_ROWS * TS * 2);
                // (FIX: the tails exist in a diagonal tile only, nibbles and steps below the diagonal only -- descriptors of size zero drop the rest)
                const bool fx = FIX && ty != 0;
                const __amdgpu_buffer_rsrc_t tails = FIX ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(Mp + (size_t)v.t * kTileBytes), 0, fx ? 0 : (int)kTileBytes, 0x00020000) : tile;
                raw[Q][0] = __builtin_amdgcn_raw_buffer_load_b128(tile, lo_a, oh, 0);
                raw[Q][1] = __builtin_amdgcn_raw_buffer_load_b128(tile, lo_a + 256, oh, 0);
                raw[Q][2] = __builtin_amdgcn_raw_buffer_load_b128(tails, lo_t, ot, 0);
                raw[Q][3] = __builtin_amdgcn_raw_buffer_load_b128(tile, lo_a, oh + 16 * TS * 4, 0);
                raw[Q][4] = __builtin_amdgcn_raw_buffer_load_b128(tile, lo_a + 256, oh + 16 * TS * 4, 0);
                raw[Q][5] = __builtin_amdgcn_raw_buffer_load_b128(tails, lo_t, ot + 16 * TS * 2, 0);
                if constexpr (FIX) {
                    const __amdgpu_buffer_rsrc_t aux = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(Mp + (size_t)v.t * kTileBytes), 0, fx ? (int)kMixedFixedTileBytes : 0, 0x00020000);
                    raw[Q][6][0] = __builtin_amdgcn_raw_buffer_load_b32(aux, lo_nib, qp * (64 * 8 * 4), 0);
                    raw[Q][6][1] = __builtin_amdgcn_raw_buffer_load_b32(aux, lo_nib + 16, qp * (64 * 8 * 4), 0);
                    raw[Q][6][2] = __builtin_amdgcn_raw_buffer_load_b32(aux, lo_stp, qp * (4 * 8 * 4), 0);
                    raw[Q][6][3] = __builtin_amdgcn_raw_buffer_load_b32(aux, lo_stp + 16, qp * (4 * 8 * 4), 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        raw[Q][4 * k + j] = __builtin_amdgcn_raw_buffer_load_b128(tile, lo_a + j * 256, qp * (MT_ROWS * TS * 8) + k * (16 * TS * 8), 0);
            }
            const int s0 = s0_of(v);
            const int64_t left = (int64_t)(ns - s0) * np * 8;
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(rhs_all + (int64_t)s0 * np), 0,
                                                                              (int)(left < 0x7fffffff ? left : 0x7fffffff), 0x00020000);
            const int oI = (v.I * TS + MT_ROWS * qp) * 8;
#pragma unroll
            for (int k = 0; k < RI; ++k) pri[Q][k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rr, go_ri[k], oI, 0));
            if constexpr (Q == 0) {
                const int oJ = v.J * TS * 8;
#pragma unroll
                for (int k = 0; k < RJ; ++k) prj[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rr, go_rj[k], oJ, 0));
            }
        };
        auto put = [&](auto qc, int vtp, bool fx) {  // step's registers -> the LDS images of parity Q & 1 (J slice: visit parity); fx: a fixed-point tile
            constexpr int Q = decltype(qc)::value;
            constexpr unsigned par = Q & 1;
            unsigned char *sp = stg + par * kStgB + wo_stg;
            if (FIX && fx) {                         // (uniform)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const u32x4 ha = raw[Q][3 * k], hb = raw[Q][3 * k + 1];
                    const unsigned int nw = raw[Q][FIX ? 6 : 0][k];
                    const double step = (double)__uint_as_float(raw[Q][FIX ? 6 : 0][2 + k]), off = -(0x1p52 + 0x1p35) * step;
                    auto dec = [&](unsigned int hi, int j) -> double {      // (2^52 + q) * step - (2^52 + 2^35) * step, exact
                        const unsigned int top = __builtin_amdgcn_alignbit(0x04330000u, hi, 28);
                        unsigned int lo = (nw >> (4 * j)) & 15u;
                        asm("v_lshl_or_b32 %0, %1, 4, %0" : "+v"(lo) : "v"(hi));
                        return fma(__hiloint2double((int)top, (int)lo), step, off);
                    };
                    f64x2 *row = reinterpret_cast<f64x2 *>(sp + k * (16 * MT_RS * 8));
                    row[0] = (f64x2){dec(ha.x, 0), dec(ha.y, 1)};
                    row[1] = (f64x2){dec(ha.z, 2), dec(ha.w, 3)};
                    row[32] = (f64x2){dec(hb.x, 4), dec(hb.y, 5)};
                    row[33] = (f64x2){dec(hb.z, 6), dec(hb.w, 7)};
                }
            } else if constexpr (SPLIT) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const u32x4 ha = raw[Q][3 * k], hb = raw[Q][3 * k + 1], lq = raw[Q][3 * k + 2];
                    f64x2 *row = reinterpret_cast<f64x2 *>(sp + k * (16 * MT_RS * 8));
                    row[0] = (f64x2){split_decode_lo(__uint_as_float(ha.x), lq.x), split_decode_hi(__uint_as_float(ha.y), lq.x)};
                    row[1] = (f64x2){split_decode_lo(__uint_as_float(ha.z), lq.y), split_decode_hi(__uint_as_float(ha.w), lq.y)};
                    row[32] = (f64x2){split_decode_lo(__uint_as_float(hb.x), lq.z), split_decode_hi(__uint_as_float(hb.y), lq.z)};
                    row[33] = (f64x2){split_decode_lo(__uint_as_float(hb.z), lq.w), split_decode_hi(__uint_as_float(hb.w), lq.w)};
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4 *>(sp + k * (16 * MT_RS * 8) + j * 256) = raw[Q][4 * k + j];
            }
#pragma unroll
            for (int k = 0; k < RI; ++k) *reinterpret_cast<double *>(ri + par * kRiB + wo_ri[k]) = pri[Q][k];
            if constexpr (Q == 0) {
                unsigned char *rjp = rj + vtp * kRjB;
#pragma unroll
                for (int k = 0; k < RJ; ++k) *reinterpret_cast<double *>(rjp + wo_rj[k]) = prj[k];
            }
        };
        using std::integral_constant;
        StreamVisit v1 = next(cv);                   // the visit after the current one; `src` = the visit the ring is refilled from
        unsigned tyc = type_of(cv), ty1 = type_of(v1);   // formats of the current and the next visit's tiles (scalar)
        // prologue: the first visit's four steps in flight, step 0 staged, slot 0 refilled from the next visit
        fetch(integral_constant<int, 0>{}, cv, tyc);
        fetch(integral_constant<int, 1>{}, cv, tyc);
        fetch(integral_constant<int, 2>{}, cv, tyc);
        fetch(integral_constant<int, 3>{}, cv, tyc);
        WS_TL_PUT_BEGIN(); put(integral_constant<int, 0>{}, 0, tyc != 0); WS_TL_PUT_END();
        fetch(integral_constant<int, 0>{}, v1.t < tend ? v1 : cv, v1.t < tend ? ty1 : tyc);
        WS_BARRIER();
#pragma unroll 1
        for (;;) {
            // while the MFMA waves multiply step q of the current visit, stage step q + 1 and refill its slot from the next visit
            const StreamVisit v2 = next(v1);         // (its format byte is requested here and used at the end of this visit)
            const unsigned ty2 = type_of(v2);
            const StreamVisit src = v1.t < tend ? v1 : cv;
            const unsigned tys = v1.t < tend ? ty1 : tyc;
            const bool fxc = tyc != 0;
            WS_TL_PUT_BEGIN(); put(integral_constant<int, 1>{}, tp, fxc); WS_TL_PUT_END(); fetch(integral_constant<int, 1>{}, src, tys); WS_BARRIER();
            WS_TL_PUT_BEGIN(); put(integral_constant<int, 2>{}, tp, fxc); WS_TL_PUT_END(); fetch(integral_constant<int, 2>{}, src, tys); WS_BARRIER();
            WS_TL_PUT_BEGIN(); put(integral_constant<int, 3>{}, tp, fxc); WS_TL_PUT_END(); fetch(integral_constant<int, 3>{}, src, tys); WS_BARRIER();
            WS_TL_PUT_BEGIN(); if (v1.t < tend) put(integral_constant<int, 0>{}, tp ^ 1, ty1 != 0); WS_TL_PUT_END();
            fetch(integral_constant<int, 0>{}, v2.t < tend ? v2 : cv, v2.t < tend ? ty2 : tyc);
            WS_BARRIER();
            if (v1.t >= tend) break;
            cv = v1; v1 = v2; tyc = ty1; ty1 = ty2; tp ^= 1;
        }
        if (wave == 4) WS_TL_FINISH(2);
        return;
    }

    // ---- MFMA waves: loop-invariant lane offsets (bytes)
    const bool p1 = wave < 2;
    const unsigned small_lane = Q4 ? (lk * NS + (li & 3) * 2) * 8 : (lk * NS + li) * 8;                     // the right-hand-side operand (Q4: two quads = 16 bytes)
    const unsigned a_lane = p1 ? ((16 * wave + li) * MT_RS + lk) * 8 : small_lane;                         // P1: A = tile rows; P2: A = ri
    const unsigned b_lane = p1 ? small_lane : (lk * MT_RS + 64 * (wave - 2) + li) * 8;                     // P1: B = rj;        P2: B = tile rows
    // partials: buffer stores, a lane whose signal is beyond the pass's valid ones gets an out-of-range offset (store dropped)
    //   16x16x4: part1 [signal li][tile][row lk + 4k], part2 [signal lk + 4k][tile][col 16u + li]
    //   Q4:      part1 [signal (li & 3) + 4h][tile][row 4 (li >> 2) + lk], part2 [signal lk + 4h][tile][col 16u + li]
    const int s_lane = Q4 ? (p1 ? (int)(((int64_t)(li & 3) * ntiles * TS + 16 * wave + 4 * (li >> 2) + lk) * 8) : (int)(((int64_t)lk * ntiles * TS + 64 * (wave - 2) + li) * 8))
                          : (p1 ? (li < nvalid ? (int)(((int64_t)li * ntiles * TS + 16 * wave + lk) * 8) : (int)0x80000000u)
                                : (int)(((int64_t)lk * ntiles * TS + 64 * (wave - 2) + li) * 8));
    const bool hi_valid = Q4 && (p1 ? (li & 3) + 4 : lk + 4) < nvalid, lo_valid = !Q4 || (p1 ? (li & 3) : lk) < nvalid;
    const int64_t pass_bytes = (int64_t)nvalid * ntiles * TS * 8;
    const int part_records = (int)(pass_bytes < 0x7fffffff ? pass_bytes : 0x7fffffff);
    f64x4 acc2[4];                                   // waves 2-3: P2 blocks, columns 64*(wave-2) + 16*u .., over the tile's stages
    WS_BARRIER();                                    // step 0 of the first visit staged
    // One continuous software pipeline over the steps: a step's 32 MFMAs run as four groups of eight, each group's operands
    // read from LDS while the previous group multiplies.  The step's barrier sits BEFORE its last group (whose operands are in
    // registers by then: nobody reads the step's LDS images after it), and the next step's first operands are requested
    // right after it, so the matrix pipe does not drain at step boundaries.
    if (p1) {
        // P1: rows 16*wave .. of a stage, all 128 columns.  A[i = li][k = lk], B[k = lk][j = s = li]
        // Q4: the right-hand-side operand (B: the J slice, 32 pivot steps x two signal quads) does not change over the four steps of a
        // visit -- it is read from LDS with step 0's groups only and kept in registers (128 of them): the LDS operand reads of the four
        // MFMA waves cost as much time as their matrix instructions, and these were a third of them
        using BT = std::conditional_t<Q4, f64x2, double>;
        double A[2][8]; BT B[Q4 ? 1 : 2][8]; BT Bv[Q4 ? 32 : 1];
        auto load = [&](auto gc, unsigned par, int vtp, int buf, bool with_b) {      // operands of group G of the stage with parity par
            constexpr int g = decltype(gc)::value;
            const unsigned char *ap = stg + par * kStgB + a_lane, *bp = rj + vtp * kRjB + b_lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                A[buf][j] = *reinterpret_cast<const double *>(ap + 32 * (8 * g + j));
                if constexpr (Q4) { if (with_b) Bv[Q4 ? 8 * g + j : 0] = *reinterpret_cast<const BT *>(bp + 4 * NS * 8 * (8 * g + j)); }
                else B[Q4 ? 0 : buf][j] = *reinterpret_cast<const BT *>(bp + 4 * NS * 8 * (8 * g + j));
            }
        };
        using std::integral_constant;
        load(integral_constant<int, 0>{}, 0, 0, 0, true);
        f64x4 run0[RUNS ? NQ : 1], run1[RUNS && !Q4 ? NQ : 1];   // RUNS: the sums of a run of tiles of one row block, step by step
        bool fresh = true;                           // (scalar) the visit starts a run
        auto store1 = [&](const f64x4 &a0, const f64x4 &a1, int so) {   // a stage's 16 rows of this wave; a converged signal's partials are never read
            const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(part1_all + (int64_t)s0_of(cv) * ntiles * TS, 0, part_records, 0x00020000);
            if constexpr (Q4) {                      // D: row = 4 (li >> 2) + lk, signal = (li & 3) + 4h
                store_f64(a0[0] + a0[2], pr, lo_valid ? s_lane : (int)0x80000000u, so);
                store_f64(a0[1] + a0[3], pr, hi_valid ? s_lane : (int)0x80000000u, so + 4 * ntiles * TS * 8);
            } else {                                 // D: col = li = s, row = lk + 4*reg
#pragma unroll
                for (int k = 0; k < 4; ++k) store_f64(a0[k] + a1[k], pr, s_lane + 32 * k, so);
            }
        };
        auto step = [&](auto qc, int next_tp) {
            constexpr int Q = decltype(qc)::value;
            constexpr unsigned par = Q & 1;
            f64x4 t0 = (f64x4){0.0, 0.0, 0.0, 0.0}, t1 = t0;
            if constexpr (RUNS) { if (fresh) { run0[RUNS ? Q : 0] = t0; if constexpr (!Q4) run1[RUNS && !Q4 ? Q : 0] = t0; } }
            f64x4 &a0 = RUNS ? run0[RUNS ? Q : 0] : t0, &a1 = (RUNS && !Q4) ? run1[RUNS && !Q4 ? Q : 0] : t1;   // (Q4: elements 0/1 = signals lo/hi of the even k steps, 2/3 of the odd ones)
            auto mul = [&](int buf, auto gc) {
                constexpr int g = decltype(gc)::value;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    if constexpr (Q4) {
                        a0[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j], Bv[Q4 ? 8 * g + j : 0][0], a0[0], 0, 0, 0);
                        a0[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j], Bv[Q4 ? 8 * g + j : 0][1], a0[1], 0, 0, 0);
                        a0[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j + 1], Bv[Q4 ? 8 * g + j + 1 : 0][0], a0[2], 0, 0, 0);
                        a0[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j + 1], Bv[Q4 ? 8 * g + j + 1 : 0][1], a0[3], 0, 0, 0);
                    } else {
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[buf][j], B[Q4 ? 0 : buf][j], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[buf][j + 1], B[Q4 ? 0 : buf][j + 1], a1, 0, 0, 0);
                    }
                }
            };
            constexpr bool first = Q == 0, last = Q == NQ - 1;   // (B is read with the groups of a visit's step 0; group 0 of step 0 rides on the previous step 3)
            load(integral_constant<int, 1>{}, par, tp, 1, first); __builtin_amdgcn_sched_barrier(0); mul(0, integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0);
            load(integral_constant<int, 2>{}, par, tp, 0, first); __builtin_amdgcn_sched_barrier(0); mul(1, integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0);
            load(integral_constant<int, 3>{}, par, tp, 1, first); __builtin_amdgcn_sched_barrier(0); mul(0, integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0);
            WS_BARRIER();                            // the next step is staged; this step's images are free
            load(integral_constant<int, 0>{}, par ^ 1, next_tp, 0, last); __builtin_amdgcn_sched_barrier(0); mul(1, integral_constant<int, 3>{}); __builtin_amdgcn_sched_barrier(0);
            if constexpr (!RUNS) store1(a0, a1, (cv.t * TS + MT_ROWS * ((Q + q0) & (NQ - 1))) * 8);
        };
#pragma unroll 1
        for (;;) {
            step(integral_constant<int, 0>{}, tp);
            step(integral_constant<int, 1>{}, tp);
            step(integral_constant<int, 2>{}, tp);
            step(integral_constant<int, 3>{}, tp ^ 1);
            const StreamVisit nv = next(cv);
            if constexpr (RUNS) {
                fresh = PANEL ? (nv.t >= tend || nv.u != cv.u) : (nv.t >= tend || nv.I != cv.I || nv.k != cv.k);   // the run ends with this tile: record I + k (PANEL: the unit's, u)
                if (fresh) {
                    const int rec = PANEL ? cv.u : cv.I + cv.k;
#pragma unroll
                    for (int Q = 0; Q < NQ; ++Q) store1(run0[Q], run1[Q4 ? 0 : Q], (rec * TS + MT_ROWS * ((Q + q0) & (NQ - 1))) * 8);
                }
            }
            cv = nv;
            if (cv.t >= tend) break;
            tp ^= 1;
        }
        if (wave == 0) WS_TL_FINISH(0);
    } else {
        // P2: columns 64*(wave-2) + 16*u .., the 32 rows of a stage.  A[s = li][k = lk], B[k = lk][j = c = li]
        using AT = std::conditional_t<Q4, f64x2, double>;
        AT A[2][2]; double B[2][8];
        f64x2 acc4[4];                               // Q4: P2 blocks of signals lo/hi, columns 64*(wave-2) + 16*u ..
        auto load = [&](auto gc, unsigned par, int buf) {
            constexpr int g = decltype(gc)::value;
            const unsigned char *ap = ri + par * kRiB + a_lane, *bp = stg + par * kStgB + b_lane;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                A[buf][j] = *reinterpret_cast<const AT *>(ap + 4 * NS * 8 * (2 * g + j));
#pragma unroll
                for (int u = 0; u < 4; ++u) B[buf][4 * j + u] = *reinterpret_cast<const double *>(bp + 4 * MT_RS * 8 * (2 * g + j) + 128 * u);
            }
        };
        using std::integral_constant;
        load(integral_constant<int, 0>{}, 0, 0);
        auto step = [&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            constexpr unsigned par = Q & 1;
            auto mul = [&](int buf, bool first) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if constexpr (Q4) {
                            const f64x2 cin = (first && j == 0) ? (f64x2){0.0, 0.0} : acc4[u];
                            acc4[u][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j][0], B[buf][4 * j + u], cin[0], 0, 0, 0);
                            acc4[u][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j][1], B[buf][4 * j + u], cin[1], 0, 0, 0);
                        } else {
                            const f64x4 cin = (first && j == 0) ? (f64x4){0.0, 0.0, 0.0, 0.0} : acc2[u];   // a tile's first product starts the sums
                            acc2[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[buf][j], B[buf][4 * j + u], cin, 0, 0, 0);
                        }
                    }
            };
            load(integral_constant<int, 1>{}, par, 1); __builtin_amdgcn_sched_barrier(0); mul(0, Q == 0); __builtin_amdgcn_sched_barrier(0);
            load(integral_constant<int, 2>{}, par, 0); __builtin_amdgcn_sched_barrier(0); mul(1, false); __builtin_amdgcn_sched_barrier(0);
            load(integral_constant<int, 3>{}, par, 1); __builtin_amdgcn_sched_barrier(0); mul(0, false); __builtin_amdgcn_sched_barrier(0);
            WS_BARRIER();                            // the next step is staged; this step's images are free
            load(integral_constant<int, 0>{}, par ^ 1, 0); __builtin_amdgcn_sched_barrier(0); mul(1, false); __builtin_amdgcn_sched_barrier(0);
            if constexpr (Q == NQ - 1 && !PANEL) {   // P2 of the tile is complete: D row = s = lk + 4*reg, col = li
                if (cv.I != cv.J) {
                    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(part2_all + (int64_t)s0_of(cv) * ntiles * TS, 0, part_records, 0x00020000);
#pragma unroll
                    for (int k = 0; k < NS / 4; ++k) {
                        const int so = (int)(((int64_t)4 * k * ntiles + cv.t) * TS * 8);   // (beyond the pass's valid signals: out of range, dropped)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if constexpr (Q4) store_f64(acc4[u][k], pr, (k ? hi_valid : lo_valid) ? s_lane + 128 * u : (int)0x80000000u, so);
                            else store_f64(acc2[u][k], pr, s_lane + 128 * u, so);
                        }
                    }
                }
            }
        };
        // PANEL: the panel's column sums stay in REGISTERS (one set of four f64x2 per column block: the workgroup's 512 threads own the
        // CU, the P2 waves have a hundred registers to spare); the slot c = J - k C is wave-uniform, so picking the set is a scalar branch.
        // (A first version kept them in LDS: the read-modify-write after every tile sat on the P2 waves' way to the next barrier and cost
        // the launch 38 us of its 611.)
        f64x2 col[kPanelC][4];
        int nflush = 0;
        if constexpr (PANEL) {
#pragma unroll
            for (int c = 0; c < kPanelC; ++c)
#pragma unroll
                for (int u = 0; u < 4; ++u) col[c][u] = (f64x2){0.0, 0.0};
        }
#pragma unroll 1
        for (;;) {
            step(integral_constant<int, 0>{});
            step(integral_constant<int, 1>{});
            step(integral_constant<int, 2>{});
            step(integral_constant<int, 3>{});
            const StreamVisit nv = next(cv);
            if constexpr (PANEL) {
                if (cv.I != cv.J) {                  // (a diagonal tile's transposed product is its own P1)
                    const int c = cv.J - cv.k * kPanelC;
#pragma unroll
                    for (int cc = 0; cc < kPanelC; ++cc)
                        if (c == cc) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) col[cc][u] += acc4[u];
                        }
                }
                if (nv.t >= tend || nv.k != cv.k) {  // the workgroup leaves the panel: one record per column block, id (f0 + nflush) C + c
                    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(part2_all, 0, part_records, 0x00020000);
#pragma unroll
                    for (int c = 0; c < kPanelC; ++c) {
                        const int rec = (pf0 + nflush) * kPanelC + c;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
#pragma unroll
                            for (int k = 0; k < 2; ++k)
                                store_f64(col[c][u][k], pr, (k ? hi_valid : lo_valid) ? s_lane + 128 * u : (int)0x80000000u, (int)(((int64_t)4 * k * ntiles + rec) * TS * 8));
                            col[c][u] = (f64x2){0.0, 0.0};
                        }
                    }
                    ++nflush;
                }
            }
            cv = nv;
            if (cv.t >= tend) break;
        }
        if (wave == 2) WS_TL_FINISH(1);
    }
}

// Same tile product for a batch of problems that each own their matrix (windows): blockIdx.y = matrix; a matrix serves
// nrhs right-hand sides (problems nrhs*blockIdx.y ..; e.g. the two signals of ls_windowcsd share a window's Gram), the tile
// is read once and applied to each of them.
__global__ void __launch_bounds__(256)
symv_tile_batch_kernel(const double *__restrict__ Mp_all, int64_t mp_stride, const double *__restrict__ rhs_all, int64_t np,
                       int ntiles, int nrhs, double *__restrict__ part1_all, double *__restrict__ part2_all, const AdmmStatus *status) {
    const int mat = blockIdx.y;
    {
        bool all = status != nullptr;
        for (int r = 0; r < nrhs && all; ++r) all = status[mat * nrhs + r].converged != 0;
        if (all) return;
    }
    __shared__ double sI[TS], sJ[TS], sT[4][TS];
    const int t = blockIdx.x;
    int I, J;
    tile_index(t, I, J);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double2 *base = reinterpret_cast<const double2 *>(Mp_all + (int64_t)mat * mp_stride + (int64_t)t * TS * TS + wave * 32 * TS) + lane;
    double2 m[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) m[r] = base[r * (TS / 2)];
    for (int rr = 0; rr < nrhs; ++rr) {
        const int sg = mat * nrhs + rr;
        if (status != nullptr && status[sg].converged) continue;   // uniform
        const double *rhs = rhs_all + (int64_t)sg * np;
        double *part1 = part1_all + (int64_t)sg * ntiles * TS, *part2 = part2_all + (int64_t)sg * ntiles * TS;
        if (rr > 0) __syncthreads();   // previous right-hand side's readers are done with sI / sJ / sT
        if (threadIdx.x < TS) sI[threadIdx.x] = rhs[(int64_t)I * TS + threadIdx.x];
        else sJ[threadIdx.x - TS] = rhs[(int64_t)J * TS + threadIdx.x - TS];
        __syncthreads();
        const double rj0 = sJ[2 * lane], rj1 = sJ[2 * lane + 1];
        double t0 = 0, t1 = 0, v[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const double ri = sI[wave * 32 + r];
            t0 = fma(m[r].x, ri, t0);
            t1 = fma(m[r].y, ri, t1);
            v[r] = fma(m[r].x, rj0, m[r].y * rj1);
        }
#pragma unroll
        for (int w = 32, cnt = 16; w >= 2; w >>= 1, cnt >>= 1) {
            const bool hi = (lane & w) != 0;
#pragma unroll
            for (int k = 0; k < cnt; ++k) {
                const double send = hi ? v[k] : v[k + cnt];
                const double keep = hi ? v[k + cnt] : v[k];
                v[k] = keep + __shfl_xor(send, w, 64);
            }
        }
        v[0] += __shfl_xor(v[0], 1, 64);
        if ((lane & 1) == 0) {
            const int row = ((lane & 32) ? 16 : 0) + ((lane & 16) ? 8 : 0) + ((lane & 8) ? 4 : 0) + ((lane & 4) ? 2 : 0) + ((lane & 2) ? 1 : 0);
            part1[(int64_t)t * TS + wave * 32 + row] = v[0];
        }
        if (I != J) {
            sT[wave][2 * lane] = t0; sT[wave][2 * lane + 1] = t1;
            __syncthreads();
            if (threadIdx.x < TS)
                part2[(int64_t)t * TS + threadIdx.x] = ((sT[0][threadIdx.x] + sT[1][threadIdx.x]) + sT[2][threadIdx.x]) + sT[3][threadIdx.x];
        }
    }
}

// The part1 records of row block I: the tiles (I, 0..I) -- or, when the multi-signal kernel wrote one record per RUN (runs_G = its
// number of segments), the records I + k for the segments k = [k ntiles / nseg, (k+1) ntiles / nseg) that meet the row.
struct Part1Range { int first, count; };
// PANEL layout of the multi-signal kernel's partials (runs_G = -C, ptab = panel_plan's table; F0 behind the 5 G workgroup rows, whose count
// sits in F0[-1]): row block I's contributions are its P1 records u(k, I), k = 0 .. I / C, then the P2 records (F0[kI] .. F0[kI + 1]) C + I - kI C.
struct PanelList { int n1, nent, f0, c, C, nblk; };
__device__ __forceinline__ PanelList panel_list(int I, int nblk, int runs_G, const int *__restrict__ ptab) {
    const int C = -runs_G, kI = I / C;
    const int *F0 = ptab + 1 + 5 * ptab[0];
    const int f0 = F0[kI], f1 = F0[kI + 1];
    return {kI + 1, kI + 1 + f1 - f0, f0, I - kI * C, C, nblk};
}
__device__ __forceinline__ unsigned panel_entry(const PanelList &l, int I, int e, unsigned p2off) {   // element offset of entry e's 128 values from part1
    return e < l.n1 ? (unsigned)(e * l.nblk - l.C * (e * (e - 1) / 2) + I - e * l.C) * TS
                    : p2off + (unsigned)((l.f0 + e - l.n1) * l.C + l.c) * TS;
}

__device__ __forceinline__ Part1Range part1_range(int I, int ntiles, int runs_G) {
    const int t0 = I * (I + 1) / 2;
    if (runs_G == 0) return {t0, I + 1};
    const int g0 = (int)((((int64_t)t0 + 1) * runs_G - 1) / ntiles), g1 = (int)((((int64_t)t0 + I + 1) * runs_G - 1) / ntiles);
    return {I + g0, g1 - g0 + 1};
}

// x[I*128+i] = sum_{J<=I} part1[(I,J)][i] + sum_{K>I} part2[(K,I)][i]; both sums in fixed order.
// 256 threads: 0..127 walk part1, 128..255 walk part2, 16 independent loads in flight each.
__device__ __forceinline__ double gather_x(const double *__restrict__ part1, const double *__restrict__ part2, int nblk, int I,
                                           double *sh /*[128]*/, Part1Range r1) {
    const int i = threadIdx.x & 127, half = threadIdx.x >> 7;
    double s = 0;
    if (half == 0) {
        const double *p = part1 + (int64_t)r1.first * TS + i;
        int J = 0;
        for (; J + 16 <= r1.count; J += 16) {
            double a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = p[(int64_t)(J + q) * TS];
#pragma unroll
            for (int q = 0; q < 16; ++q) s += a[q];
        }
        for (; J < r1.count; ++J) s += p[(int64_t)J * TS];
    } else {
        int K = I + 1;
        for (; K + 16 <= nblk; K += 16) {
            double a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = part2[((int64_t)(K + q) * (K + q + 1) / 2 + I) * TS + i];
#pragma unroll
            for (int q = 0; q < 16; ++q) s += a[q];
        }
        for (; K < nblk; ++K) s += part2[((int64_t)K * (K + 1) / 2 + I) * TS + i];
        sh[i] = s;
    }
    __syncthreads();
    return half == 0 ? s + sh[i] : 0.0;
}

// 512-thread variant: the nblk contributions of a row block (part1 entries for J <= I, part2 entries for K > I) form
// one list; four groups of 128 lanes each sum a contiguous quarter (16 independent loads in flight at nblk = 64:
// one memory latency), then a fixed-order combine through LDS.
__device__ __forceinline__ double gather_x4(const double *__restrict__ part1, const double *__restrict__ part2, int nblk, int I,
                                            double *sh /*[3*128]*/) {
    const int i = threadIdx.x & 127, g = threadIdx.x >> 7;
    const int per = (nblk + 3) / 4, e0 = g * per, e1 = e0 + per < nblk ? e0 + per : nblk;
    auto at = [&](int e) -> const double * {
        return e <= I ? part1 + ((int64_t)I * (I + 1) / 2 + e) * TS + i : part2 + ((int64_t)e * (e + 1) / 2 + I) * TS + i;
    };
    double s = 0;
    int e = e0;
    for (; e + 16 <= e1; e += 16) {
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = *at(e + q);
#pragma unroll
        for (int q = 0; q < 16; ++q) s += a[q];
    }
    for (; e < e1; ++e) s += *at(e);
    if (g > 0) sh[(g - 1) * TS + i] = s;
    __syncthreads();
    return g == 0 ? ((s + sh[i]) + sh[TS + i]) + sh[2 * TS + i] : 0.0;
}

__global__ void __launch_bounds__(256)
symv_reduce_kernel(const double *__restrict__ part1, const double *__restrict__ part2, int nblk, int ntiles, int64_t np,
                   double *__restrict__ x, const AdmmStatus *status, const double *__restrict__ xb, int runs_G) {
    const int sg = blockIdx.y;
    if (status != nullptr && status[sg].converged) return;
    __shared__ double sh[TS];
    const double s = gather_x(part1 + (int64_t)sg * ntiles * TS, part2 + (int64_t)sg * ntiles * TS, nblk, blockIdx.x, sh,
                              part1_range(blockIdx.x, ntiles, runs_G));
    if (threadIdx.x < TS) {
        const int64_t gi = (int64_t)sg * np + (int64_t)blockIdx.x * TS + threadIdx.x;
        x[gi] = xb ? xb[gi] + s : s;
    }
}

// The same for the multi-signal kernel's partials per RUN: a row block has a few part1 records and up to nblk - 1 part2 tiles, so the
// contributions are taken as ONE list and summed by eight groups of 128 lanes, a contiguous eighth each (16 loads in flight per lane),
// then combined in fixed order.
__global__ void __launch_bounds__(1024)
symv_reduce_runs_kernel(const double *__restrict__ part1_all, const double *__restrict__ part2_all, int nblk, int ntiles, int64_t np,
                        double *__restrict__ x, const AdmmStatus *status, const double *__restrict__ xb, int runs_G, const int *__restrict__ ptab) {
    const int sg = blockIdx.y, I = blockIdx.x;
    if (status != nullptr && status[sg].converged) return;
    __shared__ double sh[7 * TS];
    const double *part1 = part1_all + (int64_t)sg * ntiles * TS, *part2 = part2_all + (int64_t)sg * ntiles * TS;
    const int i = threadIdx.x & 127, g = threadIdx.x >> 7;
    const bool panel = runs_G < 0;
    const Part1Range r1 = part1_range(I, ntiles, panel ? 0 : runs_G);
    const PanelList pl = panel ? panel_list(I, nblk, runs_G, ptab) : PanelList{0, 0, 0, 0, 1, nblk};
    const int nent = panel ? pl.nent : r1.count + nblk - 1 - I, eshift = I + 1 - r1.count;
    const int per = (nent + 7) / 8, e0 = g * per, e1 = e0 + per < nent ? e0 + per : nent;
    const unsigned p2off = (unsigned)(part2 - part1);
    auto at = [&](int e) -> const double * {
        if (panel) return part1 + panel_entry(pl, I, e, p2off) + i;
        const int K = e + eshift;
        return e < r1.count ? part1 + ((int64_t)r1.first + e) * TS + i : part2 + ((int64_t)K * (K + 1) / 2 + I) * TS + i;
    };
    double s = 0;
    int e = e0;
    for (; e + 16 <= e1; e += 16) {
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = *at(e + q);
#pragma unroll
        for (int q = 0; q < 16; ++q) s += a[q];
    }
    {   // the remainder, also in one batch (clamped addresses, masked values)
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = *at(e + q < e1 ? e + q : (nent > 0 ? nent - 1 : 0));
#pragma unroll
        for (int q = 0; q < 16; ++q) s += e + q < e1 ? a[q] : 0.0;
    }
    if (g > 0) sh[(g - 1) * TS + i] = s;
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int q = 0; q < 7; ++q) s += sh[q * TS + i];
        const int64_t gi = (int64_t)sg * np + (int64_t)I * TS + i;
        x[gi] = xb ? xb[gi] + s : s;
    }
}

// ---- fused: gather x from the tile partials + prox_g + dual update + next rhs, one workgroup per
// 128-row block; ||x-z||^2 is combined by the last-arriving workgroup in fixed block order (deterministic).
// Valid for element-wise prox (L1, L0) and for group prox with 128 % group_len == 0, n % group_len == 0.
__global__ void __launch_bounds__(512)
admm_fused_update_kernel(AdmmParams p, const double *__restrict__ part1_all, const double *__restrict__ part2_all, int nblk,
                         int ntiles, double *__restrict__ blocknorm_all, unsigned int *__restrict__ ticket_all) {
    const int sg = blockIdx.y;
    AdmmStatus *status = p.status + sg;
    if (status->converged) return;
    __shared__ double sh[3 * TS], sq[TS], gs[TS];
    __shared__ int last;
    const double *part1 = part1_all + (int64_t)sg * ntiles * TS, *part2 = part2_all + (int64_t)sg * ntiles * TS;
    double *blocknorm = blocknorm_all + (int64_t)sg * nblk;
    unsigned int *ticket = ticket_all + sg;
    const int I = blockIdx.x, i = threadIdx.x & 127;
    const int64_t li_ = (int64_t)I * TS + i, gi = (int64_t)sg * p.np + li_;
    const bool row = threadIdx.x < TS, ok = row && li_ < p.n;
    const bool offset_form = p.xb != nullptr;
    const double ui = ok ? p.u[gi] : 0.0, bi = ok ? (offset_form ? p.xb[gi] : p.b[gi]) : 0.0;   // in flight together with the partials
    double xi = gather_x4(part1, part2, nblk, I, sh);
    if (offset_form) xi += bi;                                   // (bi holds xb here)
    const double v = xi + ui;
    double zi = 0.0, d2 = 0.0;
    if (p.prox_kind == LPVS_PROX_L1) {
        const double gl = p.mu * p.prox_param;
        zi = v + (v <= -gl ? gl : (v >= gl ? -gl : -v));
    } else if (p.prox_kind == LPVS_PROX_L0) {
        zi = fabs(v) > sqrt(2.0 * p.mu * p.prox_param) ? v : 0.0;
    } else {  // group: block soft-threshold, norms through LDS
        const int gl = (int)p.group_len;
        if (row) sq[i] = v * v;
        __syncthreads();
        if (threadIdx.x < TS / gl) {
            double s2 = 0;
            for (int q = 0; q < gl; ++q) s2 += sq[threadIdx.x * gl + q];   // sequential, as norm() on the slice
            double scale = 1.0 - p.prox_param * p.mu / sqrt(s2);           // s2 == 0 -> -inf -> 0
            if (!(scale > 0)) scale = 0.0;
            gs[threadIdx.x] = scale;
        }
        __syncthreads();
        if (row) zi = gs[i / gl] * v;
    }
    if (row) {
        if (!ok) zi = 0.0;
        const double d = xi - zi, un = ui + d;     // src/lasso.jl:154-155
        p.x[gi] = xi; p.z[gi] = zi; p.u[gi] = un;
        p.rhs[gi] = ok ? (offset_form ? (zi - un) / p.mu : bi + (zi - un) / p.mu) : 0.0;
        d2 = ok ? d * d : 0.0;
    }
    // block sum of d2: the two row waves reduce by shuffles (fixed pattern -> reproducible)
    const double wsum = wave_sum(d2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&blocknorm[I], sh[0] + sh[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (tk == (unsigned)nblk - 1);
    }
    __syncthreads();
    if (last) {  // every other workgroup has published its block norm
        if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        if (threadIdx.x < 64) {   // lane q sums blocks q, q+64, ...; then the wave's fixed shuffle pattern
            double part = 0;
            for (int q = threadIdx.x; q < nblk; q += 64) part += __hip_atomic_load(&blocknorm[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double w = wave_sum(part);
            if (threadIdx.x == 0) sq[0] = w;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double nxz = sqrt(sq[0]);                               // norm(tmp)   src/lasso.jl:157
            status->iters += 1;
            status->nxz = nxz;
            if (nxz < p.tol) status->converged = 1;                       //             src/lasso.jl:164
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}



// ---- the same update with the convergence test DEFERRED by one launch (single-problem path) -------------------------
// The ticket above serialises every iteration on an agent-scope fence, an atomic and a last-arriving workgroup.  Here
// every workgroup writes its block's ||x-z||^2 into the buffer of the iteration's parity and is done.  The NEXT
// launch first sums the previous iteration's block norms (every workgroup, identically, with loads that fly together
// with its partial sums), commits iteration count / norm / convergence (workgroup 0), and if that iteration had
// converged nobody writes anything -- the iterates stay those of the converged iteration, exactly as with the
// ticket.  Whether a previous iteration is pending is known to the HOST (every launch of a chunk but the first has
// one), so it is a kernel argument: no workgroup reads a flag that another workgroup of the same launch writes.
// admm_commit_kernel commits the last iteration of a chunk.  Mat-vec launches issued past the (not yet visible)
// convergence are harmless.
__device__ __forceinline__ double pending_norm(const double *__restrict__ bn, int nblk, double *slot) {
    if (threadIdx.x < 64) {   // lane q sums blocks q, q+64, ...; then the wave's fixed shuffle pattern
        double part = 0;
        for (int q = threadIdx.x; q < nblk; q += 64) part += bn[q];
        const double w = wave_sum(part);
        if (threadIdx.x == 0) *slot = w;
    }
    __syncthreads();
    return sqrt(*slot);                                               // norm(tmp)   src/lasso.jl:157
}

__global__ void __launch_bounds__(512)
admm_fused_update2_kernel(AdmmParams p, const double *__restrict__ part1_all, const double *__restrict__ part2_all, int nblk,
                          int ntiles, double *__restrict__ blocknorm_all, int parity, int commit_prev, int runs_G, const int *__restrict__ ptab) {
    const int sg = blockIdx.y;
    AdmmStatus *status = p.status + sg;
    __shared__ double sh[3 * TS], sq[TS], gs[TS], slot;
    const double *part1 = part1_all + (int64_t)sg * ntiles * TS, *part2 = part2_all + (int64_t)sg * ntiles * TS;
    double *bn_prev = blocknorm_all + ((int64_t)sg * 2 + (parity ^ 1)) * nblk, *bn_cur = blocknorm_all + ((int64_t)sg * 2 + parity) * nblk;
    const int I = blockIdx.x, i = threadIdx.x & 127;
    const int64_t li_ = (int64_t)I * TS + i, gi = (int64_t)sg * p.np + li_;
    const bool row = threadIdx.x < TS, ok = row && li_ < p.n;
    const bool offset_form = p.xb != nullptr;                       // x = xb + M (z-u)/mu
    // EVERY load of the prologue is issued before anything is waited for -- the convergence flag, u, xb (or b), the first 16 tile
    // partials of this thread's quarter of the row block and the previous iteration's block norms -- and all of them
    // unconditionally (clamped addresses, values selected afterwards): a load under a branch makes the compiler drain the memory
    // pipe right there, and the kernel is nothing but memory latency (measured: three dependent round trips, 5.7 us per launch).
    const int conv_flag = __builtin_nontemporal_load(&status->converged);
    const double u_raw = p.u[gi], b_raw = (offset_form ? p.xb : p.b)[gi];
    const int gq = threadIdx.x >> 7;
    // the row block's contributions as one list: its part1 records (tiles (I, 0..I), or the runs of the multi-signal kernel), then part2 of tiles (K > I, I)
    const bool panel = runs_G < 0;                                            // (the multi-signal kernel's panel walk: uniform)
    const Part1Range r1 = part1_range(I, ntiles, panel ? 0 : runs_G);
    const PanelList pl = panel ? panel_list(I, nblk, runs_G, ptab) : PanelList{0, 0, 0, 0, 1, nblk};
    const int nent = panel ? pl.nent : r1.count + nblk - 1 - I, eshift = I + 1 - r1.count;      // (runs_G == 0: nent = nblk, eshift = 0)
    const int per = (nent + 3) / 4, e0 = gq * per, e1 = e0 + per < nent ? e0 + per : nent;
    // (32-bit element offsets from part1: both partial arrays live in one buffer, part2 behind part1; the 64-bit form of this
    // address arithmetic was 300 instructions ahead of the first load)
    const unsigned p2off = (unsigned)(part2 - part1);
    const unsigned rowoff = (unsigned)r1.first * TS + i;
    auto at = [&](int e) -> const double * {
        if (panel) return part1 + panel_entry(pl, I, e, p2off) + i;
        const int K = e + eshift;
        return part1 + (e < r1.count ? rowoff + (unsigned)e * TS : p2off + ((unsigned)(K * (K + 1) / 2 + I)) * TS + i);
    };
    double pre[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) pre[q] = *at(e0 + q < nent ? e0 + q : nent - 1);
    const int lane64 = threadIdx.x & 63;
    const double bn_raw = bn_prev[lane64 < nblk ? lane64 : nblk - 1];
    // (a use of every loaded value BEFORE the early exit: otherwise the compiler tests the flag first and sinks the loads below
    // the branch -- two dependent scalar round trips ahead of everything else)
    asm volatile("" :: "v"(u_raw), "v"(b_raw), "v"(bn_raw), "v"(pre[0]), "v"(pre[1]), "v"(pre[2]), "v"(pre[3]), "v"(pre[4]), "v"(pre[5]),
                 "v"(pre[6]), "v"(pre[7]), "v"(pre[8]), "v"(pre[9]), "v"(pre[10]), "v"(pre[11]), "v"(pre[12]), "v"(pre[13]), "v"(pre[14]),
                 "v"(pre[15]), "s"(conv_flag));
    // `converged` is written by workgroup 0 of a launch only when that launch's commit finds convergence, and then
    // every workgroup of the launch (whether it reads the flag before or after that write) returns without writing.
    if (conv_flag) return;
    const double ui = ok ? u_raw : 0.0, bi = ok ? b_raw : 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) pre[q] = e0 + q < e1 ? pre[q] : 0.0;
    if (commit_prev) {                                               // uniform (host-known): commit the previous iteration
        if (threadIdx.x < 64) {   // lane q sums blocks q, q+64, ...; then the wave's fixed shuffle pattern (as pending_norm)
            double part = 0;
            part += lane64 < nblk ? bn_raw : 0.0;
            for (int q = lane64 + 64; q < nblk; q += 64) part += bn_prev[q];
            const double w = wave_sum(part);
            if (threadIdx.x == 0) slot = w;
        }
        __syncthreads();
        const double nxz = sqrt(slot);                               // norm(tmp)   src/lasso.jl:157
        const bool conv = nxz < p.tol;                               //             src/lasso.jl:164
        if (I == 0 && threadIdx.x == 0) {
            status->iters += 1;
            status->nxz = nxz;
            if (conv) status->converged = 1;
        }
        if (conv) return;                                            // every workgroup takes the same decision
    }
    double xi;
    {   // same summation order as gather_x4: the quarter's contributions in order, then the four quarters in order
        double sacc = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) sacc += pre[q];                 // (entries past e1 are +0.0: they do not change the sum's bits)
        int e = e0 + 16;
        for (; e + 16 <= e1; e += 16) {
            double a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = *at(e + q);
#pragma unroll
            for (int q = 0; q < 16; ++q) sacc += a[q];
        }
        for (; e < e1; ++e) sacc += *at(e);
        if (gq > 0) sh[(gq - 1) * TS + i] = sacc;
        __syncthreads();
        xi = gq == 0 ? ((sacc + sh[i]) + sh[TS + i]) + sh[2 * TS + i] : 0.0;
    }
    if (offset_form) xi += bi;                                       // (bi holds xb here)
    const double v = xi + ui;
    double zi = 0.0, d2 = 0.0;
    if (p.prox_kind == LPVS_PROX_L1) {
        const double gl = p.mu * p.prox_param;
        zi = v + (v <= -gl ? gl : (v >= gl ? -gl : -v));
    } else if (p.prox_kind == LPVS_PROX_L0) {
        zi = fabs(v) > sqrt(2.0 * p.mu * p.prox_param) ? v : 0.0;
    } else {  // group: block soft-threshold, norms through LDS
        const int gl = (int)p.group_len;
        if (row) sq[i] = v * v;
        __syncthreads();
        if (threadIdx.x < TS / gl) {
            double s2 = 0;
            for (int q = 0; q < gl; ++q) s2 += sq[threadIdx.x * gl + q];   // sequential, as norm() on the slice
            double scale = 1.0 - p.prox_param * p.mu / sqrt(s2);           // s2 == 0 -> -inf -> 0
            if (!(scale > 0)) scale = 0.0;
            gs[threadIdx.x] = scale;
        }
        __syncthreads();
        if (row) zi = gs[i / gl] * v;
    }
    if (row) {
        if (!ok) zi = 0.0;
        const double d = xi - zi, un = ui + d;     // src/lasso.jl:154-155
        p.x[gi] = xi; p.z[gi] = zi; p.u[gi] = un;
        p.rhs[gi] = ok ? (offset_form ? (zi - un) / p.mu : bi + (zi - un) / p.mu) : 0.0;
        d2 = ok ? d * d : 0.0;
    }
    const double wsum = wave_sum(d2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) bn_cur[I] = sh[0] + sh[1];
}

// commits the last executed iteration of a chunk (one workgroup per signal)
__global__ void __launch_bounds__(64)
admm_commit_kernel(AdmmParams p, int nblk, double *__restrict__ blocknorm_all, int parity_last) {
    const int sg = blockIdx.x;
    AdmmStatus *status = p.status + sg;
    if (status->converged) return;   // otherwise the chunk's last iteration is pending (the host launches this only after >= 1 iteration)
    __shared__ double slot;
    const double nxz = pending_norm(blocknorm_all + ((int64_t)sg * 2 + parity_last) * nblk, nblk, &slot);
    if (threadIdx.x == 0) {
        status->iters += 1;
        status->nxz = nxz;
        if (nxz < p.tol) status->converged = 1;
    }
}

// Small systems of a batch (np <= 1024, e.g. the windows of ls_windowpsd): ONE workgroup per problem does what
// admm_fused_update_kernel does with nblk workgroups -- thread group g (128 lanes) owns row block g -- so the norm
// ||x-z|| needs no cross-workgroup ticket (no agent-scope fences, no atomics).  blockDim = 128 * nblk.
__global__ void __launch_bounds__(1024)
admm_window_update_kernel(AdmmParams p, const double *__restrict__ part1_all, const double *__restrict__ part2_all, int nblk, int ntiles) {
    const int sg = blockIdx.x;
    AdmmStatus *status = p.status + sg;
    if (status->converged) return;
    __shared__ double sq[8 * TS], gs[8 * TS], wsum_s[16];
    const double *part1 = part1_all + (int64_t)sg * ntiles * TS, *part2 = part2_all + (int64_t)sg * ntiles * TS;
    const int I = threadIdx.x >> 7, i = threadIdx.x & 127;
    const int64_t li_ = (int64_t)I * TS + i, gi = (int64_t)sg * p.np + li_;
    const bool ok = li_ < p.n;
    const double ui = ok ? p.u[gi] : 0.0, bi = ok ? p.b[gi] : 0.0;
    double a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        a[e] = e < nblk ? (e <= I ? part1[((int64_t)I * (I + 1) / 2 + e) * TS + i] : part2[((int64_t)e * (e + 1) / 2 + I) * TS + i]) : 0.0;
    double xi = a[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) xi += a[e];                    // fixed order: tile column 0, 1, ...
    const bool offset_form = p.xb != nullptr;                  // x = xb + M~ (z-u)/mu (reduced-precision copy of M, see the split kernel)
    if (offset_form) xi += ok ? p.xb[gi] : 0.0;
    const double v = xi + ui;
    double zi = 0.0;
    if (p.prox_kind == LPVS_PROX_L1) {
        const double gl = p.mu * p.prox_param;
        zi = v + (v <= -gl ? gl : (v >= gl ? -gl : -v));
    } else if (p.prox_kind == LPVS_PROX_L0) {
        zi = fabs(v) > sqrt(2.0 * p.mu * p.prox_param) ? v : 0.0;
    } else {  // group: block soft-threshold, norms through LDS (128 % group_len == 0)
        const int gl = (int)p.group_len;
        sq[threadIdx.x] = v * v;
        __syncthreads();
        if (i < TS / gl) {
            double s2 = 0;
            for (int q = 0; q < gl; ++q) s2 += sq[I * TS + i * gl + q];
            double scale = 1.0 - p.prox_param * p.mu / sqrt(s2);
            if (!(scale > 0)) scale = 0.0;
            gs[I * TS + i] = scale;
        }
        __syncthreads();
        zi = gs[I * TS + i / gl] * v;
    }
    if (!ok) zi = 0.0;
    const double d = xi - zi, un = ui + d;                       // src/lasso.jl:154-155
    p.x[gi] = xi; p.z[gi] = zi; p.u[gi] = un;
    p.rhs[gi] = ok ? (offset_form ? (zi - un) / p.mu : bi + (zi - un) / p.mu) : 0.0;
    const double w = wave_sum(ok ? d * d : 0.0);
    if ((threadIdx.x & 63) == 0) wsum_s[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0;
        for (int q = 0; q < 2 * nblk; ++q) tot += wsum_s[q];     // wave order = row order
        const double nxz = sqrt(tot);                            // norm(tmp)   src/lasso.jl:157
        status->iters += 1;
        status->nxz = nxz;
        if (nxz < p.tol) status->converged = 1;                  //             src/lasso.jl:164
    }
}

// ---- batch of independent small problems (windows of ls_windowpsd): problem q = blockIdx.y ----------------
__global__ void __launch_bounds__(256)
admm_batch_init_kernel(AdmmBatch p) {
    const int q = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < p.np) {
        const int64_t o = (int64_t)q * p.np + i;
        const double x0 = (p.x0 != nullptr && i < p.n) ? p.x0[o] : 0.0;   // init = true (src/lasso.jl:112): x = z = the ridge solution, u = 0
        p.x[o] = x0; p.z[o] = x0; p.u[o] = 0.0;
        p.rhs[o] = i < p.n ? (p.xb == nullptr ? p.b[o] + x0 / p.mu : x0 / p.mu) : 0.0;   // b + (z-u)/mu (offset form: (z-u)/mu alone)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { p.status[q].iters = 0; p.status[q].converged = 0; p.status[q].nxz = 0.0; }
}

__global__ void __launch_bounds__(256)
symv_batch_kernel(AdmmBatch p) {
    const int q = blockIdx.y;
    if (p.status[q].converged) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= p.np) return;
    const double2 *m2 = reinterpret_cast<const double2 *>(p.M + ((int64_t)(q / p.nrhs) * p.np + row) * p.np);
    const double2 *r2 = reinterpret_cast<const double2 *>(p.rhs + (int64_t)q * p.np);
    double acc = 0;
    const int64_t nv = p.np / 2;
#pragma unroll 4
    for (int64_t j = lane; j < nv; j += 64) {
        const double2 m = m2[j], v = r2[j];
        acc = fma(m.x, v.x, acc);
        acc = fma(m.y, v.y, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) p.x[(int64_t)q * p.np + row] = acc;
}

// one workgroup per problem: prox_g + dual update + ||x-z|| + next rhs + that problem's convergence flag
__global__ void __launch_bounds__(256)
admm_batch_prox_kernel(AdmmBatch p) {
    __shared__ double sh[4];
    const int q = blockIdx.x;
    if (p.status[q].converged) return;
    const int64_t o = (int64_t)q * p.np, n = p.n;
    const double mu = p.mu;
    const double *__restrict__ X = p.x + o;
    const double *__restrict__ B = p.b + o;
    double *__restrict__ Z = p.z + o;
    double *__restrict__ U = p.u + o;
    double *__restrict__ R = p.rhs + o;
    double ss = 0;
    auto finish = [&](int64_t i, double xi, double ui, double zi) {
        const double d = xi - zi, un = ui + d;
        Z[i] = zi; U[i] = un;
        R[i] = B[i] + (zi - un) / mu;
        ss = fma(d, d, ss);
    };
    if (p.prox_kind == LPVS_PROX_GROUP_L2) {
        const int64_t gl = p.group_len, ng = n / gl;
        const double lm = p.prox_param * mu;
        for (int64_t g = threadIdx.x; g < ng; g += 256) {
            double s2 = 0;
            for (int64_t k = 0; k < gl; ++k) { const double v = X[g * gl + k] + U[g * gl + k]; s2 += v * v; }
            double scale = 1.0 - lm / sqrt(s2);
            if (!(scale > 0)) scale = 0.0;
            for (int64_t k = 0; k < gl; ++k) {
                const int64_t i = g * gl + k;
                const double xi = X[i], ui = U[i];
                finish(i, xi, ui, scale * (xi + ui));
            }
        }
        for (int64_t i = ng * gl + threadIdx.x; i < n; i += 256) finish(i, X[i], U[i], Z[i]);
    } else {
        const double gl1 = mu * p.prox_param, th0 = sqrt(2.0 * mu * p.prox_param);
        for (int64_t i = threadIdx.x; i < n; i += 256) {
            const double xi = X[i], ui = U[i], v = xi + ui;
            const double zi = p.prox_kind == LPVS_PROX_L1 ? v + (v <= -gl1 ? gl1 : (v >= gl1 ? -gl1 : -v)) : (fabs(v) > th0 ? v : 0.0);
            finish(i, xi, ui, zi);
        }
    }
    const double w = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nxz = sqrt(((sh[0] + sh[1]) + sh[2]) + sh[3]);
        p.status[q].iters += 1;
        p.status[q].nxz = nxz;
        if (nxz < p.tol) p.status[q].converged = 1;
    }
}

}  // namespace

int32_t launch_admm_batch_init(const AdmmBatch &p, hipStream_t s) {
    hipLaunchKernelGGL(admm_batch_init_kernel, dim3((unsigned)ceil_div(p.np, 256), (unsigned)p.nbatch), dim3(256), 0, s, p);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_pack_tiles_batch(const double *M, int64_t np, int nbatch, double *Mp, hipStream_t s) {
    const int nblk = (int)(np / TS);
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned)(nblk * (nblk + 1) / 2), (unsigned)nbatch), dim3(256), 0, s, M, np, Mp);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

bool fused_ok(const AdmmParams &p);

static void launch_mixed_batch(const AdmmBatch &p, unsigned ntiles, unsigned ns, double *part1, double *part2, const AdmmStatus *status, hipStream_t s) {
    // non-temporal loads once the inverses of the batch no longer fit the Infinity Cache (LPVS_OPT_NT_LOADS forces either)
    const int nto = option_in_effect(LPVS_OPT_NT_LOADS, p.opt_nt_loads);
    const bool nt = nto ? nto == LPVS_NT_ON : (size_t)ntiles * kSplitTileBytes * ns > ((size_t)240 << 20);
    if (nt)
        hipLaunchKernelGGL(symv_tile_mixed_batch_kernel<true>, dim3(ntiles, ns), dim3(256), 0, s, reinterpret_cast<const unsigned char *>(p.Mp), p.mp_types,
                           (size_t)ntiles * kSplitTileBytes, p.rhs, p.np, (int)ntiles, part1, part2, status);
    else
        hipLaunchKernelGGL(symv_tile_mixed_batch_kernel<false>, dim3(ntiles, ns), dim3(256), 0, s, reinterpret_cast<const unsigned char *>(p.Mp), p.mp_types,
                           (size_t)ntiles * kSplitTileBytes, p.rhs, p.np, (int)ntiles, part1, part2, status);
}

enum { FI_FIRST = 0, FI_MID = 1, FI_LAST = 2 };   // modes of admm_iter_mixed_kernel (further down)
static int32_t launch_fi_chunk(const AdmmParams &p, int64_t iters, bool batch, size_t mp_stride, bool prefetch_all, hipStream_t s);
typedef void (*FiKernel)(AdmmParams, const unsigned char *, const unsigned char *, int, int, long long, int, int, int, size_t, int);
static FiKernel fi_kernel(int mode, bool small, bool batch, bool nt, bool pa, bool f32);
static AdmmParams batch_as_params(const AdmmBatch &p) {   // AdmmParams with ns = nbatch has the layout the fused kernels expect
    AdmmParams q{p.M, p.np, p.n, p.b, p.x, p.z, p.u, p.rhs, p.mu, p.tol, p.prox_kind, p.prox_param, p.group_len, p.status,
                 p.scratch, p.part, p.Mp, p.nbatch};
    q.xb = p.xb; q.mp_split = p.mp_split; q.mp_types = p.mp_types; q.fi = p.fi; q.fi_base = p.fi_base; q.fi_prefetch_all = p.fi_prefetch_all;
    q.opt_iteration = p.opt_iteration; q.opt_nt_loads = p.opt_nt_loads;
    q.nib_period = p.nib_period; q.nib_ramp = p.nib_ramp; q.xb_corr = p.xb_corr; q.nib_acc = p.nib_acc;
    q.mp_fix32 = p.nib_period > 0 && p.nib_acc != nullptr && p.xb_corr != nullptr ? 1 : 0;
    return q;
}
bool fi_batch_applicable(const AdmmBatch &p) {
    const AdmmParams q = batch_as_params(p);
    return option_in_effect(LPVS_OPT_ITERATION, p.opt_iteration) != LPVS_ITERATION_TWO && p.fi != nullptr && p.nrhs <= 1 && p.mp_split && p.mp_types != nullptr && p.xb != nullptr && p.part != nullptr &&
           p.Mp != nullptr && fused_ok(q) && p.np <= 8192 && (int64_t)p.nbatch * p.np * 8 < ((int64_t)1 << 40);
}
int32_t launch_fi_batch_setup(const AdmmBatch &p, hipStream_t s) { return launch_fi_setup(batch_as_params(p), 0, true, s); }

int32_t launch_admm_batch_iterations(const AdmmBatch &p, int64_t iters, hipStream_t s) {
    // packed-symmetric form (half the matrix bytes per iteration) when the batch has tile-packed matrices and the
    // prox can be fused; AdmmParams with ns = nbatch has the layout the fused update kernel expects
    AdmmParams q = batch_as_params(p);
    if (iters > 0 && fi_batch_applicable(p)) {        // one launch per iteration for the whole batch
        const int nblk_ = (int)(p.np / TS);
        return launch_fi_chunk(q, iters, true, (size_t)(nblk_ * (nblk_ + 1) / 2) * kSplitTileBytes, p.fi_prefetch_all != 0, s);
    }
    if (p.Mp != nullptr && p.part != nullptr) {
        const int nblk = (int)(p.np / TS);
        const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2), ns = (unsigned)p.nbatch;
        double *part1 = p.part, *part2 = part1 + (size_t)ntiles * TS * ns;
        double *blocknorm = part2 + (size_t)ntiles * TS * ns;
        unsigned int *ticket = reinterpret_cast<unsigned int *>(blocknorm + (size_t)nblk * ns);
        const int nrhs = p.nrhs > 0 ? p.nrhs : 1;
        const bool fusable = fused_ok(q);
        for (int64_t i = 0; i < iters; ++i) {
            if (p.mp_split && p.mp_types && nrhs == 1)
                launch_mixed_batch(p, ntiles, ns, part1, part2, p.status, s);
            else if (p.mp_split)
                hipLaunchKernelGGL(symv_tile_split_batch_kernel, dim3(ntiles, ns / (unsigned)nrhs), dim3(256), 0, s, reinterpret_cast<const unsigned char *>(p.Mp),
                                   (size_t)ntiles * kSplitTileBytes, p.rhs, p.np, (int)ntiles, nrhs, part1, part2, p.status);
            else
                hipLaunchKernelGGL(symv_tile_batch_kernel, dim3(ntiles, ns / (unsigned)nrhs), dim3(256), 0, s, p.Mp, (int64_t)ntiles * TS * TS, p.rhs, p.np,
                                   (int)ntiles, nrhs, part1, part2, p.status);
            if (!fusable) {
                // IndBallL0 (the top-r selection needs the whole vector) and group lengths that do not divide 128: gather x from the tile
                // partials, then one workgroup per problem (radix select in LDS / block soft-threshold) -- the kernels of the single handles
                hipLaunchKernelGGL(symv_reduce_kernel, dim3((unsigned)nblk, ns), dim3(256), 0, s, part1, part2, nblk, (int)ntiles, p.np, p.x, p.status, p.xb, 0);
                hipLaunchKernelGGL(admm_prox_kernel, dim3(ns), dim3(1024), 0, s, q);
            } else if (nblk <= 8)
                hipLaunchKernelGGL(admm_window_update_kernel, dim3(ns), dim3((unsigned)(TS * nblk)), 0, s, q, part1, part2, nblk, (int)ntiles);
            else
                hipLaunchKernelGGL(admm_fused_update_kernel, dim3((unsigned)nblk, ns), dim3(512), 0, s, q, part1, part2, nblk, (int)ntiles, blocknorm, ticket);
        }
    } else {
        for (int64_t i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(symv_batch_kernel, dim3((unsigned)ceil_div(p.np, 4), (unsigned)p.nbatch), dim3(256), 0, s, p);
            if (p.prox_kind == LPVS_PROX_BALL_L0) hipLaunchKernelGGL(admm_prox_kernel, dim3((unsigned)p.nbatch), dim3(1024), 0, s, q);
            else hipLaunchKernelGGL(admm_batch_prox_kernel, dim3((unsigned)p.nbatch), dim3(256), 0, s, p);
        }
    }
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

// the batch mat-vec alone, `reps` times (benchmark instrumentation; iterates are not modified)
int32_t launch_admm_batch_matvec_only(const AdmmBatch &p, int reps, hipStream_t s) {
    if (p.Mp == nullptr || p.part == nullptr) { set_error("batch mat-vec timing needs the tile-packed form"); return LPVS_ESTATE; }
    const int nblk = (int)(p.np / TS);
    const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2), ns = (unsigned)p.nbatch;
    const int nrhs = p.nrhs > 0 ? p.nrhs : 1;
    double *part1 = p.part, *part2 = part1 + (size_t)ntiles * TS * ns;
    for (int i = 0; i < reps; ++i) {
        if (p.mp_split && p.mp_types && nrhs == 1)
            launch_mixed_batch(p, ntiles, ns, part1, part2, nullptr, s);
        else if (p.mp_split)
            hipLaunchKernelGGL(symv_tile_split_batch_kernel, dim3(ntiles, ns / (unsigned)nrhs), dim3(256), 0, s, reinterpret_cast<const unsigned char *>(p.Mp),
                               (size_t)ntiles * kSplitTileBytes, p.rhs, p.np, (int)ntiles, nrhs, part1, part2, (const AdmmStatus *)nullptr);
        else
            hipLaunchKernelGGL(symv_tile_batch_kernel, dim3(ntiles, ns / (unsigned)nrhs), dim3(256), 0, s, p.Mp, (int64_t)ntiles * TS * TS, p.rhs, p.np,
                               (int)ntiles, nrhs, part1, part2, (const AdmmStatus *)nullptr);
    }
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

// ---- dense (ridge) estimator on a batch of windows: out = A v per problem, matrix of problem q = A_all[q / nrhs] ----------
__global__ void __launch_bounds__(256)
batch_matvec_kernel(const double *__restrict__ A_all, int64_t np, int nrhs, const double *__restrict__ v_all, double *__restrict__ out_all) {
    const int q = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= np) return;
    const double2 *m2 = reinterpret_cast<const double2 *>(A_all + ((int64_t)(q / nrhs) * np + row) * np);
    const double2 *r2 = reinterpret_cast<const double2 *>(v_all + (int64_t)q * np);
    double acc = 0;
    for (int64_t j = lane; j < np / 2; j += 64) {
        const double2 m = m2[j], v = r2[j];
        acc = fma(m.x, v.x, acc);
        acc = fma(m.y, v.y, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) out_all[(int64_t)q * np + row] = acc;
}
__global__ void __launch_bounds__(256)
batch_ridge_residual_kernel(const double *__restrict__ b, const double *__restrict__ Gx, const double *__restrict__ x, double ridge, int64_t n,
                            int64_t np, double *__restrict__ r) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, o = (int64_t)blockIdx.y * np + i;
    if (i < np) r[o] = i < n ? b[o] - fma(ridge, x[o], Gx[o]) : 0.0;
}
__global__ void __launch_bounds__(256)
batch_vec_add_kernel(double *__restrict__ x, const double *__restrict__ d, int64_t np) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, o = (int64_t)blockIdx.y * np + i;
    if (i < np) x[o] += d[o];
}

// x = M b refined `steps` times against H = Q + ridge I, for nprob problems (problem q uses matrices q / nrhs); t1, t2 scratch
int32_t launch_batch_ridge_solve(const double *Q, const double *M, int64_t np, int64_t n, int nprob, int nrhs, const double *b, double ridge,
                                 int steps, double *x, double *t1, double *t2, hipStream_t s) {
    const dim3 gm((unsigned)ceil_div(np, 4), (unsigned)nprob), gv((unsigned)ceil_div(np, 256), (unsigned)nprob);
    hipLaunchKernelGGL(batch_matvec_kernel, gm, dim3(256), 0, s, M, np, nrhs, b, x);
    for (int k = 0; k < steps; ++k) {
        hipLaunchKernelGGL(batch_matvec_kernel, gm, dim3(256), 0, s, Q, np, nrhs, (const double *)x, t1);
        hipLaunchKernelGGL(batch_ridge_residual_kernel, gv, dim3(256), 0, s, b, (const double *)t1, (const double *)x, ridge, n, np, t2);
        hipLaunchKernelGGL(batch_matvec_kernel, gm, dim3(256), 0, s, M, np, nrhs, (const double *)t2, t1);
        hipLaunchKernelGGL(batch_vec_add_kernel, gv, dim3(256), 0, s, x, (const double *)t1, np);
    }
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_admm_init(const AdmmParams &p, hipStream_t s) {
    hipLaunchKernelGGL(admm_init_kernel, dim3((unsigned)ceil_div(p.np, 256), (unsigned)p.ns), dim3(256), 0, s, p);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_admm_restate(const AdmmParams &p, int64_t iters, hipStream_t s) {
    hipLaunchKernelGGL(admm_restate_kernel, dim3((unsigned)ceil_div(p.np, 256), (unsigned)p.ns), dim3(256), 0, s, p, (long long)iters);
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

bool admm_batch_uses_tiles(const AdmmBatch &p) {
    AdmmParams q{p.M, p.np, p.n, p.b, p.x, p.z, p.u, p.rhs, p.mu, p.tol, p.prox_kind, p.prox_param, p.group_len, p.status,
                 nullptr, p.part, p.Mp, p.nbatch};
    (void)q;
    return p.Mp != nullptr && p.part != nullptr;          // (launch_admm_batch_iterations' own test: non-fusable prox operators ride the tiles too)
}

int32_t launch_batch_matvec(const double *A, int64_t np, int nprob, int nrhs, const double *v, double *out, hipStream_t s) {
    hipLaunchKernelGGL(batch_matvec_kernel, dim3((unsigned)ceil_div(np, 4), (unsigned)nprob), dim3(256), 0, s, A, np, nrhs, v, out);
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
        hipLaunchKernelGGL(symv_reduce_kernel, dim3((unsigned)nblk, 1u), dim3(256), 0, s, abs_part, abs_part + (size_t)ntiles * TS, nblk, (int)ntiles, np, rows, nullptr, nullptr, 0);
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

bool fused_ok(const AdmmParams &p) {
    if (p.prox_kind == LPVS_PROX_L1 || p.prox_kind == LPVS_PROX_L0) return true;
    return p.prox_kind == LPVS_PROX_GROUP_L2 && p.group_len <= TS && TS % p.group_len == 0 && p.n % p.group_len == 0;
}

static void launch_split(const unsigned char *Mp, const unsigned char *types, const double *rhs, int64_t np, unsigned ntiles, double *part1,
                         double *part2, const AdmmStatus *status, hipStream_t s, int fmode = 0) {
    if (types != nullptr) hipLaunchKernelGGL(symv_tile_mixed_kernel, dim3(ntiles), dim3(256), 0, s, Mp, types, rhs, np, (int)ntiles, part1, part2, status, fmode);
    else hipLaunchKernelGGL(symv_tile_split_kernel, dim3(ntiles), dim3(256), 0, s, Mp, rhs, np, (int)ntiles, part1, part2, status);
}

// persistent grid of the wave-specialised multi-signal kernel: one workgroup per CU, evened out over the rounds so that
// every workgroup walks the same number of tiles (+-1)
static unsigned stream_cus() {
    static const unsigned slots = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        return (unsigned)cus;
    }();
    return slots;
}
static unsigned stream_grid(unsigned ntiles) {
    const unsigned slots = stream_cus(), rounds = (ntiles + slots - 1) / slots;
    return (ntiles + rounds - 1) / rounds;
}
static int stream_runs(const AdmmParams &p);

// The panel walk's plan (symv_tile_mfma_ws_kernel<.., PANEL>): the unit list (row I of panel k; kPanelC tile columns per panel) cut into G
// ranges of equal tile counts.  Device table (ints), cached per (device, nblk, G):  [0] = G;  then G rows {u0, u1, k0, I0, f0};  then
// F0[npanel + 1], the first flush index of every panel (a workgroup flushes the panel's column sums once per panel it walks in).
struct PanelPlan { const int *dev = nullptr; int G = 0, nflush = 0, nunits = 0; };
static std::mutex g_panel_mu;
static std::map<std::tuple<int, int, int>, PanelPlan> g_panel_cache;
// (lpvs_release_cached_memory: the plans' device tables are caches like the pool's blocks)
void release_panel_plans() {
    std::lock_guard<std::mutex> lk(g_panel_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &kv : g_panel_cache) { (void)hipSetDevice(std::get<0>(kv.first)); (void)hipFree(const_cast<int *>(kv.second.dev)); }
    (void)hipSetDevice(cur);
    g_panel_cache.clear();
}
static PanelPlan panel_plan(int nblk, int G) {
    std::mutex &mu = g_panel_mu;
    auto &cache = g_panel_cache;
    int device = 0;
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_tuple(device, nblk, G);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    constexpr int C = kPanelC;
    const int npanel = (nblk + C - 1) / C;
    const long long ntiles = (long long)nblk * (nblk + 1) / 2;
    struct Unit { int k, I, cnt; };
    std::vector<Unit> units;
    for (int k = 0; k < npanel; ++k)
        for (int I = k * C; I < nblk; ++I) units.push_back({k, I, std::min({C, I - k * C + 1, nblk - k * C})});
    std::vector<int> tab(1 + 5 * (size_t)G + (size_t)npanel + 1, 0);
    tab[0] = G;
    std::vector<int> flushes_of_panel((size_t)npanel, 0);
    size_t u = 0;
    long long done = 0;
    int f = 0;
    for (int g = 0; g < G; ++g) {
        const long long want = ntiles * (g + 1) / G;            // cumulative tiles after workgroup g (the last one takes the rest)
        const size_t u0 = u;
        while (u < units.size() && (g == G - 1 || done + units[u].cnt / 2 < want)) { done += units[u].cnt; ++u; }
        int *row = tab.data() + 1 + 5 * (size_t)g;
        row[0] = (int)u0; row[1] = (int)u; row[4] = f;
        if (u > u0) {
            row[2] = units[u0].k; row[3] = units[u0].I;
            for (size_t q = u0; q < u; ++q)
                if (q == u0 || units[q].k != units[q - 1].k) { ++flushes_of_panel[(size_t)units[q].k]; ++f; }
        }
    }
    // flush order = (workgroup, panel), and a workgroup's panels follow the previous workgroup's: panel k's flushes are consecutive
    int *F0 = tab.data() + 1 + 5 * (size_t)G;
    for (int k = 0; k < npanel; ++k) F0[k + 1] = F0[k] + flushes_of_panel[(size_t)k];
    PanelPlan pl;
    int *dev = nullptr;
    if (hipMalloc(&dev, tab.size() * sizeof(int)) != hipSuccess || hipMemcpy(dev, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        return pl;                                              // (not cached: the caller falls back to the run walk)
    }
    pl.dev = dev; pl.G = G; pl.nflush = f; pl.nunits = (int)units.size();
    cache[key] = pl;
    return pl;
}
static bool stream_panel(const AdmmParams &p);

template <bool SPLIT, bool Q4, bool RUNS, bool FIX, bool PANEL = false>
static void launch_mfma_stream(const AdmmParams &p, unsigned ntiles, double *part1, double *part2, const AdmmStatus *status, hipStream_t s) {
    const size_t lds = symv_ws_lds() + (PANEL ? kPanelLds : 0);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&symv_tile_mfma_ws_kernel<SPLIT, Q4, RUNS, FIX, PANEL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);   // per device; cheap
    if constexpr (PANEL) {
        const PanelPlan pl = panel_plan((int)(p.np / TS), (int)stream_cus());
        hipLaunchKernelGGL((symv_tile_mfma_ws_kernel<SPLIT, Q4, RUNS, FIX, PANEL>), dim3((unsigned)pl.G), dim3(512), lds, s,
                           reinterpret_cast<const unsigned char *>(p.Mp), p.rhs, p.np, p.ns, (int)ntiles, part1, part2, status, pl.G, FIX ? p.mp_types : nullptr, pl.dev + 1);
        return;
    }
    const int nseg = RUNS ? stream_runs(p) : 0;
    hipLaunchKernelGGL((symv_tile_mfma_ws_kernel<SPLIT, Q4, RUNS, FIX, PANEL>), dim3(RUNS ? std::min(stream_cus(), (unsigned)nseg) : stream_grid(ntiles)), dim3(512), lds, s,
                       reinterpret_cast<const unsigned char *>(p.Mp), p.rhs, p.np, p.ns, (int)ntiles, part1, part2, status, nseg, FIX ? p.mp_types : nullptr, nullptr);
}

// multi-signal handles: stream (default: persistent, register-staged MFMA kernel), dma (LDS-DMA staged MFMA kernel, 8-byte
// storage only), valu (no matrix cores)
static int multi_matvec_choice() {
    static const int multi = [] {
        const char *e = getenv("LPVS_MULTI_MATVEC");
        return !e ? 2 : (std::string(e) == "valu" ? 0 : (std::string(e) == "dma" ? 1 : 2));
    }();
    return multi;
}
// (api.hip: may a handle with several right-hand sides keep its off-diagonal tiles in the fixed-point format?  Only the stream kernel reads them.)
bool multi_signal_fixed_tiles_ok(int64_t np) { return multi_matvec_choice() == 2 && np <= 49152; }
static bool uses_stream_kernel(const AdmmParams &p) {
    return p.ns > 1 && !p.mp_f32 && p.Mp != nullptr && (p.mp_split || (multi_matvec_choice() == 2 && p.np <= 49152));   // (31-bit byte offsets into the partials: np <= 49152)
}
// The stream kernel's number of segments when it writes its P1 partials per RUN of tiles (one signal pass; every workgroup the same
// number of segments of about L tiles; room for nseg + nblk records in the per-tile record area), else 0: the consumers of the partials
// (symv_reduce_kernel, admm_fused_update2_kernel) take it as `runs_G`.  LPVS_MULTI_RUNS=L sets the segment length (default 8), 0 keeps
// one record per tile.
static int stream_runs(const AdmmParams &p) {
    const unsigned L = [] { const char *e = getenv("LPVS_MULTI_RUNS"); const int v = e ? atoi(e) : 8; return (unsigned)(v < 0 ? 0 : (v > 64 ? 64 : v)); }();   // (read per call: tests switch it)
    if (L < 2 || !uses_stream_kernel(p) || p.ns > WS_NS) return 0;
    const int nblk = (int)(p.np / TS);
    const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2), cus = stream_cus();
    if (ntiles < 2 * cus * L) return 0;              // (small matrices: a tile per visit)
    const unsigned rounds = (ntiles + cus * L - 1) / (cus * L), nseg = cus * rounds;
    return nseg + (unsigned)nblk <= ntiles ? (int)nseg : 0;
}
// The panel walk (round 5; LPVS_MULTI_WALK=panel) where the run walk applies and the four-block MFMA form does (up to 8 signals).  NOT the
// default: measured at cfg5 on one box, same process (tools/cfg5_ab.py, profiles/r05_cfg5_walks_ab.txt) the product takes 650 us against
// 605 with the run walk -- every unit of four tiles is a jump of I x 96 KB for its workgroup, where the run walk's 256 workgroups stream one
// contiguous window between them -- and the gather of the partials 40 us less (10 MB of column sums instead of 270 MB): 0.731 ms per
// iteration either way.  The consumers get runs_G = -kPanelC and the plan's table.
static bool stream_panel(const AdmmParams &p) {
    const bool off = [] { const char *e = getenv("LPVS_MULTI_WALK"); return !(e && std::string(e) == "panel"); }();   // (read per call: tests switch it)
    const bool q4_off = [] { const char *e = getenv("LPVS_MULTI_MFMA"); return e && std::string(e) == "16"; }();
    if (off || q4_off || stream_runs(p) == 0 || p.ns > 8 || !p.mp_split) return false;
    // the P1 records (one per unit) and the P2 records (ids (flush index) * kPanelC + c) live in the per-tile record areas, ntiles records per
    // signal each: a plan that would index past them (short triangles: ~45 row blocks with LPVS_MULTI_RUNS=2) takes the run walk instead
    const int nblk = (int)(p.np / TS);
    const long long ntiles = (long long)nblk * (nblk + 1) / 2;
    const PanelPlan pl = panel_plan(nblk, (int)stream_cus());
    return pl.dev != nullptr && (long long)pl.nflush * kPanelC <= ntiles && (long long)pl.nunits <= ntiles;
}
static int stream_layout(const AdmmParams &p) { return stream_panel(p) ? -kPanelC : stream_runs(p); }
static const int *stream_table(const AdmmParams &p) { return stream_panel(p) ? panel_plan((int)(p.np / TS), (int)stream_cus()).dev : nullptr; }

// the mat-vec of one iteration on the packed symmetric form (tile partials -> part1 / part2)
static void launch_sym_matvec(const AdmmParams &p, const AdmmStatus *status, hipStream_t s) {
    const int nblk = (int)(p.np / TS);
    const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2);
    const unsigned ns = (unsigned)p.ns;
    double *part1 = p.part, *part2 = part1 + (size_t)ntiles * TS * ns;
    const int multi = multi_matvec_choice();
    if (uses_stream_kernel(p)) {
        // up to 8 signals: the 4x4x4 four-block MFMA (no padded columns); LPVS_MULTI_MFMA=16 keeps the 16-column instruction
        const bool q4_off = [] { const char *e = getenv("LPVS_MULTI_MFMA"); return e && std::string(e) == "16"; }();
        const bool q4 = p.ns <= 8 && !q4_off, runs = stream_runs(p) != 0;
        if (stream_panel(p)) {                       // (6-byte / mixed tiles, up to 8 signals, a triangle large enough for runs)
            if (p.mp_types != nullptr) launch_mfma_stream<true, true, true, true, true>(p, ntiles, part1, part2, status, s);
            else launch_mfma_stream<true, true, true, false, true>(p, ntiles, part1, part2, status, s);
            return;
        }
        auto go = [&](auto split, auto q4c, auto runsc) {
            if constexpr (decltype(split)::value) {
                if (p.mp_types != nullptr) { launch_mfma_stream<true, decltype(q4c)::value, decltype(runsc)::value, true>(p, ntiles, part1, part2, status, s); return; }
            }
            launch_mfma_stream<decltype(split)::value, decltype(q4c)::value, decltype(runsc)::value, false>(p, ntiles, part1, part2, status, s);
        };
        using T = std::true_type; using F = std::false_type;
        if (p.mp_split) { if (q4) { if (runs) go(T{}, T{}, T{}); else go(T{}, T{}, F{}); } else { if (runs) go(T{}, F{}, T{}); else go(T{}, F{}, F{}); } }
        else            { if (q4) { if (runs) go(F{}, T{}, T{}); else go(F{}, T{}, F{}); } else { if (runs) go(F{}, F{}, T{}); else go(F{}, F{}, F{}); } }
    } else if (p.ns > 8 && !p.mp_f32 && multi == 1) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&symv_tile_mfma_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)symv_mfma_lds<16>());   // per device; cheap
        hipLaunchKernelGGL(symv_tile_mfma_kernel<16>, dim3(ntiles), dim3(256), symv_mfma_lds<16>(), s, p.Mp, p.rhs, p.np, p.ns, (int)ntiles, part1, part2, status);
    } else if (p.ns > 1 && !p.mp_f32 && multi == 1) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&symv_tile_mfma_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)symv_mfma_lds<8>());
        hipLaunchKernelGGL(symv_tile_mfma_kernel<8>, dim3(ntiles), dim3(256), symv_mfma_lds<8>(), s, p.Mp, p.rhs, p.np, p.ns, (int)ntiles, part1, part2, status);
    } else if (p.ns > 1 && !p.mp_f32)
        hipLaunchKernelGGL((symv_tile_multi_kernel<double, 8>), dim3(ntiles), dim3(256), 0, s, p.Mp, p.rhs, p.np, p.ns, (int)ntiles, part1, part2, status);
    else if (p.mp_f32 && p.ns == 1)
        hipLaunchKernelGGL(symv_tile_f32_kernel, dim3(ntiles), dim3(256), 0, s, reinterpret_cast<const float *>(p.Mp), p.rhs, p.np, (int)ntiles, part1,
                           part2, status);
    else if (p.mp_f32)
        hipLaunchKernelGGL(symv_tile_kernel<float>, dim3(ntiles), dim3(256), 0, s, reinterpret_cast<const float *>(p.Mp), p.rhs, p.np, p.ns,
                           (int)ntiles, part1, part2, status);
    else if (p.mp_split)
        launch_split(reinterpret_cast<const unsigned char *>(p.Mp), p.mp_types, p.rhs, p.np, ntiles, part1, part2, status, s, p.mp_fix32);
    else
        hipLaunchKernelGGL(symv_tile_kernel<double>, dim3(ntiles), dim3(256), 0, s, p.Mp, p.rhs, p.np, p.ns, (int)ntiles, part1, part2, status);
}

// one ADMM iteration on the packed symmetric form
static int32_t launch_iteration_sym(const AdmmParams &p, hipStream_t s, int it) {
    const int nblk = (int)(p.np / TS);
    const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2);
    const unsigned ns = (unsigned)p.ns;
    double *part1 = p.part, *part2 = part1 + (size_t)ntiles * TS * ns;
    double *blocknorm = part2 + (size_t)ntiles * TS * ns;
    launch_sym_matvec(p, p.status, s);
    // (the stale nibble product: refresh R_g sits between the product of rhs_g and the update that adds xb to it -- where the one-launch scheme has it)
    if (p.nib_period > 0 && p.ns == 1 && p.mp_types != nullptr) {
        const long long g = p.fi_base + it;
        if (nib_refresh_due(g, p.nib_period, p.nib_ramp)) LPVS_TRY(launch_nibble_refresh(p, false, nullptr, s));   // (LPVS_ESTATE without its buffers: not a stale offset vector)
    }
    if (fused_ok(p)) {
        hipLaunchKernelGGL(admm_fused_update2_kernel, dim3((unsigned)nblk, ns), dim3(512), 0, s, p, part1, part2, nblk, (int)ntiles, blocknorm, it & 1, it > 0 ? 1 : 0, stream_layout(p), stream_table(p));
    } else {
        if (const int runs = stream_layout(p))
            hipLaunchKernelGGL(symv_reduce_runs_kernel, dim3((unsigned)nblk, ns), dim3(1024), 0, s, part1, part2, nblk, (int)ntiles, p.np, p.x, p.status, p.xb, runs, stream_table(p));
        else
            hipLaunchKernelGGL(symv_reduce_kernel, dim3((unsigned)nblk, ns), dim3(256), 0, s, part1, part2, nblk, (int)ntiles, p.np, p.x, p.status, p.xb, 0);
        hipLaunchKernelGGL(admm_prox_kernel, dim3(ns), dim3(1024), 0, s, p);
    }
    return LPVS_OK;
}

static void launch_symv_raw(const double *M, int64_t np, const double *rhs, double *x, const AdmmStatus *st, int ns,
                            hipStream_t s) {
    if (np >= 4096)
        hipLaunchKernelGGL(symv_kernel<4>, dim3((unsigned)ceil_div(np, 16), (unsigned)ns), dim3(256), 0, s, M, np, rhs, x, st);
    else
        hipLaunchKernelGGL(symv_kernel<1>, dim3((unsigned)ceil_div(np, 4), (unsigned)ns), dim3(256), 0, s, M, np, rhs, x, st);
}

// out[r] = sum_c A[r*ld + c] v[c], r < rows, c < cols (cols % 2 == 0, 16-B aligned rows): one wave per row
__global__ void __launch_bounds__(256)
rect_matvec_kernel(const double *__restrict__ A, int64_t rows, int64_t cols, int64_t ld, const double *__restrict__ v,
                   double *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const double2 *a2 = reinterpret_cast<const double2 *>(A + r * ld);
    const double2 *v2 = reinterpret_cast<const double2 *>(v);
    double acc = 0;
    for (int64_t j = lane; j < cols / 2; j += 64) {
        const double2 m = a2[j], w = v2[j];
        acc = fma(m.x, w.x, acc);
        acc = fma(m.y, w.y, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[r] = acc;
}

int32_t launch_rect_matvec(const double *A, int64_t rows, int64_t cols, int64_t ld, const double *v, double *out, hipStream_t s) {
    hipLaunchKernelGGL(rect_matvec_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, A, rows, cols, ld, v, out);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

// r = b - (G x + ridge x)  (Gx supplied) on the first n entries, 0 on the pad;   x += d
__global__ void __launch_bounds__(256)
ridge_residual_kernel(const double *__restrict__ b, const double *__restrict__ Gx, const double *__restrict__ x, double ridge, int64_t n,
                      int64_t np, double *__restrict__ r) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < np) r[i] = i < n ? b[i] - fma(ridge, x[i], Gx[i]) : 0.0;
}
__global__ void __launch_bounds__(256)
vec_add_kernel(double *__restrict__ x, const double *__restrict__ d, int64_t np) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < np) x[i] += d[i];
}

// x = M b followed by `steps` rounds of iterative refinement against H = G + ridge I (M is H^-1 up to the sweep's rounding;
// the normal equations square cond(A), so the explicit inverse alone loses about cond(H) * eps):  x += M (b - H x);
// on return t2 holds the final residual b - H x
int32_t launch_ridge_solve_refined(const double *G, const double *M, int64_t np, int64_t n, const double *b, double ridge, int steps,
                                   double *x, double *t1, double *t2, hipStream_t s) {
    launch_symv_raw(M, np, b, x, nullptr, 1, s);
    const unsigned nb = (unsigned)ceil_div(np, 256);
    for (int k = 0; k < steps; ++k) {
        launch_symv_raw(G, np, x, t1, nullptr, 1, s);                                     // t1 = G x
        hipLaunchKernelGGL(ridge_residual_kernel, dim3(nb), dim3(256), 0, s, b, t1, x, ridge, n, np, t2);   // t2 = b - H x
        launch_symv_raw(M, np, t2, t1, nullptr, 1, s);                                    // t1 = M r
        hipLaunchKernelGGL(vec_add_kernel, dim3(nb), dim3(256), 0, s, x, t1, np);
    }
    launch_symv_raw(G, np, x, t1, nullptr, 1, s);                                         // final residual left in t2
    hipLaunchKernelGGL(ridge_residual_kernel, dim3(nb), dim3(256), 0, s, b, t1, x, ridge, n, np, t2);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_symv(const double *M, int64_t np, const double *rhs, double *x, hipStream_t s, int ns) {
    launch_symv_raw(M, np, rhs, x, nullptr, ns, s);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

// ---- accurate accumulation for the two places that need it: the offset vector's residual and the x-update's correction ---------
// Accumulation as accurate as if carried with twice the mantissa and rounded once (Ogita, Rump, Oishi: Dot2): error-free products
// (fma), error-free sums (Knuth's two-sum), the error terms added up in a second double.  -ffp-contract=off (Makefile) and the _rn
// intrinsics keep the compiler from fusing or reassociating any of it.
struct dot2_t { double s, c; };
__device__ __forceinline__ void dot2_add(dot2_t &a, double p, double e) {      // a += p + e, p the leading term
    const double s = __dadd_rn(a.s, p), bb = __dadd_rn(s, -a.s);
    const double err = __dadd_rn(__dadd_rn(a.s, -__dadd_rn(s, -bb)), __dadd_rn(p, -bb));
    a.s = s;
    a.c = __dadd_rn(a.c, __dadd_rn(e, err));
}
__device__ __forceinline__ void dot2_fma(dot2_t &a, double x, double y) {      // a += x * y
    const double p = __dmul_rn(x, y);
    dot2_add(a, p, __fma_rn(x, y, -p));
}

// r[sg][i] = bsign b[sg][i] - sum_j A[i][j] x[sg][j] - shift x[sg][i] in that arithmetic, rounded once; 0 on the pad (b may be NULL: 0).
// One wave per row (lane covers columns lane + 64 t), four rows per workgroup; the row is read ONCE for up to 8 signals (blockIdx.y =
// group of 8 signals): the matrix of a multi-signal handle is 8 GiB.
constexpr int kDdSignals = 8;
__global__ void __launch_bounds__(256)
shifted_residual_dd_kernel(const double *__restrict__ A, int64_t np, int64_t n, int ns, const double *__restrict__ b_all, double bsign,
                           const double *__restrict__ x_all, double shift, double *__restrict__ r_all) {
    const int sg0 = blockIdx.y * kDdSignals, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nsg = ns - sg0 < kDdSignals ? ns - sg0 : kDdSignals;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= np) return;
    const double *a = A + row * np;
    dot2_t acc[kDdSignals];
#pragma unroll
    for (int q = 0; q < kDdSignals; ++q) acc[q] = {0.0, 0.0};
    if (row < n)
        for (int64_t j = lane; j < np; j += 64) {
            const double m = -a[j];
#pragma unroll
            for (int q = 0; q < kDdSignals; ++q)
                if (q < nsg) dot2_fma(acc[q], m, x_all[(int64_t)(sg0 + q) * np + j]);
        }
#pragma unroll
    for (int q = 0; q < kDdSignals; ++q) {
        if (q >= nsg) break;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double os = __shfl_xor(acc[q].s, o, 64), oc = __shfl_xor(acc[q].c, o, 64);
            dot2_add(acc[q], os, oc);
        }
        if (lane == 0) {
            double r = 0.0;
            if (row < n) {
                const int64_t gi = (int64_t)(sg0 + q) * np + row;
                dot2_fma(acc[q], -shift, x_all[gi]);
                if (b_all != nullptr) dot2_add(acc[q], bsign * b_all[gi], 0.0);
                r = __dadd_rn(acc[q].s, acc[q].c);
            }
            r_all[(int64_t)(sg0 + q) * np + row] = r;
        }
    }
}
static void launch_residual_dd(const double *A, int64_t np, int64_t n, int ns, const double *b, double bsign, const double *x, double shift, double *r,
                               hipStream_t s) {
    hipLaunchKernelGGL(shifted_residual_dd_kernel, dim3((unsigned)ceil_div(np, 4), (unsigned)ceil_div(ns, kDdSignals)), dim3(256), 0, s, A, np, n, ns, b, bsign, x, shift, r);
}

// ---- the offset vector xb = (G + shift I)^-1 b of the x-update's offset form: xb = M b, then `steps` rounds  xb += M (b - (G + shift I) xb)
// with the residual accumulated as above -- the forward error of the explicit inverse (|M H - I| ~ 2e-13 at n = 8192) leaves xb, whatever
// it multiplies in the iteration.  G, b are the problem's data, exact as given.  t1, t2: [ns][np] scratch.
int32_t launch_offset_vector_refined(const double *G, const double *M, int64_t np, int64_t n, int ns, const double *b, double shift, int steps,
                                     double *xb, double *t1, double *t2, hipStream_t s) {
    launch_symv_raw(M, np, b, xb, nullptr, ns, s);
    const int64_t total = np * (int64_t)ns;
    const unsigned nb = (unsigned)ceil_div(total, 256);
    for (int k = 0; k < steps; ++k) {
        launch_residual_dd(G, np, n, ns, b, 1.0, xb, shift, t2, s);                                              // t2 = b - H xb
        launch_symv_raw(M, np, t2, t1, nullptr, ns, s);                                                          // t1 = M r
        hipLaunchKernelGGL(vec_add_kernel, dim3(nb), dim3(256), 0, s, xb, t1, total);
    }
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

// ---- the x-update's systematic error, removed (round 5).
// The iteration applies M~ = (I + E) H^-1 -- the explicit inverse with its forward error, |E| = |M H - I| ~ 2e-13 elementwise at cfg3,
// and the 2^-40 rounding of the packed copy on top -- instead of H^-1:  x_{k+1} = xb + M~ v_k = H^-1 (b + v_k) + E w_{k+1},  w = M~ v.
// E w is the SAME vector iteration after iteration once the iterates move slowly, i.e. a constant forcing of the map, and at cfg3 the
// map's slowest mode amplifies it ~10^3 times on the way to the fixed point: every f64 evaluation of the iteration -- this library's
// with any storage of M and either launch scheme, and a CPU restatement's Cholesky solves alike -- sits 0.3 .. 1.3e-9 from the
// extended-precision iterates after 2000 iterations, all along one direction (profiles/r05_cfg3_error_directions.txt), where one
// ulp of INPUT uncertainty moves the answer by 1e-10.
// The cure is one step of iterative refinement -- per CORRECTION, not per iteration: with v the right-hand side the next x-update is
// about to multiply,   w = M~ v,   r = v - H w  (accumulated in twice the mantissa: formed in doubles it would drown in its own
// rounding, eps cond(H)),   d = M~ r,   and the offset vector becomes  xb_eff = xb + d  = xb - E w.  Between corrections the error
// is E (w_k - w), second order; the fixed point of the corrected map is the exact one whatever M~ is (it only preconditions), so the
// packed copy's rounding is corrected along with the inverse's.  Both M~ products go through the handle's own packed mat-vec.
// t: 3 x [ns][np] scratch.
__global__ void __launch_bounds__(256)
vec_sum_kernel(const double *a, const double *b, double *out, int64_t total) {   // (out may be one of the inputs)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = a[i] + b[i];
}
// out = M~ rhs through the packed tiles (the handle's stand-alone mat-vec + the gather of its partials), no offset, no status
// ---- the stale nibble product (round 5) --------------------------------------------------------------------------------------------
// A handle whose x-update is corrected may stream only the 32 leading bits of its fixed-point tiles (AdmmParams::mp_fix32: 4 B per element
// instead of 4.5, -11 % of the bytes of an iteration at cfg3).  The correction removes the systematic part of what that leaves out from x and
// z -- but the DUAL variable integrates the rest over the 500+ iterations between two corrections (u: 2e-9 .. 7e-9 from the exact iterates
// where the 36-bit tiles give 1e-10 .. 5e-10).  So the part left out, N = nibble x step of every element, is multiplied into the right-hand side
// every nib_period iterations by a pass over the nibble planes alone (16 MB at cfg3, not 157) and carried in the offset vector:
//     xb = xb_corr + N rhs_g          after the launches g = 1 and g = 0 (mod nib_period)
// The error left in x is N (rhs_k - rhs_g), k - g < nib_period -- it telescopes over a run instead of integrating.  Absolute launch indices:
// the iterates do not depend on the chunking of lpvs_admm_run.
__global__ void __launch_bounds__(256)
rhs_from_state_kernel(const double *__restrict__ z, const double *__restrict__ u, double mu, int64_t n, int64_t np, double *__restrict__ rhs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < np) rhs[i] = i < n ? (z[i] - u[i]) / mu : 0.0;            // (the very expression the update kernels write: bit-identical to p.rhs at a chunk's end)
}
// split = true (after a correction wrote xb = xb0 + d for the right-hand side in memory): xb_corr = xb - N rhs instead, xb stays
__global__ void __launch_bounds__(256)
vec_diff_kernel(const double *a, const double *b, double *out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = a[i] - b[i];
}
int32_t launch_nibble_refresh(const AdmmParams &p, bool from_state, const double *u_src, hipStream_t s, bool split) {
    if (p.ns != 1 || p.mp_types == nullptr || p.xb_corr == nullptr || p.nib_rhs == nullptr || p.nib_part == nullptr) { set_error("nibble refresh: not a single-signal mixed-storage handle"); return LPVS_ESTATE; }
    const int nblk = (int)(p.np / TS);
    const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2);
    double *part1 = p.nib_part, *part2 = part1 + (size_t)ntiles * TS;
    const double *rhs = p.rhs;
    if (from_state) {
        hipLaunchKernelGGL(rhs_from_state_kernel, dim3((unsigned)ceil_div(p.np, 256)), dim3(256), 0, s, p.z, u_src, p.mu, p.n, p.np, p.nib_rhs);
        rhs = p.nib_rhs;
    }
    launch_split(reinterpret_cast<const unsigned char *>(p.Mp), p.mp_types, rhs, p.np, ntiles, part1, part2, nullptr, s, /*fmode=*/2);
    if (split) {
        hipLaunchKernelGGL(symv_reduce_kernel, dim3((unsigned)nblk, 1u), dim3(256), 0, s, part1, part2, nblk, (int)ntiles, p.np, p.nib_rhs, nullptr, (const double *)nullptr, 0);   // N rhs
        hipLaunchKernelGGL(vec_diff_kernel, dim3((unsigned)ceil_div(p.np, 256)), dim3(256), 0, s, p.xb, (const double *)p.nib_rhs, const_cast<double *>(p.xb_corr), p.np);
    } else
        hipLaunchKernelGGL(symv_reduce_kernel, dim3((unsigned)nblk, 1u), dim3(256), 0, s, part1, part2, nblk, (int)ntiles, p.np, const_cast<double *>(p.xb), nullptr, p.xb_corr, 0);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

static void launch_packed_apply(const AdmmParams &p, const double *rhs, double *out, hipStream_t s) {
    AdmmParams q = p;
    q.rhs = const_cast<double *>(rhs);
    launch_sym_matvec(q, nullptr, s);
    const int nblk = (int)(p.np / TS);
    const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2), ns = (unsigned)p.ns;
    double *part1 = p.part, *part2 = part1 + (size_t)ntiles * TS * ns;
    if (const int runs = stream_layout(p))
        hipLaunchKernelGGL(symv_reduce_runs_kernel, dim3((unsigned)nblk, ns), dim3(1024), 0, s, part1, part2, nblk, (int)ntiles, p.np, out, nullptr, nullptr, runs, stream_table(p));
    else
        hipLaunchKernelGGL(symv_reduce_kernel, dim3((unsigned)nblk, ns), dim3(256), 0, s, part1, part2, nblk, (int)ntiles, p.np, out, nullptr, nullptr, 0);
}
// b != nullptr: the refinement step is taken for the WHOLE right-hand side b + v -- x = xb0 + M~ v is what the next x-update would produce,
// r = b + v - H x -- so that the error of xb0 = M b itself is corrected along with E w (handles whose offset vector was not refined at
// lpvs_admm_init: one accurate product per correction instead of one more per solve).
int32_t launch_xupdate_correction(const AdmmParams &p, const double *G, double shift, const double *b, const double *xb0, double *xb_eff, double *t, hipStream_t s) {
    if (p.part == nullptr || p.Mp == nullptr) { set_error("the x-update correction needs the packed inverse"); return LPVS_ESTATE; }
    const int64_t total = p.np * (int64_t)p.ns;
    const unsigned nb = (unsigned)ceil_div(total, 256);
    double *w = t, *r = t + total, *d = t + 2 * total;
    launch_packed_apply(p, p.rhs, w, s);                                             // w = M~ v
    if (b != nullptr) {
        hipLaunchKernelGGL(vec_sum_kernel, dim3(nb), dim3(256), 0, s, xb0, (const double *)w, w, total);          // w = xb0 + M~ v = the next x
        hipLaunchKernelGGL(vec_sum_kernel, dim3(nb), dim3(256), 0, s, b, (const double *)p.rhs, d, total);        // d = b + v (scratch)
        launch_residual_dd(G, p.np, p.n, p.ns, d, 1.0, w, shift, r, s);              // r = b + v - H x
    } else
        launch_residual_dd(G, p.np, p.n, p.ns, p.rhs, 1.0, w, shift, r, s);          // r = v - H w
    launch_packed_apply(p, r, d, s);                                                 // d = M~ r
    hipLaunchKernelGGL(vec_sum_kernel, dim3(nb), dim3(256), 0, s, xb0, (const double *)d, xb_eff, total);   // xb_eff = xb0 + d
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

// =====================================================================================================================
// ONE launch per ADMM iteration (single-signal handles with the mixed storage, offset form, fusable prox).
//
// The two-launch iteration spends 6.9 of 31.7 us per iteration at cfg3 on the end of the mat-vec launch and on the update kernel
// (two memory round trips: it gathers 64 tile partials per row).  Here the tile workgroups do not store partials: they ADD them into
// the next x with 64-bit FIXED-POINT global atomics -- integer addition is associative, so the sum does not depend on the order the
// memory side serves the adds in (the device nufft.hip uses for its grids) -- and the NEXT launch's tile workgroups rebuild their two
// right-hand-side blocks from x in their prologue (prox and dual update of 256 elements: a few instructions, redundantly per tile,
// while the tile's own bytes are in flight).  The workgroup of a diagonal tile also writes the block's x, z, u, its ||x-z||^2 and the
// maxima the next quantum needs, and zeroes the block of the accumulator after next.  Launch boundaries are the only synchronisation.
//
//   launch with index g (the right-hand side rhs_g it multiplies):   update u_{g-1} in the prologue (x_{g-1} = xb + q_{g-1} * acc,
//   z, u, rhs_g), then acc' += round(M~ rhs_g / q_g).   The first launch of a chunk takes rhs from memory (no update); the chunk's last
//   update is a launch of the same kernel without the tile part (one workgroup per row block), which also leaves rhs in memory.
//
// The quantum needs a bound on |M~ rhs_g| BEFORE the launch: every prox of the fused set shrinks, |2z - v| <= |v|, so
// |rhs_g| <= |x_{g-1} + u_{g-2}| / mu <= (max|xb| + R max|rhs_{g-1}| + max|u_{g-2}|) / mu =: V_g with R the largest absolute row sum of
// M~ -- all three maxima are left behind by the update two launches back -- and |M~ rhs_g| <= R V_g.  q_g = 2^(e-62) with R V_g < 2^e:
// no overflow for any input, and at cfg3 q is still ~2^-55 of |x| (the bound is loose by 2^6..2^7; an int64 has ten bits more than a
// double's mantissa).  Everything is deterministic: the quantum is computed identically by every workgroup from values written by an
// earlier launch, and no workgroup reads what another workgroup of the same launch writes (u is double-buffered, the accumulators
// rotate through three buffers, per-parity slots hold block norms, maxima and quanta).
// =====================================================================================================================
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// max over the wave of a NON-NEGATIVE double, the same value in every lane: row scans and row broadcasts through DPP (a dozen
// cycles per step) instead of six ds_bpermute round trips -- this sits between the arrival of the state and the first product of every
// tile workgroup.  Zero is the identity: lanes without a source read 0 (bound_ctrl), disabled rows keep 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take0(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_nonneg(double v) {
    v = fmax(v, dpp_take0<0x111, 0xf>(v));           // row_shr:1   lane i: max over [i-1, i] of its row of 16
    v = fmax(v, dpp_take0<0x112, 0xf>(v));           // row_shr:2            [i-3, i]
    v = fmax(v, dpp_take0<0x114, 0xf>(v));           // row_shr:4            [i-7, i]
    v = fmax(v, dpp_take0<0x118, 0xf>(v));           // row_shr:8            [i-15, i]: lane 15 of a row holds the row's maximum
    v = fmax(v, dpp_take0<0x142, 0xa>(v));           // row_bcast:15 into rows 1 and 3
    v = fmax(v, dpp_take0<0x143, 0xc>(v));           // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's maximum
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

struct FiBufs {   // views into AdmmParams::fi (8-byte units): see fi_doubles()
    long long *acc0; int64_t np;
    double *ualt, *bn, *qbuf, *consts;
    double2 *rec;                                    // per parity and row block: {max|rhs|, max|u|} left by the block's last update
    __host__ __device__ __forceinline__ long long *acc(int slot) const { return acc0 + (int64_t)slot * np; }   // (no array: a dynamically indexed one lives in scratch)
};
__host__ __device__ __forceinline__ FiBufs fi_views(double *fi, int64_t np /* all problems' rows */, int nblk /* all problems' row blocks */, int nprob = 1) {
    FiBufs f;
    f.acc0 = reinterpret_cast<long long *>(fi); f.np = np;
    f.ualt = fi + 3 * np;
    f.bn = fi + 4 * np; f.rec = reinterpret_cast<double2 *>(f.bn + 2 * nblk);   // (16-byte aligned: np is a multiple of 128)
    f.qbuf = reinterpret_cast<double *>(f.rec + 2 * nblk); f.consts = f.qbuf + 2 * nprob;
    return f;
}
// Worst relative excess of a row sum of |M~| (the packed copy the product streams) over the same row sum of |M|:
//   single-precision copy (_f32 handles): 2^-24 per element;  float head + 16-bit tail: 2^-40 per element;
//   36-bit fixed point: <= step/2 per element with step <= 2^-44 max|M| sqrt(8192/np) (pack_tiles_mixed_kernel's admission), so
//   <= np * step/2 = 2^-45 sqrt(8192 np) max|M| <= 2^-30 max|M| <= 2^-30 R for np <= 49152 (fi_applicable); the clamp to +-(2^35 - 1)
//   only shrinks.  The bound multiplies two row sums, so (1 + slack)^2 must stay under the 1.000001 the kernel uses.
constexpr double kFiPackedRowSlack = 0x1p-24 + 0x1p-30;
static_assert((1.0 + kFiPackedRowSlack) * (1.0 + kFiPackedRowSlack) * (1.0 + 0x1p-40) < 1.000001,
              "the quantum bound of admm_iter_mixed_kernel no longer covers the rounding of the packed inverse: raise its 1.000001");
size_t fi_doubles(int64_t np, int64_t nprob) { return (size_t)((4 * np + 6 * (np / TS) + 4) * nprob + 2); }
bool fi_applicable(const AdmmParams &p) {
    const bool on = option_in_effect(LPVS_OPT_ITERATION, p.opt_iteration) != LPVS_ITERATION_TWO;   // (resolved per call: tests and tools switch it between handles)
    return on && p.fi != nullptr && p.ns == 1 && (p.mp_types != nullptr || p.mp_f32) && p.xb != nullptr && p.part != nullptr && p.Mp != nullptr && fused_ok(p) &&
           p.np <= 49152;                            // (six clamped loads per lane cover the block norms / maxima of 384 row blocks)
}

// R = max_i sum_j |M_ij| (one wave per row) -> consts[2 sg] as the bit pattern of a non-negative double (integer max: order-independent);
// blockIdx.y = problem of a batch (matrices np x np apart)
__global__ void __launch_bounds__(256)
fi_rowsum_kernel(const double *__restrict__ M, int64_t np, int64_t n, unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= n) return;                              // (valid rows only: a pad row holds a 1 on its diagonal and multiplies a zero -- counting
                                                     //  it would loosen the bound, and with it the quantum, by 1 / mu)
    const double2 *m2 = reinterpret_cast<const double2 *>(M + ((int64_t)blockIdx.y * np + r) * np);
    double acc = 0;
    for (int64_t j = lane; j < np / 2; j += 64) { const double2 m = m2[j]; acc += fabs(m.x) + fabs(m.y); }
    acc = wave_sum(acc);
    if (lane == 0) atomicMax(out + 2 * blockIdx.y, (unsigned long long)__double_as_longlong(acc));
}
// max|xb| -> consts[2 sg + 1]; the records an update two / one launches before iteration `base` would have left, from rhs and u in memory:
//   rec[(base-1)&1] = {max|rhs|, max|u|};  rec[(base-2)&1] = {0, mu max|rhs|}  (so that V_base = (max|xb| + mu max|rhs|) / mu >= max|rhs|).
// One workgroup per problem (blockIdx.x).
__global__ void __launch_bounds__(256)
fi_state_kernel(AdmmParams p, int nblk, long long base, int with_consts) {
    const int sg = blockIdx.x, nprob = gridDim.x;
    const FiBufs f = fi_views(p.fi, (int64_t)nprob * p.np, nprob * nblk, nprob);
    const int64_t voff = (int64_t)sg * p.np;
    __shared__ double sh[3][4];
    double mx = 0, mr = 0, mu_ = 0;
    int bad = 0;
    for (int64_t e = threadIdx.x; e < p.np; e += 256) {
        const double a = p.xb[voff + e], b = p.rhs[voff + e], c = p.u[voff + e];
        mx = fmax(mx, fabs(a)); mr = fmax(mr, fabs(b)); mu_ = fmax(mu_, fabs(c));
        bad |= !(fabs(b) < 0x1p1000) || !(fabs(c) < 0x1p1000);        // NaN or Inf
    }
    mx = wave_max(mx); mr = wave_max(mr); mu_ = wave_max(mu_);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = mx; sh[1][threadIdx.x >> 6] = mr; sh[2][threadIdx.x >> 6] = mu_; }
    __syncthreads();
    mx = fmax(fmax(sh[0][0], sh[0][1]), fmax(sh[0][2], sh[0][3]));
    mr = fmax(fmax(sh[1][0], sh[1][1]), fmax(sh[1][2], sh[1][3]));
    mu_ = fmax(fmax(sh[2][0], sh[2][1]), fmax(sh[2][2], sh[2][3]));
    // (fmax drops NaNs: a NaN in the state handed over -- x0, a restored u -- is marked by an infinite maximum instead, which the
    // iteration kernel's bound turns into NaN iterates; see there)
    if (__syncthreads_or(bad)) { mr = __longlong_as_double(0x7ff0000000000000ll); mu_ = mr; }
    const int p1 = (int)((base + 1) & 1), p2 = (int)(base & 1);      // parities of base - 1 and base - 2
    const int nbt = nprob * nblk, boff = sg * nblk;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        f.rec[p1 * nbt + boff + b] = make_double2(mr, mu_);
        f.rec[p2 * nbt + boff + b] = make_double2(0.0, p.mu * mr);
        f.bn[boff + b] = 0.0; f.bn[nbt + boff + b] = 0.0;
    }
    if (threadIdx.x == 0) {
        if (with_consts) f.consts[2 * sg + 1] = mx;
        f.qbuf[sg] = 0.0; f.qbuf[nprob + sg] = 0.0;
    }
}

// v[rg] = the lane's partial row sums of its 8 row groups, tc[k] = its partial column sums of its 8 columns (fix_tile_product without
// its tail): row group by row group, as the bytes arrive
__device__ __forceinline__ void fi_fixed_product(const FixRaw &fr, const double *sI, const double *sJ, int wave, int lane, double (&v)[8], double (&tc)[8]) {
    const int c = lane & 15, gq = lane >> 4;
    double rj[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) tc[k] = 0.0;
    const float stv[8] = {fr.st[0].x, fr.st[0].y, fr.st[0].z, fr.st[0].w, fr.st[1].x, fr.st[1].y, fr.st[1].z, fr.st[1].w};
    const unsigned int nw[8] = {fr.nq[0].x, fr.nq[0].y, fr.nq[0].z, fr.nq[0].w, fr.nq[1].x, fr.nq[1].y, fr.nq[1].z, fr.nq[1].w};
    double ri = (double)stv[0] * sI[wave * 32 + gq];
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        const double step = (double)stv[rg];
        const double ri_next = rg + 1 < 8 ? (double)stv[rg + 1] * sI[wave * 32 + 4 * (rg + 1) + gq] : 0.0;
        const int hh[8] = {fr.ha[rg].x, fr.ha[rg].y, fr.ha[rg].z, fr.ha[rg].w, fr.hb[rg].x, fr.hb[rg].y, fr.hb[rg].z, fr.hb[rg].w};
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const double m0 = fix_decode((unsigned int)hh[k], (nw[rg] >> (4 * k)) & 15u);
            const double m1 = fix_decode((unsigned int)hh[k + 1], (nw[rg] >> (4 * k + 4)) & 15u);
            tc[k] = opaque(fma(m0, ri, tc[k]));
            tc[k + 1] = opaque(fma(m1, ri, tc[k + 1]));
            a0 = fma(m0, rj[k], a0);
            a1 = fma(m1, rj[k + 1], a1);
        }
        v[rg] = step * (a0 + a1);
        ri = ri_next;
        __builtin_amdgcn_sched_barrier(0);           // (no hoisting of later groups' decodes: they would wait for later bytes)
    }
}


#if defined(LPVS_TIMELINE) && LPVS_TIMELINE < 3
// Debug build only (make timeline -> liblpvspectral_timeline.so; tools/iter_timeline.py): every workgroup of the single-problem one-launch
// iteration leaves wall-clock stamps (s_memrealtime, 100 MHz) of its phases, 8 words per workgroup and launch parity:
//   {g, entry, update done, prologue barrier passed, tile consumed, last atomic issued, XCC_ID, HW_ID}
__device__ unsigned long long *g_lpvs_tl = nullptr;
extern "C" int32_t lpvs_debug_set_timeline(unsigned long long *dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lpvs_tl), &dev_buf, sizeof(dev_buf)) == hipSuccess ? LPVS_OK : LPVS_EDEVICE;
}
// LPVS_TIMELINE=1: entry and end only (the stamps between them make the kernel wait for its scalar loads in the middle of the
// overlapped load / compute sequence: 39 us instead of 27); =2: all five
#define LPVS_TL_STAMP(k) do { if ((LPVS_TIMELINE >= 2 || (k) == 5) && tl_on && threadIdx.x == 0) tl_rec[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LPVS_TL_STAMP(k) do { } while (0)
#endif

// g: FIRST / MID -- index of the right-hand side this launch multiplies (the update it performs is u_{g-1});  LAST -- g - 1 is the
// update it performs (it multiplies nothing).  aslot: accumulator this launch adds into (FIRST / MID) resp. would have (LAST).
// BATCH: blockIdx.y = problem of a batch that each own their matrix (the windows of ls_windowpsd; p.ns = problems, vectors [ns][np],
// mp_stride = bytes between their packed matrices); NT: non-temporal tile loads (batches beyond the Infinity Cache).
// prefetch_all: every tile of the batch is in the fixed format, so the diagonal tiles (and their double diagonals) are requested up
// front like the others (cfg4: four of a window's ten tiles); otherwise diagonal tiles are loaded after the prologue (cfg3: float-head).
// F32: the packed inverse is the plain single-precision copy of the _f32 handles (64 KB tiles, every tile requested up front; PA ignored).
// NIBR (single problems that iterate on 32-bit reads, launches after which the stale nibble product is due): the launch ALSO multiplies the
// 4-bit planes of its fixed-point tile into the same right-hand side and adds those sums, as integers of the launch's quantum, into
// p.nib_acc -- nib_acc_commit_kernel turns them into the offset vector of the launches that follow (see "the stale nibble product").
template <int MODE, int NK, bool BATCH, bool NT, bool PA, bool F32, bool NIBR = false>
__device__ __forceinline__ void
fi_one_tile_body(const AdmmParams &p, const unsigned char *__restrict__ Mp, const unsigned char *__restrict__ types, int ntiles, int nblk, long long g, int aslot,
                 int uslot /* u is read from: 0 = p.u, 1 = the alternate buffer */, int commit_prev, size_t mp_stride) {
    constexpr bool prefetch_all = PA;               // (a template parameter: the two cases need different register sets, together they spill)
#if defined(LPVS_TIMELINE) && LPVS_TIMELINE < 3
    const bool tl_on = MODE == FI_MID && !BATCH && !F32 && g_lpvs_tl != nullptr;
    unsigned long long *tl_rec = tl_on ? g_lpvs_tl + ((size_t)(g & 1) * (size_t)ntiles + blockIdx.x) * 8 : nullptr;
    if (tl_on && threadIdx.x == 0) {
        tl_rec[0] = (unsigned long long)g; tl_rec[1] = __builtin_amdgcn_s_memrealtime();
        tl_rec[6] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);      // HW_REG_XCC_ID[3:0]
        tl_rec[7] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID
    }
#endif
    __shared__ double sI[TS], sJ[TS], sT[4][TS], sq[2 * TS], red[3][4];
    const int sg = BATCH ? (int)blockIdx.y : 0, nprob = BATCH ? (int)gridDim.y : 1;
    const int64_t voff = (int64_t)sg * p.np;                           // this problem's vectors
    const int boff = sg * nblk, nbt = nprob * nblk;                    // ... and its slots among the per-block records
    const FiBufs f = fi_views(p.fi, (int64_t)nprob * p.np, nbt, nprob);
    Mp += (size_t)sg * mp_stride; types += (size_t)sg * (size_t)ntiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Launch order: the diagonal tiles first.  Their workgroups own the row blocks' state (the longest prologue) and at cfg3 they are
    // the float-head tiles (96 KB, loaded in two halves after the prologue): dealt out in tile order the last workgroup of the launch
    // would be the slowest one.
    int I, J;
    if (MODE == FI_LAST || (int)blockIdx.x < nblk) { I = J = blockIdx.x; }
    else {
        const int k = (int)blockIdx.x - nblk;                           // k-th tile below the diagonal: k = I (I - 1) / 2 + J, J < I
        I = (int)((1.0 + sqrt(1.0 + 8.0 * (double)k)) * 0.5);
        while (I * (I - 1) / 2 > k) --I;
        while ((I + 1) * I / 2 <= k) ++I;
        J = k - I * (I - 1) / 2;
    }
    const int t = I * (I + 1) / 2 + J;
    const unsigned char *tile = Mp + (size_t)t * (F32 ? (size_t)TS * TS * 4 : kSplitTileBytes);
    const unsigned char ttype = (MODE == FI_LAST || F32) ? 0 : types[t];
    AdmmStatus *status = p.status + sg;
    // ---- every load before the first wait, all of them unconditional (a load under a branch or in a loop of unknown length makes the
    // compiler wait for EVERYTHING at the next use): the state of this thread's element FIRST (loads return in order: the update then
    // runs while the tile is still streaming in), block norms and maxima as six clamped loads per lane (np <= 49152), then the tile
    // through a buffer descriptor of size 0 for a float-head tile (its loads are dropped; that format is read in two halves below).
    const int conv_flag = __builtin_nontemporal_load(&status->converged);
    const int blk = threadIdx.x < TS ? I : J, i = threadIdx.x & (TS - 1);
    const int64_t e = (int64_t)blk * TS + i;
    const bool ok = e < p.n;
    const int pg = (int)(g & 1), pg1 = pg ^ 1;                         // parities of g (= g - 2) and of g - 1
    double rhs_mem = 0, xbv = 0, uv = 0, qprev = 0;
    long long accp = 0;
    if (MODE == FI_FIRST) rhs_mem = p.rhs[voff + e];
    else {
        accp = f.acc((aslot + 2) % 3)[voff + e];                       // sums of the previous launch
        xbv = p.xb[voff + e];
        uv = (uslot ? f.ualt : p.u)[voff + e];
        qprev = f.qbuf[pg1 * nprob + sg];
    }
    double bnv[NK];
    double2 recv[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {                                     // NK = 1 (up to 64 row blocks: np <= 8192) or 6
        const int b = lane + 64 * k < nblk ? lane + 64 * k : nblk - 1;
        bnv[k] = MODE != FI_FIRST ? f.bn[pg * nbt + boff + b] : 0.0;   // ||x-z||^2 blocks of update u_{g-2}
        recv[k] = MODE != FI_LAST ? f.rec[pg * nbt + boff + b] : make_double2(0.0, 0.0);
    }
    // (single problems: host copies, kernel arguments instead of loads)
    const double Rrow = BATCH ? f.consts[2 * sg] : p.fi_R, xbmax = BATCH ? f.consts[2 * sg + 1] : p.fi_xbmax;
    __builtin_amdgcn_sched_barrier(0);               // (the scheduler must not sink state loads below the tile's: they are wanted first)
    FixRaw fr;
    float4 fha[8], fhb[8];                           // F32: the lane's 8 row groups x 8 columns
    double diag_pre = 0.0;
    if (MODE != FI_LAST && F32) {
        typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(tile), 0, TS * TS * 4, 0x00020000);
        const int off_head = ((wave * 32 + (lane >> 4)) * TS + 4 * (lane & 15)) * 4;
#pragma unroll
        for (int rg = 0; rg < 8; ++rg) {
            fha[rg] = __builtin_bit_cast(float4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(rs, off_head, rg * (4 * TS * 4), 0));
            fhb[rg] = __builtin_bit_cast(float4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(rs, off_head, rg * (4 * TS * 4) + 256, 0));
        }
    }
    if (MODE != FI_LAST && !F32) {
        typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
        // (issued for every tile BELOW the diagonal without waiting for its format byte -- a float-head tile there, none at cfg3,
        // costs 74 KB of wasted reads; a diagonal tile is requested here only when the whole batch is in the fixed format, with the
        // 1 KiB of its double diagonal behind the steps -- otherwise it is loaded further down)
        constexpr int aux = NT ? 2 : 0;
        const int tbytes = I != J ? (int)kMixedFixedTileBytes : (prefetch_all ? (int)kMixedFixedTileBytes + TS * 8 : 0);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(tile), 0, tbytes, 0x00020000);
        // (32-bit tiles: the nibbles are zero and are not read -- the same two loads through a descriptor of size zero return them without traffic)
        const __amdgpu_buffer_rsrc_t rs_nq = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(tile), 0, p.mp_fix32 ? 0 : tbytes, 0x00020000);
        const int gq_ = lane >> 4, c_ = lane & 15;
        const int off_head = ((wave * 32 + gq_) * TS + 4 * c_) * 4;
        const int off_nq = (int)kFixHeadBytes + (wave * 64 + lane) * 32, off_st = (int)(kFixHeadBytes + kFixNibBytes) + (wave * 4 + gq_) * 32;
        fr.nq[0] = __builtin_bit_cast(uint4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(rs_nq, off_nq, 0, aux));    // (nibbles and steps first: fix_load)
        fr.nq[1] = __builtin_bit_cast(uint4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(rs_nq, off_nq, 16, aux));
        fr.st[0] = __builtin_bit_cast(float4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(rs, off_st, 0, aux));
        fr.st[1] = __builtin_bit_cast(float4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(rs, off_st, 16, aux));
#pragma unroll
        for (int rg = 0; rg < 8; ++rg) {
            fr.ha[rg] = __builtin_bit_cast(int4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(rs, off_head, rg * (4 * TS * 4), aux));
            fr.hb[rg] = __builtin_bit_cast(int4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(rs, off_head, rg * (4 * TS * 4) + 256, aux));
        }
        if (PA) {   // the lane's row after the row butterfly: its entry of a diagonal tile's double diagonal (dropped for every other tile)
            const int rgo = ((c_ & 8) ? 4 : 0) + ((c_ & 4) ? 2 : 0) + ((c_ & 2) ? 1 : 0);
            typedef unsigned int u32x2b __attribute__((ext_vector_type(2)));
            const u32x2b dw = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(kFixHeadBytes + kFixNibBytes) + TS * 4 + (wave * 32 + 4 * rgo + gq_) * 8, 0, 0);
            diag_pre = __builtin_bit_cast(double, dw);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    double mR = 0, mU = 0;
#pragma unroll
    for (int k = 0; k < NK; ++k) { mR = fmax(mR, recv[k].x); mU = fmax(mU, recv[k].y); }
    if (conv_flag) return;
    // (tol <= 0 can never stop: only the workgroup that keeps the status needs the norm then -- a dependent shuffle chain less in every other prologue)
    if (MODE != FI_FIRST && commit_prev && (p.tol > 0.0 || blockIdx.x == 0)) {   // (uniform, host-known) commit update u_{g-2}
        double part = 0.0;                                             // lane q sums blocks q, q + 64, ...; then the wave's fixed shuffle pattern
#pragma unroll
        for (int k = 0; k < NK; ++k) part += lane + 64 * k < nblk ? bnv[k] : 0.0;
        const double nxz = sqrt(wave_sum(part));                       // every wave, identically     norm(tmp)   src/lasso.jl:157
        const bool conv = nxz < p.tol;                                 //                             src/lasso.jl:164
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            status->iters += 1;
            status->nxz = nxz;
            if (conv) status->converged = 1;
        }
        if (conv) return;                                              // every workgroup takes the same decision
    }
    // ---- update u_{g-1} for this thread's element (threads < 128: block I, the others: block J)
    double rhs_v;
    if (MODE == FI_FIRST) rhs_v = rhs_mem;
    else {
        const double xi = ok ? xbv + (double)accp * qprev : 0.0;       // x = xb + M~ (z-u)/mu: the exact integer sum, scaled once
        const double ui = ok ? uv : 0.0;
        const double v = xi + ui;
        double zi = 0.0;
        if (p.prox_kind == LPVS_PROX_L1) {
            const double gl = p.mu * p.prox_param;
            zi = v + (v <= -gl ? gl : (v >= gl ? -gl : -v));
        } else if (p.prox_kind == LPVS_PROX_L0) {
            zi = fabs(v) > sqrt(2.0 * p.mu * p.prox_param) ? v : 0.0;
        } else {  // group: block soft-threshold, norms through LDS (as admm_fused_update2_kernel)
            // every lane sums its own group (LDS broadcast reads, the same sequential order as norm() on the slice): one barrier
            // instead of two, no lanes idling behind eight of them
            const int gl = (int)p.group_len;
            sq[threadIdx.x] = v * v;
            __syncthreads();
            const double *grp = sq + (threadIdx.x & TS) + (i / gl) * gl;
            double s2 = 0;
            for (int q = 0; q < gl; ++q) s2 += grp[q];
            double scale = 1.0 - p.prox_param * p.mu / sqrt(s2);                          // s2 == 0 -> -inf -> 0
            if (!(scale > 0)) scale = 0.0;
            zi = scale * v;
        }
        if (!ok) zi = 0.0;
        const double d = xi - zi, un = ui + d;                         // src/lasso.jl:154-155
        rhs_v = ok ? (zi - un) / p.mu : 0.0;
        if (I == J) {                                                  // the block's owner (uniform): state, norm, maxima, next accumulator
            const bool own = threadIdx.x < TS;
            if (own) {
                p.x[voff + e] = xi; p.z[voff + e] = zi;
                (MODE == FI_LAST ? p.u : (uslot ? p.u : f.ualt))[voff + e] = un;
                if (MODE == FI_LAST) p.rhs[voff + e] = rhs_v;
            }
            const double d2 = own && ok ? d * d : 0.0;
            // (a NaN would drop out of fmax: it is recorded as an infinite maximum, which makes the next bound infinite and the iterates NaN)
            const double inf_ = __longlong_as_double(0x7ff0000000000000ll);
            const double ar = fabs(rhs_v) < inf_ ? fabs(rhs_v) : inf_, au = fabs(un) < inf_ ? fabs(un) : inf_;
            const double w0 = wave_sum(d2), w1 = wave_max(own ? ar : 0.0), w2 = wave_max(own ? au : 0.0);
            if (lane == 0) { red[0][wave] = w0; red[1][wave] = w1; red[2][wave] = w2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                f.bn[pg1 * nbt + boff + I] = red[0][0] + red[0][1];
                f.rec[pg1 * nbt + boff + I] = make_double2(fmax(red[1][0], red[1][1]), fmax(red[2][0], red[2][1]));
            }
        }
    }
    if (MODE == FI_LAST) return;
    LPVS_TL_STAMP(2);
    if (I == J && threadIdx.x < TS) f.acc((aslot + 1) % 3)[voff + e] = 0;   // the accumulator of the next launch
    if (threadIdx.x < TS) sI[i] = rhs_v; else sJ[i] = rhs_v;
    // ---- this launch's quantum (identical in every workgroup)
    mR = wave_max_nonneg(mR); mU = wave_max_nonneg(mU);
    // The factor 1.000001: Rrow is the largest absolute row sum of the FULL-PRECISION inverse (fi_rowsum_kernel), the product streams
    // its packed copy M~, whose row sums may exceed it by kFiPackedRowSlack (relative) -- see the static_assert at its definition.
    double B = Rrow * ((xbmax + Rrow * mR + mU) / p.mu) * 1.000001;
    // An infinite bound (an overflowed iterate, or fi_state_kernel's marker for a non-finite entry in the state it was given) has no
    // quantum: NaN then, which the next prologue's x = xb + acc * quantum spreads over every element -- as the two-launch iteration
    // and the reference's own arithmetic would (a NaN partial converted to an integer would otherwise silently vanish from x).
    const bool bound_ok = B < 0x1p1000;
    if (!(B > 0x1p-900)) B = 0x1p-900;
    int eb = 0;
    (void)frexp(bound_ok ? B : 1.0, &eb);                              // B < 2^eb
    const double quantum = bound_ok ? ldexp(1.0, eb - 62) : __longlong_as_double(0x7ff8000000000000ll), invq = bound_ok ? ldexp(1.0, 62 - eb) : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) f.qbuf[pg * nprob + sg] = quantum;
    __syncthreads();
    LPVS_TL_STAMP(3);
    // ---- tile product
    const int c = lane & 15, gq = lane >> 4;
    double rj[8], tc[8], v[8];
    const double *diag = nullptr;
    if (F32) {
        // single-precision tile (symv_tile_f32_kernel's product), row group by row group as the bytes arrive
#pragma unroll
        for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) tc[k] = 0.0;
#pragma unroll
        for (int rg = 0; rg < 8; ++rg) {
            const double ri = sI[wave * 32 + 4 * rg + gq];
            const float hh[8] = {fha[rg].x, fha[rg].y, fha[rg].z, fha[rg].w, fhb[rg].x, fhb[rg].y, fhb[rg].z, fhb[rg].w};
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const double m0 = (double)hh[k], m1 = (double)hh[k + 1];
                tc[k] = opaque(fma(m0, ri, tc[k]));
                tc[k + 1] = opaque(fma(m1, ri, tc[k + 1]));
                a0 = fma(m0, rj[k], a0);
                a1 = fma(m1, rj[k + 1], a1);
            }
            v[rg] = a0 + a1;
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (PA || ttype != 0) {
        // (two call sites, two register sets: a tile loaded under a branch into the registers of the prefetched one would make the
        // compiler wait for everything before the first product)
        if (!PA && I == J) {                         // (uniform) a diagonal tile in the fixed format that was not requested up front: only now
            FixRaw fd;
            fix_load(tile, wave, lane, fd, p.mp_fix32 != 0 ? 1 : 0);
            fi_fixed_product(fd, sI, sJ, wave, lane, v, tc);
        } else fi_fixed_product(fr, sI, sJ, wave, lane, v, tc);
        if (ttype == 2 && !prefetch_all) diag = reinterpret_cast<const double *>(tile + kFixHeadBytes + kFixNibBytes + TS * 4);
    } else {
        // float head + 16-bit tail, two halves of four row groups (as symv_tile_mixed_kernel)
#pragma unroll
        for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) tc[k] = 0.0;
        const float *head = reinterpret_cast<const float *>(tile) + (wave * 32 + gq) * TS + 4 * c;
        const unsigned short *tail = reinterpret_cast<const unsigned short *>(tile + (size_t)TS * TS * 4) + (wave * 32 + gq) * TS + 8 * c;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 ha[4], hb[4];
            uint4 lq[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rg = 4 * half + r4;
                ha[r4] = load16<NT, float4>(head + rg * 4 * TS);
                hb[r4] = load16<NT, float4>(head + rg * 4 * TS + 64);
                lq[r4] = load16<NT, uint4>(tail + rg * 4 * TS);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rg = 4 * half + r4;
                const double ri = sI[wave * 32 + 4 * rg + gq];
                const float hh[8] = {ha[r4].x, ha[r4].y, ha[r4].z, ha[r4].w, hb[r4].x, hb[r4].y, hb[r4].z, hb[r4].w};
                const unsigned int qq[8] = {lq[r4].x & 0xffffu, lq[r4].x >> 16, lq[r4].y & 0xffffu, lq[r4].y >> 16,
                                            lq[r4].z & 0xffffu, lq[r4].z >> 16, lq[r4].w & 0xffffu, lq[r4].w >> 16};
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const double m0 = split_decode(hh[k], qq[k]), m1 = split_decode(hh[k + 1], qq[k + 1]);
                    tc[k] = opaque(fma(m0, ri, tc[k]));
                    tc[k + 1] = opaque(fma(m1, ri, tc[k + 1]));
                    a0 = fma(m0, rj[k], a0);
                    a1 = fma(m1, rj[k + 1], a1);
                }
                v[rg] = a0 + a1;
            }
        }
    }
    // ---- row sums (halving butterfly over the 16 column lanes), column sums (four row lanes, then the four waves), added into x
    LPVS_TL_STAMP(4);
#pragma unroll
    for (int m = 8, cnt = 4; m >= 2; m >>= 1, cnt >>= 1) {
        const bool up = (c & m) != 0;
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            const double lo_ = opaque(v[k]), hi_ = opaque(v[k + cnt]);
            v[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
        }
    }
    v[0] += __shfl_xor(v[0], 1, 64);
    unsigned long long *acc_cur = reinterpret_cast<unsigned long long *>(f.acc(aslot));
    if ((c & 1) == 0) {
        const int rg = ((c & 8) ? 4 : 0) + ((c & 4) ? 2 : 0) + ((c & 2) ? 1 : 0);
        const int row = wave * 32 + 4 * rg + gq;
        const double r1 = diag != nullptr ? fma(diag[row], sI[row], v[0]) : (PA && I == J ? fma(diag_pre, sI[row], v[0]) : v[0]);   // (PA: every diagonal tile keeps its diagonal apart)
        atomicAdd(acc_cur + voff + (int64_t)I * TS + row, (unsigned long long)__double2ll_rn(r1 * invq));
    }
    if (I != J) {
#pragma unroll
        for (int m = 32, cnt = 4; m >= 16; m >>= 1, cnt >>= 1) {
            const bool up = (lane & m) != 0;
#pragma unroll
            for (int k = 0; k < cnt; ++k) {
                const double lo_ = opaque(tc[k]), hi_ = opaque(tc[k + cnt]);
                tc[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
            }
        }
        const int col = ((lane & 32) ? 64 : 0) + 4 * c + ((lane & 16) ? 2 : 0);
        sT[wave][col] = tc[0]; sT[wave][col + 1] = tc[1];
        __syncthreads();
        if (threadIdx.x < TS) {
            const double r2 = ((sT[0][threadIdx.x] + sT[1][threadIdx.x]) + sT[2][threadIdx.x]) + sT[3][threadIdx.x];
            atomicAdd(acc_cur + voff + (int64_t)J * TS + threadIdx.x, (unsigned long long)__double2ll_rn(r2 * invq));
        }
    }
    if constexpr (NIBR && !F32) {
        // ---- the nibble planes of a fixed-point tile against the same two blocks (fix_load's mode 2: heads = the bias, so an element decodes
        // to nibble x step), reduced as above, into the second accumulator.  A rare launch (one in nib_period): the planes are requested only now.
        if (ttype != 0) {                                              // (uniform)
            __syncthreads();                                           // sT is read above
            FixRaw fb;
            fix_load(tile, wave, lane, fb, 2);
            double vn[8], tn[8];
            fi_fixed_product(fb, sI, sJ, wave, lane, vn, tn);
            unsigned long long *acc_n = reinterpret_cast<unsigned long long *>(p.nib_acc);
#pragma unroll
            for (int m = 8, cnt = 4; m >= 2; m >>= 1, cnt >>= 1) {
                const bool up = (c & m) != 0;
#pragma unroll
                for (int k = 0; k < cnt; ++k) {
                    const double lo_ = opaque(vn[k]), hi_ = opaque(vn[k + cnt]);
                    vn[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
                }
            }
            vn[0] += __shfl_xor(vn[0], 1, 64);
            if ((c & 1) == 0) {
                const int rg = ((c & 8) ? 4 : 0) + ((c & 4) ? 2 : 0) + ((c & 2) ? 1 : 0);
                atomicAdd(acc_n + voff + (int64_t)I * TS + wave * 32 + 4 * rg + gq, (unsigned long long)__double2ll_rn(vn[0] * invq));
            }
            if (I != J) {
#pragma unroll
                for (int m = 32, cnt = 4; m >= 16; m >>= 1, cnt >>= 1) {
                    const bool up = (lane & m) != 0;
#pragma unroll
                    for (int k = 0; k < cnt; ++k) {
                        const double lo_ = opaque(tn[k]), hi_ = opaque(tn[k + cnt]);
                        tn[k] = (up ? hi_ : lo_) + __shfl_xor(up ? lo_ : hi_, m, 64);
                    }
                }
                const int col = ((lane & 32) ? 64 : 0) + 4 * c + ((lane & 16) ? 2 : 0);
                sT[wave][col] = tn[0]; sT[wave][col + 1] = tn[1];
                __syncthreads();
                if (threadIdx.x < TS) {
                    const double r2 = ((sT[0][threadIdx.x] + sT[1][threadIdx.x]) + sT[2][threadIdx.x]) + sT[3][threadIdx.x];
                    atomicAdd(acc_n + voff + (int64_t)J * TS + threadIdx.x, (unsigned long long)__double2ll_rn(r2 * invq));
                }
            }
        }
    }
    LPVS_TL_STAMP(5);
}

template <int MODE, int NK, bool BATCH, bool NT, bool PA, bool F32, bool NIBR = false>
__global__ void __launch_bounds__(256, 3)
admm_iter_mixed_kernel(AdmmParams p, const unsigned char *__restrict__ Mp, const unsigned char *__restrict__ types, int ntiles, int nblk, long long g, int aslot,
                       int uslot, int commit_prev, size_t mp_stride, int /* PA as a run-time value: unused */) {
    fi_one_tile_body<MODE, NK, BATCH, NT, PA, F32, NIBR>(p, Mp, types, ntiles, nblk, g, aslot, uslot, commit_prev, mp_stride);
}
// xb = xb_corr + (sums of the nibble planes' product, integers of the launch's quantum) x quantum; the accumulator is left zeroed for the next refresh
// (blockIdx.y = problem of a batch: vectors np apart, one quantum each)
__global__ void __launch_bounds__(256)
nib_acc_commit_kernel(long long *__restrict__ acc, const double *__restrict__ quantum, const double *__restrict__ xb_corr, double *__restrict__ xb, int64_t np) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    const int64_t e = (int64_t)blockIdx.y * np + i;
    xb[e] = xb_corr[e] + (double)acc[e] * quantum[blockIdx.y];
    acc[e] = 0;
}

// constants and records of the one-launch iteration (after lpvs_admm_init / set_state; base = iterations done so far); p.ns problems
__global__ void set_double_kernel(double *dst, double v) { *dst = v; }
int32_t launch_fi_setup(const AdmmParams &p, long long base, bool with_consts, hipStream_t s, double r_known) {
    const int nblk = (int)(p.np / TS), nprob = p.ns;
    const FiBufs f = fi_views(p.fi, (int64_t)nprob * p.np, nprob * nblk, nprob);
    if (with_consts) {
        LPVS_HIP(hipMemsetAsync(f.consts, 0, sizeof(double) * 2 * (size_t)nprob, s));
        if (r_known > 0 && nprob == 1) hipLaunchKernelGGL(set_double_kernel, dim3(1), dim3(1), 0, s, f.consts, r_known);   // (the packing pass left it)
        else hipLaunchKernelGGL(fi_rowsum_kernel, dim3((unsigned)ceil_div(p.np, 4), (unsigned)nprob), dim3(256), 0, s, p.M, p.np, p.n, reinterpret_cast<unsigned long long *>(f.consts));
    }
    hipLaunchKernelGGL(fi_state_kernel, dim3((unsigned)nprob), dim3(256), 0, s, p, nblk, base, with_consts ? 1 : 0);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

// after a chunk: if an update of THIS chunk converged before the chunk's last one, its u may sit in the alternate buffer
// (blockIdx.y = problem)
__global__ void __launch_bounds__(256)
fi_fixup_kernel(AdmmParams p, int nblk, long long base, long long iters) {
    const int sg = blockIdx.y, nprob = gridDim.y;
    const AdmmStatus *status = p.status + sg;
    if (!status->converged) return;
    const long long ic = status->iters - base - 1;                     // chunk-local index of the converged update
    if (ic < 0 || ic > iters - 2 || ((ic + 1) & 1) == 0) return;       // (u_ic was written to slot (ic + 1) & 1; the last update writes p.u itself)
    const FiBufs f = fi_views(p.fi, (int64_t)nprob * p.np, nprob * nblk, nprob);
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < p.np) p.u[(int64_t)sg * p.np + e] = f.ualt[(int64_t)sg * p.np + e];
}
// commits the chunk's last update (one workgroup of 64 lanes per problem; block norms of parity `par`)
__global__ void __launch_bounds__(64)
fi_commit_kernel(AdmmParams p, int nblk, int par) {
    const int sg = blockIdx.x, nprob = gridDim.x;
    AdmmStatus *status = p.status + sg;
    if (status->converged) return;
    const FiBufs f = fi_views(p.fi, (int64_t)nprob * p.np, nprob * nblk, nprob);
    __shared__ double slot;
    const double nxz = pending_norm(f.bn + (int64_t)par * nprob * nblk + (int64_t)sg * nblk, nblk, &slot);
    if (threadIdx.x == 0) {
        status->iters += 1;
        status->nxz = nxz;
        if (nxz < p.tol) status->converged = 1;
    }
}

int32_t fi_read_consts(const AdmmParams &p, double out[2], hipStream_t s) {
    const FiBufs f = fi_views(p.fi, p.np, (int)(p.np / TS));
    LPVS_HIP(hipMemcpyAsync(out, f.consts, sizeof(double) * 2, hipMemcpyDeviceToHost, s));
    LPVS_HIP(hipStreamSynchronize(s));
    return LPVS_OK;
}

// a chunk of `iters` iterations: first launch (mat-vec of the right-hand side in memory), iters - 1 fused launches, the last update.
// Single problem (p.ns == 1, batch == false) or a batch of p.ns problems that each own their matrix (mp_stride bytes apart).
template <int NK, bool BATCH, bool NT, bool PA, bool F32 = false>
static FiKernel fi_kernel_mode(int mode) {
    return mode == FI_FIRST ? admm_iter_mixed_kernel<FI_FIRST, NK, BATCH, NT, PA, F32> : mode == FI_MID ? admm_iter_mixed_kernel<FI_MID, NK, BATCH, NT, PA, F32>
                                                                                                       : admm_iter_mixed_kernel<FI_LAST, NK, BATCH, NT, PA, F32>;
}
static FiKernel fi_kernel(int mode, bool small, bool batch, bool nt, bool pa, bool f32 = false) {
    if (f32) return small ? fi_kernel_mode<1, false, false, true, true>(mode) : fi_kernel_mode<6, false, false, true, true>(mode);
    if (batch) {
        if (nt) return pa ? fi_kernel_mode<1, true, true, true>(mode) : fi_kernel_mode<1, true, true, false>(mode);
        return pa ? fi_kernel_mode<1, true, false, true>(mode) : fi_kernel_mode<1, true, false, false>(mode);
    }
    if (small) return pa ? fi_kernel_mode<1, false, false, true>(mode) : fi_kernel_mode<1, false, false, false>(mode);
    // (more than 64 row blocks: np >= 8320.  A single inverse beyond the Infinity Cache -- np >= 10752 -- streams with non-temporal loads)
    if (nt) return pa ? fi_kernel_mode<6, false, true, true>(mode) : fi_kernel_mode<6, false, true, false>(mode);
    return pa ? fi_kernel_mode<6, false, false, true>(mode) : fi_kernel_mode<6, false, false, false>(mode);
}
bool nib_fused_applies(const AdmmParams &p) {
    const char *e = getenv("LPVS_NIB_FUSED");
    return !(e && e[0] == '0') && p.nib_period > 0 && p.nib_acc != nullptr && p.mp_types != nullptr && p.fi_prefetch_all == 0 && fi_applicable(p);
}
// (the launches that also multiply the nibble planes: single problems, no prefetch of diagonal tiles; FIRST or MID)
template <bool NT, bool PA> static FiKernel fi_kernel_nibr_batch(int mode) {
    return mode == FI_FIRST ? admm_iter_mixed_kernel<FI_FIRST, 1, true, NT, PA, false, true> : admm_iter_mixed_kernel<FI_MID, 1, true, NT, PA, false, true>;
}
static FiKernel fi_kernel_nibr(int mode, bool small, bool nt, bool batch = false, bool pa = false) {
    if (batch) { if (nt) return pa ? fi_kernel_nibr_batch<true, true>(mode) : fi_kernel_nibr_batch<true, false>(mode);
                 return pa ? fi_kernel_nibr_batch<false, true>(mode) : fi_kernel_nibr_batch<false, false>(mode); }
    if (small) return mode == FI_FIRST ? admm_iter_mixed_kernel<FI_FIRST, 1, false, false, false, false, true> : admm_iter_mixed_kernel<FI_MID, 1, false, false, false, false, true>;
    if (nt) return mode == FI_FIRST ? admm_iter_mixed_kernel<FI_FIRST, 6, false, true, false, false, true> : admm_iter_mixed_kernel<FI_MID, 6, false, true, false, false, true>;
    return mode == FI_FIRST ? admm_iter_mixed_kernel<FI_FIRST, 6, false, false, false, false, true> : admm_iter_mixed_kernel<FI_MID, 6, false, false, false, false, true>;
}
static int32_t launch_fi_chunk(const AdmmParams &p, int64_t iters, bool batch, size_t mp_stride, bool prefetch_all, hipStream_t s) {
    const int nblk = (int)(p.np / TS), nprob = batch ? p.ns : 1;
    const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2);
    const FiBufs f = fi_views(p.fi, (int64_t)nprob * p.np, nprob * nblk, nprob);
    const unsigned char *Mp = reinterpret_cast<const unsigned char *>(p.Mp);
    LPVS_HIP(hipMemsetAsync(f.acc(0), 0, sizeof(long long) * (size_t)p.np * (size_t)nprob, s));
    const long long base = p.fi_base;
    const bool small = nblk <= 64;                    // one load per lane covers the block norms / maxima
    const int nto = option_in_effect(LPVS_OPT_NT_LOADS, p.opt_nt_loads);
    // (single problems: the bytes the launch really reads -- the fixed-point tiles are shorter than their slots; only the kernel for more than 64 row blocks has the variant)
    const size_t stream_bytes = batch ? (size_t)ntiles * kSplitTileBytes * (size_t)nprob : (size_t)ntiles * kMixedFixedTileBytes;
    const bool nt = (batch || !small) && !p.mp_f32 && (nto ? nto == LPVS_NT_ON : stream_bytes > ((size_t)240 << 20));
    auto launch = [&](int mode, unsigned grid, long long g, int aslot, int uslot, int commit_prev) {
        hipLaunchKernelGGL(fi_kernel(mode, small, batch, nt, prefetch_all, p.mp_f32 != 0), dim3(grid, (unsigned)nprob), dim3(256), 0, s, p, Mp, p.mp_types, (int)ntiles, nblk, g, aslot, uslot,
                           commit_prev, mp_stride, prefetch_all ? 1 : 0);
    };
    // stale nibble product: launch g (which multiplies rhs_g) also multiplies the nibble planes when g = 1 or g = 0 mod the period, and the
    // commit kernel behind it forms the offset vector of the launches from g + 1 on (LPVS_NIB_FUSED=0: the three stand-alone kernels of
    // launch_nibble_refresh instead -- the two-launch iteration's way, for A/B runs)
    // (a batch: only inside the launch -- launch_nibble_refresh is a single-problem routine; the engine enables the stale product only then)
    const bool nib = batch ? (p.nib_period > 0 && p.nib_acc != nullptr && p.xb_corr != nullptr && p.mp_types != nullptr) : (!prefetch_all && nib_fused_applies(p));
    auto refresh_due = [&](long long g) { return p.nib_period > 0 && nib_refresh_due(g, p.nib_period, p.nib_ramp); };
    auto launch_step = [&](int mode, long long g, int aslot, int uslot, int commit_prev, const double *u_after) -> int32_t {
        const bool due = refresh_due(g);
        if (due && nib) {
            hipLaunchKernelGGL(fi_kernel_nibr(mode, small, nt, batch, prefetch_all), dim3(ntiles, (unsigned)nprob), dim3(256), 0, s, p, Mp, p.mp_types, (int)ntiles, nblk, g, aslot, uslot, commit_prev,
                               mp_stride, prefetch_all ? 1 : 0);
            hipLaunchKernelGGL(nib_acc_commit_kernel, dim3((unsigned)ceil_div(p.np, 256), (unsigned)nprob), dim3(256), 0, s, p.nib_acc, (const double *)(f.qbuf + (g & 1) * nprob), p.xb_corr,
                               const_cast<double *>(p.xb), p.np);
            return LPVS_OK;
        }
        launch(mode, ntiles, g, aslot, uslot, commit_prev);
        if (due && !batch) return launch_nibble_refresh(p, u_after != nullptr, u_after, s);
        return LPVS_OK;
    };
    LPVS_TRY(launch_step(FI_FIRST, base, 0, 0, 0, nullptr));                                    // rhs_base is the one in memory
    for (int64_t j = 1; j < iters; ++j)     // launch j: update u_{j-1} (reads u from slot (j-1) & 1, writes the other), mat-vec of rhs_j = (z - u) / mu of the state it leaves
        LPVS_TRY(launch_step(FI_MID, base + j, (int)(j % 3), (int)((j - 1) & 1), j >= 2 ? 1 : 0, ((j - 1) & 1) ? p.u : f.ualt));
    // the chunk's last update u_{iters-1}: sums of launch iters - 1, u from slot (iters - 1) & 1, everything back in the handle's vectors
    launch(FI_LAST, (unsigned)nblk, base + iters, (int)(iters % 3), (int)((iters - 1) & 1), iters >= 2 ? 1 : 0);
    hipLaunchKernelGGL(fi_fixup_kernel, dim3((unsigned)ceil_div(p.np, 256), (unsigned)nprob), dim3(256), 0, s, p, nblk, base, (long long)iters);
    // commit the chunk's last update (deferred convergence test, as in the two-launch iteration)
    hipLaunchKernelGGL(fi_commit_kernel, dim3((unsigned)nprob), dim3(64), 0, s, p, nblk, (int)((base + iters - 1) & 1));
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

// ---- one launch per iteration for the full-matrix path (np < kSymmetricMinNp: cfg2's n = 1024) ---------------------------------------
// The two-launch iteration of these sizes (symv_kernel + admm_batch_prox_kernel) is bound by its two dependent launch boundaries and
// two lone-workgroup latency chains: 6.6 us per iteration for 8 MB of matrix.  Here an iteration is ONE launch of np / 4 workgroups:
//   * every workgroup redoes the WHOLE update of the previous iteration from x, u, b (np <= 2047 elements: a few per thread) -- z =
//     prox(x + u), u += x - z, the next right-hand side into LDS, and ||x - z|| in a fixed order, so that every workgroup takes the
//     SAME stopping decision without talking to another (the reference's test, src/lasso.jl:164, in the iteration it belongs to: no
//     deferral); the workgroup that owns four elements writes their z and u;
//   * then its four rows of x = M rhs (one wave per row; the rows were requested before anything else and are in flight during the
//     update) go to the OTHER x buffer: x and u are double-buffered by launch parity, nothing is read and written in the same launch;
//   * "converged before this launch" travels in a per-parity control word written by workgroup 0 of the previous launch.
// A chunk is: first launch (rhs from memory, no update), iters - 1 fused launches, a last update-only launch, and a fix-up that
// brings x and u back to the handle's vectors when the final state sits in the alternate buffers.  Bit-reproducible (fixed orders).
enum { SM_FIRST = 0, SM_MID = 1, SM_LAST = 2 };
constexpr int kSmallMaxNp = 2048;                      // (LDS image of the right-hand side; np < kSymmetricMinNp anyway)
template <int MODE, int NW /* waves = rows of M per workgroup */, int NPMAX /* np <= NPMAX: 1024 or kSmallMaxNp */>
__global__ void __launch_bounds__(64 * NW)
admm_small_iter_kernel(AdmmParams p, int j /* launch of the chunk: 0 = first */) {
    constexpr int NT = 64 * NW, EPT = NPMAX / NT, MT = NPMAX / 128;   // threads, elements of the state per thread (i = tid + NT k), 16-byte pieces of a row per lane
    typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2b __attribute__((ext_vector_type(2)));
    __shared__ double srhs[NPMAX], sv[NPMAX], red[NW];
    const int sg = blockIdx.y;
    const int64_t o = (int64_t)sg * p.np;
    const int np = (int)p.np, n = (int)p.n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *ctl = p.sm_ctl + 4 * sg;
    AdmmStatus *status = p.status + sg;
    double *const Xb[2] = {p.x + o, p.scratch + (int64_t)sg * 2 * p.np};
    double *const Ub[2] = {p.u + o, p.scratch + (int64_t)sg * 2 * p.np + p.np};
    const int was_conv = MODE == SM_FIRST ? __builtin_nontemporal_load(&status->converged) : __builtin_nontemporal_load(&ctl[(j - 1) & 1]);
    // ---- the state, eight elements per thread (i = tid + 256 k), through descriptors of np doubles
    double xv[EPT], uv[EPT], bv[EPT];
    if (MODE == SM_FIRST) {
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(p.rhs + o, 0, np * 8, 0x00020000);
#pragma unroll
        for (int k = 0; k < EPT; ++k) xv[k] = __builtin_bit_cast(double, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(rr, (int)threadIdx.x * 8, k * NT * 8, 0));
    } else {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(Xb[j & 1], 0, np * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(Ub[(j - 1) & 1], 0, np * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.b + o), 0, np * 8, 0x00020000);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            xv[k] = __builtin_bit_cast(double, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(rx, (int)threadIdx.x * 8, k * NT * 8, 0));
            uv[k] = __builtin_bit_cast(double, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(ru, (int)threadIdx.x * 8, k * NT * 8, 0));
            bv[k] = __builtin_bit_cast(double, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(rb, (int)threadIdx.x * 8, k * NT * 8, 0));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- then this workgroup's rows of M (one wave per row, a lane covers columns 2 lane + 128 t): unconditional loads through a
    // descriptor of the row's size -- columns past np read as zero.  Loads return in order: the state (requested first) arrives
    // first, and the update runs while the rows are still in flight (rows first: 5.73 us per iteration at cfg2; state first: see DESIGN)
    const int row = (int)blockIdx.x * NW + wave;
    double2 mrow[MT];
    if (MODE != SM_LAST) {
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.M + (int64_t)row * p.np), 0, row < np ? np * 8 : 0, 0x00020000);
#pragma unroll
        for (int t = 0; t < MT; ++t) mrow[t] = __builtin_bit_cast(double2, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(rm, lane * 16, t * 1024, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == SM_FIRST) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl[0] = was_conv; ctl[2] = was_conv ? 0 : -1; }
        if (was_conv) return;
#pragma unroll
        for (int k = 0; k < EPT; ++k) srhs[threadIdx.x + NT * k] = xv[k];
    } else {
        if (was_conv) { if (blockIdx.x == 0 && threadIdx.x == 0) ctl[j & 1] = 1; return; }
        // ---- update of iteration (base + j): z = prox(x + u), u += x - z, rhs = b + (z - u)/mu, ||x - z||     src/lasso.jl:153-157
        double zv[EPT], vv[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) { const int i = threadIdx.x + NT * k; vv[k] = i < n ? xv[k] + uv[k] : 0.0; }
        if (p.prox_kind == LPVS_PROX_GROUP_L2) {
            const int gl = (int)p.group_len;
#pragma unroll
            for (int k = 0; k < EPT; ++k) sv[threadIdx.x + NT * k] = vv[k] * vv[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int i = threadIdx.x + NT * k;
                double s2 = 0;
                if (i < n) { const double *grp = sv + (i / gl) * gl; for (int q = 0; q < gl; ++q) s2 += grp[q]; }   // the same sequential order as norm() on the slice
                double scale = 1.0 - p.prox_param * p.mu / sqrt(s2);
                if (!(scale > 0)) scale = 0.0;
                zv[k] = i < n ? scale * vv[k] : 0.0;
            }
        } else {
            const double gl1 = p.mu * p.prox_param, th0 = sqrt(2.0 * p.mu * p.prox_param);
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const double v = vv[k];
                zv[k] = p.prox_kind == LPVS_PROX_L1 ? v + (v <= -gl1 ? gl1 : (v >= gl1 ? -gl1 : -v)) : (fabs(v) > th0 ? v : 0.0);
            }
        }
        double ss = 0, un[EPT], rh[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = threadIdx.x + NT * k;
            const bool ok = i < n;
            const double xi = ok ? xv[k] : 0.0, d = xi - zv[k];
            un[k] = (ok ? uv[k] : 0.0) + d;
            rh[k] = ok ? bv[k] + (zv[k] - un[k]) / p.mu : 0.0;
            ss = fma(d, d, ss);
            srhs[i] = rh[k];
        }
        double nxz = 0.0;
        if (p.tol > 0.0 || blockIdx.x == 0) {         // (uniform; tol <= 0 can never stop: only the workgroup that keeps the status needs the norm then)
            const double w = wave_sum(ss);
            if (lane == 0) red[wave] = w;
            __syncthreads();
            double tot = 0;
#pragma unroll
            for (int q = 0; q < NW; ++q) tot += red[q];                         // fixed order
            nxz = sqrt(tot);              // every workgroup, identically     norm(tmp)   src/lasso.jl:157
        }
        const bool conv = nxz < p.tol;                                        //                                  src/lasso.jl:164
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            status->iters += 1;
            status->nxz = nxz;
            if (conv) { status->converged = 1; ctl[2] = j; }
            ctl[j & 1] = conv ? 1 : 0;
        }
        // the owner of elements [NW b, NW b + NW) writes their z and u (and the right-hand side when the chunk -- or the run -- ends here)
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = threadIdx.x + NT * k;
            if (i / NW == (int)blockIdx.x && i < np) {
                p.z[o + i] = zv[k];
                Ub[j & 1][i] = un[k];
                if (conv || MODE == SM_LAST) p.rhs[o + i] = rh[k];
            }
        }
        if (conv || MODE == SM_LAST) return;
    }
    __syncthreads();
    // ---- this wave's row of x = M rhs into the other buffer
    double acc = 0;
    const double2 *r2 = reinterpret_cast<const double2 *>(srhs);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        if (lane + 64 * t < np / 2) {                 // (uniform per t up to the last partial group; the loads above were unconditional)
            const double2 v = r2[lane + 64 * t];
            acc = fma(mrow[t].x, v.x, acc);
            acc = fma(mrow[t].y, v.y, acc);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0 && row < np) Xb[(j + 1) & 1][row] = acc;
}
// final state of a chunk of `iters` launches back into the handle's vectors when it sits in the alternate buffers (blockIdx.y = signal)
__global__ void __launch_bounds__(256)
admm_small_fixup_kernel(AdmmParams p, int iters) {
    const int sg = blockIdx.y;
    const int jc = p.sm_ctl[4 * sg + 2] >= 0 ? p.sm_ctl[4 * sg + 2] : iters;   // the launch that wrote the final u (read the final x)
    if ((jc & 1) == 0) return;
    const int64_t o = (int64_t)sg * p.np, e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < p.np) {
        p.x[o + e] = p.scratch[(int64_t)sg * 2 * p.np + e];
        p.u[o + e] = p.scratch[(int64_t)sg * 2 * p.np + p.np + e];
    }
}
bool small_iter_applicable(const AdmmParams &p) {
    const bool on = option_in_effect(LPVS_OPT_ITERATION, p.opt_iteration) != LPVS_ITERATION_TWO;
    if (!on || p.sm_ctl == nullptr || p.Mp != nullptr || p.scratch == nullptr || p.np > kSmallMaxNp || p.np < 128 || p.xb != nullptr) return false;
    if (p.prox_kind == LPVS_PROX_L1 || p.prox_kind == LPVS_PROX_L0) return true;
    return p.prox_kind == LPVS_PROX_GROUP_L2 && p.group_len >= 1 && p.group_len <= 256 && p.n % p.group_len == 0;
}
template <int NW, int NPMAX>
static void launch_small_launches(const AdmmParams &p, int64_t iters, hipStream_t s) {
    const dim3 grid((unsigned)(p.np / NW), (unsigned)p.ns), blk(64 * NW);
    hipLaunchKernelGGL((admm_small_iter_kernel<SM_FIRST, NW, NPMAX>), grid, blk, 0, s, p, 0);
    for (int64_t j = 1; j < iters; ++j) hipLaunchKernelGGL((admm_small_iter_kernel<SM_MID, NW, NPMAX>), grid, blk, 0, s, p, (int)j);
    hipLaunchKernelGGL((admm_small_iter_kernel<SM_LAST, NW, NPMAX>), grid, blk, 0, s, p, (int)iters);
}
static int32_t launch_small_chunk(const AdmmParams &p, int64_t iters, hipStream_t s) {
    // rows of M per workgroup (= waves): 4 (measured at cfg2: 4 rows 5.37 us, 8 rows 5.47, 16 rows 6.47 per iteration); LPVS_SMALL_ROWS = 8 for A/B runs
    const bool nw8 = [] { const char *e = getenv("LPVS_SMALL_ROWS"); return e && atoi(e) == 8; }();
    if (p.np <= 1024) { if (nw8) launch_small_launches<8, 1024>(p, iters, s); else launch_small_launches<4, 1024>(p, iters, s); }
    else { if (nw8) launch_small_launches<8, kSmallMaxNp>(p, iters, s); else launch_small_launches<4, kSmallMaxNp>(p, iters, s); }
    hipLaunchKernelGGL(admm_small_fixup_kernel, dim3((unsigned)ceil_div(p.np, 256), (unsigned)p.ns), dim3(256), 0, s, p, (int)iters);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_admm_matvec_only(const AdmmParams &p, int reps, hipStream_t s) {
    const bool sym = p.part != nullptr && p.Mp != nullptr;
    // (handles that iterate in one launch are timed on the stand-alone mat-vec of the two-launch scheme: the same product, no update)
    for (int i = 0; i < reps; ++i) {
        if (sym) {
            launch_sym_matvec(p, nullptr, s);
        } else {
            launch_symv_raw(p.M, p.np, p.rhs, p.x, nullptr, p.ns, s);
        }
    }
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_admm_iterations(const AdmmParams &p, int64_t iters, hipStream_t s) {
    const bool sym = p.part != nullptr && p.Mp != nullptr;
    if (sym && iters > 0 && fi_applicable(p)) return launch_fi_chunk(p, iters, false, 0, p.fi_prefetch_all != 0, s);
    if (!sym && iters > 0 && small_iter_applicable(p)) {
        for (int64_t done = 0; done < iters; done += 1 << 20) LPVS_TRY(launch_small_chunk(p, std::min<int64_t>(iters - done, 1 << 20), s));   // (the launch index is an int)
        return LPVS_OK;
    }
    for (int64_t i = 0; i < iters; ++i) {
        if (sym) {
            LPVS_TRY(launch_iteration_sym(p, s, (int)i));
        } else {
            launch_symv_raw(p.M, p.np, p.rhs, p.x, p.status, p.ns, s);
            if (p.prox_kind != LPVS_PROX_BALL_L0 && p.n <= 4096) {
                // small problems: the light 256-thread kernel (no 128 KiB LDS image) has the shorter latency
                AdmmBatch q{p.M, p.np, p.n, p.ns, p.b, p.x, p.z, p.u, p.rhs, p.mu, p.tol, p.prox_kind, p.prox_param, p.group_len,
                            p.status, nullptr, nullptr, 1};
                hipLaunchKernelGGL(admm_batch_prox_kernel, dim3((unsigned)p.ns), dim3(256), 0, s, q);
            } else {
                hipLaunchKernelGGL(admm_prox_kernel, dim3((unsigned)p.ns), dim3(1024), 0, s, p);
            }
        }
    }
    if (sym && iters > 0 && fused_ok(p)) {   // commit the last iteration of the chunk (deferred convergence test)
        const int nblk = (int)(p.np / TS);
        const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2);
        double *blocknorm = p.part + (size_t)ntiles * TS * 2 * (unsigned)p.ns;
        hipLaunchKernelGGL(admm_commit_kernel, dim3((unsigned)p.ns), dim3(64), 0, s, p, nblk, blocknorm, (int)((iters - 1) & 1));
    }
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

}  // namespace lpvs
