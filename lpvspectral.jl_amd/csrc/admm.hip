// admm.hip -- the scaled-form ADMM loop of src/lasso.jl:136-171 on the resident Gram.
//
//   x <- prox_{mu f}(z - u)   = M (b + (z-u)/mu),  M = (G + I/mu)^-1     (src/lasso.jl:150-151)
//   z <- prox_{mu g}(x + u)                                               (src/lasso.jl:152-153)
//   u <- u + (x - z);  stop when ||x - z||_2 < tol                        (src/lasso.jl:154-157,164)
//
// The ADMM family is six translation units around two headers (round 6; the kernels' code is what round 5's single admm.hip compiled to,
// instruction for instruction):
//   admm.hip             THIS FILE: the two-launch iteration -- tile mat-vecs on every storage of the inverse (doubles, floats, 6-byte, mixed; single
//                        problems and window batches), the gather of the tile partials, the fused updates with the deferred convergence test, the
//                        general prox kernel (L1 / L0 / group / top-r) --, the stale nibble product's refresh, and the dispatch (launch_admm_iterations)
//   admm_pack.hip        storages of the inverse: packing kernels (doubles, floats, 6-byte, mixed 36-bit / 40-bit per tile)
//   admm_multi.hip       several right-hand sides sharing M on the f64 matrix cores (symv_tile_mfma_ws_kernel; cfg5)
//   admm_one_launch.hip  the whole iteration in ONE launch: fixed-point accumulation of the tile partials (cfg3, cfg4)
//   admm_small.hip       one launch per iteration below np = 2048 (cfg2)
//   admm_refine.hip      refined ridge solves, the residual in twice the mantissa, the x-update's offset vector and its scheduled correction
//   admm_device.h        device helpers shared by the units (reductions, tile index map, decoders / tile products of the storages)
//   admm_host.h          the host entry points that cross the units (what api.hip calls is in lpvs_internal.h)
// Kernels of iterations after convergence see the device-side flag and exit, so a chunk of iterations can be enqueued without a host round
// trip and still stop at exactly the reference's iteration.
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

enum { FI_FIRST = 0, FI_MID = 1, FI_LAST = 2 };   // modes of admm_iter_mixed_kernel (admm_one_launch.hip)
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

bool admm_batch_uses_tiles(const AdmmBatch &p) {
    AdmmParams q{p.M, p.np, p.n, p.b, p.x, p.z, p.u, p.rhs, p.mu, p.tol, p.prox_kind, p.prox_param, p.group_len, p.status,
                 nullptr, p.part, p.Mp, p.nbatch};
    (void)q;
    return p.Mp != nullptr && p.part != nullptr;          // (launch_admm_batch_iterations' own test: non-fusable prox operators ride the tiles too)
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


// the mat-vec of one iteration on the packed symmetric form (tile partials -> part1 / part2)
void launch_sym_matvec(const AdmmParams &p, const AdmmStatus *status, hipStream_t s) {
    const int nblk = (int)(p.np / TS);
    const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2);
    const unsigned ns = (unsigned)p.ns;
    double *part1 = p.part, *part2 = part1 + (size_t)ntiles * TS * ns;
    if (multi_matvec_on_matrix_cores(p))             // several right-hand sides: admm_multi.hip
        launch_multi_matvec(p, ntiles, part1, part2, status, s);
    else if (p.ns > 1 && !p.mp_f32)
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

void launch_symv_raw(const double *M, int64_t np, const double *rhs, double *x, const AdmmStatus *st, int ns,
                            hipStream_t s) {
    if (np >= 4096)
        hipLaunchKernelGGL(symv_kernel<4>, dim3((unsigned)ceil_div(np, 16), (unsigned)ns), dim3(256), 0, s, M, np, rhs, x, st);
    else
        hipLaunchKernelGGL(symv_kernel<1>, dim3((unsigned)ceil_div(np, 4), (unsigned)ns), dim3(256), 0, s, M, np, rhs, x, st);
}

int32_t launch_symv(const double *M, int64_t np, const double *rhs, double *x, hipStream_t s, int ns) {
    launch_symv_raw(M, np, rhs, x, nullptr, ns, s);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}
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

// out[np] = the gathered tile partials (part1: row sums per tile, part2: column sums per tile) of ONE signal, nothing added
void launch_gather_tile_partials(const double *part1, const double *part2, int nblk, unsigned ntiles, int64_t np, double *out, hipStream_t s) {
    hipLaunchKernelGGL(symv_reduce_kernel, dim3((unsigned)nblk, 1u), dim3(256), 0, s, part1, part2, nblk, (int)ntiles, np, out, (const AdmmStatus *)nullptr, (const double *)nullptr, 0);
}
void launch_packed_apply(const AdmmParams &p, const double *rhs, double *out, hipStream_t s) {
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
