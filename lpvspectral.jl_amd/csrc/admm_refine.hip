// admm_refine.hip -- the accurate linear algebra around the iteration: refined ridge solves (ls_spectral, ls_spectral_lpv, init = true), the
// residual accumulated in twice the mantissa, the x-update's offset vector xb = M b and its scheduled correction, and the dense
// estimator of window batches.
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

int32_t launch_batch_matvec(const double *A, int64_t np, int nprob, int nrhs, const double *v, double *out, hipStream_t s) {
    hipLaunchKernelGGL(batch_matvec_kernel, dim3((unsigned)ceil_div(np, 4), (unsigned)nprob), dim3(256), 0, s, A, np, nrhs, v, out);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
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
    if (row < n) {
        // 16-byte loads, four in flight per lane (round 6: scalar 8-byte loads one at a time left the pass latency-bound, 129 us for the 512 MB of cfg3's
        // Gram); a lane takes the column pairs 2 lane + 128 t.  The sums are carried in twice the mantissa: their order is immaterial to the rounded result.
        const double2 *a2 = reinterpret_cast<const double2 *>(a);
        const int64_t nv = np / 2;                                   // (np is a multiple of 128)
#pragma unroll 4
        for (int64_t j = lane; j < nv; j += 64) {
            const double2 m = a2[j];
#pragma unroll
            for (int q = 0; q < kDdSignals; ++q)
                if (q < nsg) {
                    const double2 xv = reinterpret_cast<const double2 *>(x_all + (int64_t)(sg0 + q) * np)[j];
                    dot2_fma(acc[q], -m.x, xv.x);
                    dot2_fma(acc[q], -m.y, xv.y);
                }
        }
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
// x_q = M rhs_q for up to 8 right-hand sides in ONE pass over the full symmetric matrix (one wave per row; launch_symv_raw's kernels re-read M for
// every signal: 8 x 8.6 GB = 11 ms per product at cfg5, two products per lpvs_admm_init since round 6 -- 3.6 ms each this way: 8.66 GB fetched, the right-hand sides re-read from the L2 per row).  Per signal the
// arithmetic is symv_kernel's, operation for operation (lane j takes the column pairs j, j + 64, ... in order, then the wave's fixed shuffle sum).
template <int NSB>
__global__ void __launch_bounds__(256)
symv_rows_multi_kernel(const double *__restrict__ M, int64_t np, const double *__restrict__ rhs_all, double *__restrict__ x_all, int ns) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= np) return;
    const double2 *m2 = reinterpret_cast<const double2 *>(M + row * np);
    const double2 *r2 = reinterpret_cast<const double2 *>(rhs_all);
    const int64_t nv = np / 2;
    double acc[NSB];
#pragma unroll
    for (int q = 0; q < NSB; ++q) acc[q] = 0.0;
    for (int64_t j = lane; j < nv; j += 64) {
        const double2 m = m2[j];
#pragma unroll
        for (int q = 0; q < NSB; ++q)
            if (q < ns) {
                const double2 v = r2[(int64_t)q * nv + j];
                acc[q] = fma(m.x, v.x, acc[q]);
                acc[q] = fma(m.y, v.y, acc[q]);
            }
    }
#pragma unroll
    for (int q = 0; q < NSB; ++q) {
        const double sum = wave_sum(acc[q]);
        if (lane == 0 && q < ns) x_all[(int64_t)q * np + row] = sum;
    }
}
static void launch_symv_all_signals(const double *M, int64_t np, const double *rhs, double *x, int ns, hipStream_t s) {
    if (ns == 1) { launch_symv_raw(M, np, rhs, x, nullptr, 1, s); return; }
    for (int q0 = 0; q0 < ns; q0 += 8)
        hipLaunchKernelGGL(symv_rows_multi_kernel<8>, dim3((unsigned)ceil_div(np, 4)), dim3(256), 0, s, M, np, rhs + (int64_t)q0 * np, x + (int64_t)q0 * np, std::min(8, ns - q0));
}

int32_t launch_offset_vector_refined(const double *G, const double *M, int64_t np, int64_t n, int ns, const double *b, double shift, int steps,
                                     double *xb, double *t1, double *t2, hipStream_t s) {
    launch_symv_all_signals(M, np, b, xb, ns, s);
    const int64_t total = np * (int64_t)ns;
    const unsigned nb = (unsigned)ceil_div(total, 256);
    for (int k = 0; k < steps; ++k) {
        launch_residual_dd(G, np, n, ns, b, 1.0, xb, shift, t2, s);                                              // t2 = b - H xb
        launch_symv_all_signals(M, np, t2, t1, ns, s);                                                           // t1 = M r
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

}  // namespace lpvs
