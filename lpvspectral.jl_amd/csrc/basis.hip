// basis.hip -- regressor / basis-table assembly kernels (gfx950).
//
// Every product is rounded separately, in the reference's operation order (this file is
// compiled with -ffp-contract=off): the Fourier phase is (2pi*f)*t (src/lsfft.jl:41), the LPV
// phase is w*x (src/lasso.jl:39), the activation is exp((-gamma)*(d*d)) (src/lsfft.jl:195).
// These kernels are HBM-write bound (one sincos / exp per 16 / 8 output bytes); all stores are
// coalesced along the fastest-varying output index.
#include "lpvs_internal.h"

#include <cstdlib>
#include <string>

namespace lpvs {

namespace {

constexpr double kTwoPi = 6.283185307179586;  // T(2pi), src/lsfft.jl:33

// ---- a2: column-major N x Nreg regressor (the layout get_fourier_regressor returns) -------
__global__ void __launch_bounds__(256)
fourier_regressor_colmajor_kernel(const double *__restrict__ t, int64_t N,
                                  const double *__restrict__ f, int64_t Nf, int zerofreq,
                                  double *__restrict__ A) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t fn = blockIdx.y;
    if (n >= N) return;
    const double dd = 1.0 / sqrt((double)(2 * Nf));       // src/lsfft.jl:35
    const int64_t sinoffset = zerofreq ? Nf - 1 : Nf;     // src/lsfft.jl:32,37-39
    const double phi = (kTwoPi * f[fn]) * t[n];
    double s, c;
    sincos(phi, &s, &c);
    A[n + fn * N] = c * dd;
    if (!(zerofreq && fn == 0)) A[n + (fn + sinoffset) * N] = -s * dd;
}

// ---- a2 transposed: D[col][ldn] (one regressor column per row, samples along the row) ---------------
// = the column-major regressor with leading dimension ldn; it is the k-major panel of the DUAL Gram
// A A' (contraction over regressor columns).  Rows >= Nreg and samples >= N are zero.
__global__ void __launch_bounds__(256)
fourier_dual_panel_kernel(const double *__restrict__ t, int64_t N, const double *__restrict__ f, int64_t Nf, int zerofreq,
                          double *__restrict__ D, int64_t ldn, int64_t nrows) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = blockIdx.y;                                   // row of D = regressor column
    if (n >= ldn) return;
    const int64_t nreg = zerofreq ? 2 * Nf - 1 : 2 * Nf;
    double v = 0.0;
    if (r < nreg && n < N) {
        const double dd = 1.0 / sqrt((double)(2 * Nf));
        const bool is_sin = r >= Nf;
        const int64_t fn = is_sin ? r - (zerofreq ? Nf - 1 : Nf) : r;
        const double phi = (kTwoPi * f[fn]) * t[n];
        double s, c;
        sincos(phi, &s, &c);
        v = is_sin ? -s * dd : c * dd;
    }
    (void)nrows;
    D[r * ldn + n] = v;
}

// ---- a2 as a k-major panel P[n][ld]: the operand layout of the Gram kernel ---------------
__global__ void __launch_bounds__(256)
fourier_panel_kernel(const double *__restrict__ t, int64_t N, const double *__restrict__ f,
                     int64_t Nf, int zerofreq, double *__restrict__ P, int64_t ld) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * Nf) return;
    const int64_t n = idx / Nf, fn = idx - n * Nf;
    const double dd = 1.0 / sqrt((double)(2 * Nf));
    const int64_t sinoffset = zerofreq ? Nf - 1 : Nf;
    const int64_t nreg = zerofreq ? 2 * Nf - 1 : 2 * Nf;
    const double phi = (kTwoPi * f[fn]) * t[n];
    double s, c;
    sincos(phi, &s, &c);
    double *row = P + n * ld;
    row[fn] = c * dd;
    if (!(zerofreq && fn == 0)) row[fn + sinoffset] = -s * dd;
    for (int64_t cpad = nreg + fn; cpad < ld; cpad += Nf) row[cpad] = 0.0;  // zero the pad columns
}

// ---- batch of windows: P[q][r][ld], window q covers samples toff[q] .. toff[q]+n (src/windows.jl:33-34) -----
// rows r >= n (up to nrows) and columns >= Nreg are zero-filled
__global__ void __launch_bounds__(256)
window_panel_kernel(const double *__restrict__ t, const int64_t *__restrict__ toff, int64_t n, int64_t nrows,
                    const double *__restrict__ f, int64_t Nf, int zerofreq, double *__restrict__ P, int64_t ld) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nrows * Nf) return;
    const int64_t r = idx / Nf, fn = idx - r * Nf;
    const int q = blockIdx.y;
    const double dd = 1.0 / sqrt((double)(2 * Nf));
    const int64_t sinoffset = zerofreq ? Nf - 1 : Nf;
    const int64_t nreg = zerofreq ? 2 * Nf - 1 : 2 * Nf;
    double *row = P + ((int64_t)q * nrows + r) * ld;
    double s = 0, c = 0;
    if (r < n) {
        const double phi = (kTwoPi * f[fn]) * t[toff[q] + r];
        sincos(phi, &s, &c);
        c = c * dd; s = -s * dd;
    }
    row[fn] = c;
    if (!(zerofreq && fn == 0)) row[fn + sinoffset] = s;
    for (int64_t cpad = nreg + fn; cpad < ld; cpad += Nf) row[cpad] = 0.0;
}

// ---- a4: trig table T[n][f] = (cos(w_f x_n), -sin(w_f x_n)) --------------------------------
// conj(exp(i w x)) of src/lasso.jl:39; the minus sign is exact, so T.y * K == -(sin * K).
// EXACT (diagnostic, LPVS_PHASE=exact): the phase of the REAL product w_f x_n instead of its rounding fl(w_f x_n) -- the
// rounded product's sincos corrected to first order by the product's exact error d = w x - fl(w x) (|d| <= ulp(w x)/2, the neglected
// d^2/2 < 1e-19).  The reference rounds (src/lasso.jl:39, exp.(im .* w .* X)); the structured Gram (nudft.hip, nufft.hip) does not,
// and this knob lets a test separate that half-ulp of phase from everything else the two Gram paths do differently.
template <bool EXACT>
__global__ void __launch_bounds__(256)
trig_table_kernel(const double *__restrict__ X, int64_t N, const double *__restrict__ w, int64_t Nf,
                  double2 *__restrict__ T) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * Nf) return;
    const int64_t n = idx / Nf, fn = idx - n * Nf;
    const double phi = w[fn] * X[n];
    double s, c;
    sincos(phi, &s, &c);
    if (EXACT) {
        const double d = fma(w[fn], X[n], -phi);
        const double c2 = fma(-d, s, c), s2 = fma(d, c, s);
        c = c2; s = s2;
    }
    T[idx] = make_double2(c, -s);
}

// ---- a3: activation table K[n][ldk] -------------------------------------------------------
__device__ inline double sgn(double x) { return (double)((x > 0) - (x < 0)); }

__global__ void __launch_bounds__(256)
basis_table_kernel(const double *__restrict__ V, int64_t N, const double *__restrict__ vc, int64_t nb,
                   double gamma, int normalize, int coulomb, double *__restrict__ K, int64_t ldk) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const double v = V[n];
    double *row = K + n * ldk;
    double sum = 0;
    for (int64_t j = 0; j < nb; ++j) {
        const double d = v - vc[j];
        double k = exp(-gamma * (d * d));
        if (coulomb) k = k * (sgn(v) == sgn(vc[j]) ? 1.0 : 0.0);  // src/lsfft.jl:202
        row[j] = k;
        sum = j == 0 ? k : sum + k;  // sequential, as Base.sum on a short vector
    }
    if (normalize)
        for (int64_t j = 0; j < nb; ++j) row[j] = row[j] / sum;  // src/lsfft.jl:199
    for (int64_t j = nb; j < ldk; ++j) row[j] = 0.0;
}

__global__ void __launch_bounds__(256)
minmax_kernel(const double *__restrict__ V, int64_t N, double *__restrict__ part) {
    __shared__ double slo[256], shi[256], sam[256];
    double lo = INFINITY, hi = -INFINITY, am = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const double v = V[i];
        lo = fmin(lo, v); hi = fmax(hi, v); am = fmax(am, fabs(v));
    }
    slo[threadIdx.x] = lo; shi[threadIdx.x] = hi; sam[threadIdx.x] = am;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + s]);
            shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + s]);
            sam[threadIdx.x] = fmax(sam[threadIdx.x], sam[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = slo[0];
        part[blockIdx.x * 3 + 1] = shi[0];
        part[blockIdx.x * 3 + 2] = sam[0];
    }
}

// ---- a4+a5 materialised: Phi column-major N x 2*Nf*nb -------------------------------------
// One block = 64 samples x one frequency; the T / K rows are staged through LDS so that both
// the table reads (along f / j) and the Phi stores (along n) are coalesced.
__global__ void __launch_bounds__(256)
lpv_regressor_colmajor_kernel(const double2 *__restrict__ T, const double *__restrict__ K, int64_t ldk,
                              int64_t N, int64_t Nf, int64_t nb, int permuted,
                              double *__restrict__ Phi) {
    extern __shared__ double sh[];  // [64][nb] activations, then [64] cos, [64] -sin
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int64_t fn = blockIdx.y;
    double *sK = sh, *sc = sh + 64 * nb, *ss = sc + 64;
    for (int64_t i = threadIdx.x; i < 64 * nb; i += 256) {
        const int64_t r = i / nb, j = i - r * nb;
        sK[i] = (n0 + r < N) ? K[(n0 + r) * ldk + j] : 0.0;
    }
    if (threadIdx.x < 64 && n0 + threadIdx.x < N) {
        const double2 tv = T[(n0 + threadIdx.x) * Nf + fn];
        sc[threadIdx.x] = tv.x;
        ss[threadIdx.x] = tv.y;
    }
    __syncthreads();
    const int r = threadIdx.x & 63;
    if (n0 + r >= N) return;
    for (int64_t c = threadIdx.x >> 6; c < 2 * nb; c += 4) {
        const int64_t j = c < nb ? c : c - nb;
        const double val = (c < nb ? sc[r] : ss[r]) * sK[r * nb + j];
        int64_t col;
        if (permuted) col = fn * 2 * nb + c;                       // inds of src/lasso.jl:47
        else col = (c < nb ? 0 : Nf * nb) + fn + j * Nf;           // [Re As, Im As]
        Phi[(n0 + r) + col * N] = val;
    }
}

// column-major m x n  ->  k-major panel P[m][ld] (columns >= n zero-filled); 32x32 LDS tiles
__global__ void __launch_bounds__(256)
transpose_to_panel_kernel(const double *__restrict__ A, int64_t m, int64_t n, double *__restrict__ P, int64_t ld) {
    __shared__ double tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {  // read along rows (contiguous in column-major)
        const int64_t r = r0 + tx, c = c0 + j;
        tile[j][tx] = (r < m && c < n) ? A[r + c * m] : 0.0;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {  // write along columns (contiguous in the panel)
        const int64_t r = r0 + i, c = c0 + tx;
        if (r < m && c < ld) P[r * ld + c] = tile[tx][i];
    }
}

}  // namespace

int32_t launch_transpose_to_panel(const double *A, int64_t m, int64_t n, double *P, int64_t ld, hipStream_t s) {
    if (m == 0) return LPVS_OK;
    dim3 grid((unsigned)ceil_div(m, 32), (unsigned)ceil_div(ld, 32));
    hipLaunchKernelGGL(transpose_to_panel_kernel, grid, dim3(256), 0, s, A, m, n, P, ld);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_fourier_regressor_colmajor(const double *t, int64_t N, const double *f, int64_t Nf,
                                          int zerofreq, double *A, hipStream_t s) {
    if (N == 0 || Nf == 0) return LPVS_OK;
    dim3 grid((unsigned)ceil_div(N, 256), (unsigned)Nf);
    hipLaunchKernelGGL(fourier_regressor_colmajor_kernel, grid, dim3(256), 0, s, t, N, f, Nf, zerofreq, A);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_fourier_dual_panel(const double *t, int64_t N, const double *f, int64_t Nf, int zerofreq, double *D, int64_t ldn,
                                  int64_t nrows, hipStream_t s) {
    dim3 grid((unsigned)ceil_div(ldn, 256), (unsigned)nrows);
    hipLaunchKernelGGL(fourier_dual_panel_kernel, grid, dim3(256), 0, s, t, N, f, Nf, zerofreq, D, ldn, nrows);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_fourier_panel(const double *t, int64_t N, const double *f, int64_t Nf, int zerofreq,
                             double *P, int64_t ld, hipStream_t s) {
    if (N == 0 || Nf == 0) return LPVS_OK;
    hipLaunchKernelGGL(fourier_panel_kernel, dim3((unsigned)ceil_div(N * Nf, 256)), dim3(256), 0, s, t, N, f,
                       Nf, zerofreq, P, ld);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_window_panels(const double *t, const int64_t *toff_dev, int nbatch, int64_t n, int64_t nrows, const double *f,
                             int64_t Nf, int zerofreq, double *P, int64_t ld, hipStream_t s) {
    dim3 grid((unsigned)ceil_div(nrows * Nf, 256), (unsigned)nbatch);
    hipLaunchKernelGGL(window_panel_kernel, grid, dim3(256), 0, s, t, toff_dev, n, nrows, f, Nf, zerofreq, P, ld);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_trig_table(const double *X, int64_t N, const double *w, int64_t Nf, double2 *T,
                          hipStream_t s) {
    if (N == 0 || Nf == 0) return LPVS_OK;
    const char *e = experiment_env("LPVS_PHASE");            // diagnostic (read per call): "exact" = phases of the unrounded products w x
    if (e != nullptr && std::string(e) == "exact")
        hipLaunchKernelGGL(trig_table_kernel<true>, dim3((unsigned)ceil_div(N * Nf, 256)), dim3(256), 0, s, X, N, w, Nf, T);
    else
        hipLaunchKernelGGL(trig_table_kernel<false>, dim3((unsigned)ceil_div(N * Nf, 256)), dim3(256), 0, s, X, N, w, Nf, T);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_basis_table(const double *V, int64_t N, const double *vc, int64_t nb, double gamma,
                           int normalize, int coulomb, double *K, int64_t ldk, hipStream_t s) {
    if (N == 0) return LPVS_OK;
    hipLaunchKernelGGL(basis_table_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, V, N, vc, nb,
                       gamma, normalize, coulomb, K, ldk);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t device_minmax(const double *V, int64_t N, double *lo, double *hi, double *amax, hipStream_t s) {
    const int blocks = (int)(N < 256 * 256 ? ceil_div(N, 256) : 256);
    DevBuf part;
    LPVS_TRY(part.alloc(sizeof(double) * 3 * blocks));
    hipLaunchKernelGGL(minmax_kernel, dim3(blocks), dim3(256), 0, s, V, N, part.as<double>());
    LPVS_HIP(hipGetLastError());
    std::string tmp(sizeof(double) * 3 * blocks, '\0');
    double *h = reinterpret_cast<double *>(&tmp[0]);
    LPVS_HIP(hipMemcpyAsync(h, part.p, sizeof(double) * 3 * blocks, hipMemcpyDeviceToHost, s));
    LPVS_HIP(hipStreamSynchronize(s));
    *lo = INFINITY; *hi = -INFINITY; *amax = 0;
    for (int i = 0; i < blocks; ++i) {
        *lo = fmin(*lo, h[3 * i]); *hi = fmax(*hi, h[3 * i + 1]); *amax = fmax(*amax, h[3 * i + 2]);
    }
    return LPVS_OK;
}

int32_t launch_lpv_regressor_colmajor(const double2 *T, const double *K, int64_t ldk, int64_t N, int64_t Nf,
                                      int64_t nb, int permuted, double *Phi, hipStream_t s) {
    if (N == 0 || Nf == 0) return LPVS_OK;
    const size_t lds = sizeof(double) * (64 * nb + 128);
    dim3 grid((unsigned)ceil_div(N, 64), (unsigned)Nf);
    hipLaunchKernelGGL(lpv_regressor_colmajor_kernel, grid, dim3(256), lds, s, T, K, ldk, N, Nf, nb, permuted, Phi);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

}  // namespace lpvs
