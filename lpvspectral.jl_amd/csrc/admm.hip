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
symv_kernel(const double *__restrict__ M, int64_t np, const double *__restrict__ rhs, double *__restrict__ x,
            const AdmmStatus *status) {
    if (status != nullptr && status->converged) return;
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
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.np; i += (int64_t)gridDim.x * 256) {
        const bool ok = i < p.n;
        const double xi = ok ? p.x[i] : 0.0;
        p.x[i] = xi;
        p.z[i] = xi;                                    // z = copy(x)      src/lasso.jl:146
        p.u[i] = 0.0;                                   // u = zeros        src/lasso.jl:147
        p.rhs[i] = ok ? p.b[i] + (xi - 0.0) / p.mu : 0.0;  // b + (z-u)/mu
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.status->iters = 0; p.status->converged = 0; p.status->nxz = 0.0;
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

// Threshold key of the r-th largest |v| (radix select, 8 bits per pass) and the number of
// equal-key elements to keep (lowest indices first).  All 1024 threads participate.
__device__ void topr_threshold(const double *v, int64_t n, int64_t r, unsigned long long *thr,
                               long long *keep_equal, unsigned int *hist /*[256]*/, long long *shll /*[2]*/) {
    unsigned long long prefix = 0, mask = 0;
    long long remaining = r;  // how many of the candidates (matching prefix) we still need
    for (int pass = 7; pass >= 0; --pass) {
        const int shift = pass * 8;
        for (int i = threadIdx.x; i < 256; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int64_t i = threadIdx.x; i < n; i += 1024) {
            const unsigned long long k = abs_key(v[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long need = remaining;
            int b = 255;
            for (; b > 0; --b) {
                if ((long long)hist[b] >= need) break;
                need -= hist[b];
            }
            shll[0] = b; shll[1] = need;
        }
        __syncthreads();
        prefix |= ((unsigned long long)shll[0]) << shift;
        mask |= 255ull << shift;
        remaining = shll[1];
        __syncthreads();
    }
    *thr = prefix;
    *keep_equal = remaining;
}

__global__ void __launch_bounds__(1024)
admm_prox_kernel(AdmmParams p) {
    __shared__ double sh[16];
    __shared__ unsigned int hist[256];
    __shared__ long long shll[2];
    __shared__ int scan[1024];
    if (p.status->converged) return;
    const int64_t n = p.n;
    const double mu = p.mu;
    double ss = 0;  // sum (x-z)^2 over this thread's elements

    auto finish = [&](int64_t i, double xi, double ui, double zi) {
        const double d = xi - zi;              // tmp = x - z            src/lasso.jl:154
        const double un = ui + d;              // u += tmp               src/lasso.jl:155
        p.z[i] = zi; p.u[i] = un;
        p.rhs[i] = p.b[i] + (zi - un) / mu;    // next x-update: b + (z-u)/mu
        ss = fma(d, d, ss);
    };

    if (p.prox_kind == LPVS_PROX_L1) {
        const double gl = mu * p.prox_param;
        for (int64_t i = threadIdx.x; i < n; i += 1024) {
            const double xi = p.x[i], ui = p.u[i], v = xi + ui;
            const double zi = v + (v <= -gl ? gl : (v >= gl ? -gl : -v));
            finish(i, xi, ui, zi);
        }
    } else if (p.prox_kind == LPVS_PROX_L0) {
        const double th = sqrt(2.0 * mu * p.prox_param);
        for (int64_t i = threadIdx.x; i < n; i += 1024) {
            const double xi = p.x[i], ui = p.u[i], v = xi + ui;
            finish(i, xi, ui, fabs(v) > th ? v : 0.0);
        }
    } else if (p.prox_kind == LPVS_PROX_GROUP_L2) {
        const int64_t gl = p.group_len, ng = n / gl;
        const double lm = p.prox_param * mu;
        for (int64_t g = threadIdx.x; g < ng; g += 1024) {
            double s2 = 0;
            for (int64_t q = 0; q < gl; ++q) { const double v = p.x[g * gl + q] + p.u[g * gl + q]; s2 += v * v; }
            double scale = 1.0 - lm / sqrt(s2);   // s2 == 0 -> -inf -> 0
            if (!(scale > 0)) scale = 0.0;
            for (int64_t q = 0; q < gl; ++q) {
                const int64_t i = g * gl + q;
                const double xi = p.x[i], ui = p.u[i];
                finish(i, xi, ui, scale * (xi + ui));
            }
        }
        for (int64_t i = ng * gl + threadIdx.x; i < n; i += 1024) {  // entries outside every slice: prox leaves z
            const double xi = p.x[i], ui = p.u[i];
            finish(i, xi, ui, p.z[i]);
        }
    } else {  // LPVS_PROX_BALL_L0
        double *v = p.scratch;
        for (int64_t i = threadIdx.x; i < n; i += 1024) v[i] = p.x[i] + p.u[i];
        __syncthreads();
        long long r = (long long)p.prox_param;
        if (r >= n) {
            for (int64_t i = threadIdx.x; i < n; i += 1024) finish(i, p.x[i], p.u[i], v[i]);
        } else if (r <= 0) {
            for (int64_t i = threadIdx.x; i < n; i += 1024) finish(i, p.x[i], p.u[i], 0.0);
        } else {
            unsigned long long thr; long long keep_eq;
            topr_threshold(v, n, r, &thr, &keep_eq, hist, shll);
            long long eq_seen = 0;  // equal-key elements at lower indices (uniform across threads)
            for (int64_t base = 0; base < n; base += 1024) {
                const int64_t i = base + threadIdx.x;
                const bool in = i < n;
                const unsigned long long k = in ? abs_key(v[i]) : 0ull;
                const int iseq = in && k == thr;
                // inclusive block scan of iseq (Hillis-Steele in LDS; n/1024 rounds only)
                scan[threadIdx.x] = iseq;
                __syncthreads();
                for (int o = 1; o < 1024; o <<= 1) {
                    const int t = threadIdx.x >= o ? scan[threadIdx.x - o] : 0;
                    __syncthreads();
                    scan[threadIdx.x] += t;
                    __syncthreads();
                }
                const long long rank_eq = eq_seen + scan[threadIdx.x];  // 1-based rank among equals
                const long long tot = scan[1023];
                if (in) {
                    const bool keep = k > thr || (iseq && rank_eq <= keep_eq);
                    finish(i, p.x[i], p.u[i], keep ? v[i] : 0.0);
                }
                eq_seen += tot;
                __syncthreads();
            }
        }
    }

    const double tot = block_sum_1024(ss, sh);
    if (threadIdx.x == 0) {
        const double nxz = sqrt(tot);          // norm(tmp)               src/lasso.jl:157
        p.status->iters += 1;
        p.status->nxz = nxz;
        if (nxz < p.tol) p.status->converged = 1;  //                    src/lasso.jl:164
    }
}

}  // namespace

int32_t launch_admm_init(const AdmmParams &p, hipStream_t s) {
    hipLaunchKernelGGL(admm_init_kernel, dim3((unsigned)ceil_div(p.np, 256)), dim3(256), 0, s, p);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

static void launch_symv_raw(const double *M, int64_t np, const double *rhs, double *x, const AdmmStatus *st,
                            hipStream_t s) {
    if (np >= 4096)
        hipLaunchKernelGGL(symv_kernel<4>, dim3((unsigned)ceil_div(np, 16)), dim3(256), 0, s, M, np, rhs, x, st);
    else
        hipLaunchKernelGGL(symv_kernel<1>, dim3((unsigned)ceil_div(np, 4)), dim3(256), 0, s, M, np, rhs, x, st);
}

int32_t launch_symv(const double *M, int64_t np, const double *rhs, double *x, hipStream_t s) {
    launch_symv_raw(M, np, rhs, x, nullptr, s);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_admm_iterations(const AdmmParams &p, int64_t iters, hipStream_t s) {
    for (int64_t i = 0; i < iters; ++i) {
        launch_symv_raw(p.M, p.np, p.rhs, p.x, p.status, s);
        hipLaunchKernelGGL(admm_prox_kernel, dim3(1), dim3(1024), 0, s, p);
    }
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

}  // namespace lpvs
