// nudft.hip -- structured Gram for frequency grids that are arithmetic progressions.
//
// With w_f = a + f*D + eps_f (eps_f = the rounding residuals of the caller's doubles), products of the
// regressor's trig factors (src/lasso.jl:39) reduce by the product-to-sum identities to
//     cos/sin((w_f - w_f') x)   and   cos/sin((w_f + w_f') x),
// whose frequencies are m*D + (eps_f - eps_f') and 2a + s*D + (eps_f + eps_f'), m = f-f', s = f+f'.  So
//     G[(f,c,j),(f',c',j')] = sum_n Phi[n,(f,c,j)] Phi[n,(f',c',j')]
// is a signed half-sum of 3Nf-1 non-uniform Fourier sums per activation pair {j,j'}:
//     C(omega)[q] = sum_n KK[n][q] cos(omega x_n),   S(omega)[q] = sum_n KK[n][q] sin(omega x_n),
// plus their x-weighted twins (the omega-derivatives), which carry the first-order correction in eps.
// The neglected second order is (eps * x)^2 / 2 <= 5e-15 under the admission test |eps| max|x| <= 1e-7.
// Slot frequencies are kept in double-double and the phase omega*x is formed with an FMA-exact product, so
// the sums are those of the exact real frequencies w_f +- w_f' of the given doubles.
//
// Work: N*(3Nf-1) sincos + 4*N*(3Nf-1)*P fma  (cfg3: 1.6e9 sincos + 2.3e11 flop) instead of the 4.2e13 flop
// of the dense symmetric-pair contraction.  VALU / transcendental bound; deterministic (chunk partials are
// summed in fixed order).
#include "lpvs_internal.h"

namespace lpvs {

namespace {

constexpr int SLOTS = 64;   // slots per workgroup
constexpr int QPT = 9;      // weights (activation pairs) per thread
constexpr int RB = 4;       // rows per inner block (256 threads = 64 slots x 4 rows of sincos)

// out[chunk][slot][q][v], v = {cos, sin, x*cos, x*sin} (NV = 4) or {cos, sin} (NV = 2)
template <int NV>
__global__ void __launch_bounds__(256)
nudft_accumulate_kernel(const double *__restrict__ x, const double *__restrict__ y, int64_t N, const double *__restrict__ Wt,
                        int64_t ldw, int nq, const double *__restrict__ om_hi, const double *__restrict__ om_lo, int nslots,
                        int64_t rows_per_chunk, double *__restrict__ out) {
    __shared__ double trig[RB][SLOTS][4];
    __shared__ double wrow[RB][40];
    const int sl = threadIdx.x & (SLOTS - 1), g = threadIdx.x >> 6;        // slot within block; row (phase 1) / q-group (phase 2)
    const int slot = blockIdx.x * SLOTS + sl;
    const bool live = slot < nslots;
    const double wh = live ? om_hi[slot] : 0.0, wl = (live && om_lo) ? om_lo[slot] : 0.0;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < N ? r0 + rows_per_chunk : N;
    const int qbase = blockIdx.z * (4 * QPT);                               // pair block (nq > 36: several passes over the samples)
    const int nqb = nq - qbase < 4 * QPT ? nq - qbase : 4 * QPT;            // pairs of this block
    const int q0 = g * QPT;
    double acc[QPT][NV];
#pragma unroll
    for (int q = 0; q < QPT; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[q][v] = 0.0;
    for (int64_t rb = r0; rb < r1; rb += RB) {
        {   // phase 1: thread (slot, row g) evaluates the trig factor of that (sample, frequency)
            const int64_t r = rb + g;
            double c = 0.0, s = 0.0, xv = 0.0;
            if (r < r1) {
                xv = x[r];
                const double p = wh * xv;                       // rounded product
                const double d = fma(wh, xv, -p) + wl * xv;     // its exact error + the low word: omega*x = p + d
                sincos(p, &s, &c);
                const double c2 = fma(-d, s, c), s2 = fma(d, c, s);
                c = c2; s = s2;
            }
            trig[g][sl][0] = c; trig[g][sl][1] = s;
            if (NV == 4) { trig[g][sl][2] = xv * c; trig[g][sl][3] = xv * s; }
            if (threadIdx.x < RB * 40) {                        // stage the weights of the RB rows
                const int rr = threadIdx.x / 40, q = threadIdx.x % 40;
                const int64_t r2 = rb + rr;
                double wv = 0.0;
                if (r2 < r1 && q < nqb) { wv = Wt[r2 * ldw + qbase + q]; if (y) wv = wv * y[r2]; }
                wrow[rr][q] = wv;
            }
        }
        __syncthreads();
        if (q0 < nqb) {   // phase 2: thread (slot, q-group g) accumulates its QPT weights over the RB rows
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                double tv[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) tv[v] = trig[rr][sl][v];
#pragma unroll
                for (int q = 0; q < QPT; ++q) {
                    const double wv = wrow[rr][q0 + q];         // zero beyond nq
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[q][v] = fma(wv, tv[v], acc[q][v]);
                }
            }
        }
        __syncthreads();
    }
    if (live && q0 < nqb) {
        double *o = out + (((int64_t)blockIdx.y * nslots + slot) * nq + qbase) * NV;
#pragma unroll
        for (int q = 0; q < QPT; ++q)
            if (q0 + q < nqb)
#pragma unroll
                for (int v = 0; v < NV; ++v) o[(q0 + q) * NV + v] = acc[q][v];
    }
}

__global__ void __launch_bounds__(256)
nudft_reduce_kernel(const double *__restrict__ part, int nchunks, int64_t count, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double s = 0;
    for (int c = 0; c < nchunks; ++c) s += part[(int64_t)c * count + i];
    out[i] = s;
}

// G[a][b] = G[b][a], a >= b, from the slot tables tab[slot][q][4]; slots 0..Nf-1 are the differences m = f-f',
// slots Nf..3Nf-2 the sums s = f+f'.
__global__ void __launch_bounds__(256)
ap_assemble_kernel(const double *__restrict__ tab, const double *__restrict__ eps, int Nf, int nb, int P, int64_t n,
                   double *__restrict__ G, int64_t ldg) {
    const int64_t ga = blockIdx.y;
    const int64_t gb = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gb > ga || ga >= n) return;
    const int g2 = 2 * nb;
    const int fa = (int)(ga / g2), ra = (int)(ga - (int64_t)fa * g2), ca = ra >= nb, ja0 = ra - ca * nb;
    const int fb = (int)(gb / g2), rbm = (int)(gb - (int64_t)fb * g2), cb = rbm >= nb, jb0 = rbm - cb * nb;
    int ja = ja0, jb = jb0;
    if (ja > jb) { const int t = ja; ja = jb; jb = t; }
    const int q = ja * nb - ja * (ja - 1) / 2 + (jb - ja);
    const int m = fa - fb, s = fa + fb;                     // fa >= fb because a >= b
    const double dm = eps[fa] - eps[fb], dp = eps[fa] + eps[fb];
    const double *tm = tab + ((int64_t)m * P + q) * 4;
    const double *tp = tab + ((int64_t)(Nf + s) * P + q) * 4;
    const double cm = fma(-dm, tm[3], tm[0]), sm = fma(dm, tm[2], tm[1]);   // first order in the residuals
    const double cp = fma(-dp, tp[3], tp[0]), sp = fma(dp, tp[2], tp[1]);
    double v;
    if (!ca && !cb) v = 0.5 * (cm + cp);                    //  cos *  cos
    else if (ca && cb) v = 0.5 * (cm - cp);                 // -sin * -sin
    else if (!ca && cb) v = -0.5 * (sp - sm);               //  cos(w_f x) * -sin(w_f' x)
    else v = -0.5 * (sp + sm);                              // -sin(w_f x) *  cos(w_f' x)
    G[ga * ldg + gb] = v;
    G[gb * ldg + ga] = v;
}

// b[f*2nb + j] = sum y K_j cos(w_f x),  b[f*2nb + nb + j] = -sum y K_j sin(w_f x)   from tab[f][j][2]
__global__ void __launch_bounds__(256)
ap_rhs_kernel(const double *__restrict__ tab, int Nf, int nb, double *__restrict__ b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nf * nb) return;
    const int f = i / nb, j = i - f * nb;
    b[(int64_t)f * 2 * nb + j] = tab[((int64_t)f * nb + j) * 2 + 0];
    b[(int64_t)f * 2 * nb + nb + j] = -tab[((int64_t)f * nb + j) * 2 + 1];
}

}  // namespace

size_t nudft_chunks(int64_t N) { return (size_t)ceil_div(N, 8192); }
size_t nudft_partial_bytes(int64_t N, int64_t nslots, int64_t nq, int nv) {
    return sizeof(double) * nudft_chunks(N) * (size_t)nslots * (size_t)nq * (size_t)nv;
}

// tab[slot][q][nv] = sum_n (y_n) Wt[n][q] {cos, sin, (x cos, x sin)}(omega_slot x_n); om_lo may be null (plain phase)
int32_t launch_nudft(const double *x, const double *y, int64_t N, const double *Wt, int64_t ldw, int nq, const double *om_hi,
                     const double *om_lo, int nslots, int nv, double *partial, double *tab, hipStream_t s) {
    const int64_t rpc = 8192;
    const unsigned nchunks = (unsigned)nudft_chunks(N);
    dim3 grid((unsigned)ceil_div(nslots, SLOTS), nchunks, (unsigned)ceil_div(nq, 4 * QPT));
    if (nv == 4) hipLaunchKernelGGL(nudft_accumulate_kernel<4>, grid, dim3(256), 0, s, x, y, N, Wt, ldw, nq, om_hi, om_lo, nslots, rpc, partial);
    else hipLaunchKernelGGL(nudft_accumulate_kernel<2>, grid, dim3(256), 0, s, x, y, N, Wt, ldw, nq, om_hi, om_lo, nslots, rpc, partial);
    LPVS_HIP(hipGetLastError());
    const int64_t count = (int64_t)nslots * nq * nv;
    hipLaunchKernelGGL(nudft_reduce_kernel, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, s, partial, (int)nchunks, count, tab);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_ap_assemble(const double *tab, const double *eps, int64_t Nf, int64_t nb, int64_t n, double *G, int64_t ldg, hipStream_t s) {
    const int P = (int)(nb * (nb + 1) / 2);
    dim3 grid((unsigned)ceil_div(n, 256), (unsigned)n);
    hipLaunchKernelGGL(ap_assemble_kernel, grid, dim3(256), 0, s, tab, eps, (int)Nf, (int)nb, P, n, G, ldg);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_ap_rhs(const double *tab, int64_t Nf, int64_t nb, double *b, hipStream_t s) {
    hipLaunchKernelGGL(ap_rhs_kernel, dim3((unsigned)ceil_div(Nf * nb, 256)), dim3(256), 0, s, tab, (int)Nf, (int)nb, b);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

}  // namespace lpvs
