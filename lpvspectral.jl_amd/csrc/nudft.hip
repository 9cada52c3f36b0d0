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
// The right-hand side b = Phi' y uses the same sums at the Nf slots a + f*D with weights y K_j.
//
// Slot frequencies are exact arithmetic progressions kept in double-double, eight slots to a group: per sample the
// kernel evaluates sincos only for the group anchors (omega_{8a} x, phase formed with an FMA-exact product) and for
// the eight in-group offsets (b D x), and gets every slot by one complex multiplication
//     cis(omega_{8a+b} x) = cis(omega_{8a} x) cis(b D x)
// -- 16 sincos per 64 slots instead of 64.  The weights of a (sample, pair) are wave-uniform and come through
// scalar loads, so the inner loop is one LDS read of the slot's four trig values and 4 FMAs per pair.
//
// Work: 4*N*(3Nf-1)*P FMAs (cfg3: 2.3e11) + N*(3Nf-1)/4 sincos instead of the 4.2e13 flop of the dense
// symmetric-pair contraction.  VALU bound; deterministic (chunk partials are summed in fixed order).
#include "lpvs_internal.h"

#include <cstdlib>

namespace lpvs {

namespace {

constexpr int RB = 16;      // samples per inner block

// out[chunk][slot][q][4] = sum over the chunk's samples of Wt[n][q] (y_n) {cos, sin, x cos, x sin}(omega_slot x_n)
//
// A workgroup owns 64*S consecutive slots (8*S groups of 8) and one q-block of 4*QPT pairs; thread (lane sl, wave g)
// accumulates slots sl, sl+64, ... (S of them) against the wave's QPT pairs.  Per inner block of 16 samples:
//   step A  one sincos per thread: the 8*S group anchors and the 8 in-group offsets of each sample; the samples'
//           weights are staged in LDS
//   step B  every (sample, slot) by one complex product -> trig[sample][slot] = y {c, s, x c, x s}
//   step C  per sample: S LDS reads of 32 B (trig) + QPT/2 reads of 16 B (weights, same address in every lane),
//           4*S*QPT FMAs.  S = 2 keeps the LDS behind the FMA pipe (4 (16 S + 4 QPT) <= 16 S QPT cycles).
template <int QPT, int S>
__global__ void __launch_bounds__(256)
nudft_accumulate_kernel(const double *__restrict__ x, const double *__restrict__ y, int64_t N, const double *__restrict__ Wt,
                        int64_t ldw, int nq, const double *__restrict__ om_hi, const double *__restrict__ om_lo, int nslots,
                        const ApStep step, double *__restrict__ out, const int64_t *__restrict__ seg, int64_t rows_per_chunk) {
    constexpr int SLOTS = 64 * S, NA = 8 * S;                    // slots and anchors per workgroup
    constexpr int WSTR = (QPT + 1) & ~1;                         // weight stride of a wave inside a staged sample (16-B aligned)
    constexpr int NEF = NA + 8;                                  // sincos per sample: anchors + offsets
    __shared__ __attribute__((aligned(16))) double wrow[2][RB][4 * WSTR];
    __shared__ __attribute__((aligned(16))) double ef[2][RB][NEF][2];
    __shared__ double xs[2][RB], ys[2][RB];
    const int t = threadIdx.x;
    const int sl = t & 63;
    const int g = __builtin_amdgcn_readfirstlane(t >> 6);        // wave = q-group of the accumulation phase
    const int slot0 = blockIdx.x * SLOTS;
    // sample range of this y-block: a chunk of the signal, or (windows) seg = {begin, end, first sample of the window}:
    // weights are indexed relative to the window start, x and y by the absolute sample
    int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    int64_t r1 = r0 + rows_per_chunk < N ? r0 + rows_per_chunk : N;
    int64_t wbase = 0;
    if (seg != nullptr) { r0 = seg[3 * blockIdx.y]; r1 = seg[3 * blockIdx.y + 1]; wbase = seg[3 * blockIdx.y + 2]; }
    const int qbase = blockIdx.z * (4 * QPT);
    const int nqb = nq - qbase < 4 * QPT ? nq - qbase : 4 * QPT;  // pairs of this q-block
    const int q0 = g * QPT;
    const int nvalid = nqb - q0 < QPT ? nqb - q0 : QPT;           // pairs of this wave (<= 0: the wave only helps with step A)

    double acc[S][QPT][4];
#pragma unroll
    for (int u = 0; u < S; ++u)
#pragma unroll
        for (int q = 0; q < QPT; ++q)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][q][v] = 0.0;

    int buf = 0;
    for (int64_t rb = r0; rb < r1; rb += RB, buf ^= 1) {
        // step A: sincos of (sample, anchor | offset), RB*NEF of them over 256 threads; tables are double-buffered, so
        // one barrier per block orders both this block's reads and the overwrite two blocks later
        for (int e = t; e < RB * NEF; e += 256) {
            const int pr = e / NEF, pk = e - pr * NEF;
            double wh, wl;
            if (pk < NA) {
                const int as = slot0 + 8 * pk;
                const bool ok = as < nslots;
                wh = ok ? om_hi[as] : 0.0; wl = ok ? om_lo[as] : 0.0;
            } else {
                wh = step.hi[pk - NA]; wl = step.lo[pk - NA];
            }
            const int64_t r = rb + pr;
            double c = 1.0, sn = 0.0;
            if (r < r1) {
                const double xv = x[r];
                const double p = wh * xv;                       // rounded product
                const double d = fma(wh, xv, -p) + wl * xv;     // its exact error + the low word: omega*x = p + d
                sincos(p, &sn, &c);
                const double c2 = fma(-d, sn, c), s2 = fma(d, c, sn);
                c = c2; sn = s2;
                if (pk == 0) { xs[buf][pr] = xv; ys[buf][pr] = y ? y[r] : 1.0; }
            } else if (pk == 0) {
                xs[buf][pr] = 0.0; ys[buf][pr] = 0.0;           // samples past the end contribute nothing
            }
            ef[buf][pr][pk][0] = c; ef[buf][pr][pk][1] = sn;
        }
        for (int e = t; e < RB * 4 * QPT; e += 256) {           // weights of the block: [sample][wave][QPT (+pad)]
            const int pr = e / (4 * QPT), q = e - pr * (4 * QPT);
            const int64_t r = rb + pr;
            const double wv = (r < r1 && q < nqb) ? (Wt ? Wt[(r - wbase) * ldw + qbase + q] : 1.0) : 0.0;   // Wt == NULL: unit weights
            wrow[buf][pr][(q / QPT) * WSTR + q % QPT] = wv;
        }
        __syncthreads();
        if (nvalid > 0) {   // steps B + C: the slot's cis by one complex product (recomputed by every q-group: 14 ops against
                            // 16 QPT FMAs), then the wave's QPT weights
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                const double yv = ys[buf][rr], xv = xs[buf][rr];
                double tv[S][4];
#pragma unroll
                for (int u = 0; u < S; ++u) {
                    const int ls = sl + 64 * u;
                    const double2 e2 = *reinterpret_cast<const double2 *>(&ef[buf][rr][ls >> 3][0]);
                    const double2 f2 = *reinterpret_cast<const double2 *>(&ef[buf][rr][NA + (ls & 7)][0]);
                    const double c = fma(e2.x, f2.x, -(e2.y * f2.y)) * yv, sn = fma(e2.y, f2.x, e2.x * f2.y) * yv;
                    tv[u][0] = c; tv[u][1] = sn; tv[u][2] = xv * c; tv[u][3] = xv * sn;
                }
                double wq[WSTR];
#pragma unroll
                for (int q = 0; q < WSTR; q += 2) {
                    const double2 w2 = *reinterpret_cast<const double2 *>(&wrow[buf][rr][g * WSTR + q]);
                    wq[q] = w2.x; wq[q + 1] = w2.y;
                }
#pragma unroll
                for (int q = 0; q < QPT; ++q)
#pragma unroll
                    for (int u = 0; u < S; ++u) {
                        acc[u][q][0] = fma(wq[q], tv[u][0], acc[u][q][0]);
                        acc[u][q][1] = fma(wq[q], tv[u][1], acc[u][q][1]);
                        acc[u][q][2] = fma(wq[q], tv[u][2], acc[u][q][2]);
                        acc[u][q][3] = fma(wq[q], tv[u][3], acc[u][q][3]);
                    }
            }
        }
    }
    if (nvalid > 0) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int slot = slot0 + sl + 64 * u;
            if (slot >= nslots) continue;
            double *o = out + (((int64_t)blockIdx.y * nslots + slot) * nq + qbase + q0) * 4;
#pragma unroll
            for (int q = 0; q < QPT; ++q)
                if (q < nvalid) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) o[q * 4 + v] = acc[u][q][v];
                }
        }
    }
}


// Single weight per sample (Fourier problems: W_n or 1, times y_n for the right-hand side).  With one weight there is
// nothing to share between q-groups, so the trig values never go through LDS: per block of 16 samples the workgroup
// evaluates the anchors / offsets (step A, as above), then wave g takes samples g, g+4, ... and every thread forms its
// S slots' cis by one complex product and accumulates in registers.  The four waves' sums are combined in wave order
// at the end of the chunk.  One barrier per block (the anchor table is double-buffered).
template <int S>
__global__ void __launch_bounds__(256)
nudft_single_kernel(const double *__restrict__ x, const double *__restrict__ y, int64_t N, const double *__restrict__ Wt, int64_t ldw,
                    const double *__restrict__ om_hi, const double *__restrict__ om_lo, int nslots, const ApStep step,
                    double *__restrict__ out, const int64_t *__restrict__ seg, int64_t rows_per_chunk) {
    constexpr int NA = 8 * S, NEF = NA + 8;
    __shared__ __attribute__((aligned(16))) double ef[2][RB][NEF][2];
    __shared__ double xs[2][RB], ws[2][RB];
    __shared__ __attribute__((aligned(16))) double comb[3][64 * S][4];
    const int t = threadIdx.x, sl = t & 63;
    const int g = __builtin_amdgcn_readfirstlane(t >> 6);
    const int slot0 = blockIdx.x * 64 * S;
    int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    int64_t r1 = r0 + rows_per_chunk < N ? r0 + rows_per_chunk : N;
    int64_t wbase = 0;
    if (seg != nullptr) { r0 = seg[3 * blockIdx.y]; r1 = seg[3 * blockIdx.y + 1]; wbase = seg[3 * blockIdx.y + 2]; }
    double acc[S][4];
#pragma unroll
    for (int u = 0; u < S; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
    int buf = 0;
    for (int64_t rb = r0; rb < r1; rb += RB, buf ^= 1) {
        for (int e = t; e < RB * NEF; e += 256) {               // step A
            const int pr = e / NEF, pk = e - pr * NEF;
            double wh, wl;
            if (pk < NA) {
                const int as = slot0 + 8 * pk;
                const bool ok = as < nslots;
                wh = ok ? om_hi[as] : 0.0; wl = ok ? om_lo[as] : 0.0;
            } else {
                wh = step.hi[pk - NA]; wl = step.lo[pk - NA];
            }
            const int64_t r = rb + pr;
            double c = 1.0, sn = 0.0;
            if (r < r1) {
                const double xv = x[r];
                const double p = wh * xv;
                const double d = fma(wh, xv, -p) + wl * xv;     // omega*x = p + d
                sincos(p, &sn, &c);
                const double c2 = fma(-d, sn, c), s2 = fma(d, c, sn);
                c = c2; sn = s2;
                if (pk == 0) { xs[buf][pr] = xv; ws[buf][pr] = (Wt ? Wt[(r - wbase) * ldw] : 1.0) * (y ? y[r] : 1.0); }
            } else if (pk == 0) {
                xs[buf][pr] = 0.0; ws[buf][pr] = 0.0;           // samples past the end contribute nothing
            }
            ef[buf][pr][pk][0] = c; ef[buf][pr][pk][1] = sn;
        }
        __syncthreads();   // also orders this block's reads of buffer `buf` against the writes two blocks later
#pragma unroll
        for (int i = 0; i < RB / 4; ++i) {                      // steps B + C fused, samples g, g+4, ...
            const int rr = g + 4 * i;
            const double wv = ws[buf][rr], xv = xs[buf][rr];
#pragma unroll
            for (int u = 0; u < S; ++u) {
                const int ls = sl + 64 * u;
                const double2 e2 = *reinterpret_cast<const double2 *>(&ef[buf][rr][ls >> 3][0]);
                const double2 f2 = *reinterpret_cast<const double2 *>(&ef[buf][rr][NA + (ls & 7)][0]);
                const double c = fma(e2.x, f2.x, -(e2.y * f2.y)) * wv, sn = fma(e2.y, f2.x, e2.x * f2.y) * wv;
                acc[u][0] += c; acc[u][1] += sn;
                acc[u][2] = fma(xv, c, acc[u][2]); acc[u][3] = fma(xv, sn, acc[u][3]);
            }
        }
    }
    __syncthreads();
    if (g > 0) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            double4 o; o.x = acc[u][0]; o.y = acc[u][1]; o.z = acc[u][2]; o.w = acc[u][3];
            *reinterpret_cast<double4 *>(&comb[g - 1][sl + 64 * u][0]) = o;
        }
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int slot = slot0 + sl + 64 * u;
            if (slot >= nslots) continue;
            double *o = out + ((int64_t)blockIdx.y * nslots + slot) * 4;
#pragma unroll
            for (int v = 0; v < 4; ++v) o[v] = ((acc[u][v] + comb[0][sl + 64 * u][v]) + comb[1][sl + 64 * u][v]) + comb[2][sl + 64 * u][v];
        }
    }
}

// blockIdx.y = problem of a batch (its nchunks partials are consecutive)
__global__ void __launch_bounds__(256)
nudft_reduce_kernel(const double *__restrict__ part_all, int nchunks, int64_t count, double *__restrict__ out_all) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const double *part = part_all + (int64_t)blockIdx.y * nchunks * count;
    double s = 0;
    for (int c = 0; c < nchunks; ++c) s += part[(int64_t)c * count + i];
    out_all[(int64_t)blockIdx.y * count + i] = s;
}

// Fourier layout (src/lsfft.jl:26-49): column a < Nf is cos(w_a x) dd, column a >= Nf is -sin(w_k x) dd with
// k = a - Nf + zf (zf = 1 when the zero frequency is present and has no sine column), dd = 1/sqrt(2 Nf).
// One activation "pair" (weight W_n or 1).  blockIdx.z = problem of a batch.
__global__ void __launch_bounds__(256)
ap_assemble_fourier_kernel(const double *__restrict__ tab_all, const double *__restrict__ eps, int Nf, int s0, double delta, int zf, double scale,
                           int64_t n, double *__restrict__ G_all, int64_t ldg, int64_t tab_stride, int64_t g_stride) {
    const int64_t ga = blockIdx.y;
    const int64_t gb = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gb > ga || ga >= n) return;
    const double *tab = tab_all + (int64_t)blockIdx.z * tab_stride;
    double *G = G_all + (int64_t)blockIdx.z * g_stride;
    const int ca = ga >= Nf, cb = gb >= Nf;
    const int ka = ca ? (int)ga - Nf + zf : (int)ga, kb = cb ? (int)gb - Nf + zf : (int)gb;
    const int hi = ka >= kb ? ka : kb, lo = ka >= kb ? kb : ka;
    const double dm = eps[hi] - eps[lo], dp = (eps[ka] + eps[kb]) + delta;
    const double *tm = tab + (int64_t)(hi - lo) * 4;
    const double *tp = tab + (int64_t)(s0 + ka + kb) * 4;
    const double cm = fma(-dm, tm[3], tm[0]);
    double sm = fma(dm, tm[2], tm[1]);                       // sin((w_hi - w_lo) x)
    if (ka < kb) sm = -sm;                                   // -> sin((w_a - w_b) x)
    const double cp = fma(-dp, tp[3], tp[0]), sp = fma(dp, tp[2], tp[1]);
    double v;
    if (!ca && !cb) v = 0.5 * (cm + cp);
    else if (ca && cb) v = 0.5 * (cm - cp);
    else if (!ca && cb) v = -0.5 * (sp - sm);
    else v = -0.5 * (sp + sm);
    v *= scale;
    G[ga * ldg + gb] = v;
    G[gb * ldg + ga] = v;
}

__global__ void __launch_bounds__(256)
ap_rhs_fourier_kernel(const double *__restrict__ tab_all, const double *__restrict__ eps, int Nf, int zf, double dd,
                      double *__restrict__ b_all, int64_t tab_stride, int64_t b_stride) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= Nf) return;
    const double *t = tab_all + (int64_t)blockIdx.y * tab_stride + (int64_t)k * 4;
    double *b = b_all + (int64_t)blockIdx.y * b_stride;
    const double e = eps[k];
    b[k] = dd * fma(-e, t[3], t[0]);
    if (k >= zf) b[Nf + k - zf] = -dd * fma(e, t[2], t[1]);
}

// G[a][b] = G[b][a], a >= b, from the slot tables tab[slot][q][4]; slots 0..Nf-1 are the differences m = f-f',
// slots s0..s0+2Nf-2 the sums s = f+f' (s0 = Nf rounded up to a multiple of 8).
__global__ void __launch_bounds__(256)
ap_assemble_kernel(const double *__restrict__ tab, const double *__restrict__ eps, int Nf, int s0, double delta, int nb, int P, int64_t n,
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
    const double dm = eps[fa] - eps[fb], dp = (eps[fa] + eps[fb]) + delta;
    const double *tm = tab + ((int64_t)m * P + q) * 4;
    const double *tp = tab + ((int64_t)(s0 + s) * P + q) * 4;
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

// b[f*2nb + j] = sum y K_j cos(w_f x),  b[f*2nb + nb + j] = -sum y K_j sin(w_f x)   from tab[f][j][4] at the
// slots a + f*D, corrected to first order for w_f = a + f*D + eps_f
__global__ void __launch_bounds__(256)
ap_rhs_kernel(const double *__restrict__ tab, const double *__restrict__ eps, int Nf, int nb, double *__restrict__ b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nf * nb) return;
    const int f = i / nb, j = i - f * nb;
    const double *t = tab + ((int64_t)f * nb + j) * 4;
    const double e = eps[f];
    b[(int64_t)f * 2 * nb + j] = fma(-e, t[3], t[0]);
    b[(int64_t)f * 2 * nb + nb + j] = -fma(e, t[2], t[1]);
}

template <int QPT, int S>
void launch_accumulate(unsigned nchunks, hipStream_t s, const double *x, const double *y, int64_t N, const double *Wt, int64_t ldw, int nq,
                       const double *om_hi, const double *om_lo, int nslots, const ApStep &step, double *partial, const int64_t *seg,
                       int64_t rpc) {
    dim3 grid((unsigned)ceil_div(nslots, 64 * S), nchunks, (unsigned)ceil_div(nq, 4 * QPT));
    hipLaunchKernelGGL((nudft_accumulate_kernel<QPT, S>), grid, dim3(256), 0, s, x, y, N, Wt, ldw, nq, om_hi, om_lo, nslots, step, partial, seg, rpc);
}

}  // namespace

// samples per workgroup: 8192, fewer for short signals so that (slot groups) x (chunks) still fills the chip -- down to 64: a
// workgroup walks its samples one after the other (~0.19 us each), so the short records of ls_windowpsd_lpv's windows (600 samples:
// two chunks of 512 took 95 us per launch, three launches per window) are latency, not work
int64_t nudft_rows_per_chunk(int64_t N, int64_t nslots) {
    const int64_t groups = ceil_div(nslots, 128);
    int64_t rpc = 8192;
    while (rpc > 64 && groups * ceil_div(N, rpc) < 2048) rpc >>= 1;
    return rpc;
}
size_t nudft_chunks(int64_t N, int64_t nslots) { return (size_t)ceil_div(N, nudft_rows_per_chunk(N, nslots)); }
size_t nudft_partial_bytes(int64_t N, int64_t nslots, int64_t nq) {
    return sizeof(double) * nudft_chunks(N, nslots) * (size_t)nslots * (size_t)nq * 4;
}

// tab[slot][q][4] = sum_n (y_n) Wt[n][q] {cos, sin, x cos, x sin}(omega_slot x_n).  The slots must form arithmetic
// progressions with step D inside every aligned group of eight (nslots % 8 == 0); step holds b*D, b = 0..7.
// Wt == NULL means unit weights (nq = 1).  seg != NULL: nseg y-blocks {begin, end, weight base} (device array) grouped
// nbatch x (nseg / nbatch) -- windows of a signal; tab then holds one table per batch entry.
static int32_t nudft_impl(const double *x, const double *y, int64_t N, const double *Wt, int64_t ldw, int nq, const double *om_hi,
                          const double *om_lo, int nslots, const ApStep &step, double *partial, double *tab, const int64_t *seg,
                          unsigned nseg, unsigned nbatch, hipStream_t s) {
    if (nslots % 8 != 0) { set_error("nudft: slot count %d is not a multiple of 8", nslots); return LPVS_ESTATE; }
    const unsigned nchunks = seg ? nseg : (unsigned)nudft_chunks(N, nslots);
    const int64_t rpc = seg ? 0 : nudft_rows_per_chunk(N, nslots);
    if (nq == 1) {
        dim3 grid((unsigned)ceil_div(nslots, 128), nchunks, 1);
        hipLaunchKernelGGL(nudft_single_kernel<2>, grid, dim3(256), 0, s, x, y, N, Wt, ldw, om_hi, om_lo, nslots, step, partial, seg, rpc);
    } else if (nq <= 8) launch_accumulate<2, 2>(nchunks, s, x, y, N, Wt, ldw, nq, om_hi, om_lo, nslots, step, partial, seg, rpc);
    else if (nq <= 16) launch_accumulate<4, 2>(nchunks, s, x, y, N, Wt, ldw, nq, om_hi, om_lo, nslots, step, partial, seg, rpc);
    else launch_accumulate<9, 2>(nchunks, s, x, y, N, Wt, ldw, nq, om_hi, om_lo, nslots, step, partial, seg, rpc);
    LPVS_HIP(hipGetLastError());
    const int64_t count = (int64_t)nslots * nq * 4;
    hipLaunchKernelGGL(nudft_reduce_kernel, dim3((unsigned)ceil_div(count, 256), nbatch), dim3(256), 0, s, partial, (int)(nchunks / nbatch), count, tab);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_nudft(const double *x, const double *y, int64_t N, const double *Wt, int64_t ldw, int nq, const double *om_hi,
                     const double *om_lo, int nslots, const ApStep &step, double *partial, double *tab, hipStream_t s) {
    return nudft_impl(x, y, N, Wt, ldw, nq, om_hi, om_lo, nslots, step, partial, tab, nullptr, 0, 1, s);
}

int32_t launch_nudft_windows(const double *x, const double *y, const double *Wt, const double *om_hi, const double *om_lo, int nslots,
                             const ApStep &step, const int64_t *seg_dev, int nwin, int segs_per_window, double *partial, double *tab,
                             hipStream_t s) {
    return nudft_impl(x, y, 0, Wt, 1, 1, om_hi, om_lo, nslots, step, partial, tab, seg_dev, (unsigned)(nwin * segs_per_window), (unsigned)nwin, s);
}

int32_t launch_ap_assemble_fourier(const double *tab, const double *eps, int64_t Nf, int64_t s0, double delta, int zf, int64_t n, double *G,
                                   int64_t ldg, int nbatch, int64_t tab_stride, int64_t g_stride, hipStream_t s) {
    dim3 grid((unsigned)ceil_div(n, 256), (unsigned)n, (unsigned)nbatch);
    hipLaunchKernelGGL(ap_assemble_fourier_kernel, grid, dim3(256), 0, s, tab, eps, (int)Nf, (int)s0, delta, zf, 1.0 / (double)(2 * Nf), n, G, ldg,
                       tab_stride, g_stride);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_ap_rhs_fourier(const double *tab, const double *eps, int64_t Nf, int zf, double *b, int nbatch, int64_t tab_stride,
                              int64_t b_stride, hipStream_t s) {
    hipLaunchKernelGGL(ap_rhs_fourier_kernel, dim3((unsigned)ceil_div(Nf, 256), (unsigned)nbatch), dim3(256), 0, s, tab, eps, (int)Nf, zf,
                       1.0 / sqrt((double)(2 * Nf)), b, tab_stride, b_stride);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_ap_assemble(const double *tab, const double *eps, int64_t Nf, int64_t s0, double delta, int64_t nb, int64_t n, double *G,
                           int64_t ldg, hipStream_t s) {
    const int P = (int)(nb * (nb + 1) / 2);
    dim3 grid((unsigned)ceil_div(n, 256), (unsigned)n);
    hipLaunchKernelGGL(ap_assemble_kernel, grid, dim3(256), 0, s, tab, eps, (int)Nf, (int)s0, delta, (int)nb, P, n, G, ldg);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

int32_t launch_ap_rhs(const double *tab, const double *eps, int64_t Nf, int64_t nb, double *b, hipStream_t s) {
    hipLaunchKernelGGL(ap_rhs_kernel, dim3((unsigned)ceil_div(Nf * nb, 256)), dim3(256), 0, s, tab, eps, (int)Nf, (int)nb, b);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

}  // namespace lpvs
