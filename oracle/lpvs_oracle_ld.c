/*
 * lpvs_oracle_ld.c -- the Gram-form ADMM of lpvs_oracle.c (lpvo_admm_gram: src/lasso.jl:136-171 with the
 * x-update of ProximalOperators' LeastSquares solved exactly) carried in x87 EXTENDED precision (64-bit mantissa).
 *
 * TEST INFRASTRUCTURE ONLY, like lpvs_oracle.c: an ADJUDICATOR.  When the f64 device path and the f64 oracle differ by more
 * than the stated tolerance after thousands of iterations of a map that has not converged, somebody has to say which of
 * the two is further from the iterate the reference's mathematics defines.  Same algorithm, same operation order as
 * lpvo_admm_gram (Cholesky of G + I/mu, two triangular solves per iteration, prox, dual update), every quantity --
 * factor, state vectors, prox arithmetic -- a `long double`; inputs and outputs are doubles.  2^-64 / 2^-53 = 1/2048 of the
 * double path's rounding, so at the sizes in question its iterates are exact to ~1e-12 where the f64 paths sit at ~1e-9.
 *
 * Nothing under lpvspectral.jl_amd/ links or calls this.  gcc -O2 -fopenmp (x86-64: long double = x87 80-bit).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef long double ld;

int lpvo_ld_mantissa_bits(void) { return LDBL_MANT_DIG; }

/* dot of two contiguous vectors, four independent chains (the x87 add has a 3-5 cycle latency) */
static inline ld dot_ld(const ld *a, const ld *b, int64_t n) {
    ld s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int64_t k = 0;
    for (; k + 4 <= n; k += 4) {
        s0 += a[k] * b[k]; s1 += a[k + 1] * b[k + 1]; s2 += a[k + 2] * b[k + 2]; s3 += a[k + 3] * b[k + 3];
    }
    for (; k < n; ++k) s0 += a[k] * b[k];
    return (s0 + s1) + (s2 + s3);
}

/* L row-major, lower triangle; column panels of CB so that a row's prefix is read once per CB columns */
#define CB 32
static int chol_lower_ld(ld *L, int64_t n, int verbose) {
    for (int64_t j0 = 0; j0 < n; j0 += CB) {
        const int64_t j1 = j0 + CB < n ? j0 + CB : n;
        /* the panel's own rows first (sequential in j), then every row below it in parallel */
        for (int64_t j = j0; j < j1; ++j) {
            for (int64_t c = j0; c <= j; ++c) {
                ld s = L[j * n + c] - dot_ld(L + j * n, L + c * n, c);
                if (c == j) { if (!(s > 0)) return -1; L[j * n + j] = sqrtl(s); }
                else L[j * n + c] = s / L[c * n + c];
            }
        }
#pragma omp parallel for schedule(dynamic, 8)
        for (int64_t i = j1; i < n; ++i)
            for (int64_t c = j0; c < j1; ++c)
                L[i * n + c] = (L[i * n + c] - dot_ld(L + i * n, L + c * n, c)) / L[c * n + c];
        if (verbose && (j0 / CB) % 32 == 0) { fprintf(stderr, "[ld] cholesky column %lld of %lld\n", (long long)j0, (long long)n); fflush(stderr); }
    }
    return 0;
}

/* L y = b, then L' x = y.  Row blocks of RB: the part left of (right of) the diagonal block is a mat-vec whose rows are
 * independent (threads), the diagonal block is sequential. */
#define RB 64
static void chol_solve_ld(const ld *L, const ld *Lt, int64_t n, ld *b) {
    for (int64_t i0 = 0; i0 < n; i0 += RB) {
        const int64_t i1 = i0 + RB < n ? i0 + RB : n;
#pragma omp parallel for schedule(static) if (i0 >= 1024)
        for (int64_t i = i0; i < i1; ++i) b[i] -= dot_ld(L + i * n, b, i0);
        for (int64_t i = i0; i < i1; ++i) b[i] = (b[i] - dot_ld(L + i * n + i0, b + i0, i - i0)) / L[i * n + i];
    }
    for (int64_t i1 = n; i1 > 0; i1 -= RB) {
        const int64_t i0 = i1 - RB > 0 ? i1 - RB : 0;
#pragma omp parallel for schedule(static) if (n - i1 >= 1024)
        for (int64_t i = i0; i < i1; ++i) b[i] -= dot_ld(Lt + i * n + i1, b + i1, n - i1);
        for (int64_t i = i1 - 1; i >= i0; --i) b[i] = (b[i] - dot_ld(Lt + i * n + i + 1, b + i + 1, i1 - i - 1)) / L[i * n + i];
    }
}

/* the prox operators of lpvs_oracle.c in extended precision (kinds: 1 L1, 2 L0, 3 IndBallL0, 4 group L2) */
static void prox_ld(int kind, ld *z, const ld *v, int64_t n, double param, int64_t glen, double g) {
    if (kind == 1) {
        const ld gl = (ld)g * (ld)param;
        for (int64_t i = 0; i < n; ++i) z[i] = v[i] + (v[i] <= -gl ? gl : (v[i] >= gl ? -gl : -v[i]));
    } else if (kind == 2) {
        const ld th = sqrtl(2 * (ld)g * (ld)param);
        for (int64_t i = 0; i < n; ++i) z[i] = fabsl(v[i]) > th ? v[i] : 0;
    } else if (kind == 3) {
        const int64_t r = (int64_t)param;
        if (r >= n) { memcpy(z, v, sizeof(ld) * n); return; }
        char *keep = (char *)calloc(n, 1);
        for (int64_t k = 0; k < r; ++k) {
            int64_t best = -1; ld bv = -1;
            for (int64_t i = 0; i < n; ++i) if (!keep[i] && fabsl(v[i]) > bv) { bv = fabsl(v[i]); best = i; }
            keep[best] = 1;
        }
        for (int64_t i = 0; i < n; ++i) z[i] = keep[i] ? v[i] : 0;
        free(keep);
    } else {
        for (int64_t s = 0; s + glen <= n; s += glen) {
            ld ss = 0;
            for (int64_t i = 0; i < glen; ++i) ss += v[s + i] * v[s + i];
            const ld nv = sqrtl(ss);
            ld scale = 1 - (ld)param * (ld)g / nv;
            if (!(scale > 0)) scale = 0;
            for (int64_t i = 0; i < glen; ++i) z[s + i] = scale * v[s + i];
        }
    }
}

/* G: n x n symmetric (doubles), b: n.  Runs snaps[nsnap-1] iterations from x0 (NULL: 0) with z = x0, u = 0 and writes the
 * iterates after snaps[k] iterations (ascending) to x_out / z_out / u_out [nsnap][n], rounded to double once.  No stopping test
 * (tol = 0 of the parity runs).  Returns the iterations done, or < 0. */
int64_t lpvo_admm_gram_ld(const double *G, int64_t n, const double *b, const double *x0, int prox_kind, double prox_param, int64_t glen,
                          double mu, const int64_t *snaps, int64_t nsnap, double *x_out, double *z_out, double *u_out, int verbose) {
    if (!(mu > 0 && mu <= 1) || nsnap < 1) return -2;
    ld *L = (ld *)malloc(sizeof(ld) * n * n), *Lt = (ld *)malloc(sizeof(ld) * n * n);
    ld *x = (ld *)calloc(n, sizeof(ld)), *z = (ld *)calloc(n, sizeof(ld)), *u = (ld *)calloc(n, sizeof(ld)), *t = (ld *)calloc(n, sizeof(ld));
    if (!L || !Lt || !x || !z || !u || !t) { free(L); free(Lt); free(x); free(z); free(u); free(t); return -4; }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = 0; k < n; ++k) L[i * n + k] = k <= i ? (ld)G[i * n + k] + (i == k ? (ld)1 / (ld)mu : (ld)0) : (ld)0;
    if (chol_lower_ld(L, n, verbose)) { free(L); free(Lt); free(x); free(z); free(u); free(t); return -3; }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = 0; k < n; ++k) Lt[i * n + k] = k >= i ? L[k * n + i] : (ld)0;
    if (x0) for (int64_t k = 0; k < n; ++k) x[k] = z[k] = x0[k];
    int64_t it = 0, snap = 0;
    const int64_t iters = snaps[nsnap - 1];
    for (int64_t i = 1; i <= iters; ++i) {
        for (int64_t k = 0; k < n; ++k) x[k] = (ld)b[k] + (z[k] - u[k]) / (ld)mu;
        chol_solve_ld(L, Lt, n, x);
        for (int64_t k = 0; k < n; ++k) t[k] = x[k] + u[k];
        prox_ld(prox_kind, z, t, n, prox_param, glen, mu);
        for (int64_t k = 0; k < n; ++k) u[k] += x[k] - z[k];
        it = i;
        while (snap < nsnap && snaps[snap] == i) {
            for (int64_t k = 0; k < n; ++k) { x_out[snap * n + k] = (double)x[k]; z_out[snap * n + k] = (double)z[k]; u_out[snap * n + k] = (double)u[k]; }
            ++snap;
        }
        if (verbose && i % 100 == 0) { fprintf(stderr, "[ld] iteration %lld of %lld\n", (long long)i, (long long)iters); fflush(stderr); }
    }
    free(L); free(Lt); free(x); free(z); free(u); free(t);
    return it;
}
