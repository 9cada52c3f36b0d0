/*
 * lpvspectral.h -- C-ABI of the MI355X-native regression-matrix + ADMM hot path of
 * LPVSpectral.jl.  This is the drop-in boundary: a Julia wrapper keeps the reference's
 * function signatures and reaches these entry points through @ccall;
 * the Python host mirror in lpvspectral.jl_amd/ reaches them through ctypes.
 *
 * Conventions
 *   - every function returns an int32 status (LPVS_OK or a negative LPVS_E* code);
 *     lpvs_last_error() gives the thread-local message of the last failure.
 *   - sizes are int64_t (Julia Int); matrices are COLUMN-MAJOR (Julia layout).
 *   - every floating-point array argument may be a HOST pointer or a DEVICE (HIP) pointer of the
 *     current device; the library detects which (hipPointerGetAttributes).  Integer outputs
 *     (int64_t* counts, offsets, iteration counts) and scalar outputs are HOST pointers.  The
 *     caller owns all argument memory; it is never retained or freed here, and is
 *     only read/written for the duration of the call.  Device arguments are read on the
 *     library's own streams: work the caller has queued on other streams to produce them
 *     must be complete before the call (the Python mirror synchronises torch's current
 *     stream; a Julia wrapper calls AMDGPU.synchronize()).  Device outputs are complete
 *     when the call returns.
 *   - a handle owns one HIP stream on one device; handles are not thread-safe,
 *     different handles may be used concurrently.
 *   - arithmetic type: fp64 (suffix _f64), the eltype of every reference test: assembly, Gram,
 *     factorisation, prox operators and every accumulation are double.  STORAGE CAVEAT: for
 *     n >= 2048 the ADMM mat-vec of an _f64 handle streams, by default, a reduced-width COPY of
 *     (G + I/mu)^-1 with >= 36 significant bits per element (LPVS_STORAGE_MIXED below), applied in
 *     offset form x = M b + M~ (z-u)/mu with M b from the full doubles.  Measured cost: 1.1e-10
 *     rel-L2 in z after 2000 iterations at N = 2^20, n = 8192 -- below the reference's own CG
 *     tolerance sqrt(eps) = 1.5e-8 (src/lasso.jl:151, ProximalOperators iterative=true).
 *     lpvs_problem_set_option(h, LPVS_OPT_M_STORAGE, LPVS_STORAGE_F64) streams the doubles
 *     themselves (1.55x the bytes per iteration).
 *
 * File:line citations are into the reference (baggepinnen/LPVSpectral.jl v0.3.4).
 */
#ifndef LPVSPECTRAL_H
#define LPVSPECTRAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPVS_VERSION 100 /* 0.1.0 */

/* status codes -> exception the Julia/Python wrapper raises */
#define LPVS_OK 0
#define LPVS_EARGUMENT (-1)    /* ArgumentError: zero frequency not first        src/lsfft.jl:22   */
#define LPVS_EASSERT (-2)      /* AssertionError: mu outside [0,1], length mismatch src/lasso.jl:143, src/windows.jl:31 */
#define LPVS_EDOMAIN (-3)      /* DomainError: noverlap >= n (DSP.arraysplit)    src/windows.jl:33 */
#define LPVS_ENOMEM (-4)       /* host or device allocation failed */
#define LPVS_EDEVICE (-5)      /* HIP runtime error / no gfx950 device */
#define LPVS_EUNSUPPORTED (-6) /* e.g. coulomb=true in the sparse LPV path (note ++) */
#define LPVS_ENUMERIC (-7)     /* (G + I/mu) not positive definite */
#define LPVS_ESTATE (-8)       /* call order violated (run before init, ...) */

/* prox_g kinds (ProximalOperators.jl objects at src/lasso.jl:53-55,88,108; README.md:70-83) */
#define LPVS_PROX_L1 1       /* NormL1(lam):  soft threshold at mu*lam */
#define LPVS_PROX_L0 2       /* NormL0(lam):  keep |v| > sqrt(2*mu*lam) */
#define LPVS_PROX_BALL_L0 3  /* IndBallL0(r): keep the r largest |v| (param = r) */
#define LPVS_PROX_GROUP_L2 4 /* SlicedSeparableSum(NormL2(lam)) on contiguous groups of group_len */

/* linear-term convention of the x-update, lpvs_admm_init(..., linear_sign) */
#define LPVS_LINEAR_LEAST_SQUARES (+1) /* LeastSquares(A,y): (G+I/mu)x =  b + v/mu   src/lasso.jl:51,98 */
#define LPVS_LINEAR_QUADRATIC_AS_WRITTEN (-1) /* Quadratic(Q,q=+A'Wy): (Q+I/mu)x = -q + v/mu   src/lasso.jl:119-121 */

typedef struct lpvs_problem lpvs_problem;

/* ---- library ------------------------------------------------------------------------ */
int32_t lpvs_version(void);
int32_t lpvs_device_count(void);          /* number of visible HIP devices (0 if none) */
const char *lpvs_last_error(void);        /* thread-local, valid until the next failing call */
/* work buffers are cached per device between calls (cap: LPVS_POOL_GIB, default 128); this returns them to the driver */
int32_t lpvs_release_cached_memory(void);

/* ---- options: how a handle stores the inverse its ADMM mat-vec streams, how it iterates, which Gram form its constructor
 * takes.  Value 0 (LPVS_OPT_DEFAULT) = the library's choice, which an environment variable of the same name may override for
 * experiments (LPVS_M_STORAGE, LPVS_ITERATION, LPVS_GRAM_FORM, LPVS_NT_LOADS, LPVS_NUDFT, LPVS_WINDOW_CHUNK_MB [0 = uncut],
 * LPVS_WINDOWS_IN_FLIGHT, LPVS_RESERVE_CUS [0 = none]); an explicit option wins over the
 * environment.  Results do not depend on ITERATION / NT_LOADS / SLOT_SUMS beyond rounding (tests/test_gpu_one_launch.py,
 * tests/test_gpu_nufft.py); M_STORAGE trades bytes per iteration against 1e-10 in the iterates (see the header comment).
 *   lpvs_set_default_option:  thread-local default for handles created and batched-window calls made AFTERWARDS by the calling
 *                             thread (the batched-window entry points have no handle to carry options);
 *   lpvs_problem_set_option:  one handle.  ITERATION / NT_LOADS take effect at the next lpvs_admm_run; a CHANGED M_STORAGE
 *                             invalidates the iteration state (the packed copy is rebuilt): lpvs_admm_init must follow, an
 *                             lpvs_admm_run without it returns LPVS_ESTATE;
 *                             GRAM_FORM and SLOT_SUMS are constructor-time choices, WINDOW_CHUNK_MB / WINDOWS_IN_FLIGHT / RESERVE_CUS
 *                             belong to calls without a handle or to the device (LPVS_ESTATE on a handle: set the default). */
#define LPVS_OPT_DEFAULT 0
#define LPVS_OPT_M_STORAGE 1  /* LPVS_STORAGE_*  : packed inverse of n >= 2048 handles / window batches */
#define LPVS_OPT_ITERATION 2  /* LPVS_ITERATION_*: one launch per ADMM iteration (where applicable) or mat-vec + update launches */
#define LPVS_OPT_GRAM_FORM 3  /* LPVS_GRAM_*     : structured Gram for arithmetic-progression grids, or the dense MFMA forms */
#define LPVS_OPT_NT_LOADS 4   /* LPVS_NT_*       : non-temporal tile loads (default: when a launch streams more than 240 MiB of inverses -- window batches, single problems from n = 10752) */
#define LPVS_OPT_SLOT_SUMS 5  /* LPVS_SLOTS_*    : slot sums of the structured Gram by non-uniform FFT or by direct evaluation */
#define LPVS_STORAGE_MIXED 1  /* float head + 16-bit tail (40 bits) for tiles with large entries, 36-bit fixed point elsewhere, all 36 bits read by every iteration */
#define LPVS_STORAGE_SPLIT 2  /* float head + 16-bit tail everywhere (6 bytes, 40 significant bits) */
#define LPVS_STORAGE_F64 3    /* doubles (8 bytes): the reference-width copy */
#define LPVS_STORAGE_MIXED32 4 /* MIXED, but the iteration READS the 32 leading bits of the fixed-point tiles only (4 B per element: 140 instead of 157 MB per
                                * iteration at n = 8192) and the product of the 4-bit planes is taken with a right-hand side at most 32 iterations old (1, 2, 4, ... in the first 256), carried
                                * in the x-update's offset vector ("stale nibble product").  THE DEFAULT of handles whose x-update is
                                * corrected (one right-hand side, doubles, n >= 2048 -- LPVS_OPT_XUPDATE_CORRECTION); MIXED for every other handle, also when
                                * asked for by name.  x, z and u stay where MIXED leaves them: at cfg3 after 200 .. 2000 iterations x and z are the same
                                * 1.2e-10 .. 4.8e-10 from the exact iterates and u 9.8e-11 .. 5.0e-10 (MIXED: 9.6e-11 .. 4.5e-10) --
                                * profiles/r05_cfg3_stale_nibble_product.txt.  (Without the refresh the dual variable integrates the truncation: u 2e-9 .. 7e-9.) */
#define LPVS_ITERATION_ONE 1
#define LPVS_ITERATION_TWO 2
#define LPVS_GRAM_AP 1        /* structured (error unless the grid is an arithmetic progression up to rounding) */
#define LPVS_GRAM_KRS 2       /* dense, symmetric-pair contraction */
#define LPVS_GRAM_KR 3        /* dense, Khatri-Rao contraction */
#define LPVS_NT_OFF 1
#define LPVS_NT_ON 2
#define LPVS_OPT_WINDOW_CHUNK_MB 6   /* batched-window engine: MB (10^6 bytes) of packed inverses per chunk (a chunk is re-read from the Infinity
                                       * Cache every iteration); > 0: that many MB, LPVS_WINDOW_UNCUT: every window in one launch per iteration;
                                       * default: 1.0625 x the Infinity Cache of the GPU nodes of the KFD topology (256 MiB on MI355X = 268.4 MB
                                       * -> 285 MB) */
#define LPVS_OPT_WINDOWS_IN_FLIGHT 7 /* parts of a chunk solved concurrently on streams of their own: 1 .. 4 (default 2) */
#define LPVS_OPT_RESERVE_CUS 8       /* CUs the factorisation's trailing updates leave to its pivot chain: > 0, or LPVS_RESERVE_NONE (default 8) */
#define LPVS_OPT_XUPDATE_CORRECTION 9 /* LPVS_XCORR_*: re-form the x-update's offset vector after the iterations 16, 128, 256, 512, 1024, ... (several right-hand sides: 16, 512, 1024, ...) by one step of iterative
                                       * refinement, residual accumulated in twice the mantissa (handles of n >= 2048).  Default: ON -- since
                                       * round 6 also for handles with several right-hand sides (cfg5: 4e-10 of drift in x, z after 2000 iterations without it;
                                       * a correction there is one accurate pass over the f64 Gram for all channels, 1.9 % of the run).  Per handle (before
                                       * lpvs_admm_init) or as a thread default. */
#define LPVS_XCORR_ON 1
#define LPVS_XCORR_OFF 2
#define LPVS_SLOTS_NUFFT 1
#define LPVS_SLOTS_DIRECT 2
#define LPVS_WINDOW_UNCUT (-1)
#define LPVS_RESERVE_NONE (-1)
int32_t lpvs_set_default_option(int32_t option, int32_t value);
int32_t lpvs_get_default_option(int32_t option, int32_t *value);   /* the calling thread's explicit default (0 if none) */
int32_t lpvs_problem_set_option(lpvs_problem *h, int32_t option, int32_t value);
int32_t lpvs_problem_get_option(lpvs_problem *h, int32_t option, int32_t *value);   /* the value in effect for this handle */

/* ---- a1  check_freq                                                  src/lsfft.jl:20-24
 * *zerofreq = 0 (no zero frequency) or 1 (zero frequency is first); LPVS_EARGUMENT if a
 * zero frequency sits elsewhere.  Host pointers only (tiny). */
int32_t lpvs_check_freq_f64(const double *f, int64_t Nf, int64_t *zerofreq);

/* ---- a2  get_fourier_regressor(t,f)                                  src/lsfft.jl:26-49
 * A_out is N x Nreg column-major, Nreg = 2Nf - (zerofreq ? 1 : 0):
 *   A[n,k] = cos((2pi f_k) t_n)/sqrt(2Nf);  A[n,k+sinoffset] = -sin(...)/sqrt(2Nf). */
int32_t lpvs_fourier_regressor_f64(const double *t, int64_t N, const double *f, int64_t Nf,
                                   double *A_out, int64_t *zerofreq);

/* ---- a3  basis_activation_func evaluated on all samples   src/utilities.jl:23-36, src/lsfft.jl:195-207
 * K_out is N x nb column-major, nb = Nv (2Nv if coulomb). */
int32_t lpvs_basis_activation_f64(const double *V, int64_t N, int64_t Nv, int32_t normalize,
                                  int32_t coulomb, double *K_out);

/* ---- a4+a5  LPV regressor                          src/lasso.jl:35-50, src/lsfft.jl:240-247
 * permuted != 0: Phi = [Re As, Im As][:, inds]  (N x 2*Nf*nb, each frequency's 2nb columns
 * adjacent, the matrix ls_sparse_spectral_lpv solves on); permuted == 0: [Re As, Im As]. */
int32_t lpvs_lpv_regressor_f64(const double *X, const double *V, int64_t N, const double *w,
                               int64_t Nf, int64_t Nv, int32_t normalize, int32_t coulomb,
                               int32_t permuted, double *Phi_out);

/* ---- problem handles: regressor assembly + Gram on device ----------------------------
 * Each constructor builds G (n x n) and b (n) on `device` and keeps them resident:
 *   fourier: G = A' diag(W) A, b = A' diag(W) y  (W == NULL: identity)  src/lasso.jl:91,98 / :111,118-120
 *   lpv:     G = Phi' Phi,     b = Phi' y   (Phi never materialised)    src/lasso.jl:35-51
 *   gram:    G, b supplied by the caller (any Quadratic(Q,q) / LeastSquares in Gram form).
 *   dense:   see below. */
int32_t lpvs_problem_create_fourier_f64(const double *y, const double *t, int64_t N, const double *f,
                                        int64_t Nf, const double *W, int32_t device,
                                        lpvs_problem **out);
int32_t lpvs_problem_create_lpv_f64(const double *y, const double *X, const double *V, int64_t N,
                                    const double *w, int64_t Nf, int64_t Nv, int32_t normalize,
                                    int32_t coulomb, int32_t device, lpvs_problem **out);
/* lpv_multi: ns signals sharing (X, V, w) -- Y is N x ns column-major (multichannel records): ONE Gram, ns
 *          right-hand sides b_q = Phi' y_q; every ADMM kernel then advances all ns signals in one pass over M,
 *          each with its own convergence flag and iteration count (extension: the reference has no batched
 *          entry point, SURVEY.md section 8(b) "Gaps").  x0 / iterates / params are n x ns (resp. m x ns). */
int32_t lpvs_problem_create_lpv_multi_f64(const double *Y, int64_t ns, const double *X, const double *V, int64_t N,
                                          const double *w, int64_t Nf, int64_t Nv, int32_t normalize,
                                          int32_t coulomb, int32_t device, lpvs_problem **out);

/* ---- one exchange step for a single large problem: the sample rows of one signal are sharded
 * across devices, every shard forms its partial Gram G_r = Phi_r' Phi_r and b_r = Phi_r' y_r, the partials are
 * summed (RCCL all-reduce on the device pointers below) and the ADMM runs replicated.  The basis centres of
 * src/utilities.jl:24-32 depend on min/max of V over ALL rows, so a shard is built with the global ranges:
 *   ranges4 = {min V, max V, max|V|, max|X|}   (lpvs_lpv_ranges_f64 gives a shard's own; combine with min/max)
 * max|X| only steers the choice of Gram form, which must be the same on every shard. */
int32_t lpvs_lpv_ranges_f64(const double *X, const double *V, int64_t N, double *out4);
int32_t lpvs_problem_create_lpv_rows_f64(const double *Y, int64_t ns, const double *X, const double *V,
                                         int64_t N_local, const double *w, int64_t Nf, int64_t Nv,
                                         int32_t normalize, int32_t coulomb, const double *ranges4,
                                         int32_t device, lpvs_problem **out);
/* device pointers of the handle's Gram (np x np, leading dimension np, full symmetric storage, pad rows and
 * columns zero) and right-hand sides (np per signal); valid while the handle lives.  After modifying them in
 * place call lpvs_problem_gram_modified so that the cached factorisation is dropped. */
int32_t lpvs_problem_device_gram_f64(lpvs_problem *h, double **G_dev, double **b_dev, int64_t *np);
int32_t lpvs_problem_gram_modified(lpvs_problem *h);
int32_t lpvs_problem_num_signals(const lpvs_problem *h, int64_t *ns);
/* dense:   A (m x n column-major) supplied by the caller: G = A' diag(W) A, b = A' diag(W) y --
 *          the generic ADMM(x, LeastSquares(A,y,iterative=true), proxg) plugin path, src/lasso.jl:136 */
int32_t lpvs_problem_create_dense_f64(const double *A, const double *y, int64_t m, int64_t n,
                                      const double *W, int32_t device, lpvs_problem **out);
int32_t lpvs_problem_create_gram_f64(const double *G, const double *b, int64_t n, int32_t device,
                                     lpvs_problem **out);
int32_t lpvs_problem_destroy(lpvs_problem *h);

int32_t lpvs_problem_size(const lpvs_problem *h, int64_t *n);
int32_t lpvs_problem_zerofreq(const lpvs_problem *h, int64_t *zerofreq);
/* copy out G (n x n, full symmetric) and/or b (n); either pointer may be NULL */
int32_t lpvs_problem_get_gram_f64(lpvs_problem *h, double *G_out, double *b_out);

/* all right-hand sides b_q = Phi' y_q of a (multi-signal) handle, n x ns column-major */
int32_t lpvs_problem_get_rhs_f64(lpvs_problem *h, double *b_out);
/* (G + shift*I)^-1 (n x n, symmetric) -- e.g. the parameter covariance of ls_spectral_lpv up to the factor var(e),
 * src/lsfft.jl:252-254.  Leaves the handle's factorisation cached for that shift. */
int32_t lpvs_problem_get_inverse_f64(lpvs_problem *h, double shift, double *Minv_out);

/* ---- dense (ridge) solve from the same Gram:  (G + ridge*I) x = b ---------------------
 * ls_spectral weighted form (src/lsfft.jl:77, ridge = lam), fourier_solve / real_complex_bs in
 * normal-equation form (src/utilities.jl:49-60, ridge = lam^2).  x_out has n entries in the
 * regressor's column order. */
int32_t lpvs_problem_solve_ridge_f64(lpvs_problem *h, double ridge, double *x_out);

/* ---- ls_spectral(y,t,f; lam): [A; lam I] \ [y; 0]                src/lsfft.jl:62-67, src/utilities.jl:56-60
 * The minimiser is computed from a Gram on device, in the better-conditioned of its two forms:
 *   N >= Nreg (tall):  x = (A'A + lam^2 I)^-1 A'y          (primal normal equations)
 *   N <  Nreg (fat):   x = A' (A A' + lam^2 I)^-1 y        (dual form; e.g. default_freqs of an even-length t
 *                                                           has Nreg = N+1 and a vanishing Nyquist sine column)
 * re/im: Nf entries, fourier2complex of x. */
int32_t lpvs_ls_spectral_f64(const double *y, const double *t, int64_t N, const double *f, int64_t Nf, double lam,
                             int32_t device, double *re_out, double *im_out);

/* ---- ADMM                                                          src/lasso.jl:136-171
 * set_prox: the g of z <- prox_{mu g}(x+u).  init: x <- x0 (zeros if NULL), z <- x, u <- 0,
 * factorises (G + I/mu) once (LPVS_EASSERT unless 0 <= mu <= 1).  run: up to max_iters more
 * iterations, stopping after the first with ||x-z||_2 < tol exactly as src/lasso.jl:164; it
 * returns to the caller so that printing, cb(x,z) and Ctrl-C stay on the host thread
 * (src/lasso.jl:158-163, :59-63).  *iters_done counts iterations since init. */
int32_t lpvs_problem_set_prox(lpvs_problem *h, int32_t kind, double param, int64_t group_len);
int32_t lpvs_admm_init_f64(lpvs_problem *h, const double *x0, double mu, double tol,
                           int32_t linear_sign);
int32_t lpvs_admm_run(lpvs_problem *h, int64_t max_iters, int64_t *iters_done, double *nxz,
                      int32_t *converged);
/* resume (checkpoint / re-entry): after lpvs_admm_init (same mu, tol, prox, linear_sign) install iterates
 * saved with lpvs_admm_get_f64 from an earlier run -- possibly of another process -- together with the number of
 * iterations they represent; lpvs_admm_run then continues as the uninterrupted run would (the x-update of src/lasso.jl:150-151 only
 * needs z and u) -- BIT FOR BIT for handles of n < 2048; handles of n >= 2048 also carry the x-update's offset vector between two
 * scheduled iterations (below): with it (lpvs_admm_set_offset_f64) the continuation is bit-exact, without it the vector is re-formed from
 * the state handed in and the continuation agrees to second order (~1e-12).  iters_done = 0 restarts: the offset vector of a fresh
 * lpvs_admm_init is put back.  Arrays are n (x ns) in the solver's own order, as lpvs_admm_get_f64 returns. */
int32_t lpvs_admm_set_state_f64(lpvs_problem *h, const double *x, const double *z, const double *u, int64_t iters_done);
/* Handles of n >= 2048 run the x-update in its offset form, x = xb + M (z - u)/mu, and re-form the offset vector after the iterations
 * 16, 128, 256, 512, 1024, ... (handles with several right-hand sides: 16, 512, 1024, ...) so that the systematic error of the explicit inverse leaves the iteration (one step of iterative refinement
 * with the residual accumulated in twice the mantissa).  That vector is part of the iteration's state between two
 * scheduled iterations: a checkpoint that is to continue BIT FOR BIT saves it with lpvs_admm_get_offset_f64 next to x, z, u and
 * installs it with lpvs_admm_set_offset_f64 after lpvs_admm_set_state_f64 (which, without it, re-forms the vector from the state it is
 * given -- the same to second order).  Always doubles (also for handles made by the _f32 constructors).
 * lpvs_admm_offset_len: how many -- n x ns, or 2 n for a handle that iterates on LPVS_STORAGE_MIXED32 reads (the vector with and without
 * the nibble term of the last refresh: both are state); 0 when the handle has no offset vector (n < 2048).  The buffer is opaque: hand
 * back what lpvs_admm_get_offset_f64 wrote.  `len` = the number of doubles the caller's buffer holds and must EQUAL lpvs_admm_offset_len
 * (LPVS_EARGUMENT otherwise: a checkpoint written by a 36-bit handle does not go into a handle that reads 32 bits, or the reverse -- nothing
 * is read or written past a buffer of the wrong size).  get / set: LPVS_ESTATE when the handle has no offset vector, or before lpvs_admm_init. */
int32_t lpvs_admm_offset_len(lpvs_problem *h, int64_t *len);
int32_t lpvs_admm_get_offset_f64(lpvs_problem *h, double *xb_out, int64_t len);
int32_t lpvs_admm_set_offset_f64(lpvs_problem *h, const double *xb, int64_t len);
/* per-signal state of a multi-signal handle (lpvs_admm_run reports the slowest signal / the largest ||x-z||) */
int32_t lpvs_admm_status(lpvs_problem *h, int64_t signal, int64_t *iters_done, double *nxz, int32_t *converged);
/* iterates in the solver's own (regressor-column) order; any pointer may be NULL */
int32_t lpvs_admm_get_f64(lpvs_problem *h, double *x_out, double *z_out, double *u_out);

/* ---- a14 result packing ---------------------------------------------------------------
 * fourier: fourier2complex(z, zerofreq)                         src/utilities.jl:62-73
 * lpv:     z[sortperm(inds)] -> complex, index f+(v-1)Nf        src/lasso.jl:67-68
 * which = 0: pack z (what the estimators return), 1: pack x.  re/im have Nf (fourier) or
 * Nf*nb (lpv) entries. */
int32_t lpvs_problem_get_params_f64(lpvs_problem *h, int32_t which, double *re_out, double *im_out);
/* same packing applied to a caller-supplied coefficient vector (e.g. a ridge solution) */
int32_t lpvs_problem_pack_params_f64(lpvs_problem *h, const double *coef, double *re_out,
                                     double *im_out);

/* ---- timing: HIP-event durations (ms) of the handle's phases, measured on its stream ---
 * out[0] basis tables, out[1] Gram kernel(s), out[2] Gram reduce + rhs, out[3] factorisation,
 * out[4] ADMM iterations (sum over lpvs_admm_run calls), out[5] flops the Gram kernel's MFMA core
 * actually issues (tiles*128*256*2*Npad; the structured form: 8*N*(3Nf-1)*P VALU flops), out[6]
 * algorithmic Gram flops N*n*(n+1), out[7] ADMM iterations timed in out[4], out[8] Gram form used:
 * 0 none (Gram given), 1 n x n lower triangle, 2 symmetric-pair, 3 k-major panel, 4 structured
 * (arithmetic-progression w, nudft.hip), 5 structured with the slot sums by non-uniform FFT; out[9] the time of the x-update
 * corrections inside out[4] (HIP events around each), out[10] their count; out[11] the refreshes of the stale nibble product
 * enqueued inside out[4] (LPVS_STORAGE_MIXED32), out[12] the duration of one (us) where a refresh is three kernels of its own (the two-launch
 * iteration; measured stand-alone by lpvs_admm_time_matvec) -- 0 before that call and for the one-launch iteration, whose refresh is part
 * of the launch it follows (that launch also multiplies the 4-bit planes) plus one vector kernel. */
int32_t lpvs_problem_get_timing(lpvs_problem *h, double *out, int32_t n_out);

/* average duration (microseconds) of the ADMM mat-vec kernel of this handle over `reps` back-to-back launches, from
 * HIP events on the handle's stream (benchmark instrumentation; needs lpvs_admm_init; iterates are not modified).
 * *bytes_per_launch = bytes of M the kernel streams per launch (tile-packed lower triangle, or the full matrix). */
int32_t lpvs_admm_time_matvec(lpvs_problem *h, int32_t reps, double *us_per_launch, double *bytes_per_launch);

/* ---- single-precision entry points ------------------------------------------------------
 * The reference is eltype-generic (src/lasso.jl:85,91,144: Float32 inputs run in Float32).  The
 * _f32 functions take and return float arrays (host or device).  Inputs are widened exactly to
 * double; assembly, Gram and factorisation run in double (at least as accurate as a Float32 run
 * of the reference); a handle created here streams a single-precision copy of (G + I/mu)^-1 in
 * the ADMM mat-vec of large problems (half the bytes per iteration, double accumulation);
 * outputs are rounded to float.  Handle functions without a type suffix (set_prox, admm_run,
 * admm_status, timing, destroy, ...) are shared with the _f64 handles. */
int32_t lpvs_check_freq_f32(const float *f, int64_t Nf, int64_t *zerofreq);
int32_t lpvs_fourier_regressor_f32(const float *t, int64_t N, const float *f, int64_t Nf,
                                   float *A_out, int64_t *zerofreq);
int32_t lpvs_lpv_regressor_f32(const float *X, const float *V, int64_t N, const float *w, int64_t Nf,
                               int64_t Nv, int32_t normalize, int32_t coulomb, int32_t permuted,
                               float *Phi_out);
int32_t lpvs_problem_create_fourier_f32(const float *y, const float *t, int64_t N, const float *f,
                                        int64_t Nf, const float *W, int32_t device,
                                        lpvs_problem **out);
int32_t lpvs_problem_create_lpv_f32(const float *y, const float *X, const float *V, int64_t N,
                                    const float *w, int64_t Nf, int64_t Nv, int32_t normalize,
                                    int32_t coulomb, int32_t device, lpvs_problem **out);
int32_t lpvs_problem_create_lpv_multi_f32(const float *Y, int64_t ns, const float *X, const float *V, int64_t N, const float *w,
                                          int64_t Nf, int64_t Nv, int32_t normalize, int32_t coulomb, int32_t device,
                                          lpvs_problem **out);
/* lpvs_windows_estimate_f64 for Float32 callers (float in / out, double arithmetic) */
int32_t lpvs_windows_estimate_f32(const float *Y, int64_t ns, const float *t, int64_t L, int64_t n, int64_t noverlap,
                                  const float *W, const float *freqs, int64_t Nf, int32_t estimator, double lam,
                                  int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol, int64_t iters,
                                  int32_t linear_sign, int64_t win_lo, int64_t win_hi, int32_t device, float *x_re, float *x_im,
                                  int64_t *iters_out);
/* lpvs_windows_estimate_multi_f64 / lpvs_windowcsd_f64 / lpvs_problem_create_lpv_rows_f64 / lpvs_problem_solve_ridge_f64 /
 * lpvs_admm_set_state_f64 for Float32 callers */
int32_t lpvs_windows_estimate_multi_f32(const float *Y, int64_t ns, const float *t, int64_t L, int64_t n, int64_t noverlap,
                                        const float *W, const float *freqs, int64_t Nf, int32_t estimator, double lam,
                                        int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol,
                                        int64_t iters, int32_t linear_sign, const int32_t *devices, int32_t ngpus, float *x_re,
                                        float *x_im, int64_t *iters_out);
int32_t lpvs_windowcsd_f32(const float *y, const float *u, const float *t, int64_t L, int64_t n, int64_t noverlap,
                           const float *W, const float *freqs, int64_t Nf, int32_t estimator, double lam, int32_t prox_kind,
                           double prox_param, int64_t group_len, double mu, double tol, int64_t iters, int32_t linear_sign,
                           int64_t win_lo, int64_t win_hi, int32_t device, float *Syu_re, float *Syu_im, float *Syy,
                           float *Suu, float *x_re, float *x_im, int64_t *iters_out);
int32_t lpvs_problem_create_lpv_rows_f32(const float *Y, int64_t ns, const float *X, const float *V, int64_t N_local,
                                         const float *w, int64_t Nf, int64_t Nv, int32_t normalize, int32_t coulomb,
                                         const double *ranges4, int32_t device, lpvs_problem **out);
int32_t lpvs_problem_solve_ridge_f32(lpvs_problem *h, double ridge, float *x_out);
int32_t lpvs_admm_set_state_f32(lpvs_problem *h, const float *x, const float *z, const float *u, int64_t iters_done);
int32_t lpvs_admm_init_f32(lpvs_problem *h, const float *x0, double mu, double tol,
                           int32_t linear_sign);
int32_t lpvs_admm_get_f32(lpvs_problem *h, float *x_out, float *z_out, float *u_out);
int32_t lpvs_problem_get_params_f32(lpvs_problem *h, int32_t which, float *re_out, float *im_out);
int32_t lpvs_ls_spectral_f32(const float *y, const float *t, int64_t N, const float *f, int64_t Nf,
                             double lam, int32_t device, float *re_out, float *im_out);

/* ---- a15 window bookkeeping (DSP.arraysplit as used by src/windows.jl:27-36) ---------- */
int32_t lpvs_window_count(int64_t L, int64_t n, int64_t noverlap, int64_t *count);
int32_t lpvs_window_offsets(int64_t L, int64_t n, int64_t noverlap, int64_t *offsets /* 0-based */,
                            int64_t capacity, int64_t *count);
/* merge (src/windows.jl:57-70): yf is count x n (window-major, row w = window w) -> ym[L] */
int32_t lpvs_merge_f64(const double *yf, int64_t count, int64_t n, int64_t noverlap, int64_t L,
                       double *ym);

/* ---- batched windows: ls_windowpsd(y,t,freqs; estimator = ls_sparse_spectral)  ----------------------
 * src/lsfft.jl:112-126 driving src/lasso.jl:105-126 for every window of src/windows.jl:27-36, all windows
 * [win_lo, win_hi) (0-based, of the k = (L-n) div (n-noverlap) + 1 windows) solved together on `device`:
 * one batch of regressor panels, Gram matrices, factorisations and ADMM iterations per launch instead of a
 * sequential loop.  Windows are independent, so ranks of a multi-GPU job take disjoint [win_lo, win_hi).
 *   W          n window weights (NULL = rect/ones); the estimator is always the weighted 4-argument method
 *   linear_sign LPVS_LINEAR_QUADRATIC_AS_WRITTEN reproduces Quadratic(Q, q=+A'Wy) of src/lasso.jl:119-121
 *   prox_kind  L1, L0, GROUP_L2 or BALL_L0 (README.md:79-83; one workgroup per window selects its r largest)
 *   x_re,x_im  (win_hi-win_lo) x Nf, window-major: fourier2complex(z) of each window
 *   S_out      Nf: sum over these windows, in window order, of |x|^2 (NOT yet divided by k^2); may be NULL
 *   iters_out  per-window iteration count (stops per window at ||x-z|| < tol); may be NULL */
int32_t lpvs_windowpsd_sparse_f64(const double *y, const double *t, int64_t L, int64_t n, int64_t noverlap,
                                  const double *W, const double *freqs, int64_t Nf, int32_t prox_kind,
                                  double prox_param, int64_t group_len, double mu, double tol, int64_t iters,
                                  int32_t linear_sign, int64_t win_lo, int64_t win_hi, int32_t device,
                                  double *x_re, double *x_im, double *S_out, int64_t *iters_out);

/* phase times of the calling thread's last batched-window call (HIP events on the library's stream), out[0..9]:
 * Gram + rhs ms, inverse ms, ADMM / dense-solve ms, windows, batch mat-vec microseconds per launch (only when the environment
 * has LPVS_WINDOW_MATVEC_TIMING: 200 extra launches after the last pass), windows of that pass, passes, Gram form (0 dense, 1 structured with direct
 * slot sums, 2 structured with the slot sums by non-uniform FFT), bytes of packed inverses one iteration's launch reads, 1 if the pass ran one launch per iteration, 2 if that launch reads 32 of the
 * fixed-point tiles' 36 bits (the stale nibble product, LPVS_STORAGE_MIXED32: the default) (the timed launches are then the stand-alone batch
 * mat-vec of the two-launch scheme: the same product without the update, all 36 bits);
 * out[10..11] (when n_out >= 12): after lpvs_windows_estimate_multi_*: ranks of the RCCL communicator that gathered the
 * coefficients (0 = no collective ran: one device, or shards sharing a device), devices driven */
int32_t lpvs_windowpsd_last_timing(double *out, int32_t n_out);

/* estimator of the batched-window engine */
#define LPVS_EST_SPARSE 1 /* ls_sparse_spectral(y,t,f,W): Quadratic(Q,q) + ADMM        src/lasso.jl:105-126 */
#define LPVS_EST_DENSE 2  /* ls_spectral(y,t,f,W): (A'WA + lam I) \ A'Wy               src/lsfft.jl:74-80  */
#define LPVS_EST_SPARSE_INIT 3 /* ls_sparse_spectral(y,t,f,W; init=true): as LPVS_EST_SPARSE, started from
                                  fourier_solve(A,y,zerofreq,lam) = (A'A + lam^2 I) \ A'y -- the UNWEIGHTED ridge solution, as written (src/lasso.jl:112) */

/* ---- the engine itself: ns signals sharing the sampling points t (Y is L x ns column-major) --------------------------
 * Every window's Gram A'WA and factorisation are formed ONCE and serve all ns right-hand sides A'W y_s -- the y and u of
 * ls_windowcsd / ls_cohere (src/lsfft.jl:150-151,184-185 call the estimator twice per window on the same t).
 *   estimator   LPVS_EST_SPARSE (prox_*, mu, tol, iters, linear_sign as in lpvs_windowpsd_sparse_f64; lam unused),
 *               LPVS_EST_SPARSE_INIT (the same with init = true: lam = the lambda of the starting ridge solve) or
 *               LPVS_EST_DENSE (lam = ridge; the ADMM arguments are ignored)
 *   x_re, x_im  ns x (win_hi - win_lo) x Nf, signal-major then window-major: fourier2complex of every solution
 *   iters_out   ns x (win_hi - win_lo) iteration counts (0 for the dense estimator); may be NULL.  HOST memory only
 *               (as every int64_t* output of this header: iteration counts are bookkeeping the host loop consumes) */
int32_t lpvs_windows_estimate_f64(const double *Y, int64_t ns, const double *t, int64_t L, int64_t n, int64_t noverlap,
                                  const double *W, const double *freqs, int64_t Nf, int32_t estimator, double lam,
                                  int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol, int64_t iters,
                                  int32_t linear_sign, int64_t win_lo, int64_t win_hi, int32_t device, double *x_re,
                                  double *x_im, int64_t *iters_out);

/* The same call returning the raw ADMM state of every problem instead of the packed coefficients: the (x, z) that ADMM returns
 * (src/lasso.jl:170) and the scaled dual variable u (:147,:155), each ns x (win_hi - win_lo) x Nreg, signal-major then window-major,
 * Nreg = 2 Nf (2 Nf - 1 with the zero frequency first) in the reference's ordering [re; im] (src/lasso.jl:93-97).  Sparse estimators
 * only.  x_out / z_out / u_out are HOST arrays; any of them (and iters_out) may be NULL.  What the parity tests hold the window
 * batches' iteration to the oracle with (x, z AND u at the bench's own iteration count); fourier2complex(z) is what
 * lpvs_windows_estimate_f64 returns for the same arguments. */
int32_t lpvs_windows_estimate_state_f64(const double *Y, int64_t ns, const double *t, int64_t L, int64_t n, int64_t noverlap,
                                        const double *W, const double *freqs, int64_t Nf, int32_t estimator, double lam,
                                        int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol, int64_t iters,
                                        int32_t linear_sign, int64_t win_lo, int64_t win_hi, int32_t device, double *x_out,
                                        double *z_out, double *u_out, int64_t *iters_out);

/* ---- the same engine driven over several devices by ONE host process (SURVEY.md section 8(e) "Process model") --------
 * The k windows are split into contiguous ranges over `ngpus` devices (devices[r], or 0..ngpus-1 when devices == NULL;
 * ngpus <= 0: every visible device); one host thread, stream and engine pass per device, no data-path collective; the
 * per-window coefficients are gathered with ONE RCCL all-gather over xGMI (librccl.so.1 is bound at run time, only when
 * ngpus > 1) and returned for ALL k windows: x_re / x_im are ns x k x Nf, iters_out ns x k.  The caller accumulates
 * |x|^2 (src/lsfft.jl:122) or xy conj(xu) (:152, :187-189) in window order.  Window shards reproduce the single-device
 * run bit for bit (fixed segment lengths, one admission decision for the structured Gram).  Arguments may be host or
 * device pointers; device arguments are staged through the host once. */
int32_t lpvs_windows_estimate_multi_f64(const double *Y, int64_t ns, const double *t, int64_t L, int64_t n, int64_t noverlap,
                                        const double *W, const double *freqs, int64_t Nf, int32_t estimator, double lam,
                                        int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol,
                                        int64_t iters, int32_t linear_sign, const int32_t *devices, int32_t ngpus, double *x_re,
                                        double *x_im, int64_t *iters_out);

/* ---- multichannel LPV batches over several devices by ONE host process (BASELINE.json config 5) ------
 * Y is N x ns (one column per channel sharing X, V, w): the channels are split into contiguous ranges over `ngpus` devices
 * (devices[r], or 0..ngpus-1 when NULL; ngpus <= 0: every visible device), one host thread per device builds its shard's Gram and
 * factorisation ONCE and advances the shard's channels together (lpvs_problem_create_lpv_multi_f64); every channel stops at its own
 * ||x-z||_2 < tol.  No data-path collective.  prox as lpvs_problem_set_prox (the reference's ls_sparse_spectral_lpv is
 * LPVS_PROX_GROUP_L2 with group_len = 2 Nv, src/lasso.jl:53-55; BASELINE's cfg5 asks LPVS_PROX_BALL_L0, an extension).
 * re_out / im_out: (Nf*Nv) x ns column-major HOST arrays, channel q in column q, parameter index f + (v-1) Nf (src/lasso.jl:67-68);
 * iters_out: ns iteration counts (HOST, may be NULL).  Inputs may be host or device pointers (staged through the host once). */
int32_t lpvs_lpv_batch_multi_f64(const double *Y, int64_t ns, const double *X, const double *V, int64_t N, const double *w, int64_t Nf,
                                 int64_t Nv, int32_t normalize, int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol,
                                 int64_t iters, const int32_t *devices, int32_t ngpus, double *re_out, double *im_out, int64_t *iters_out);

/* ---- independent LPV signals, several devices, several solves in flight per device (extension: BASELINE.json config 3 as a batch)
 * The loop  [ls_sparse_spectral_lpv(Y[:,q], X[:,q], V[:,q], w, Nv; ...) for q in 1:nsig]  (src/lasso.jl:27-70 per signal) inside the
 * library: every signal has its OWN X and V (N x nsig column-major, like Y), hence its own Gram, factorisation and ADMM run.
 * Contiguous signal ranges go to the devices (ngpus <= 0: every visible device; devices may be NULL); per device `in_flight` host
 * threads (1 .. 8) each solve one signal at a time on a handle and stream of their own -- the matrix-core-bound Gram and factorisation
 * of one solve run under the HBM-bound iterations of another (one MI355X at N = 2^20, n = 8192: 13.9 signals/s with in_flight = 1,
 * 16.4-16.8 with 2).  No collective.  Results do not depend on ngpus / in_flight.  re_out / im_out: (Nf*Nv) x nsig column-major HOST
 * arrays (signal q in column q, parameter index f + (v-1) Nf); iters_out: nsig iteration counts (HOST, may be NULL).
 * Y, X, V may be host or device pointers; a signal whose columns live on a device other than the one that solves it is copied there
 * (peer copy) by the constructor -- kernels never read another device's memory (this holds for every lpvs_problem_create_*). */
int32_t lpvs_lpv_signals_multi_f64(const double *Y, const double *X, const double *V, int64_t N, int64_t nsig, const double *w, int64_t Nf,
                                   int64_t Nv, int32_t normalize, int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol,
                                   int64_t iters, const int32_t *devices, int32_t ngpus, int32_t in_flight, double *re_out, double *im_out,
                                   int64_t *iters_out);

/* ---- ls_windowpsd_lpv                                                                    src/lsfft.jl:267-277
 * S[Nf] = sum over the windows of Windows3(Y, X, V, n, noverlap, rect), in window order, of abs2.(sum(reshape_params(x_i, Nf), dims=2))
 * with x_i = ls_spectral_lpv(y_i, x_i, v_i, w, Nv; lam, coulomb, normalize).x (:239-250: every window its own basis centres and Gram).
 * The windows' Grams are built `in_flight` (1 .. 8) at a time on streams of their own into ONE batch; their factorisations and
 * refined ridge solves then run for all windows at once (the batch machinery of the window engine).  n = length(Y) / nw as the caller
 * computes it (:269).  fva_out (HOST, k = window count entries, may be NULL): each window's fraction of variance explained
 * 1 - var(e)/var(y) (:255) -- the reference warns when it is below 0.9, the bindings do the same.  It is formed from the Gram
 * (x'Gx - 2x'b + y'y), i.e. from separately rounded sums: good to ~1e-8 of var(y), enough for the 0.9 threshold and no more; a constant
 * window (var(y) = 0) reports -Inf, so that it warns.  The windows are solved in chunks sized from the device's free memory; a window's
 * coefficients depend on the chunking to rounding only.
 * LPVS_ENUMERIC when a window's normal equations are singular to working precision (the reference's QR route is the wrapper's).
 * S_out: HOST array. */
int32_t lpvs_windowpsd_lpv_f64(const double *Y, const double *X, const double *V, int64_t N, const double *w, int64_t Nf, int64_t Nv,
                               int64_t n, int64_t noverlap, double lam, int32_t normalize, int32_t coulomb, int32_t device, int32_t in_flight,
                               double *S_out, double *fva_out);

/* ---- ls_windowcsd / ls_cohere on the engine                                              src/lsfft.jl:140-156, :176-193
 * Accumulators over the windows [win_lo, win_hi) in window order (NOT yet normalised: ls_windowcsd returns Syu / k,
 * ls_cohere |Syu|^2 / (Suu Syy)):  Syu += xy .* conj.(xu),  Syy += abs2.(xy),  Suu += abs2.(xu).  Any output may be NULL;
 * x_re / x_im (2 x nwin x Nf: xy of every window, then xu) and iters_out (2 x nwin) are optional. */
int32_t lpvs_windowcsd_f64(const double *y, const double *u, const double *t, int64_t L, int64_t n, int64_t noverlap,
                           const double *W, const double *freqs, int64_t Nf, int32_t estimator, double lam, int32_t prox_kind,
                           double prox_param, int64_t group_len, double mu, double tol, int64_t iters, int32_t linear_sign,
                           int64_t win_lo, int64_t win_hi, int32_t device, double *Syu_re, double *Syu_im, double *Syy,
                           double *Suu, double *x_re, double *x_im, int64_t *iters_out);

/* ---- storage of the inverse the ADMM mat-vec streams (after lpvs_admm_init) ------------------------------------------
 * *kind = 0 full symmetric doubles (n < 2048), 1 tile-packed lower triangle in doubles, 2 in floats (_f32 handles),
 * 3 in 6-byte elements (float head + 16-bit tail = 40 significant bits; LPVS_M_STORAGE=split, or a matrix of which fewer than
 * half of the tiles qualify for 4), 4 mixed: as 3, but tiles whose entries are all small against max|M| are 36-bit fixed point with a
 * per-row step (the default of _f64 handles -- with several right-hand sides the diagonal tiles always stay in the 6-byte format;
 * lpvs_admm_time_matvec reports the bytes a launch reads);
 * LPVS_M_STORAGE=f64 selects 1.  Bit 4 (+16) is set when, with the prox operator currently set, the ADMM iteration runs as ONE
 * launch (kind 4, one right-hand side, L1 / L0 / group prox whose groups divide 128: the tile partials are added into x with 64-bit
 * fixed-point atomics -- exact and order-independent -- and the next launch's tile workgroups apply prox and dual update in their
 * prologue); LPVS_ITERATION=two keeps the separate mat-vec and update launches.  Bit 5 (+32): the fixed-point tiles of kind 4 keep
 * 32 significant bits (4 B per element: handles whose x-update is corrected -- one right-hand side, n >= 2048 -- where the storage
 * error's systematic part leaves the iteration with the inverse's) */
int32_t lpvs_admm_matvec_kind(lpvs_problem *h, int32_t *kind);

#ifdef __cplusplus
}
#endif
#endif /* LPVSPECTRAL_H */
