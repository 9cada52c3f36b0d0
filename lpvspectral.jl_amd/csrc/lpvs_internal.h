// lpvs_internal.h -- shared declarations of the gfx950 implementation behind include/lpvspectral.h
#pragma once
#include <hip/hip_runtime.h>
#include <exception>
#include <new>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/lpvspectral.h"

namespace lpvs {

// ---- environment knobs ------------------------------------------------------------------------------------------------------------
// Three kinds.  (1) Fallbacks of the OPTIONS of the interface (include/lpvspectral.h: LPVS_M_STORAGE, LPVS_ITERATION, LPVS_GRAM_FORM,
// LPVS_NT_LOADS, LPVS_NUDFT, LPVS_WINDOW_CHUNK_MB, LPVS_WINDOWS_IN_FLIGHT, LPVS_RESERVE_CUS, LPVS_XUPDATE_CORRECTION=0) and (2) operational
// ones that change no result (LPVS_POOL_GIB, LPVS_BATCH_PANEL_GIB, LPVS_TRACE, LPVS_NO_GRAPH, LPVS_WINDOW_MATVEC_TIMING, LPVS_MULTI_FORCE_RCCL,
// LPVS_MULTI_ALLOW_SHARED_DEVICE): plain getenv.  (3) EXPERIMENT knobs -- schedules and kernel variants kept for the A/B measurements the
// design documents quote, and the numerics studies (LPVS_FIX_BITS, LPVS_NIB_*, LPVS_XB_REFINE, LPVS_XUPDATE_CORRECTION schedules, LPVS_PHASE,
// LPVS_FACTOR*, LPVS_PIVOT*, LPVS_KW, LPVS_LOOKAHEAD, LPVS_CHAIN, LPVS_BAND_TILE, LPVS_RU_STAGE, LPVS_MULTI_*): read through experiment_env, which
// answers nullptr unless the process ALSO has LPVS_EXPERIMENTS=1 (the test-suite and tools/ set it; a production process that inherits a
// stray LPVS_FIX_BITS does not change its results), and is compiled out altogether with -DLPVS_NO_EXPERIMENTS.
inline const char *experiment_env(const char *name) {
#ifdef LPVS_NO_EXPERIMENTS
    (void)name;
    return nullptr;
#else
    const char *on = getenv("LPVS_EXPERIMENTS");
    return (on && on[0] == '1') ? getenv(name) : nullptr;
#endif
}

// ---- error plumbing ---------------------------------------------------------------------
void set_error(const char *fmt, ...);
#define LPVS_HIP(call)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            lpvs::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                            __LINE__);                                                       \
            (void)hipGetLastError(); /* clear the sticky error state */                      \
            return e_ == hipErrorOutOfMemory ? LPVS_ENOMEM : LPVS_EDEVICE;                   \
        }                                                                                    \
    } while (0)
#define LPVS_TRY(call)                \
    do {                              \
        int32_t rc_ = (call);         \
        if (rc_ != LPVS_OK) return rc_; \
    } while (0)

// ---- options (include/lpvspectral.h LPVS_OPT_*; api.hip) ------------------------------------------------------------
// value in effect: the explicit value (a handle's, or a job's captured copy of its caller's defaults), else the calling thread's
// default, else what the environment variable of the same name says, else 0 (the library's own choice)
constexpr int kOptCount = 10;
int option_in_effect(int option, int explicit_value = 0);
void capture_default_options(int *opt /*[kOptCount]*/);
double infinity_cache_bytes();   // the device's last-level (Infinity) cache from the KFD topology, 256 MiB when it cannot be read   // the calling thread's defaults (for work handed to other threads)

// synchronises a stream when the scope is left: DevBufs declared BEFORE it return to the pool only after the stream is idle
struct DrainOnExit {
    hipStream_t s;
    explicit DrainOnExit(hipStream_t s_) : s(s_) {}
    ~DrainOnExit() { (void)hipStreamSynchronize(s); }
};

// Body of a worker std::thread: an exception leaving it would end the process (std::terminate) -- it becomes a status instead.
template <class F, class OnFail>
static inline void run_guarded(F &&f, OnFail &&on_fail) {
    try { f(); }
    catch (const std::bad_alloc &) { set_error("out of host memory in a worker thread of the library"); on_fail(LPVS_ENOMEM); }
    catch (const std::exception &e) { set_error("worker thread of the library: %s", e.what()); on_fail(LPVS_EDEVICE); }
    catch (...) { set_error("worker thread of the library: unknown exception"); on_fail(LPVS_EDEVICE); }
}

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device buffers ---------------------------------------------------------------------
// RAII device allocation; copy_in accepts a host or a device source pointer.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;   // true size of the block (>= the size asked for when it came from the cache)
    int dev = -1;
    int32_t alloc(size_t nbytes);
    void release();
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <class T> T *as() const { return static_cast<T *>(p); }
};
size_t pool_cached_bytes(int dev);   // bytes of freed blocks the caching allocator holds for a device (handed out again before hipMalloc is asked)
bool is_device_ptr(const void *p);
int device_of_ptr(const void *p);   // owning device, -1 for host memory
// dst device <- src (host or device); dst (host or device) <- src device
int32_t copy_to_device(void *dst_dev, const void *src, size_t bytes, hipStream_t s);
int32_t copy_from_device(void *dst, const void *src_dev, size_t bytes, hipStream_t s);

// ---- basis tables (basis.hip) -----------------------------------------------------------
// Fourier regressor, column-major N x Nreg (reference layout)          src/lsfft.jl:26-49
int32_t launch_fourier_regressor_colmajor(const double *t, int64_t N, const double *f, int64_t Nf,
                                          int zerofreq, double *A, hipStream_t s);
// Fourier regressor as a row-major panel P[n][ld] (k-major operand of the Gram kernel);
// columns >= Nreg up to ld are zero-filled.
int32_t launch_fourier_panel(const double *t, int64_t N, const double *f, int64_t Nf, int zerofreq,
                             double *P, int64_t ld, hipStream_t s);
// batch of window panels P[q][nrows][ld] (window q starts at sample toff[q], n samples, zero pad rows)
int32_t launch_window_panels(const double *t, const int64_t *toff_dev, int nbatch, int64_t n, int64_t nrows, const double *f,
                             int64_t Nf, int zerofreq, double *P, int64_t ld, hipStream_t s);
// trig table T[n][f] = (cos(w_f x_n), -sin(w_f x_n))                   src/lasso.jl:39
int32_t launch_trig_table(const double *X, int64_t N, const double *w, int64_t Nf, double2 *T,
                          hipStream_t s);
// activation table K[n][ldk] (row-major, nb valid entries per row)     src/lsfft.jl:195-207
int32_t launch_basis_table(const double *V, int64_t N, const double *vc, int64_t nb, double gamma,
                           int normalize, int coulomb, double *K, int64_t ldk, hipStream_t s);
// min / max / max|.| of V (device), results to host
int32_t device_minmax(const double *V, int64_t N, double *lo, double *hi, double *amax,
                      hipStream_t s);
// materialised LPV regressor, column-major N x 2*Nf*nb                 src/lasso.jl:35-50
int32_t launch_lpv_regressor_colmajor(const double2 *T, const double *K, int64_t ldk, int64_t N,
                                      int64_t Nf, int64_t nb, int permuted, double *Phi,
                                      hipStream_t s);
// column-major m x n -> k-major panel P[m][ld], pad columns zero-filled
int32_t launch_transpose_to_panel(const double *A, int64_t m, int64_t n, double *P, int64_t ld,
                                  hipStream_t s);

// ---- Gram (gram.hip) --------------------------------------------------------------------
struct GramPlan {
    int64_t n = 0;       // unknowns (valid columns)
    int64_t N = 0;       // samples
    int64_t tiles = 0;   // lower-triangle 128x256 tiles
    int64_t ksplit = 0;  // sample chunks
    int64_t rows_per_chunk = 0;
    size_t slab_bytes = 0;
    int64_t pairs = 0;   // KRS: nb(nb+1)/2, else 0
    int64_t np2 = 0;     // KRS: row space 2Nf
};
GramPlan make_gram_plan(int64_t n, int64_t N, int64_t nbatch = 1);   // nbatch: problems solved together
GramPlan make_gram_plan_pairs(int64_t Nf, int64_t nb, int64_t N);
bool gram_krs_fits(int64_t nb);   // does the symmetric-pair form fit LDS for this many basis functions?
// symmetric-pair form of the LPV Gram (see gram.hip): KK = pair products of the activation table
int32_t launch_pair_table(const double *K, int64_t ldk, int64_t nb, int64_t Npad, double *KK, hipStream_t s);
int32_t launch_gram_krs(const GramPlan &pl, const double2 *T, int64_t Nf, const double *KK, int64_t nb,
                        double *slab, hipStream_t s);
int32_t launch_gram_reduce_krs(const GramPlan &pl, const double *slab, int64_t nb, double *G3, double *G,
                               int64_t ldg, hipStream_t s);
// Khatri-Rao form: Phi[k][f*2nb + c] = T[k][f][c>=nb] * K[k][c mod nb]
int32_t launch_gram_kr(const GramPlan &pl, const double2 *T, int64_t Nf, const double *K,
                       int64_t ldk, int64_t nb, double *slab, hipStream_t s);
// panel form: Phi[k][c] = P[k*ld + c], optional row weights W (applied once: G = P' diag(W) P)
int32_t launch_gram_panel(const GramPlan &pl, const double *P, int64_t ld, const double *W,
                          double *slab, hipStream_t s);
int32_t launch_gram_panel_batch(const GramPlan &pl, int nbatch, const double *P, int64_t batch_stride_P, int64_t ld,
                                const double *W, double *slab, hipStream_t s);
int32_t launch_gram_reduce_batch(const GramPlan &pl, int nbatch, const double *slab, double *G, int64_t ldg, hipStream_t s);
// G[a][b] (ldg x ldg, symmetric, full) = sum over chunks of the slabs
int32_t launch_gram_reduce(const GramPlan &pl, const double *slab, double *G, int64_t ldg,
                           hipStream_t s);
// b = Phi' (W .* y)
int32_t launch_rhs_kr(const double2 *T, int64_t Nf, const double *K, int64_t ldk, int64_t nb,
                      const double *y, int64_t N, double *b, double *scratch, size_t scratch_bytes,
                      hipStream_t s);
int32_t launch_rhs_panel(const double *P, int64_t ld, int64_t ncol, const double *W, const double *y,
                         int64_t N, double *b, double *scratch, size_t scratch_bytes, hipStream_t s);
size_t rhs_scratch_bytes(int64_t N, int64_t n);
int32_t launch_rhs_panel_batch(int nbatch, const double *P, int64_t strideP, int64_t ld, int64_t ncol, const double *W,
                               const double *y, const int64_t *yoff_dev, int64_t N, double *b, int64_t ldb, double *scratch,
                               size_t scratch_bytes, hipStream_t s);

// ---- structured Gram for arithmetic-progression frequency grids (nudft.hip) ----------------------------------
struct ApStep { double hi[8], lo[8]; };   // b*D in double-double, b = 0..7 (in-group offsets of the slot progressions)
size_t nudft_partial_bytes(int64_t N, int64_t nslots, int64_t nq);
int64_t nudft_rows_per_chunk(int64_t N, int64_t nslots);
int32_t launch_nudft(const double *x, const double *y, int64_t N, const double *Wt, int64_t ldw, int nq, const double *om_hi,
                     const double *om_lo, int nslots, const ApStep &step, double *partial, double *tab, hipStream_t s);
// windows of one signal: seg_dev holds nwin*segs_per_window triples {first sample, end sample, first sample of the window}
int32_t launch_nudft_windows(const double *x, const double *y, const double *Wt, const double *om_hi, const double *om_lo, int nslots,
                             const ApStep &step, const int64_t *seg_dev, int nwin, int segs_per_window, double *partial, double *tab,
                             hipStream_t s);
// the same slot sums by a type-1 non-uniform FFT when the slot frequencies are j * D (nufft.hip); work: nufft_work_bytes
int nufft_grid_size(int64_t nslots);
bool nufft_applicable(int64_t N, int64_t nslots, int64_t nq);
size_t nufft_work_bytes(int64_t N, int64_t nslots, int64_t nq);
int32_t launch_nufft_tab(const double *x, const double *y, int64_t N, double xam, const double *Wt, int64_t ldw, int nq, double D_hi, double D_lo,
                         int mode0, int nslots, int nf, bool reuse_coords, void *work, double *tab, hipStream_t s);
constexpr int32_t kNufftNonFinite = -1000;       // launch_nufft_tab: a weight vector holds NaN / Inf -- take the direct sums (not an error)
bool nufft_windows_applicable(int64_t n, int64_t nslots);
std::vector<double> nufft_window_scale(int nf, int mode0, int nslots);
int32_t launch_nufft_window_spread(const double *x, const double *W, const double *yA, const double *yB, bool hasB, const int64_t *offs_dev, int nwin,
                                   int64_t n, double D_hi, double D_lo, int nf, double *gridA, double *gridB, int64_t grid_bstride, hipStream_t s);
int32_t launch_nufft_window_modes(const double *grids, int64_t grid_bstride, int nwin, int nf, int mode0, int nslots, const double *scale_dev, double *tab,
                                  int64_t tab_bstride, hipStream_t s);
// s0 = slot of the sum frequency 2a (sum of (f, f') = slot s0 + f + f'), delta = residual of 2a against that slot
int32_t launch_ap_assemble_fourier(const double *tab, const double *eps, int64_t Nf, int64_t s0, double delta, int zf, int64_t n, double *G,
                                   int64_t ldg, int nbatch, int64_t tab_stride, int64_t g_stride, hipStream_t s);
int32_t launch_ap_rhs_fourier(const double *tab, const double *eps, int64_t Nf, int zf, double *b, int nbatch, int64_t tab_stride,
                              int64_t b_stride, hipStream_t s);
int32_t launch_ap_assemble(const double *tab, const double *eps, int64_t Nf, int64_t s0, double delta, int64_t nb, int64_t n, double *G,
                           int64_t ldg, hipStream_t s);
int32_t launch_ap_rhs(const double *tab, const double *eps, int64_t Nf, int64_t nb, double *b, hipStream_t s);

// ---- dense symmetric inverse (linalg.hip) ------------------------------------------------
// In-place inverse of the SPD matrix A (np x np, np % 64 == 0, full symmetric storage) by
// blocked symmetric sweeps; work holds 2 panels of np x 64 + one 64 x 64 block.
// Optional helper objects of the two-level sweep's look-ahead: the next pivot panel is prepared on a second
// (high-priority) stream while the bulk of the current trailing update runs on the caller's stream.
struct SweepAux {
    hipStream_t side = nullptr;
    hipStream_t bulk = nullptr;   // trailing updates of the pair schedule: a stream whose CU mask leaves a few CUs to the pivot chains (or nullptr)
    hipEvent_t panel = nullptr, rest = nullptr, band = nullptr, second = nullptr, bulkdone = nullptr;
    int32_t ensure();
    ~SweepAux();
};
size_t spd_inverse_work_bytes(int64_t np);
int32_t spd_inverse_inplace(double *A, int64_t np, double *work, int *status_dev, hipStream_t s, SweepAux *aux = nullptr);
int32_t launch_add_diag(double *M, int64_t np, int64_t n, double shift, hipStream_t s);
int32_t launch_add_diag_batch(double *M, int64_t np, int64_t n, double shift, int nbatch, hipStream_t s);
int32_t spd_inverse_inplace_batch(double *A, int64_t np, int nbatch, double *work, int *status_dev, hipStream_t s);
// C (m x m, ld) = A * B for symmetric np x np operands (test/diagnostic helper)
int32_t launch_symm_matmul(const double *A, const double *B, double *C, int64_t np, hipStream_t s);

// ---- ADMM (admm*.hip: the family map is at the head of admm.hip) --------------------------------------------------------------------
struct AdmmStatus {  // lives in device memory, copied back after each run
    long long iters;
    int converged;
    int pad;
    double nxz;
};
struct AdmmParams {
    const double *M;   // (G + I/mu)^-1, np x np
    int64_t np;        // padded size (multiple of 64)
    int64_t n;         // valid size
    const double *b;   // linear term (already signed)
    double *x, *z, *u, *rhs;
    double mu, tol;
    int prox_kind;
    double prox_param;
    int64_t group_len;
    AdmmStatus *status;
    double *scratch;   // >= 2*np doubles (top-r selection keys)
    double *part;      // symv_part_doubles(np) doubles of tile partials, or nullptr (full mat-vec)
    const double *Mp;  // tile-packed lower triangle of M (symv_packed_doubles(np)), or nullptr
    int ns;            // right-hand sides sharing M (signals of a shared-regressor batch); vectors are [ns][np]
    int mp_f32 = 0;    // Mp holds float (the _f32 entry points: M is streamed in single precision, arithmetic stays double)
    int mp_split = 0;  // Mp holds 6-byte elements (float head + 16-bit tail, 40 significant bits; see admm.hip)
    // 32-bit reads with a STALE NIBBLE PRODUCT (admm.hip, launch_nibble_refresh): the iteration streams the 32 leading bits of the fixed-point
    // tiles and the offset vector carries N rhs_g of the last refresh, xb = xb_corr + N rhs_g, re-formed after the launches g = 1 and g = 0 mod
    // nib_period.  nib_period = 0: off.  xb_corr: the offset vector without any nibble term; nib_rhs: np doubles of scratch.
    int nib_period = 0;
    int nib_ramp = 0;   // > 0: the first refreshes are denser -- the period is min(nib_period, max(1, 2^floor(log2 g) / nib_ramp)) (the right-hand side moves fastest in the first iterations)
    const double *xb_corr = nullptr;
    double *nib_rhs = nullptr;
    long long *nib_acc = nullptr;   // np integers, zero between refreshes: the one-launch iteration's sums of the nibble planes' product (admm_iter_mixed_kernel<..., NIBR>)
    double *nib_part = nullptr;   // the nibble product's own per-tile partials (2 ntiles TS doubles: the two-launch iteration's are in flight in `part` when a refresh runs)
    int mp_fix32 = 0;  // the fixed-point tiles keep 32 significant bits: their nibbles are zero and are not read (handles whose x-update is corrected: the storage error's systematic part leaves the iteration with the inverse's)
    const unsigned char *mp_types = nullptr;   // mixed storage: per-tile format, 1 = 36-bit fixed point (admm.hip); several right-hand sides: diagonal tiles always 0
    // offset form of the x-update (single-problem tile-packed path): x = xb + M (z-u)/mu with xb = M b computed once from
    // the full-precision inverse; the per-iteration product then never multiplies the large constant vector b by the
    // reduced-precision copy of M (its rounding would otherwise be amplified by cond(G + I/mu)).  nullptr: x = M (b + (z-u)/mu).
    const double *xb = nullptr;
    // one-launch iteration (admm.hip): fixed-point accumulators, alternate u, per-parity block norms / maxima / quanta; nullptr = the
    // two-launch iteration.  fi_base = iterations committed before the chunk about to be launched.
    double *fi = nullptr;
    long long fi_base = 0;
    double fi_R = 0, fi_xbmax = 0;   // largest absolute row sum of M (x 1) and max|xb|: host copies of fi's constants (single problems)
    int fi_prefetch_all = 0;         // every tile is in the fixed format: diagonal tiles are requested up front too
    int opt_iteration = 0, opt_nt_loads = 0;   // LPVS_OPT_ITERATION / LPVS_OPT_NT_LOADS of the handle (0: thread default / environment)
    // one launch per iteration of the full-matrix path (np < kSymmetricMinNp; admm_small_iter_kernel): 4 ints per signal --
    // {converged before launch parity 0, parity 1, launch of the chunk that converged (-1: none), unused}; nullptr = two launches.
    // The alternate x / u buffers of that scheme are the two halves of `scratch`.
    int *sm_ctl = nullptr;
};
bool small_iter_applicable(const AdmmParams &p);
// xb (written) = xb_corr + N rhs, N = the nibble planes of the packed copy's fixed-point tiles; rhs = p.rhs (from_state = false) or (z - u_src) / mu
// (split: xb_corr (written) = xb - N rhs instead: after a correction re-formed xb for the right-hand side in memory)
int32_t launch_nibble_refresh(const AdmmParams &p, bool from_state, const double *u_src, hipStream_t s, bool split = false);
// is the stale nibble product refreshed after launch g?  An absolute function of g: the iterates do not depend on the chunking.
inline bool nib_refresh_due(long long g, int period, int ramp) {
    if (period <= 0) return false;
    if (ramp <= 0) return g == 1 || g % period == 0;
    long long p2 = 1;
    while (2 * p2 <= g) p2 *= 2;                      // 2^floor(log2 g)   (g = 0: 1)
    long long step = p2 / ramp;
    if (step < 1) step = 1;
    if (step > period) step = period;
    return g % step == 0;
}
// the one-launch iteration multiplies the nibble planes inside the launches after which a refresh is due (otherwise: launch_nibble_refresh's three kernels)
bool nib_fused_applies(const AdmmParams &p);
// one step of iterative refinement for the right-hand side the next x-update multiplies (p.rhs), its residual in twice-the-mantissa
// accumulation against H = G + shift I; xb_eff = xb0 + M~ (v - H M~ v).  t: 3 x [ns][np] scratch (admm.hip says why)
// b != NULL: for the whole right-hand side b + v, which also corrects an unrefined xb0 = M b
int32_t launch_xupdate_correction(const AdmmParams &p, const double *G, double shift, const double *b, const double *xb0, double *xb_eff, double *t, hipStream_t s);
size_t fi_doubles(int64_t np, int64_t nprob = 1);
bool fi_applicable(const AdmmParams &p);
int32_t launch_fi_setup(const AdmmParams &p, long long base, bool with_consts, hipStream_t s, double r_known = -1.0);   // r_known > 0: the largest absolute row sum is known (p.fi_R)
int32_t fi_read_consts(const AdmmParams &p, double out[2], hipStream_t s);
size_t symv_part_doubles(int64_t np, int64_t ns = 1);
size_t symv_packed_doubles(int64_t np);
int32_t launch_pack_tiles(const double *M, int64_t np, double *Mp, hipStream_t s);
int32_t launch_pack_tiles_f32(const double *M, int64_t np, float *Mp, hipStream_t s);
int32_t launch_pack_tiles_split(const double *M, int64_t np, unsigned char *Mp, hipStream_t s);   // 6 * symv_packed_doubles(np) bytes
int32_t launch_pack_tiles_mixed(const double *M, int64_t np, unsigned char *Mp, unsigned char *types, unsigned long long *absmax, hipStream_t s,
                                bool diag_float = false, double *abs_part = nullptr, int64_t n_valid = 0, double *rows_scratch = nullptr, int fix_bits = 36);   // diag_float: diagonal tiles always in the float-head format (multi-signal handles); abs_part: also the largest absolute row sum -> absmax[1]
bool multi_signal_fixed_tiles_ok(int64_t np);               // the multi-signal tile product in use reads fixed-point off-diagonal tiles
void release_panel_plans();                                 // admm_multi.hip: frees the cached device tables of the panel walk
int32_t launch_pack_tiles_mixed_batch(const double *M, int64_t np, int nbatch, unsigned char *Mp, unsigned char *types, unsigned long long *absmax,
                                      hipStream_t s, bool diag_float = false, double *abs_part = nullptr, int64_t n_valid = 0, double *rows_scratch = nullptr, int fix_bits = 36);
constexpr size_t kMixedFixedTileBytes = 128 * 128 * 4 + 128 * 128 / 2 + 128 * 4, kMixedFloatTileBytes = 128 * 128 * 6;
constexpr size_t kMixedFixed32TileBytes = 128 * 128 * 4 + 128 * 4;   // read per 32-bit fixed-point tile: heads + steps (same slot layout, the nibble area is skipped)
// element conversions for the _f32 entry points (device buffers)
int32_t launch_cvt_f32_f64(const float *src, double *dst, int64_t count, hipStream_t s);
int32_t launch_cvt_f64_f32(const double *src, float *dst, int64_t count, hipStream_t s);
constexpr int64_t kSymmetricMinNp = 2048;  // below this the iteration is launch-latency bound: plain mat-vec
// batch of nbatch independent problems of one shape: arrays are [nbatch][np]; matrices [nbatch / nrhs][np][np]
struct AdmmBatch {
    const double *M;
    int64_t np, n;
    int nbatch;
    const double *b;
    double *x, *z, *u, *rhs;
    double mu, tol;
    int prox_kind;
    double prox_param;
    int64_t group_len;
    AdmmStatus *status;   // [nbatch]
    double *part;         // symv_part_doubles(np, nbatch) doubles, zeroed before the first iteration (or nullptr)
    const double *Mp;     // [nbatch / nrhs] tile-packed lower triangles (or nullptr: plain mat-vec)
    int nrhs = 1;         // problems per matrix: problem q uses matrix q / nrhs (signals sharing a window's Gram)
    const double *xb = nullptr;   // offset form: x = xb + M~ (z-u)/mu, xb = M b per problem ([nbatch][np]); nullptr: classic
    int mp_split = 0;     // Mp holds 6-byte elements (float head + 16-bit tail)
    const unsigned char *mp_types = nullptr;   // mixed storage: [matrix][tile] formats (1 = 36-bit fixed point), see admm.hip
    // one-launch iteration for the whole batch (admm.hip; nrhs == 1, mixed storage, offset form, fusable prox): fi_doubles(np, nbatch)
    // doubles of accumulators / records, iterations committed before the chunk about to be launched, "every tile is fixed point"
    double *fi = nullptr;
    long long fi_base = 0;
    int fi_prefetch_all = 0;
    int opt_iteration = 0, opt_nt_loads = 0;   // LPVS_OPT_ITERATION / LPVS_OPT_NT_LOADS of the call (0: thread default / environment)
    double *scratch = nullptr;                 // IndBallL0 with n > 8192: 2 * np doubles per problem (selection keys), else unused
    const double *x0 = nullptr;                // init = true: the starting point of every problem ([nbatch][np]); nullptr: zeros
    // 32-bit reads of the fixed-point tiles with the stale nibble product (AdmmParams: nib_period, nib_ramp, xb_corr, nib_acc), one-launch batches only:
    // xb is then written by the refreshes (xb = xb_corr + N rhs_g per problem), xb_corr keeps M b
    int nib_period = 0, nib_ramp = 0;
    const double *xb_corr = nullptr;           // [nbatch][np]
    long long *nib_acc = nullptr;              // [nbatch][np], zero between refreshes
};
bool fi_batch_applicable(const AdmmBatch &p);
int32_t launch_fi_batch_setup(const AdmmBatch &p, hipStream_t s);   // constants and the records of iteration 0 (after launch_admm_batch_init)
bool admm_batch_uses_tiles(const AdmmBatch &p);   // will launch_admm_batch_iterations take the tile-packed path for this batch?
int32_t launch_batch_matvec(const double *A, int64_t np, int nprob, int nrhs, const double *v, double *out, hipStream_t s);   // out_q = A[q / nrhs] v_q
int32_t launch_pack_tiles_split_batch(const double *M, int64_t np, int nbatch, unsigned char *Mp, hipStream_t s);
int32_t launch_admm_batch_matvec_only(const AdmmBatch &p, int reps, hipStream_t s);
int32_t launch_batch_ridge_solve(const double *Q, const double *M, int64_t np, int64_t n, int nprob, int nrhs, const double *b, double ridge,
                                 int steps, double *x, double *t1, double *t2, hipStream_t s);
int32_t launch_pack_tiles_batch(const double *M, int64_t np, int nbatch, double *Mp, hipStream_t s);
int32_t launch_admm_batch_init(const AdmmBatch &p, hipStream_t s);
int32_t launch_admm_batch_iterations(const AdmmBatch &p, int64_t iters, hipStream_t s);
int32_t launch_admm_init(const AdmmParams &p, hipStream_t s);              // z=x, u=0, rhs, status=0
int32_t launch_admm_iterations(const AdmmParams &p, int64_t iters, hipStream_t s);
// resume: x, z, u hold saved iterates; recompute the next right-hand side, set the iteration count, clear the flags
int32_t launch_admm_restate(const AdmmParams &p, int64_t iters, hipStream_t s);
// the mat-vec kernel alone, `reps` times (x is recomputed from the current rhs; nothing else changes)
int32_t launch_admm_matvec_only(const AdmmParams &p, int reps, hipStream_t s);
int32_t launch_rect_matvec(const double *A, int64_t rows, int64_t cols, int64_t ld, const double *v, double *out, hipStream_t s);
int32_t launch_fourier_dual_panel(const double *t, int64_t N, const double *f, int64_t Nf, int zerofreq, double *D, int64_t ldn,
                                  int64_t nrows, hipStream_t s);
// x = M b refined against H = G + ridge I:  x += M (b - H x), `steps` times (t1, t2: np doubles of scratch)
int32_t launch_offset_vector_refined(const double *G, const double *M, int64_t np, int64_t n, int ns, const double *b, double shift, int steps,
                                     double *xb, double *t1, double *t2, hipStream_t s);
int32_t launch_ridge_solve_refined(const double *G, const double *M, int64_t np, int64_t n, const double *b, double ridge, int steps,
                                   double *x, double *t1, double *t2, hipStream_t s);
// x = Minv * rhs_in (one GEMV; ridge solves)
int32_t launch_symv(const double *M, int64_t np, const double *rhs, double *x, hipStream_t s, int ns = 1);   // x[q] = M rhs[q], q < ns

// ---- batched-window engine (api.hip) ------------------------------------------------------------------------------
struct WinJob {
    const double *const *ys; int64_t ns;          // ns signals sharing the sampling points t
    const double *t; int64_t L, n, noverlap;
    const double *W, *freqs; int64_t Nf;
    int estimator; double lam;                    // LPVS_EST_SPARSE / LPVS_EST_DENSE (ridge lam)
    int prox_kind; double prox_param; int64_t group_len; double mu, tol; int64_t iters; int linear_sign;
    int64_t win_lo, win_hi; int device;
    double t_absmax = -1.0;                       // max|t| over the WHOLE record when (t, L) is only a span of it (< 0: compute)
    bool f32_grid = false;                        // the frequency grid was widened from floats: snap it to the progression it was rounded from
    int opt[kOptCount] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // the caller's default options (captured on ITS thread: multi.hip runs the job on workers)
    bool opt_captured = false;
    // lpvs_windows_estimate_state_f64: the raw ADMM state (x, z, u of src/lasso.jl:146-155, reference ordering [re; im]) of every problem, written by the
    // pass that solved it into HOST arrays [ns][st_nwin][nreg] indexed by (window - st_base); any of the three may be null
    double *st_x = nullptr, *st_z = nullptr, *st_u = nullptr; int64_t st_base = 0, st_nwin = 0;
};
// sink(window index relative to win_lo, signal, re[Nf], im[Nf], iterations): window order, signals innermost
typedef std::function<void(int64_t, int64_t, const double *, const double *, int64_t)> WinSink;
int32_t windows_engine_run(const WinJob &job, const WinSink &sink, bool chunked = false);   // chunked: in cache-sized chunks, two parts of a chunk in flight (api.hip)
void windows_last_timing(double *out10);          // the calling thread's last engine call
void windows_set_timing(const double *in10);
void windows_set_multi_info(int rccl_ranks, int devices);   // out[10], out[11] of lpvs_windowpsd_last_timing

}  // namespace lpvs
