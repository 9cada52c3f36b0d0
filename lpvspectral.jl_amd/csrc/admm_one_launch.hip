// admm_one_launch.hip -- the whole ADMM iteration in ONE launch: single-signal handles on the mixed / f32 storage and window
// batches.  The tile workgroups add their partial sums into the next x with 64-bit fixed-point atomics, the next launch's workgroups rebuild
// their right-hand-side blocks (prox + dual update, src/lasso.jl:152-155) in the prologue; launch boundaries are the only synchronisation.
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

enum { FI_FIRST = 0, FI_MID = 1, FI_LAST = 2 };   // modes of admm_iter_mixed_kernel
typedef void (*FiKernel)(AdmmParams, const unsigned char *, const unsigned char *, int, int, long long, int, int, int, size_t, int);
static FiKernel fi_kernel(int mode, bool small, bool batch, bool nt, bool pa, bool f32);

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
    const char *e = experiment_env("LPVS_NIB_FUSED");
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
int32_t launch_fi_chunk(const AdmmParams &p, int64_t iters, bool batch, size_t mp_stride, bool prefetch_all, hipStream_t s) {
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

}  // namespace lpvs
