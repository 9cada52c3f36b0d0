// admm_multi.hip -- the ADMM mat-vec of handles with several right-hand sides sharing M (lpvs_problem_create_lpv_multi_*; BASELINE cfg5): a tile is
// two small GEMMs on the f64 matrix cores.  symv_tile_mfma_ws_kernel (persistent, wave-specialised: loader waves decode the packed tiles into LDS
// images, one MFMA wave per SIMD) and its launch plan (runs of tiles, the opt-in column-panel walk); round 1's LDS-DMA kernel for comparison.
// The consumers of its partials (symv_reduce_runs_kernel, admm_fused_update2_kernel) live in admm.hip and find the records through
// stream_layout() / stream_table().
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

// ---- multi-signal tile product on the matrix cores ------------------------------------------------------------------
// With ns right-hand sides sharing M the per-signal cross-lane row reductions of symv_tile_multi_kernel, not the memory
// pipe, bound the kernel (2.9 TB/s at ns = 8).  Here a tile is two small GEMMs on v_mfma_f64_16x16x4_f64:
//     P1[i][s] = sum_c T[i][c] R_J[c][s]      (A operand = T, 16 rows x 4 cols;  B = R_J)
//     P2[s][c] = sum_i R_I[i][s] T[i][c]      (A = R_I', B = T, 4 rows x 16 cols)
// The two products need T in transposed operand layouts, so the tile goes through LDS in four 32-row stages (LDS-DMA,
// one 1 KiB row per instruction, rows padded by 16 B) and is read from there in either layout: waves 0-1 form P1 of the
// stage's two 16-row blocks, waves 2-3 accumulate P2 over the stages (four 16-column blocks each).  43 KB of LDS per
// workgroup: three workgroups per CU keep the memory pipe busy while others multiply (double-buffering the stages
// inside a workgroup at two workgroups per CU was slower: 1.36 vs 1.24 ms at ns = 8, n = 32768).  Signals are processed eight
// (ns <= 8: the MFMA's 16-wide signal dimension is half used) or sixteen at a time (no padding: the tile product costs the
// same matrix-pipe time for twice the signals).  Partials have the layout of the scalar kernels, so the update kernels
// are shared.
constexpr int MT_RS = TS + 2;                        // padded row stride of the staged rows (doubles)
constexpr int MT_ROWS = 32;                          // tile rows per stage
// NS = signals per pass: 8 (half of the MFMA's 16-wide signal dimension is padding) or 16 (none); LDS 43 / 53 KB
template <int NS> constexpr size_t symv_mfma_lds() { return sizeof(double) * ((size_t)MT_ROWS * MT_RS + (size_t)MT_ROWS * NS + (size_t)TS * NS); }

__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int NS>
__global__ void __launch_bounds__(256, 3)
symv_tile_mfma_kernel(const double *__restrict__ Mp, const double *__restrict__ rhs_all, int64_t np, int ns, int ntiles,
                      double *__restrict__ part1_all, double *__restrict__ part2_all, const AdmmStatus *status) {
    if (status != nullptr) {
        bool all = true;
        for (int q = 0; q < ns; ++q) all = all && status[q].converged;
        if (all) return;
    }
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *stg = lds;                               // [32][MT_RS]   the current 32 rows of the tile
    double *ri = stg + MT_ROWS * MT_RS;              // [32][NS]      right-hand sides of row block I, rows of the current stage
    double *rj = ri + MT_ROWS * NS;                  // [128][NS]     right-hand sides of row block J
    const int t = blockIdx.x;
    int I, J;
    tile_index(t, I, J);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const double *src = Mp + (int64_t)t * TS * TS;
    for (int s0 = 0; s0 < ns; s0 += NS) {
        const int nsb = ns - s0 < NS ? ns - s0 : NS;
        f64x4 acc2[4];                               // waves 2-3: P2 blocks, columns 64*(wave-2) + 16*u .., over all stages
#pragma unroll
        for (int u = 0; u < 4; ++u) acc2[u] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int q = 0; q < TS / MT_ROWS; ++q) {
            __syncthreads();                         // everyone is done with the previous contents of the LDS images
#pragma unroll
            for (int r = 0; r < MT_ROWS / 4; ++r) {  // wave w brings rows 8w..8w+7 of the stage: one 1 KiB row per instruction
                const int row = wave * (MT_ROWS / 4) + r;
                glds16(src + (int64_t)(MT_ROWS * q + row) * TS + 2 * lane, stg + row * MT_RS);
            }
            for (int e = tid; e < MT_ROWS * NS; e += 256) {   // R_I rows of this stage: 32 x NS values (zero beyond the pass's signals)
                const int sq = e / MT_ROWS, i = e - sq * MT_ROWS;
                ri[i * NS + sq] = sq < nsb ? rhs_all[(int64_t)(s0 + sq) * np + (int64_t)I * TS + MT_ROWS * q + i] : 0.0;
            }
            if (q == 0)
                for (int e = tid; e < TS * NS; e += 256) {
                    const int sq = e / TS, i = e - sq * TS;
                    rj[i * NS + sq] = sq < nsb ? rhs_all[(int64_t)(s0 + sq) * np + (int64_t)J * TS + i] : 0.0;
                }
            __syncthreads();                         // DMA landed (vmcnt(0)), staging visible
            if (wave < 2) {
                // P1: rows 16*wave .. of this stage, all 128 columns.  A[i = li][k = lk], B[k = lk][j = s = li]
                f64x4 a0 = (f64x4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
                const double *arow = stg + (16 * wave + li) * MT_RS;
#pragma unroll 8
                for (int kk = 0; kk < 32; kk += 2) {
                    const int c = 4 * kk + lk;
                    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[c], li < NS ? rj[c * NS + li] : 0.0, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[c + 4], li < NS ? rj[(c + 4) * NS + li] : 0.0, a1, 0, 0, 0);
                }
                // D: col = lane&15 = s, row = lk + 4*reg
                if (li < nsb && !(status != nullptr && status[s0 + li].converged)) {
                    double *p1 = part1_all + (int64_t)(s0 + li) * ntiles * TS + (int64_t)t * TS + MT_ROWS * q + 16 * wave + lk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) p1[4 * r] = a0[r] + a1[r];
                }
            } else {
                // P2: columns 64*(wave-2) + 16*u .., the 32 rows of this stage.  A[s = li][k = lk], B[k = lk][j = c = li]
#pragma unroll
                for (int kk = 0; kk < MT_ROWS / 4; ++kk) {
                    const int i = 4 * kk + lk;
                    const double a = li < NS ? ri[i * NS + li] : 0.0;
                    const double *brow = stg + i * MT_RS + 64 * (wave - 2) + li;
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc2[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, brow[16 * u], acc2[u], 0, 0, 0);
                }
            }
        }
        if (I != J && wave >= 2) {   // P2: D row = s = lk + 4*reg, col = li
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < NS / 4; ++r) {
                    const int sgl = lk + 4 * r;
                    if (sgl < nsb && !(status != nullptr && status[s0 + sgl].converged))
                        part2_all[(int64_t)(s0 + sgl) * ntiles * TS + (int64_t)t * TS + 64 * (wave - 2) + 16 * u + li] = acc2[u][r];
                }
        }
    }
}

// ---- the multi-signal tile product, wave-specialised and streamed (6-byte or 8-byte packed copy) --------------------------
// What bounds symv_tile_mfma_kernel (measured at n = 32768, ns = 8: 1.02 ms per launch, 4.2 TB/s of 8-byte tiles):
//   * the 16.8 M v_mfma_f64_16x16x4_f64 of a launch need 0.52 ms of the matrix pipe (its 66 TFLOP/s issue ceiling);
//   * on gfx950 the fp64 MFMA and the vector ALU exclude each other: every VALU instruction issued by ANY wave of the SIMD
//     costs the matrix pipe ~7 cycles (measured by adding dummy v_add_u32 to a co-resident wave: +7.1 cycles of MFMA time
//     each; scalar instructions are free).  Address arithmetic, selects for the padded signal lanes, LDS-DMA bookkeeping and
//     the decode of 6-byte elements all bill the matrix pipe;
//   * waves that alternate "stage a tile slice" / "multiply" between barriers leave either pipe idle half of the time, and
//     co-resident workgroups fall into step instead of filling each other's gaps.
// Here ONE 512-thread workgroup per CU is persistent (tiles t = blockIdx.x, += gridDim.x) and split into roles:
//   waves 4-7, LOADERS: global -> registers (a ring of D 32-row stages in flight per CU: ~100 KB) -> decode -> the LDS
//     image of the NEXT stage (double-buffered); all addresses are a scalar base plus a loop-invariant lane offset, so the
//     only vector instructions left are the three per element of the 6-byte decode;
//   waves 0-3, MFMA (one per SIMD): waves 0-1 form P1 of the stage's two 16-row blocks, waves 2-3 accumulate P2 over the
//     tile's four stages; operands come from LDS with immediate offsets (no selects: the right-hand-side images always hold
//     16 signal columns; columns beyond ns hold signal ns-1 again and their results are not stored).
// One barrier per stage.  Signals are processed 16 per pass; the last pass of ns > 16 re-covers the last 16 signals.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (a native vector: stays in registers where HIP's uint4 struct did not)
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int WS_NS = 16;                            // signal columns of the LDS images = the MFMA's N
constexpr size_t symv_ws_lds() { return sizeof(double) * 2 * ((size_t)MT_ROWS * MT_RS + (size_t)MT_ROWS * WS_NS + (size_t)TS * WS_NS); }

struct StreamVisit { int t, pass, I, J, k, end, u; };  // one (tile, signal pass): four 32-row stages; RUNS: of segment k = [.., end); PANEL: of unit u = row I of panel k, columns .. end

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_f64(double v, __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    const u32x2 w = {(unsigned int)__double2loint(v), (unsigned int)__double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, voffset, soffset, 0);
}

// Q4 (ns <= 8): the products run on v_mfma_f64_4x4x4_4b_f64 instead -- four independent 4x4x4 blocks per instruction (lane layout
// probed by tools/mfma_4x4x4_layout.hip: A[blk][i][k] in lane 16k + 4blk + i, B[blk][k][j] in lane 16k + 4blk + j, D[blk][i][j] in lane
// 16i + 4blk + j).  With the four blocks on four row quads (P1) or four column quads (P2) the TILE operand has exactly the lane layout
// of the 16x16x4 form, and the small operand is a 4 x 4 block of right-hand sides repeated in every block: signals 0-3 and 4-7 are two
// instructions on the same tile registers -- 8 signal columns cost 2 x 16 cycles where the 16-column instruction costs 64 with half of
// it padding.  The right-hand-side images hold 8 signals per row, ordered (s & 3) * 2 + (s >> 2): a lane's two quads are one 16-byte read.
// RUNS: the row-major triangle is cut into nseg SEGMENTS of consecutive tiles, segment k = [k ntiles / nseg, (k+1) ntiles / nseg)
// (about 8 tiles), workgroup g walks the segments g, g + G, ... -- at any moment the workgroups still stream one contiguous stretch of
// M between them, as with single tiles -- and the P1 waves keep the sums of a row block in registers over a RUN of tiles of the same
// row I within a segment: one record (id I + k: at most nseg + nblk of them) per run instead of one per tile -- most of one half of the
// partials (270 MB of 539 MB per launch at cfg5) is never written and never read back.  The consumers find row block I's records as
// [I + k(I,0), I + k(I,I)] (part1_range below).
// FIX (with SPLIT): the mixed storage -- tiles whose format byte is non-zero are 36-bit fixed point (74 240 of the slot's 98 304 bytes: the
// same 64 KB of leading dwords, then 8 KB of nibbles and a float step per row), the others (all diagonal tiles among them) float-head.
// A visit's format byte is requested (scalar load) a whole visit before its first stage is fetched.  Both formats' loads are issued for
// every stage, through buffer descriptors of size zero for the format the tile does not have (those loads are dropped: no load under a
// branch); a fixed-point element decodes in four vector instructions (bit-field extract, two integer instructions that assemble
// 2^52 + q, one FMA with the row's step).
// PANEL (with RUNS and Q4; round 5): the triangle is walked in COLUMN PANELS of kPanelC tile columns, a panel from its diagonal down, row by
// row -- unit (k, I) = the tiles (I, kC .. min(kC + C - 1, I)) of row I in panel k, unit index u = k nblk - C k (k - 1) / 2 + I - k C.  A
// workgroup walks a contiguous range of units [u0, u1) (the host cuts the unit list into gridDim.x ranges of equal tile counts: panel_plan).
//   P1 (row sums): one record per unit, id u -- the run logic of RUNS with the unit as the run;
//   P2 (column sums): the C column blocks of the panel keep their sums in registers of the two P2 waves across
//   ALL rows a workgroup walks in the panel, and are written out once per (workgroup, panel) as records f C + c, f = the running
//   flush index (ptab gives the workgroup's first; the flushes of panel k are the contiguous range [F0[k], F0[k+1]), which is
//   how the consumers find them).  At n = 32768: 8 224 P1 records + ~320 x 4 P2 records per signal instead of 4 370 + 32 640 -- the
//   270 MB of per-tile column sums of a launch (and their re-read by the reduction) become 10 MB.
constexpr size_t kPanelLds = 0;                                       // (the column sums live in registers)
#if defined(LPVS_TIMELINE) && LPVS_TIMELINE == 3
// Debug build only (make timeline3 -> liblpvspectral_timeline3.so; tools/ws_timeline.py): where a launch of the wave-specialised kernel spends its
// time, BY ROLE.  One wave of each role (wave 0: P1, wave 2: P2, wave 4: loader) keeps the 100-MHz wall clock (s_memrealtime) at entry and end and
// the SUM of the time it spent waiting at the stage barriers (stamp before / after every barrier); the loader also the sum of the time inside `put`
// (waiting for the stage's bytes + decoding them into the LDS image).  The role that waits least at the barriers is the one the others wait for.
// 32 words per workgroup: role r at [8 r ..]: {entry, -, end, barrier-wait ticks, barriers, put ticks (loader), of which waiting for the stage's bytes}; [24] XCC_ID, [25] HW_ID.
__device__ unsigned long long *g_lpvs_tl_ws = nullptr;
extern "C" int32_t lpvs_debug_set_timeline_ws(unsigned long long *dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lpvs_tl_ws), &dev_buf, sizeof(dev_buf)) == hipSuccess ? LPVS_OK : LPVS_EDEVICE;
}
// (32-bit tick arithmetic in VECTOR registers: the kernel has no scalar register to spare -- 102 used, any more spill --, and six vector
// instructions per stage cost the matrix pipe ~40 of a stage's ~2800 cycles)
__device__ __forceinline__ unsigned int ws_tl_now() {
    unsigned int v;
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"((unsigned int)t));
    return v;
}
#define WS_TL_DECL unsigned int tl_pre = ws_tl_now(), tl_post = tl_pre, tl_wait, tl_put, tl_t = tl_pre, tl_nbar, tl_arr; \
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(tl_wait), "=v"(tl_put), "=v"(tl_nbar), "=v"(tl_arr)); (void)tl_t; (void)tl_put; (void)tl_arr; \
    if (g_lpvs_tl_ws != nullptr && lane == 0 && (wave == 0 || wave == 2 || wave == 4)) g_lpvs_tl_ws[(size_t)blockIdx.x * 32 + 8 * (wave >> 1)] = __builtin_amdgcn_s_memrealtime();
#define WS_BARRIER() do { tl_wait += tl_post - tl_pre; tl_pre = ws_tl_now(); __syncthreads(); tl_post = ws_tl_now(); ++tl_nbar; } while (0)
#define WS_TL_PUT_BEGIN() do { tl_t = ws_tl_now(); } while (0)
#define WS_TL_PUT_END() do { tl_put += ws_tl_now() - tl_t; } while (0)
// the stage's bytes have arrived: `last` is a register of the stage's youngest load (loads return in order), touched by an instruction the compiler
// has to put its counted wait in front of; the time since WS_TL_PUT_BEGIN is the wait for the bytes alone, the rest of `put` is decode + LDS stores
#define WS_TL_ARRIVED(last) do { unsigned int t_; asm volatile("v_mov_b32 %0, %1" : "=v"(t_) : "v"(__double2loint(last))); tl_arr += ws_tl_now() - tl_t; } while (0)
#define WS_TL_FINISH(role) do { if (g_lpvs_tl_ws != nullptr && lane == 0) { unsigned long long *r_ = g_lpvs_tl_ws + (size_t)blockIdx.x * 32 + 8 * (role); \
        r_[2] = __builtin_amdgcn_s_memrealtime(); r_[3] = tl_wait + (tl_post - tl_pre); r_[4] = tl_nbar; r_[5] = tl_put; r_[6] = tl_arr; \
        if ((role) == 0) { r_[24] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20); r_[25] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4); } } } while (0)
#else
#define WS_TL_DECL
#define WS_BARRIER() __syncthreads()
#define WS_TL_PUT_BEGIN() do { } while (0)
#define WS_TL_PUT_END() do { } while (0)
#define WS_TL_ARRIVED(last) do { } while (0)
#define WS_TL_FINISH(role) do { } while (0)
#endif
template <bool SPLIT, bool Q4, bool RUNS, bool FIX, bool PANEL = false>
__global__ void __launch_bounds__(512, 1)
symv_tile_mfma_ws_kernel(const unsigned char *__restrict__ Mp, const double *__restrict__ rhs_all, int64_t np, int ns, int ntiles,
                         double *__restrict__ part1_all, double *__restrict__ part2_all, const AdmmStatus *status, int nseg,
                         const unsigned char *__restrict__ types /* FIX: per-tile formats */, const int *__restrict__ ptab = nullptr /* PANEL: panel_plan's table */) {
    static_assert(!PANEL || (RUNS && Q4), "the panel walk keeps P1 runs and needs the 8-signal LDS images (room for the column sums)");
    if (status != nullptr) {
        bool all = true;
        for (int q = 0; q < ns; ++q) all = all && status[q].converged;
        if (all) return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NS = Q4 ? 8 : WS_NS, NQ = TS / MT_ROWS;
    static_assert(NQ == 4, "a tile is four stages: ring slot = stage, LDS parity = stage & 1");
    constexpr unsigned kStgB = MT_ROWS * MT_RS * 8, kRiB = MT_ROWS * NS * 8, kRjB = TS * NS * 8;   // bytes of one image
    unsigned char *stg = lds_raw;                    // [2][32][MT_RS]   32 rows of a tile, by stage parity
    unsigned char *ri = stg + 2 * kStgB;             // [2][32][16]      right-hand sides of row block I, rows of the stage
    unsigned char *rj = ri + 2 * kRiB;               // [2][128][16]     right-hand sides of row block J, by visit parity
    constexpr size_t kTileBytes = SPLIT ? kSplitTileBytes : (size_t)TS * TS * 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ltid = tid & 255;                      // index within the role's four waves
    const int li = lane & 15, lk = lane >> 4;
    const int G = gridDim.x;
    // Step q of a tile is its stage (q + q0) & 3 with q0 = blockIdx.x & 3, so that workgroups advancing in near lockstep do
    // not all ask for the same quarter of their (power-of-two sized) tiles at the same time.
    const int q0 = blockIdx.x & (NQ - 1);
    const int npass = (ns + NS - 1) / NS, nvalid = ns < NS ? ns : NS, s0max = ns > NS ? ns - NS : 0;
    constexpr int tend_unused = 0; (void)tend_unused;
    const int tend = ntiles;
    auto seg_begin = [&](int k) -> int { return (int)(((unsigned long long)(unsigned)k * (unsigned)ntiles) / (unsigned)nseg); };
    if ((int)blockIdx.x >= (RUNS ? nseg : ntiles)) return;
    const int nblk = (int)(np / TS);
    int pu1 = 0, pf0 = 0;                            // PANEL: end of this workgroup's unit range, its first flush index
    auto next = [&](StreamVisit v) -> StreamVisit {  // scalar only; t >= ntiles after the workgroup's last visit
        if (++v.pass == npass) {
            v.pass = 0;
            if constexpr (PANEL) {
                if (++v.J <= v.end) ++v.t;
                else if (++v.u < pu1) {              // the next unit: the next row of the panel, or the first row of the next panel
                    if (++v.I == nblk) { ++v.k; v.I = v.k * kPanelC; }
                    v.J = v.k * kPanelC;
                    v.end = v.J + kPanelC - 1 < v.I ? v.J + kPanelC - 1 : v.I;
                    v.t = v.I * (v.I + 1) / 2 + v.J;
                } else v.t = ntiles;
            } else if constexpr (RUNS) {
                if (++v.t < v.end) {
                    if (++v.J > v.I) { v.J = 0; ++v.I; }
                } else if ((v.k += G) < nseg) {      // the workgroup's next segment
                    v.t = __builtin_amdgcn_readfirstlane(seg_begin(v.k)); v.end = __builtin_amdgcn_readfirstlane(seg_begin(v.k + 1));
                    tile_index(v.t, v.I, v.J);
                    v.I = __builtin_amdgcn_readfirstlane(v.I); v.J = __builtin_amdgcn_readfirstlane(v.J);
                } else v.t = ntiles;
            } else {
                v.t += G;
                v.J += G;                            // t = I(I+1)/2 + J, 0 <= J <= I
                while (v.J > v.I) { v.J -= v.I + 1; ++v.I; }
            }
        }
        return v;
    };
    auto s0_of = [&](const StreamVisit &v) -> int { const int s0 = v.pass * NS; return s0 < s0max ? s0 : s0max; };
    StreamVisit cv{RUNS ? seg_begin(blockIdx.x) : (int)blockIdx.x, 0, 0, 0, (int)blockIdx.x, RUNS ? seg_begin(blockIdx.x + 1) : 0, 0};   // the visit being multiplied
    if constexpr (PANEL) {
        const int *row = ptab + 5 * blockIdx.x;      // {u0, u1, k0, I0, f0}: scalar loads
        cv.u = __builtin_amdgcn_readfirstlane(row[0]); pu1 = __builtin_amdgcn_readfirstlane(row[1]);
        cv.k = __builtin_amdgcn_readfirstlane(row[2]); cv.I = __builtin_amdgcn_readfirstlane(row[3]); pf0 = __builtin_amdgcn_readfirstlane(row[4]);
        if (cv.u >= pu1) return;                     // (more workgroups than units: uniform)
        cv.J = cv.k * kPanelC;
        cv.end = cv.J + kPanelC - 1 < cv.I ? cv.J + kPanelC - 1 : cv.I;
        cv.t = cv.I * (cv.I + 1) / 2 + cv.J;
    } else {
        cv.t = __builtin_amdgcn_readfirstlane(cv.t); cv.end = __builtin_amdgcn_readfirstlane(cv.end);
        tile_index(cv.t, cv.I, cv.J);
        cv.I = __builtin_amdgcn_readfirstlane(cv.I); cv.J = __builtin_amdgcn_readfirstlane(cv.J);
    }
    int tp = 0;                                      // parity of the visit count
    WS_TL_DECL

    if (wave >= 4) {
        // ---- loader waves.  Thread (r = ltid >> 4, c = ltid & 15) owns rows r and r + 16 of a stage:
        //   split:  two 16-byte pieces of heads (columns 4c.., 64+4c..) and one of tails per row (pack_tiles_split_kernel's layout)
        //   double: four 16-byte pieces per row (columns 32j + 2c, 2c+1)
        const int r = ltid >> 4, c = ltid & 15;
        static_assert(!FIX || SPLIT, "fixed-point tiles live in the 6-byte slots");
        constexpr int NRAW = SPLIT ? (FIX ? 7 : 6) : 8, RI = MT_ROWS * NS / 256, RJ = TS * NS / 256;
        // fixed-point tiles: the dwords of this thread's two rows' nibbles and steps (pack_tiles_mixed_kernel's layout: row 32 w + 4 rg + g,
        // lane (g, c); this thread's rows are w = stage, rg = r >> 2 and (r >> 2) + 4, g = r & 3)
        const int lo_nib = (int)kFixHeadBytes + ((((r & 3) * 16 + c) * 8) + (r >> 2)) * 4;
        const int lo_stp = (int)(kFixHeadBytes + kFixNibBytes) + (((r & 3) * 8) + (r >> 2)) * 4;
        u32x4 raw[NQ][NRAW];                          // ring slot = step of the tile
        double pri[NQ][RI], prj[RJ];                 // (the J slice travels with step 0)
        // loop-invariant lane offsets (bytes)
        const int lo_a = SPLIT ? r * (TS * 4) + c * 16 : r * (TS * 8) + c * 16;   // heads (split) / doubles
        const int lo_t = r * (TS * 2) + c * 16;                                    // tails (split)
        int go_ri[RI], go_rj[RJ];
        unsigned wo_ri[RI], wo_rj[RJ];
#pragma unroll
        for (int k = 0; k < RI; ++k) {
            const int e = ltid + 256 * k, sq = e >> 5, i = e & 31;
            go_ri[k] = (int)(((int64_t)(sq < ns ? sq : ns - 1) * np + i) * 8);
            wo_ri[k] = (i * NS + (Q4 ? (sq & 3) * 2 + (sq >> 2) : sq)) * 8;
        }
#pragma unroll
        for (int k = 0; k < RJ; ++k) {
            const int e = ltid + 256 * k, sq = e >> 7, i = e & 127;
            go_rj[k] = (int)(((int64_t)(sq < ns ? sq : ns - 1) * np + i) * 8);
            wo_rj[k] = (i * NS + (Q4 ? (sq & 3) * 2 + (sq >> 2) : sq)) * 8;
        }
        const unsigned wo_stg = SPLIT ? (r * MT_RS + 4 * c) * 8 : (r * MT_RS + 2 * c) * 8;
        // Every load of a step is unconditional (past the last visit an earlier one is requested again): a load under a
        // branch, or registers that differ between two paths into the loop, make the compiler wait for the prefetch right
        // where it was issued.  Buffer loads: scalar descriptor + scalar offset + loop-invariant lane offset.
        auto type_of = [&](const StreamVisit &v) -> unsigned {   // (scalar load of the dword that holds the tile's format byte)
            if constexpr (!FIX) return 0u;
            const int t = v.t < tend ? v.t : tend - 1;
            return (reinterpret_cast<const unsigned int *>(types)[t >> 2] >> (8 * (t & 3))) & 255u;
        };
        auto fetch = [&](auto qc, const StreamVisit &v, unsigned ty) {
            constexpr int Q = decltype(qc)::value;
            const int qp = (Q + q0) & (NQ - 1);
            const __amdgpu_buffer_rsrc_t tile = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(Mp + (size_t)v.t * kTileBytes), 0, (int)kTileBytes, 0x00020000);
            if constexpr (SPLIT) {
                const int oh = qp * (MT_ROWS * TS * 4), ot = TS * TS * 4 + qp * (MT_ROWS * TS * 2);
                // (FIX: the tails exist in a diagonal tile only, nibbles and steps below the diagonal only -- descriptors of size zero drop the rest)
                const bool fx = FIX && ty != 0;
                const __amdgpu_buffer_rsrc_t tails = FIX ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(Mp + (size_t)v.t * kTileBytes), 0, fx ? 0 : (int)kTileBytes, 0x00020000) : tile;
                raw[Q][0] = __builtin_amdgcn_raw_buffer_load_b128(tile, lo_a, oh, 0);
                raw[Q][1] = __builtin_amdgcn_raw_buffer_load_b128(tile, lo_a + 256, oh, 0);
                raw[Q][2] = __builtin_amdgcn_raw_buffer_load_b128(tails, lo_t, ot, 0);
                raw[Q][3] = __builtin_amdgcn_raw_buffer_load_b128(tile, lo_a, oh + 16 * TS * 4, 0);
                raw[Q][4] = __builtin_amdgcn_raw_buffer_load_b128(tile, lo_a + 256, oh + 16 * TS * 4, 0);
                raw[Q][5] = __builtin_amdgcn_raw_buffer_load_b128(tails, lo_t, ot + 16 * TS * 2, 0);
                if constexpr (FIX) {
                    const __amdgpu_buffer_rsrc_t aux = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(Mp + (size_t)v.t * kTileBytes), 0, fx ? (int)kMixedFixedTileBytes : 0, 0x00020000);
                    raw[Q][6][0] = __builtin_amdgcn_raw_buffer_load_b32(aux, lo_nib, qp * (64 * 8 * 4), 0);
                    raw[Q][6][1] = __builtin_amdgcn_raw_buffer_load_b32(aux, lo_nib + 16, qp * (64 * 8 * 4), 0);
                    raw[Q][6][2] = __builtin_amdgcn_raw_buffer_load_b32(aux, lo_stp, qp * (4 * 8 * 4), 0);
                    raw[Q][6][3] = __builtin_amdgcn_raw_buffer_load_b32(aux, lo_stp + 16, qp * (4 * 8 * 4), 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        raw[Q][4 * k + j] = __builtin_amdgcn_raw_buffer_load_b128(tile, lo_a + j * 256, qp * (MT_ROWS * TS * 8) + k * (16 * TS * 8), 0);
            }
            const int s0 = s0_of(v);
            const int64_t left = (int64_t)(ns - s0) * np * 8;
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(rhs_all + (int64_t)s0 * np), 0,
                                                                              (int)(left < 0x7fffffff ? left : 0x7fffffff), 0x00020000);
            const int oI = (v.I * TS + MT_ROWS * qp) * 8;
#pragma unroll
            for (int k = 0; k < RI; ++k) pri[Q][k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rr, go_ri[k], oI, 0));
            if constexpr (Q == 0) {
                const int oJ = v.J * TS * 8;
#pragma unroll
                for (int k = 0; k < RJ; ++k) prj[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rr, go_rj[k], oJ, 0));
            }
        };
        auto put = [&](auto qc, int vtp, bool fx) {  // step's registers -> the LDS images of parity Q & 1 (J slice: visit parity); fx: a fixed-point tile
            constexpr int Q = decltype(qc)::value;
            constexpr unsigned par = Q & 1;
            unsigned char *sp = stg + par * kStgB + wo_stg;
            WS_TL_ARRIVED(pri[Q][RI - 1]);
            if (FIX && fx) {                         // (uniform)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const u32x4 ha = raw[Q][3 * k], hb = raw[Q][3 * k + 1];
                    const unsigned int nw = raw[Q][FIX ? 6 : 0][k];
                    const double step = (double)__uint_as_float(raw[Q][FIX ? 6 : 0][2 + k]), off = -(0x1p52 + 0x1p35) * step;
                    auto dec = [&](unsigned int hi, int j) -> double {      // (2^52 + q) * step - (2^52 + 2^35) * step, exact
                        const unsigned int top = __builtin_amdgcn_alignbit(0x04330000u, hi, 28);
                        unsigned int lo = (nw >> (4 * j)) & 15u;
                        asm("v_lshl_or_b32 %0, %1, 4, %0" : "+v"(lo) : "v"(hi));
                        return fma(__hiloint2double((int)top, (int)lo), step, off);
                    };
                    f64x2 *row = reinterpret_cast<f64x2 *>(sp + k * (16 * MT_RS * 8));
                    row[0] = (f64x2){dec(ha.x, 0), dec(ha.y, 1)};
                    row[1] = (f64x2){dec(ha.z, 2), dec(ha.w, 3)};
                    row[32] = (f64x2){dec(hb.x, 4), dec(hb.y, 5)};
                    row[33] = (f64x2){dec(hb.z, 6), dec(hb.w, 7)};
                }
            } else if constexpr (SPLIT) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const u32x4 ha = raw[Q][3 * k], hb = raw[Q][3 * k + 1], lq = raw[Q][3 * k + 2];
                    f64x2 *row = reinterpret_cast<f64x2 *>(sp + k * (16 * MT_RS * 8));
                    row[0] = (f64x2){split_decode_lo(__uint_as_float(ha.x), lq.x), split_decode_hi(__uint_as_float(ha.y), lq.x)};
                    row[1] = (f64x2){split_decode_lo(__uint_as_float(ha.z), lq.y), split_decode_hi(__uint_as_float(ha.w), lq.y)};
                    row[32] = (f64x2){split_decode_lo(__uint_as_float(hb.x), lq.z), split_decode_hi(__uint_as_float(hb.y), lq.z)};
                    row[33] = (f64x2){split_decode_lo(__uint_as_float(hb.z), lq.w), split_decode_hi(__uint_as_float(hb.w), lq.w)};
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4 *>(sp + k * (16 * MT_RS * 8) + j * 256) = raw[Q][4 * k + j];
            }
#pragma unroll
            for (int k = 0; k < RI; ++k) *reinterpret_cast<double *>(ri + par * kRiB + wo_ri[k]) = pri[Q][k];
            if constexpr (Q == 0) {
                unsigned char *rjp = rj + vtp * kRjB;
#pragma unroll
                for (int k = 0; k < RJ; ++k) *reinterpret_cast<double *>(rjp + wo_rj[k]) = prj[k];
            }
        };
        using std::integral_constant;
        StreamVisit v1 = next(cv);                   // the visit after the current one; `src` = the visit the ring is refilled from
        unsigned tyc = type_of(cv), ty1 = type_of(v1);   // formats of the current and the next visit's tiles (scalar)
        // prologue: the first visit's four steps in flight, step 0 staged, slot 0 refilled from the next visit
        fetch(integral_constant<int, 0>{}, cv, tyc);
        fetch(integral_constant<int, 1>{}, cv, tyc);
        fetch(integral_constant<int, 2>{}, cv, tyc);
        fetch(integral_constant<int, 3>{}, cv, tyc);
        WS_TL_PUT_BEGIN(); put(integral_constant<int, 0>{}, 0, tyc != 0); WS_TL_PUT_END();
        fetch(integral_constant<int, 0>{}, v1.t < tend ? v1 : cv, v1.t < tend ? ty1 : tyc);
        WS_BARRIER();
#pragma unroll 1
        for (;;) {
            // while the MFMA waves multiply step q of the current visit, stage step q + 1 and refill its slot from the next visit
            const StreamVisit v2 = next(v1);         // (its format byte is requested here and used at the end of this visit)
            const unsigned ty2 = type_of(v2);
            const StreamVisit src = v1.t < tend ? v1 : cv;
            const unsigned tys = v1.t < tend ? ty1 : tyc;
            const bool fxc = tyc != 0;
            WS_TL_PUT_BEGIN(); put(integral_constant<int, 1>{}, tp, fxc); WS_TL_PUT_END(); fetch(integral_constant<int, 1>{}, src, tys); WS_BARRIER();
            WS_TL_PUT_BEGIN(); put(integral_constant<int, 2>{}, tp, fxc); WS_TL_PUT_END(); fetch(integral_constant<int, 2>{}, src, tys); WS_BARRIER();
            WS_TL_PUT_BEGIN(); put(integral_constant<int, 3>{}, tp, fxc); WS_TL_PUT_END(); fetch(integral_constant<int, 3>{}, src, tys); WS_BARRIER();
            WS_TL_PUT_BEGIN(); if (v1.t < tend) put(integral_constant<int, 0>{}, tp ^ 1, ty1 != 0); WS_TL_PUT_END();
            fetch(integral_constant<int, 0>{}, v2.t < tend ? v2 : cv, v2.t < tend ? ty2 : tyc);
            WS_BARRIER();
            if (v1.t >= tend) break;
            cv = v1; v1 = v2; tyc = ty1; ty1 = ty2; tp ^= 1;
        }
        if (wave == 4) WS_TL_FINISH(2);
        return;
    }

    // ---- MFMA waves: loop-invariant lane offsets (bytes)
    const bool p1 = wave < 2;
    const unsigned small_lane = Q4 ? (lk * NS + (li & 3) * 2) * 8 : (lk * NS + li) * 8;                     // the right-hand-side operand (Q4: two quads = 16 bytes)
    const unsigned a_lane = p1 ? ((16 * wave + li) * MT_RS + lk) * 8 : small_lane;                         // P1: A = tile rows; P2: A = ri
    const unsigned b_lane = p1 ? small_lane : (lk * MT_RS + 64 * (wave - 2) + li) * 8;                     // P1: B = rj;        P2: B = tile rows
    // partials: buffer stores, a lane whose signal is beyond the pass's valid ones gets an out-of-range offset (store dropped)
    //   16x16x4: part1 [signal li][tile][row lk + 4k], part2 [signal lk + 4k][tile][col 16u + li]
    //   Q4:      part1 [signal (li & 3) + 4h][tile][row 4 (li >> 2) + lk], part2 [signal lk + 4h][tile][col 16u + li]
    const int s_lane = Q4 ? (p1 ? (int)(((int64_t)(li & 3) * ntiles * TS + 16 * wave + 4 * (li >> 2) + lk) * 8) : (int)(((int64_t)lk * ntiles * TS + 64 * (wave - 2) + li) * 8))
                          : (p1 ? (li < nvalid ? (int)(((int64_t)li * ntiles * TS + 16 * wave + lk) * 8) : (int)0x80000000u)
                                : (int)(((int64_t)lk * ntiles * TS + 64 * (wave - 2) + li) * 8));
    const bool hi_valid = Q4 && (p1 ? (li & 3) + 4 : lk + 4) < nvalid, lo_valid = !Q4 || (p1 ? (li & 3) : lk) < nvalid;
    const int64_t pass_bytes = (int64_t)nvalid * ntiles * TS * 8;
    const int part_records = (int)(pass_bytes < 0x7fffffff ? pass_bytes : 0x7fffffff);
    f64x4 acc2[4];                                   // waves 2-3: P2 blocks, columns 64*(wave-2) + 16*u .., over the tile's stages
    WS_BARRIER();                                    // step 0 of the first visit staged
    // One continuous software pipeline over the steps: a step's 32 MFMAs run as four groups of eight, each group's operands
    // read from LDS while the previous group multiplies.  The step's barrier sits BEFORE its last group (whose operands are in
    // registers by then: nobody reads the step's LDS images after it), and the next step's first operands are requested
    // right after it, so the matrix pipe does not drain at step boundaries.
    if (p1) {
        // P1: rows 16*wave .. of a stage, all 128 columns.  A[i = li][k = lk], B[k = lk][j = s = li]
        // Q4: the right-hand-side operand (B: the J slice, 32 pivot steps x two signal quads) does not change over the four steps of a
        // visit -- it is read from LDS with step 0's groups only and kept in registers (128 of them): the LDS operand reads of the four
        // MFMA waves cost as much time as their matrix instructions, and these were a third of them
        using BT = std::conditional_t<Q4, f64x2, double>;
        double A[2][8]; BT B[Q4 ? 1 : 2][8]; BT Bv[Q4 ? 32 : 1];
        auto load = [&](auto gc, unsigned par, int vtp, int buf, bool with_b) {      // operands of group G of the stage with parity par
            constexpr int g = decltype(gc)::value;
            const unsigned char *ap = stg + par * kStgB + a_lane, *bp = rj + vtp * kRjB + b_lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                A[buf][j] = *reinterpret_cast<const double *>(ap + 32 * (8 * g + j));
                if constexpr (Q4) { if (with_b) Bv[Q4 ? 8 * g + j : 0] = *reinterpret_cast<const BT *>(bp + 4 * NS * 8 * (8 * g + j)); }
                else B[Q4 ? 0 : buf][j] = *reinterpret_cast<const BT *>(bp + 4 * NS * 8 * (8 * g + j));
            }
        };
        using std::integral_constant;
        load(integral_constant<int, 0>{}, 0, 0, 0, true);
        f64x4 run0[RUNS ? NQ : 1], run1[RUNS && !Q4 ? NQ : 1];   // RUNS: the sums of a run of tiles of one row block, step by step
        bool fresh = true;                           // (scalar) the visit starts a run
        auto store1 = [&](const f64x4 &a0, const f64x4 &a1, int so) {   // a stage's 16 rows of this wave; a converged signal's partials are never read
            const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(part1_all + (int64_t)s0_of(cv) * ntiles * TS, 0, part_records, 0x00020000);
            if constexpr (Q4) {                      // D: row = 4 (li >> 2) + lk, signal = (li & 3) + 4h
                store_f64(a0[0] + a0[2], pr, lo_valid ? s_lane : (int)0x80000000u, so);
                store_f64(a0[1] + a0[3], pr, hi_valid ? s_lane : (int)0x80000000u, so + 4 * ntiles * TS * 8);
            } else {                                 // D: col = li = s, row = lk + 4*reg
#pragma unroll
                for (int k = 0; k < 4; ++k) store_f64(a0[k] + a1[k], pr, s_lane + 32 * k, so);
            }
        };
        auto step = [&](auto qc, int next_tp) {
            constexpr int Q = decltype(qc)::value;
            constexpr unsigned par = Q & 1;
            f64x4 t0 = (f64x4){0.0, 0.0, 0.0, 0.0}, t1 = t0;
            if constexpr (RUNS) { if (fresh) { run0[RUNS ? Q : 0] = t0; if constexpr (!Q4) run1[RUNS && !Q4 ? Q : 0] = t0; } }
            f64x4 &a0 = RUNS ? run0[RUNS ? Q : 0] : t0, &a1 = (RUNS && !Q4) ? run1[RUNS && !Q4 ? Q : 0] : t1;   // (Q4: elements 0/1 = signals lo/hi of the even k steps, 2/3 of the odd ones)
            auto mul = [&](int buf, auto gc) {
                constexpr int g = decltype(gc)::value;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    if constexpr (Q4) {
                        a0[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j], Bv[Q4 ? 8 * g + j : 0][0], a0[0], 0, 0, 0);
                        a0[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j], Bv[Q4 ? 8 * g + j : 0][1], a0[1], 0, 0, 0);
                        a0[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j + 1], Bv[Q4 ? 8 * g + j + 1 : 0][0], a0[2], 0, 0, 0);
                        a0[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j + 1], Bv[Q4 ? 8 * g + j + 1 : 0][1], a0[3], 0, 0, 0);
                    } else {
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[buf][j], B[Q4 ? 0 : buf][j], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[buf][j + 1], B[Q4 ? 0 : buf][j + 1], a1, 0, 0, 0);
                    }
                }
            };
            constexpr bool first = Q == 0, last = Q == NQ - 1;   // (B is read with the groups of a visit's step 0; group 0 of step 0 rides on the previous step 3)
            load(integral_constant<int, 1>{}, par, tp, 1, first); __builtin_amdgcn_sched_barrier(0); mul(0, integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0);
            load(integral_constant<int, 2>{}, par, tp, 0, first); __builtin_amdgcn_sched_barrier(0); mul(1, integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0);
            load(integral_constant<int, 3>{}, par, tp, 1, first); __builtin_amdgcn_sched_barrier(0); mul(0, integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0);
            WS_BARRIER();                            // the next step is staged; this step's images are free
            load(integral_constant<int, 0>{}, par ^ 1, next_tp, 0, last); __builtin_amdgcn_sched_barrier(0); mul(1, integral_constant<int, 3>{}); __builtin_amdgcn_sched_barrier(0);
            if constexpr (!RUNS) store1(a0, a1, (cv.t * TS + MT_ROWS * ((Q + q0) & (NQ - 1))) * 8);
        };
#pragma unroll 1
        for (;;) {
            step(integral_constant<int, 0>{}, tp);
            step(integral_constant<int, 1>{}, tp);
            step(integral_constant<int, 2>{}, tp);
            step(integral_constant<int, 3>{}, tp ^ 1);
            const StreamVisit nv = next(cv);
            if constexpr (RUNS) {
                fresh = PANEL ? (nv.t >= tend || nv.u != cv.u) : (nv.t >= tend || nv.I != cv.I || nv.k != cv.k);   // the run ends with this tile: record I + k (PANEL: the unit's, u)
                if (fresh) {
                    const int rec = PANEL ? cv.u : cv.I + cv.k;
#pragma unroll
                    for (int Q = 0; Q < NQ; ++Q) store1(run0[Q], run1[Q4 ? 0 : Q], (rec * TS + MT_ROWS * ((Q + q0) & (NQ - 1))) * 8);
                }
            }
            cv = nv;
            if (cv.t >= tend) break;
            tp ^= 1;
        }
        if (wave == 0) WS_TL_FINISH(0);
    } else {
        // P2: columns 64*(wave-2) + 16*u .., the 32 rows of a stage.  A[s = li][k = lk], B[k = lk][j = c = li]
        using AT = std::conditional_t<Q4, f64x2, double>;
        AT A[2][2]; double B[2][8];
        f64x2 acc4[4];                               // Q4: P2 blocks of signals lo/hi, columns 64*(wave-2) + 16*u ..
        auto load = [&](auto gc, unsigned par, int buf) {
            constexpr int g = decltype(gc)::value;
            const unsigned char *ap = ri + par * kRiB + a_lane, *bp = stg + par * kStgB + b_lane;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                A[buf][j] = *reinterpret_cast<const AT *>(ap + 4 * NS * 8 * (2 * g + j));
#pragma unroll
                for (int u = 0; u < 4; ++u) B[buf][4 * j + u] = *reinterpret_cast<const double *>(bp + 4 * MT_RS * 8 * (2 * g + j) + 128 * u);
            }
        };
        using std::integral_constant;
        load(integral_constant<int, 0>{}, 0, 0);
        auto step = [&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            constexpr unsigned par = Q & 1;
            auto mul = [&](int buf, bool first) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if constexpr (Q4) {
                            const f64x2 cin = (first && j == 0) ? (f64x2){0.0, 0.0} : acc4[u];
                            acc4[u][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j][0], B[buf][4 * j + u], cin[0], 0, 0, 0);
                            acc4[u][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[buf][j][1], B[buf][4 * j + u], cin[1], 0, 0, 0);
                        } else {
                            const f64x4 cin = (first && j == 0) ? (f64x4){0.0, 0.0, 0.0, 0.0} : acc2[u];   // a tile's first product starts the sums
                            acc2[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[buf][j], B[buf][4 * j + u], cin, 0, 0, 0);
                        }
                    }
            };
            load(integral_constant<int, 1>{}, par, 1); __builtin_amdgcn_sched_barrier(0); mul(0, Q == 0); __builtin_amdgcn_sched_barrier(0);
            load(integral_constant<int, 2>{}, par, 0); __builtin_amdgcn_sched_barrier(0); mul(1, false); __builtin_amdgcn_sched_barrier(0);
            load(integral_constant<int, 3>{}, par, 1); __builtin_amdgcn_sched_barrier(0); mul(0, false); __builtin_amdgcn_sched_barrier(0);
            WS_BARRIER();                            // the next step is staged; this step's images are free
            load(integral_constant<int, 0>{}, par ^ 1, 0); __builtin_amdgcn_sched_barrier(0); mul(1, false); __builtin_amdgcn_sched_barrier(0);
            if constexpr (Q == NQ - 1 && !PANEL) {   // P2 of the tile is complete: D row = s = lk + 4*reg, col = li
                if (cv.I != cv.J) {
                    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(part2_all + (int64_t)s0_of(cv) * ntiles * TS, 0, part_records, 0x00020000);
#pragma unroll
                    for (int k = 0; k < NS / 4; ++k) {
                        const int so = (int)(((int64_t)4 * k * ntiles + cv.t) * TS * 8);   // (beyond the pass's valid signals: out of range, dropped)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if constexpr (Q4) store_f64(acc4[u][k], pr, (k ? hi_valid : lo_valid) ? s_lane + 128 * u : (int)0x80000000u, so);
                            else store_f64(acc2[u][k], pr, s_lane + 128 * u, so);
                        }
                    }
                }
            }
        };
        // PANEL: the panel's column sums stay in REGISTERS (one set of four f64x2 per column block: the workgroup's 512 threads own the
        // CU, the P2 waves have a hundred registers to spare); the slot c = J - k C is wave-uniform, so picking the set is a scalar branch.
        // (A first version kept them in LDS: the read-modify-write after every tile sat on the P2 waves' way to the next barrier and cost
        // the launch 38 us of its 611.)
        f64x2 col[kPanelC][4];
        int nflush = 0;
        if constexpr (PANEL) {
#pragma unroll
            for (int c = 0; c < kPanelC; ++c)
#pragma unroll
                for (int u = 0; u < 4; ++u) col[c][u] = (f64x2){0.0, 0.0};
        }
#pragma unroll 1
        for (;;) {
            step(integral_constant<int, 0>{});
            step(integral_constant<int, 1>{});
            step(integral_constant<int, 2>{});
            step(integral_constant<int, 3>{});
            const StreamVisit nv = next(cv);
            if constexpr (PANEL) {
                if (cv.I != cv.J) {                  // (a diagonal tile's transposed product is its own P1)
                    const int c = cv.J - cv.k * kPanelC;
#pragma unroll
                    for (int cc = 0; cc < kPanelC; ++cc)
                        if (c == cc) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) col[cc][u] += acc4[u];
                        }
                }
                if (nv.t >= tend || nv.k != cv.k) {  // the workgroup leaves the panel: one record per column block, id (f0 + nflush) C + c
                    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(part2_all, 0, part_records, 0x00020000);
#pragma unroll
                    for (int c = 0; c < kPanelC; ++c) {
                        const int rec = (pf0 + nflush) * kPanelC + c;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
#pragma unroll
                            for (int k = 0; k < 2; ++k)
                                store_f64(col[c][u][k], pr, (k ? hi_valid : lo_valid) ? s_lane + 128 * u : (int)0x80000000u, (int)(((int64_t)4 * k * ntiles + rec) * TS * 8));
                            col[c][u] = (f64x2){0.0, 0.0};
                        }
                    }
                    ++nflush;
                }
            }
            cv = nv;
            if (cv.t >= tend) break;
        }
        if (wave == 2) WS_TL_FINISH(1);
    }
}

}  // namespace

// persistent grid of the wave-specialised multi-signal kernel: one workgroup per CU, evened out over the rounds so that
// every workgroup walks the same number of tiles (+-1)
static unsigned stream_cus() {
    static const unsigned slots = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        return (unsigned)cus;
    }();
    return slots;
}
static unsigned stream_grid(unsigned ntiles) {
    const unsigned slots = stream_cus(), rounds = (ntiles + slots - 1) / slots;
    return (ntiles + rounds - 1) / rounds;
}

// The panel walk's plan (symv_tile_mfma_ws_kernel<.., PANEL>): the unit list (row I of panel k; kPanelC tile columns per panel) cut into G
// ranges of equal tile counts.  Device table (ints), cached per (device, nblk, G):  [0] = G;  then G rows {u0, u1, k0, I0, f0};  then
// F0[npanel + 1], the first flush index of every panel (a workgroup flushes the panel's column sums once per panel it walks in).
struct PanelPlan { const int *dev = nullptr; int G = 0, nflush = 0, nunits = 0; };
static std::mutex g_panel_mu;
static std::map<std::tuple<int, int, int>, PanelPlan> g_panel_cache;
// (lpvs_release_cached_memory: the plans' device tables are caches like the pool's blocks)
void release_panel_plans() {
    std::lock_guard<std::mutex> lk(g_panel_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &kv : g_panel_cache) { (void)hipSetDevice(std::get<0>(kv.first)); (void)hipFree(const_cast<int *>(kv.second.dev)); }
    (void)hipSetDevice(cur);
    g_panel_cache.clear();
}
static PanelPlan panel_plan(int nblk, int G) {
    std::mutex &mu = g_panel_mu;
    auto &cache = g_panel_cache;
    int device = 0;
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_tuple(device, nblk, G);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    constexpr int C = kPanelC;
    const int npanel = (nblk + C - 1) / C;
    const long long ntiles = (long long)nblk * (nblk + 1) / 2;
    struct Unit { int k, I, cnt; };
    std::vector<Unit> units;
    for (int k = 0; k < npanel; ++k)
        for (int I = k * C; I < nblk; ++I) units.push_back({k, I, std::min({C, I - k * C + 1, nblk - k * C})});
    std::vector<int> tab(1 + 5 * (size_t)G + (size_t)npanel + 1, 0);
    tab[0] = G;
    std::vector<int> flushes_of_panel((size_t)npanel, 0);
    size_t u = 0;
    long long done = 0;
    int f = 0;
    for (int g = 0; g < G; ++g) {
        const long long want = ntiles * (g + 1) / G;            // cumulative tiles after workgroup g (the last one takes the rest)
        const size_t u0 = u;
        while (u < units.size() && (g == G - 1 || done + units[u].cnt / 2 < want)) { done += units[u].cnt; ++u; }
        int *row = tab.data() + 1 + 5 * (size_t)g;
        row[0] = (int)u0; row[1] = (int)u; row[4] = f;
        if (u > u0) {
            row[2] = units[u0].k; row[3] = units[u0].I;
            for (size_t q = u0; q < u; ++q)
                if (q == u0 || units[q].k != units[q - 1].k) { ++flushes_of_panel[(size_t)units[q].k]; ++f; }
        }
    }
    // flush order = (workgroup, panel), and a workgroup's panels follow the previous workgroup's: panel k's flushes are consecutive
    int *F0 = tab.data() + 1 + 5 * (size_t)G;
    for (int k = 0; k < npanel; ++k) F0[k + 1] = F0[k] + flushes_of_panel[(size_t)k];
    PanelPlan pl;
    int *dev = nullptr;
    if (hipMalloc(&dev, tab.size() * sizeof(int)) != hipSuccess || hipMemcpy(dev, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        return pl;                                              // (not cached: the caller falls back to the run walk)
    }
    pl.dev = dev; pl.G = G; pl.nflush = f; pl.nunits = (int)units.size();
    cache[key] = pl;
    return pl;
}
static bool stream_panel(const AdmmParams &p);

template <bool SPLIT, bool Q4, bool RUNS, bool FIX, bool PANEL = false>
static void launch_mfma_stream(const AdmmParams &p, unsigned ntiles, double *part1, double *part2, const AdmmStatus *status, hipStream_t s) {
    const size_t lds = symv_ws_lds() + (PANEL ? kPanelLds : 0);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&symv_tile_mfma_ws_kernel<SPLIT, Q4, RUNS, FIX, PANEL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);   // per device; cheap
    if constexpr (PANEL) {
        const PanelPlan pl = panel_plan((int)(p.np / TS), (int)stream_cus());
        hipLaunchKernelGGL((symv_tile_mfma_ws_kernel<SPLIT, Q4, RUNS, FIX, PANEL>), dim3((unsigned)pl.G), dim3(512), lds, s,
                           reinterpret_cast<const unsigned char *>(p.Mp), p.rhs, p.np, p.ns, (int)ntiles, part1, part2, status, pl.G, FIX ? p.mp_types : nullptr, pl.dev + 1);
        return;
    }
    const int nseg = RUNS ? stream_runs(p) : 0;
    hipLaunchKernelGGL((symv_tile_mfma_ws_kernel<SPLIT, Q4, RUNS, FIX, PANEL>), dim3(RUNS ? std::min(stream_cus(), (unsigned)nseg) : stream_grid(ntiles)), dim3(512), lds, s,
                       reinterpret_cast<const unsigned char *>(p.Mp), p.rhs, p.np, p.ns, (int)ntiles, part1, part2, status, nseg, FIX ? p.mp_types : nullptr, nullptr);
}

// multi-signal handles: stream (default: persistent, register-staged MFMA kernel), dma (LDS-DMA staged MFMA kernel, 8-byte
// storage only), valu (no matrix cores)
static int multi_matvec_choice() {
    static const int multi = [] {
        const char *e = experiment_env("LPVS_MULTI_MATVEC");
        return !e ? 2 : (std::string(e) == "valu" ? 0 : (std::string(e) == "dma" ? 1 : 2));
    }();
    return multi;
}
// (api.hip: may a handle with several right-hand sides keep its off-diagonal tiles in the fixed-point format?  Only the stream kernel reads them.)
bool multi_signal_fixed_tiles_ok(int64_t np) { return multi_matvec_choice() == 2 && np <= 49152; }
bool uses_stream_kernel(const AdmmParams &p) {
    return p.ns > 1 && !p.mp_f32 && p.Mp != nullptr && (p.mp_split || (multi_matvec_choice() == 2 && p.np <= 49152));   // (31-bit byte offsets into the partials: np <= 49152)
}
// The stream kernel's number of segments when it writes its P1 partials per RUN of tiles (one signal pass; every workgroup the same
// number of segments of about L tiles; room for nseg + nblk records in the per-tile record area), else 0: the consumers of the partials
// (symv_reduce_kernel, admm_fused_update2_kernel) take it as `runs_G`.  LPVS_MULTI_RUNS=L sets the segment length (default 8), 0 keeps
// one record per tile.
int stream_runs(const AdmmParams &p) {
    const unsigned L = [] { const char *e = experiment_env("LPVS_MULTI_RUNS"); const int v = e ? atoi(e) : 8; return (unsigned)(v < 0 ? 0 : (v > 64 ? 64 : v)); }();   // (read per call: tests switch it)
    if (L < 2 || !uses_stream_kernel(p) || p.ns > WS_NS) return 0;
    const int nblk = (int)(p.np / TS);
    const unsigned ntiles = (unsigned)(nblk * (nblk + 1) / 2), cus = stream_cus();
    if (ntiles < 2 * cus * L) return 0;              // (small matrices: a tile per visit)
    const unsigned rounds = (ntiles + cus * L - 1) / (cus * L), nseg = cus * rounds;
    return nseg + (unsigned)nblk <= ntiles ? (int)nseg : 0;
}
// The panel walk (round 5; LPVS_MULTI_WALK=panel) where the run walk applies and the four-block MFMA form does (up to 8 signals).  NOT the
// default: measured at cfg5 on one box, same process (tools/cfg5_ab.py, profiles/r05_cfg5_walks_ab.txt) the product takes 650 us against
// 605 with the run walk -- every unit of four tiles is a jump of I x 96 KB for its workgroup, where the run walk's 256 workgroups stream one
// contiguous window between them -- and the gather of the partials 40 us less (10 MB of column sums instead of 270 MB): 0.731 ms per
// iteration either way.  The consumers get runs_G = -kPanelC and the plan's table.
static bool stream_panel(const AdmmParams &p) {
    const bool off = [] { const char *e = experiment_env("LPVS_MULTI_WALK"); return !(e && std::string(e) == "panel"); }();   // (read per call: tests switch it)
    const bool q4_off = [] { const char *e = experiment_env("LPVS_MULTI_MFMA"); return e && std::string(e) == "16"; }();
    if (off || q4_off || stream_runs(p) == 0 || p.ns > 8 || !p.mp_split) return false;
    // the P1 records (one per unit) and the P2 records (ids (flush index) * kPanelC + c) live in the per-tile record areas, ntiles records per
    // signal each: a plan that would index past them (short triangles: ~45 row blocks with LPVS_MULTI_RUNS=2) takes the run walk instead
    const int nblk = (int)(p.np / TS);
    const long long ntiles = (long long)nblk * (nblk + 1) / 2;
    const PanelPlan pl = panel_plan(nblk, (int)stream_cus());
    return pl.dev != nullptr && (long long)pl.nflush * kPanelC <= ntiles && (long long)pl.nunits <= ntiles;
}
int stream_layout(const AdmmParams &p) { return stream_panel(p) ? -kPanelC : stream_runs(p); }
const int *stream_table(const AdmmParams &p) { return stream_panel(p) ? panel_plan((int)(p.np / TS), (int)stream_cus()).dev : nullptr; }

// the multi-signal branches of launch_sym_matvec (admm.hip): tile partials of all signals -> part1 / part2
void launch_multi_matvec(const AdmmParams &p, unsigned ntiles, double *part1, double *part2, const AdmmStatus *status, hipStream_t s) {
    const int multi = multi_matvec_choice();
    if (uses_stream_kernel(p)) {
        // up to 8 signals: the 4x4x4 four-block MFMA (no padded columns); LPVS_MULTI_MFMA=16 keeps the 16-column instruction
        const bool q4_off = [] { const char *e = experiment_env("LPVS_MULTI_MFMA"); return e && std::string(e) == "16"; }();
        const bool q4 = p.ns <= 8 && !q4_off, runs = stream_runs(p) != 0;
        if (stream_panel(p)) {                       // (6-byte / mixed tiles, up to 8 signals, a triangle large enough for runs)
            if (p.mp_types != nullptr) launch_mfma_stream<true, true, true, true, true>(p, ntiles, part1, part2, status, s);
            else launch_mfma_stream<true, true, true, false, true>(p, ntiles, part1, part2, status, s);
            return;
        }
        auto go = [&](auto split, auto q4c, auto runsc) {
            if constexpr (decltype(split)::value) {
                if (p.mp_types != nullptr) { launch_mfma_stream<true, decltype(q4c)::value, decltype(runsc)::value, true>(p, ntiles, part1, part2, status, s); return; }
            }
            launch_mfma_stream<decltype(split)::value, decltype(q4c)::value, decltype(runsc)::value, false>(p, ntiles, part1, part2, status, s);
        };
        using T = std::true_type; using F = std::false_type;
        if (p.mp_split) { if (q4) { if (runs) go(T{}, T{}, T{}); else go(T{}, T{}, F{}); } else { if (runs) go(T{}, F{}, T{}); else go(T{}, F{}, F{}); } }
        else            { if (q4) { if (runs) go(F{}, T{}, T{}); else go(F{}, T{}, F{}); } else { if (runs) go(F{}, F{}, T{}); else go(F{}, F{}, F{}); } }
    } else if (p.ns > 8) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&symv_tile_mfma_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)symv_mfma_lds<16>());   // per device; cheap
        hipLaunchKernelGGL(symv_tile_mfma_kernel<16>, dim3(ntiles), dim3(256), symv_mfma_lds<16>(), s, p.Mp, p.rhs, p.np, p.ns, (int)ntiles, part1, part2, status);
    } else {
        (void)multi;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&symv_tile_mfma_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)symv_mfma_lds<8>());
        hipLaunchKernelGGL(symv_tile_mfma_kernel<8>, dim3(ntiles), dim3(256), symv_mfma_lds<8>(), s, p.Mp, p.rhs, p.np, p.ns, (int)ntiles, part1, part2, status);
    }
}
bool multi_matvec_on_matrix_cores(const AdmmParams &p) { return uses_stream_kernel(p) || (p.ns > 1 && !p.mp_f32 && multi_matvec_choice() == 1); }

}  // namespace lpvs
