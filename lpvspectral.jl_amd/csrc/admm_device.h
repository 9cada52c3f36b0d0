// admm_device.h -- device-side helpers shared by the ADMM translation units (admm.hip, admm_pack.hip, admm_multi.hip, admm_one_launch.hip,
// admm_small.hip, admm_refine.hip): wave / block reductions, the tile index map, and the decoders / tile products of the three storages of the
// packed inverse (6-byte float-head elements, 36-bit fixed-point tiles, their lane ownership).  Everything lives in an anonymous namespace: every
// translation unit compiles its own copy, kernels never cross a unit.
#pragma once
#include "lpvs_internal.h"

namespace lpvs {
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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

// ---- symmetric mat-vec on the lower-triangle 128x128 tiles of M (np^2*4 bytes instead of np^2*8) -------
// M is re-stored tile-packed: Mp[t][128][128], t = I(I+1)/2 + J, I >= J, so a workgroup streams one
// contiguous 128 KiB tile.  Per tile:  part1[t][i] = sum_j T[i][j] r[J*128+j]      (rows)
//                                      part2[t][j] = sum_i T[i][j] r[I*128+i]      (transpose, I != J)
// Each wave holds 32 rows (16 B per lane per row, all 32 loads in flight); the 32 row sums are reduced
// across the 64 lanes by a halving butterfly (32 shuffles per wave instead of 32*6).
constexpr int TS = 128;

__device__ __forceinline__ void tile_index(int t, int &I, int &J) {
    I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    while (I * (I + 1) / 2 > t) --I;
    J = t - I * (I + 1) / 2;
}

// ---- 6-byte ("split") storage of the packed inverse ----------------------------------------------------------------
// The mat-vec is HBM-bound on the bytes of M, and M = (G + I/mu)^-1 comes out of the block sweep with a normwise error of
// ~1e-12 (|M H - I|_max = 2e-13 at n = 8192, tools/factor_check.py): the trailing 13 bits of its doubles carry no
// information.  An element is stored as the 48 leading bits of its double, rounded to nearest at bit 13, in two parts:
//   head = those bits down to bit 29 as a FLOAT (sign, exponent, 23 mantissa bits: exactly the double with its low 29 bits
//          cleared, which is always a float for |M| in [2^-120, 2^127]; smaller magnitudes are flushed to 0),
//   tail = the next 16 mantissa bits (bits 28..13 of the double) as an unsigned short.
// 40 significant bits, relative error <= 2^-40 = 9.1e-13 per element -- below the accuracy M has anyway -- in 6 bytes
// instead of 8: 25 % fewer bytes per iteration.  Decoding is exact and costs three VALU instructions per element:
// v_cvt_f64_f32 (whose low dword has only its top 3 bits set), extract the tail, v_lshl_or_b32 into that low dword.
//   tile layout (98304 B): head[128][128] float, then tail[128][128] uint16 with the columns of a row permuted so that the
//   8 tails a lane needs are one 16-byte load: position 8c + 4h + k holds column 64h + 4c + k  (c < 16, h < 2, k < 4).
// Lane (g = lane >> 4, c = lane & 15) of wave w owns rows 32w + 4rg + g (rg < 8) and columns {4c+k, 64+4c+k}: per row group
// two float4 and one uint4 load (every instruction covers whole 128-byte lines): 24 loads = 384 bytes in flight per lane,
// two workgroups per CU (three would need <= 168 registers and spill: measured 38.3 us against 30.4 us per launch at
// np = 8192, i.e. 6.7 TB/s of 6-byte elements; the 8-byte kernel: 41.6 us, 6.56 TB/s).  Single right-hand side only
// (multi-signal handles keep doubles for the matrix-core tile product).
//
// ACCURACY.  A reduced-precision copy of M must not multiply the large constant vector b: the rounding of the small
// eigenvalues of M (the large ones of G) would be amplified by cond(G + I/mu) -- measured 5.3e-9 rel-L2 in z after 2000
// iterations at the cfg3 size (9e-9 at n = 32768), above the 1e-9 parity bound.  So the x-update runs in OFFSET FORM
// (AdmmParams::xb): xb = M b once from the full-precision inverse, and per iteration x = xb + M~ (z-u)/mu.  Near the
// solution (z-u)/mu = x/mu - subgradient, so |dM (z-u)/mu| <= 2^-40 |M| |x| / mu <= 2^-40 |x|: no amplification.
// Measured with the offset form: 1.2e-10 rel-L2 in z against the 8-byte storage at cfg3 (2000 iterations), 6e-11 against
// an exact-solve CPU run of the same ADMM at n = 2176 (300 iterations; the 8-byte storage: 8e-13), identical supports and stopping iterations.
constexpr size_t kSplitTileBytes = (size_t)TS * TS * 6;

__device__ __forceinline__ double split_decode(float head, unsigned int tail16) {
    const double d = (double)head;
    return __hiloint2double(__double2hiint(d), (int)((tail16 << 13) | (unsigned int)__double2loint(d)));
}
// the same with the shift-or as ONE instruction (the compiler otherwise masks after shifting and ors separately): on gfx950
// every vector instruction of a wave that shares a SIMD with fp64 MFMAs costs the matrix pipe ~7 cycles
__device__ __forceinline__ double split_decode_lo(float head, unsigned int pair) {   // tail = low half of `pair`
    const double d = (double)head;
    unsigned int lo = (unsigned int)__double2loint(d), t = pair & 0xffffu;
    asm("v_lshl_or_b32 %0, %1, 13, %0" : "+v"(lo) : "v"(t));
    return __hiloint2double(__double2hiint(d), (int)lo);
}
__device__ __forceinline__ double split_decode_hi(float head, unsigned int pair) {   // tail = high half of `pair`
    const double d = (double)head;
    unsigned int lo = (unsigned int)__double2loint(d), t = pair >> 16;
    asm("v_lshl_or_b32 %0, %1, 13, %0" : "+v"(lo) : "v"(t));
    return __hiloint2double(__double2hiint(d), (int)lo);
}
// an SSA value the optimiser cannot look through: keeps `up ? a[k] : a[k+cnt]` from becoming a dynamically indexed array
// access (which the backend then lowers to an 8-way select chain per value)
__device__ __forceinline__ double opaque(double v) { asm volatile("" : "+v"(v)); return v; }

// ---- 36-bit fixed-point tiles (mixed storage of the single-signal packed inverse) -------------------------------------
// M = (G + I/mu)^-1 of the LPV / Fourier problems is strongly diagonally dominant: at cfg3 the largest entry of an
// off-diagonal tile is 2^-9.4 .. 2^-10.7 of the diagonal's.  The error of the product M~ v is then dominated by the rounding
// of the LARGE entries (diagonal tiles, 2^-41 relative); the small entries' 40 significant bits are ~10 bits more absolute
// precision than is ever felt.  A tile whose rows are all small is therefore stored as 36-bit fixed point against a per-row
// power-of-two step:   element = q * step[row],  q = 16 * hi32 + nibble  (two's complement, |q| < 2^35),
//   stored biased, q + 2^35 = 16 * hi + nibble with hi an unsigned dword;
//   tile slot (same 98304-byte stride): hi[128][128] uint32 (65536 B), nibbles (8192 B), step[128] float (512 B) = 74240 B.
// Measured on cfg3's inverse (tools/quant_study.py): |dM v| / |x| = 3.9e-13 with these tiles against 3.2e-13 with 40-bit
// elements everywhere (32-bit fixed point: 3.8e-12).  Eligibility is decided per tile when packing: every row's step must be
// <= 2^-44 * max|M| * sqrt(8192 / np) (the fixed-point errors of a row add up over ~np entries); diagonal tiles and tiles
// that fail keep the 6-byte float-head format, so a matrix without this structure loses nothing.  A DIAGONAL tile whose entries
// off the main diagonal pass the same test is stored as fixed point too, with its 128 diagonal entries apart in doubles (1024 B
// after the steps; their fixed-point value is 0) -- the nearly diagonal inverses of the Fourier windows.  tile type: 0 = float
// head + 16-bit tail, 1 = fixed point, 2 = fixed point + double diagonal.  Decoding: v_bfe_u32, v_alignbit_b32, v_lshl_or_b32, one v_add_f64 (exact integer in a double);
// the row step multiplies the row sum once and the row's right-hand-side value once (for the transposed product).
// Layouts follow the lane ownership of the split kernel (lane (g, c) of wave w: rows 32w + 4rg + g, columns 4c+k, 64+4c+k):
//   nibbles: dword (w*64 + lane)*8 + rg holds the row group's 8 nibbles, nibble k at bits 4k (k < 4: column 4c+k, else 64+4c+k-4)
//   steps:   float (w*4 + g)*8 + rg = step of row 32w + 4rg + g
constexpr size_t kFixHeadBytes = (size_t)TS * TS * 4, kFixNibBytes = (size_t)TS * TS / 2;

// The common tail of the single-signal tile products: v[rg] = the lane's partial row sums of its 8 row groups (rows 32w + 4rg + g),
// tc[k] = its partial column sums of its 8 columns; row sums by a halving butterfly over the 16 column lanes, column sums over the
// wave's four row lanes and then over the four waves through LDS.  All 256 threads call it.
__device__ __forceinline__ void tile_reduce_store(double (&v)[8], double (&tc)[8], double (*sT)[TS], bool offdiag,
                                                  double *__restrict__ part1, double *__restrict__ part2,
                                                  const double *__restrict__ diag = nullptr, const double *sI = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
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
    if ((c & 1) == 0) {
        const int rg = ((c & 8) ? 4 : 0) + ((c & 4) ? 2 : 0) + ((c & 2) ? 1 : 0);
        const int row = wave * 32 + 4 * rg + g;
        part1[row] = diag != nullptr ? fma(diag[row], sI[row], v[0]) : v[0];   // (a diagonal tile whose diagonal is kept apart in doubles)
    }
    if (offdiag) {
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
        if (threadIdx.x < TS)
            part2[threadIdx.x] = ((sT[0][threadIdx.x] + sT[1][threadIdx.x]) + sT[2][threadIdx.x]) + sT[3][threadIdx.x];
    }
}

struct FixRaw { int4 ha[8], hb[8]; uint4 nq[2]; float4 st[2]; };

// q = 16 * hi + nib (biased by 2^35) -> the double q - 2^35, exactly: the bits of 2^52 + q are assembled with two integer
// instructions (v_alignbit_b32 puts the top four bits of q under the exponent, v_lshl_or_b32 forms the low dword), then one
// subtraction.  (Integer conversions would be three double-rate instructions more per element.)
__device__ __forceinline__ double fix_decode(unsigned int hi, unsigned int nib) {
    const unsigned int top = __builtin_amdgcn_alignbit(0x04330000u, hi, 28);   // (0x04330000 << 4) | (hi >> 28) = 0x43300000 | q[35:32]
    unsigned int lo = nib;
    asm("v_lshl_or_b32 %0, %1, 4, %0" : "+v"(lo) : "v"(hi));
    return __hiloint2double((int)top, (int)lo) - (0x1p52 + 0x1p35);
}

// one 16-byte load; NT: non-temporal (streams larger than the 256 MiB Infinity Cache read ~14 % faster that way -- tools/stream_read.hip:
// 600 MB at 7.0 instead of 6.1 TB/s -- while a stream that fits it, the single-problem inverse of cfg3, gains nothing)
template <bool NT, typename V>
__device__ __forceinline__ V load16(const void *p) {
    typedef unsigned int u32x4n __attribute__((ext_vector_type(4)));
    static_assert(sizeof(V) == 16, "16-byte vectors only");
    const u32x4n r = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4n *>(p)) : *reinterpret_cast<const u32x4n *>(p);
    return __builtin_bit_cast(V, r);
}

// fmode (wave-uniform): 0 = the 36-bit element (heads + nibbles); 1 = its 32 leading bits only -- the nibbles are not read (4 B per element
// instead of 4.5) and count as zero; 2 = the NIBBLES only (the heads are the bias, so an element decodes to its nibble x step): the part
// mode 1 leaves out, for the stale nibble product of handles that iterate on 32-bit reads (launch_nibble_refresh)
template <bool NT = false>
__device__ __forceinline__ void fix_load(const unsigned char *tile, int wave, int lane, FixRaw &w, int fmode = 0) {
    const int g = lane >> 4, c = lane & 15;
    const bool fix32 = fmode == 1;
    // nibbles and steps FIRST: loads return in order, and the first row group's products need them -- requested last, they kept
    // every product waiting for the tile's last byte (all of a tile's arithmetic then sat at the end of its load)
    const uint4 *nq = reinterpret_cast<const uint4 *>(tile + kFixHeadBytes) + (wave * 64 + lane) * 2;
    if (fix32) { w.nq[0] = make_uint4(0, 0, 0, 0); w.nq[1] = w.nq[0]; }
    else { w.nq[0] = load16<NT, uint4>(nq); w.nq[1] = load16<NT, uint4>(nq + 1); }
    const float4 *st = reinterpret_cast<const float4 *>(tile + kFixHeadBytes + kFixNibBytes) + (wave * 4 + g) * 2;
    w.st[0] = load16<NT, float4>(st); w.st[1] = load16<NT, float4>(st + 1);
    const int *head = reinterpret_cast<const int *>(tile) + (wave * 32 + g) * TS + 4 * c;
    if (fmode == 2) {
        const int bias = (int)0x80000000u;           // 16 * 2^31 = 2^35: the element decodes to nibble * step
#pragma unroll
        for (int rg = 0; rg < 8; ++rg) { w.ha[rg] = make_int4(bias, bias, bias, bias); w.hb[rg] = w.ha[rg]; }
        return;
    }
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        w.ha[rg] = load16<NT, int4>(head + rg * 4 * TS);
        w.hb[rg] = load16<NT, int4>(head + rg * 4 * TS + 64);
    }
}

// the product of split_tile_product for a fixed-point tile (always off the diagonal)
__device__ __forceinline__ void fix_tile_product(const FixRaw &w, const double *sI, const double *sJ, double (*sT)[TS],
                                                 double *__restrict__ part1, double *__restrict__ part2, const double *__restrict__ diag = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    double rj[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
    double tc[8], v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tc[k] = 0.0;
    const float stv[8] = {w.st[0].x, w.st[0].y, w.st[0].z, w.st[0].w, w.st[1].x, w.st[1].y, w.st[1].z, w.st[1].w};
    const unsigned int nw[8] = {w.nq[0].x, w.nq[0].y, w.nq[0].z, w.nq[0].w, w.nq[1].x, w.nq[1].y, w.nq[1].z, w.nq[1].w};
    // the row's step folded into its right-hand-side value; requested one row group ahead, so that the column products can
    // be issued together with the row products (otherwise the compiler parks the eight decoded elements -- and spills)
    double ri = (double)stv[0] * sI[wave * 32 + g];
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        const double step = (double)stv[rg];
        const double ri_next = rg + 1 < 8 ? (double)stv[rg + 1] * sI[wave * 32 + 4 * (rg + 1) + g] : 0.0;
        const int hh[8] = {w.ha[rg].x, w.ha[rg].y, w.ha[rg].z, w.ha[rg].w, w.hb[rg].x, w.hb[rg].y, w.hb[rg].z, w.hb[rg].w};
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const double m0 = fix_decode((unsigned int)hh[k], (nw[rg] >> (4 * k)) & 15u);             // exact 36-bit integers
            const double m1 = fix_decode((unsigned int)hh[k + 1], (nw[rg] >> (4 * k + 4)) & 15u);
            tc[k] = opaque(fma(m0, ri, tc[k]));          // (pinned: left to itself the compiler parks all 64 decoded elements of the
            tc[k + 1] = opaque(fma(m1, ri, tc[k + 1]));  //  lane and issues the column products after the loop -- and spills)
            a0 = fma(m0, rj[k], a0);
            a1 = fma(m1, rj[k + 1], a1);
        }
        v[rg] = step * (a0 + a1);
        ri = ri_next;
        __builtin_amdgcn_sched_barrier(0);               // (row group by row group, as the bytes arrive)
    }
    tile_reduce_store(v, tc, sT, diag == nullptr, part1, part2, diag, sI);
}

// raw registers of one lane's share of a split tile (8 row groups: two float4 heads, one uint4 of tails)
struct SplitRaw { float4 ha[8], hb[8]; uint4 lq[8]; };

__device__ __forceinline__ void split_load(const unsigned char *tile, int wave, int g, int c, SplitRaw &w) {
    const float *head = reinterpret_cast<const float *>(tile) + (wave * 32 + g) * TS + 4 * c;
    const unsigned short *tail = reinterpret_cast<const unsigned short *>(tile + (size_t)TS * TS * 4) + (wave * 32 + g) * TS + 8 * c;
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        w.ha[rg] = *reinterpret_cast<const float4 *>(head + rg * 4 * TS);
        w.hb[rg] = *reinterpret_cast<const float4 *>(head + rg * 4 * TS + 64);
        w.lq[rg] = *reinterpret_cast<const uint4 *>(tail + rg * 4 * TS);
    }
}

// one right-hand side against the tile held in `w`: sI / sJ hold the right-hand side's blocks I and J (already visible),
// part1 / part2 point at this tile's 128 partials; sT is scratch.  All 256 threads of the workgroup call it.
__device__ __forceinline__ void split_tile_product(const SplitRaw &w, const double *sI, const double *sJ, double (*sT)[TS], bool offdiag,
                                                   double *__restrict__ part1, double *__restrict__ part2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    double rj[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { rj[k] = sJ[4 * c + k]; rj[4 + k] = sJ[64 + 4 * c + k]; }
    double tc[8], v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tc[k] = 0.0;
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
        const double ri = sI[wave * 32 + 4 * rg + g];
        const float hh[8] = {w.ha[rg].x, w.ha[rg].y, w.ha[rg].z, w.ha[rg].w, w.hb[rg].x, w.hb[rg].y, w.hb[rg].z, w.hb[rg].w};
        const unsigned int qq[8] = {w.lq[rg].x & 0xffffu, w.lq[rg].x >> 16, w.lq[rg].y & 0xffffu, w.lq[rg].y >> 16,
                                    w.lq[rg].z & 0xffffu, w.lq[rg].z >> 16, w.lq[rg].w & 0xffffu, w.lq[rg].w >> 16};
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const double m0 = split_decode(hh[k], qq[k]), m1 = split_decode(hh[k + 1], qq[k + 1]);
            tc[k] = fma(m0, ri, tc[k]);                  // (pinning these as in fix_tile_product frees 40 registers and a third
            tc[k + 1] = fma(m1, ri, tc[k + 1]);          //  workgroup per CU, but measured 31.9 us against 30.6)
            a0 = fma(m0, rj[k], a0);
            a1 = fma(m1, rj[k + 1], a1);
        }
        v[rg] = a0 + a1;
    }
    // row sums: halving butterfly over the 16 column lanes (after the step with mask m a lane keeps the row groups whose
    // bit matches its own), then the last pair
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
    if ((c & 1) == 0) {
        const int rg = ((c & 8) ? 4 : 0) + ((c & 4) ? 2 : 0) + ((c & 2) ? 1 : 0);
        part1[wave * 32 + 4 * rg + g] = v[0];
    }
    if (offdiag) {
        // column sums: over the wave's four row lanes g (masks 32, 16), then over the four waves through LDS
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
        if (threadIdx.x < TS)
            part2[threadIdx.x] = ((sT[0][threadIdx.x] + sT[1][threadIdx.x]) + sT[2][threadIdx.x]) + sT[3][threadIdx.x];
    }
}

// ||x - z|| of the iteration whose block norms are in bn[0 .. nblk): every caller (the deferred commit of the two-launch update, the commit kernels of the
// one-launch iteration) sums them in the SAME fixed order, so that all workgroups take the same stopping decision (src/lasso.jl:157,164)
__device__ __forceinline__ double pending_norm(const double *__restrict__ bn, int nblk, double *slot) {
    if (threadIdx.x < 64) {   // lane q sums blocks q, q+64, ...; then the wave's fixed shuffle pattern
        double part = 0;
        for (int q = threadIdx.x; q < nblk; q += 64) part += bn[q];
        const double w = wave_sum(part);
        if (threadIdx.x == 0) *slot = w;
    }
    __syncthreads();
    return sqrt(*slot);                                               // norm(tmp)   src/lasso.jl:157
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kPanelC = 4;                           // tile columns of a column panel of the multi-signal kernel's PANEL walk (admm_multi.hip); its consumers (admm.hip) index the records by it

}  // namespace
}  // namespace lpvs
