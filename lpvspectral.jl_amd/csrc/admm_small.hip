// admm_small.hip -- one launch per ADMM iteration for problems below np = 2048 on the full symmetric inverse (BASELINE cfg2, n = 1024).
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

// ---- one launch per iteration for the full-matrix path (np < kSymmetricMinNp: cfg2's n = 1024) ---------------------------------------
// The two-launch iteration of these sizes (symv_kernel + admm_batch_prox_kernel) is bound by its two dependent launch boundaries and
// two lone-workgroup latency chains: 6.6 us per iteration for 8 MB of matrix.  Here an iteration is ONE launch of np / 4 workgroups:
//   * every workgroup redoes the WHOLE update of the previous iteration from x, u, b (np <= 2047 elements: a few per thread) -- z =
//     prox(x + u), u += x - z, the next right-hand side into LDS, and ||x - z|| in a fixed order, so that every workgroup takes the
//     SAME stopping decision without talking to another (the reference's test, src/lasso.jl:164, in the iteration it belongs to: no
//     deferral); the workgroup that owns four elements writes their z and u;
//   * then its four rows of x = M rhs (one wave per row; the rows were requested before anything else and are in flight during the
//     update) go to the OTHER x buffer: x and u are double-buffered by launch parity, nothing is read and written in the same launch;
//   * "converged before this launch" travels in a per-parity control word written by workgroup 0 of the previous launch.
// A chunk is: first launch (rhs from memory, no update), iters - 1 fused launches, a last update-only launch, and a fix-up that
// brings x and u back to the handle's vectors when the final state sits in the alternate buffers.  Bit-reproducible (fixed orders).
enum { SM_FIRST = 0, SM_MID = 1, SM_LAST = 2 };
constexpr int kSmallMaxNp = 2048;                      // (LDS image of the right-hand side; np < kSymmetricMinNp anyway)
template <int MODE, int NW /* waves = rows of M per workgroup */, int NPMAX /* np <= NPMAX: 1024 or kSmallMaxNp */>
__global__ void __launch_bounds__(64 * NW)
admm_small_iter_kernel(AdmmParams p, int j /* launch of the chunk: 0 = first */) {
    constexpr int NT = 64 * NW, EPT = NPMAX / NT, MT = NPMAX / 128;   // threads, elements of the state per thread (i = tid + NT k), 16-byte pieces of a row per lane
    typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2b __attribute__((ext_vector_type(2)));
    __shared__ double srhs[NPMAX], sv[NPMAX], red[NW];
    const int sg = blockIdx.y;
    const int64_t o = (int64_t)sg * p.np;
    const int np = (int)p.np, n = (int)p.n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *ctl = p.sm_ctl + 4 * sg;
    AdmmStatus *status = p.status + sg;
    double *const Xb[2] = {p.x + o, p.scratch + (int64_t)sg * 2 * p.np};
    double *const Ub[2] = {p.u + o, p.scratch + (int64_t)sg * 2 * p.np + p.np};
    const int was_conv = MODE == SM_FIRST ? __builtin_nontemporal_load(&status->converged) : __builtin_nontemporal_load(&ctl[(j - 1) & 1]);
    // ---- the state, eight elements per thread (i = tid + 256 k), through descriptors of np doubles
    double xv[EPT], uv[EPT], bv[EPT];
    if (MODE == SM_FIRST) {
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(p.rhs + o, 0, np * 8, 0x00020000);
#pragma unroll
        for (int k = 0; k < EPT; ++k) xv[k] = __builtin_bit_cast(double, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(rr, (int)threadIdx.x * 8, k * NT * 8, 0));
    } else {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(Xb[j & 1], 0, np * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(Ub[(j - 1) & 1], 0, np * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.b + o), 0, np * 8, 0x00020000);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            xv[k] = __builtin_bit_cast(double, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(rx, (int)threadIdx.x * 8, k * NT * 8, 0));
            uv[k] = __builtin_bit_cast(double, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(ru, (int)threadIdx.x * 8, k * NT * 8, 0));
            bv[k] = __builtin_bit_cast(double, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(rb, (int)threadIdx.x * 8, k * NT * 8, 0));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- then this workgroup's rows of M (one wave per row, a lane covers columns 2 lane + 128 t): unconditional loads through a
    // descriptor of the row's size -- columns past np read as zero.  Loads return in order: the state (requested first) arrives
    // first, and the update runs while the rows are still in flight (rows first: 5.73 us per iteration at cfg2; state first: see DESIGN)
    const int row = (int)blockIdx.x * NW + wave;
    double2 mrow[MT];
    if (MODE != SM_LAST) {
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.M + (int64_t)row * p.np), 0, row < np ? np * 8 : 0, 0x00020000);
#pragma unroll
        for (int t = 0; t < MT; ++t) mrow[t] = __builtin_bit_cast(double2, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(rm, lane * 16, t * 1024, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == SM_FIRST) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl[0] = was_conv; ctl[2] = was_conv ? 0 : -1; }
        if (was_conv) return;
#pragma unroll
        for (int k = 0; k < EPT; ++k) srhs[threadIdx.x + NT * k] = xv[k];
    } else {
        if (was_conv) { if (blockIdx.x == 0 && threadIdx.x == 0) ctl[j & 1] = 1; return; }
        // ---- update of iteration (base + j): z = prox(x + u), u += x - z, rhs = b + (z - u)/mu, ||x - z||     src/lasso.jl:153-157
        double zv[EPT], vv[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) { const int i = threadIdx.x + NT * k; vv[k] = i < n ? xv[k] + uv[k] : 0.0; }
        if (p.prox_kind == LPVS_PROX_GROUP_L2) {
            const int gl = (int)p.group_len;
#pragma unroll
            for (int k = 0; k < EPT; ++k) sv[threadIdx.x + NT * k] = vv[k] * vv[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int i = threadIdx.x + NT * k;
                double s2 = 0;
                if (i < n) { const double *grp = sv + (i / gl) * gl; for (int q = 0; q < gl; ++q) s2 += grp[q]; }   // the same sequential order as norm() on the slice
                double scale = 1.0 - p.prox_param * p.mu / sqrt(s2);
                if (!(scale > 0)) scale = 0.0;
                zv[k] = i < n ? scale * vv[k] : 0.0;
            }
        } else {
            const double gl1 = p.mu * p.prox_param, th0 = sqrt(2.0 * p.mu * p.prox_param);
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const double v = vv[k];
                zv[k] = p.prox_kind == LPVS_PROX_L1 ? v + (v <= -gl1 ? gl1 : (v >= gl1 ? -gl1 : -v)) : (fabs(v) > th0 ? v : 0.0);
            }
        }
        double ss = 0, un[EPT], rh[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = threadIdx.x + NT * k;
            const bool ok = i < n;
            const double xi = ok ? xv[k] : 0.0, d = xi - zv[k];
            un[k] = (ok ? uv[k] : 0.0) + d;
            rh[k] = ok ? bv[k] + (zv[k] - un[k]) / p.mu : 0.0;
            ss = fma(d, d, ss);
            srhs[i] = rh[k];
        }
        double nxz = 0.0;
        if (p.tol > 0.0 || blockIdx.x == 0) {         // (uniform; tol <= 0 can never stop: only the workgroup that keeps the status needs the norm then)
            const double w = wave_sum(ss);
            if (lane == 0) red[wave] = w;
            __syncthreads();
            double tot = 0;
#pragma unroll
            for (int q = 0; q < NW; ++q) tot += red[q];                         // fixed order
            nxz = sqrt(tot);              // every workgroup, identically     norm(tmp)   src/lasso.jl:157
        }
        const bool conv = nxz < p.tol;                                        //                                  src/lasso.jl:164
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            status->iters += 1;
            status->nxz = nxz;
            if (conv) { status->converged = 1; ctl[2] = j; }
            ctl[j & 1] = conv ? 1 : 0;
        }
        // the owner of elements [NW b, NW b + NW) writes their z and u (and the right-hand side when the chunk -- or the run -- ends here)
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = threadIdx.x + NT * k;
            if (i / NW == (int)blockIdx.x && i < np) {
                p.z[o + i] = zv[k];
                Ub[j & 1][i] = un[k];
                if (conv || MODE == SM_LAST) p.rhs[o + i] = rh[k];
            }
        }
        if (conv || MODE == SM_LAST) return;
    }
    __syncthreads();
    // ---- this wave's row of x = M rhs into the other buffer
    double acc = 0;
    const double2 *r2 = reinterpret_cast<const double2 *>(srhs);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        if (lane + 64 * t < np / 2) {                 // (uniform per t up to the last partial group; the loads above were unconditional)
            const double2 v = r2[lane + 64 * t];
            acc = fma(mrow[t].x, v.x, acc);
            acc = fma(mrow[t].y, v.y, acc);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0 && row < np) Xb[(j + 1) & 1][row] = acc;
}
// final state of a chunk of `iters` launches back into the handle's vectors when it sits in the alternate buffers (blockIdx.y = signal)
__global__ void __launch_bounds__(256)
admm_small_fixup_kernel(AdmmParams p, int iters) {
    const int sg = blockIdx.y;
    const int jc = p.sm_ctl[4 * sg + 2] >= 0 ? p.sm_ctl[4 * sg + 2] : iters;   // the launch that wrote the final u (read the final x)
    if ((jc & 1) == 0) return;
    const int64_t o = (int64_t)sg * p.np, e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < p.np) {
        p.x[o + e] = p.scratch[(int64_t)sg * 2 * p.np + e];
        p.u[o + e] = p.scratch[(int64_t)sg * 2 * p.np + p.np + e];
    }
}
bool small_iter_applicable(const AdmmParams &p) {
    const bool on = option_in_effect(LPVS_OPT_ITERATION, p.opt_iteration) != LPVS_ITERATION_TWO;
    if (!on || p.sm_ctl == nullptr || p.Mp != nullptr || p.scratch == nullptr || p.np > kSmallMaxNp || p.np < 128 || p.xb != nullptr) return false;
    if (p.prox_kind == LPVS_PROX_L1 || p.prox_kind == LPVS_PROX_L0) return true;
    return p.prox_kind == LPVS_PROX_GROUP_L2 && p.group_len >= 1 && p.group_len <= 256 && p.n % p.group_len == 0;
}
template <int NW, int NPMAX>
static void launch_small_launches(const AdmmParams &p, int64_t iters, hipStream_t s) {
    const dim3 grid((unsigned)(p.np / NW), (unsigned)p.ns), blk(64 * NW);
    hipLaunchKernelGGL((admm_small_iter_kernel<SM_FIRST, NW, NPMAX>), grid, blk, 0, s, p, 0);
    for (int64_t j = 1; j < iters; ++j) hipLaunchKernelGGL((admm_small_iter_kernel<SM_MID, NW, NPMAX>), grid, blk, 0, s, p, (int)j);
    hipLaunchKernelGGL((admm_small_iter_kernel<SM_LAST, NW, NPMAX>), grid, blk, 0, s, p, (int)iters);
}
int32_t launch_small_chunk(const AdmmParams &p, int64_t iters, hipStream_t s) {
    // rows of M per workgroup (= waves): 4 (measured at cfg2: 4 rows 5.37 us, 8 rows 5.47, 16 rows 6.47 per iteration: profiles/r04_cfg2_rows_per_workgroup_before.txt)
    if (p.np <= 1024) launch_small_launches<4, 1024>(p, iters, s);
    else launch_small_launches<4, kSmallMaxNp>(p, iters, s);
    hipLaunchKernelGGL(admm_small_fixup_kernel, dim3((unsigned)ceil_div(p.np, 256), (unsigned)p.ns), dim3(256), 0, s, p, (int)iters);
    LPVS_HIP(hipGetLastError());
    return LPVS_OK;
}

}  // namespace lpvs
