// admm_host.h -- host-side entry points that cross the ADMM translation units (what api.hip / multi.hip call is in lpvs_internal.h).
#pragma once
#include "lpvs_internal.h"

namespace lpvs {

// ---- admm.hip: the two-launch iteration (tile mat-vec + update), the prox kernels, the dispatch
void launch_sym_matvec(const AdmmParams &p, const AdmmStatus *status, hipStream_t s);         // tile partials of M~ rhs -> p.part (every storage, one or several signals)
void launch_symv_raw(const double *M, int64_t np, const double *rhs, double *x, const AdmmStatus *st, int ns, hipStream_t s);   // x = M rhs on the full symmetric matrix
void launch_packed_apply(const AdmmParams &p, const double *rhs, double *out, hipStream_t s);  // out = M~ rhs through the handle's own packed mat-vec + gather
bool fused_ok(const AdmmParams &p);
void launch_gather_tile_partials(const double *part1, const double *part2, int nblk, unsigned ntiles, int64_t np, double *out, hipStream_t s);   // out = gathered tile partials of one signal

// ---- admm_multi.hip: several right-hand sides sharing M (the matrix-core tile products)
void launch_multi_matvec(const AdmmParams &p, unsigned ntiles, double *part1, double *part2, const AdmmStatus *status, hipStream_t s);
bool multi_matvec_on_matrix_cores(const AdmmParams &p);   // this handle's product is one of admm_multi.hip's kernels (else admm.hip's VALU kernels)
bool uses_stream_kernel(const AdmmParams &p);
int stream_runs(const AdmmParams &p);                     // the stream kernel writes its row-sum partials per RUN of tiles: number of segments (0: per tile)
int stream_layout(const AdmmParams &p);                   // what the consumers of the partials take as `runs_G` (< 0: the panel walk's layout, -kPanelC)
const int *stream_table(const AdmmParams &p);             // ... and the panel walk's device table (nullptr otherwise)

// ---- admm_one_launch.hip / admm_small.hip: the whole iteration in one launch
int32_t launch_fi_chunk(const AdmmParams &p, int64_t iters, bool batch, size_t mp_stride, bool prefetch_all, hipStream_t s);
int32_t launch_small_chunk(const AdmmParams &p, int64_t iters, hipStream_t s);

}  // namespace lpvs
