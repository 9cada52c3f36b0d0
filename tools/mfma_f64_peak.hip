// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 (and f32 16x16x4, v_fma_f64) on gfx950, with
// the shader clock measured in-kernel (s_memtime / s_memrealtime).  The local guides list no fp64
// MFMA peak; this measures the denominator of the Gram roofline on the box the bench runs on.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak tools/mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, rt; };

template <int NACC, int MODE>  // MODE 0: mfma f64 16x16x4, 1: mfma f32, 2: v_fma_f64, 3: mfma f64 4x4x4 (4 blocks)
__global__ void __launch_bounds__(256) kern(double *out, Stamp *st, int iters, double a0, double b0) {
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    f64x4 acc[MODE == 0 ? NACC : 1]; f32x4 accf[MODE == 1 ? NACC : 1]; double accv[MODE == 2 ? NACC * 4 : 1]; double acc4[MODE == 3 ? NACC : 1];
    for (int i = 0; i < (MODE == 3 ? NACC : 1); ++i) acc4[i] = 0.0;
    for (int i = 0; i < (MODE == 0 ? NACC : 1); ++i) acc[i] = (f64x4){0, 0, 0, 0};
    for (int i = 0; i < (MODE == 1 ? NACC : 1); ++i) accf[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < (MODE == 2 ? NACC * 4 : 1); ++i) accv[i] = i;
    float af = (float)a, bf = (float)b;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
                else if constexpr (MODE == 1) accf[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, accf[i], 0, 0, 0);
                else if constexpr (MODE == 3) acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
                else { for (int j = 0; j < 4; ++j) accv[4*i+j] = fma(a, accv[4*i+j], b); }
            }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < (MODE == 0 ? NACC : 1); ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    for (int i = 0; i < (MODE == 1 ? NACC : 1); ++i) for (int j = 0; j < 4; ++j) s += accf[i][j];
    for (int i = 0; i < (MODE == 2 ? NACC * 4 : 1); ++i) s += accv[i];
    for (int i = 0; i < (MODE == 3 ? NACC : 1); ++i) s += acc4[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, r1 - r0}; }
}

template <int NACC, int MODE>
void run(const char *name, int wg_per_cu, int iters, double *o, Stamp *st) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t ea, eb; (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    hipLaunchKernelGGL((kern<NACC, MODE>), dim3(grid), dim3(256), 0, 0, o, st, iters, 1.0, 0.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(ea);
    hipLaunchKernelGGL((kern<NACC, MODE>), dim3(grid), dim3(256), 0, 0, o, st, iters, 1.0, 0.5);
    (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
    float ms; (void)hipEventElapsedTime(&ms, ea, eb);
    std::vector<Stamp> h(grid * 4);
    (void)hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (auto &s : h) { cyc.push_back((double)s.cyc); clk.push_back((double)s.cyc / (double)s.rt * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double ninstr = (double)iters * 4 * NACC * (MODE == 2 ? 4 : 1);
    const double flop_per = MODE == 2 ? 128.0 : (MODE == 3 ? 512.0 : 2048.0);
    const double fl = (double)grid * 4 * ninstr * flop_per;
    printf("%-10s acc=%-2d waves/SIMD=%d : %8.3f ms %7.1f TFLOP/s | %.1f cyc/instr/wave  clock %.0f MHz (median)\n", name, NACC, wg_per_cu, ms,
           fl / ms * 1e-9, cyc[cyc.size() / 2] / ninstr, clk[clk.size() / 2]);
}
int main() {
    double *o; Stamp *st;
    (void)hipMalloc(&o, 8 * 256 * 4096); (void)hipMalloc(&st, sizeof(Stamp) * 4096 * 4);
    for (int w = 1; w <= 4; ++w) {
        run<1, 0>("mfma_f64", w, 8000, o, st);
        run<2, 0>("mfma_f64", w, 8000, o, st);
        run<4, 0>("mfma_f64", w, 4000, o, st);
        run<8, 0>("mfma_f64", w, 2000, o, st);
        if (w <= 2) run<16, 0>("mfma_f64", w, 1000, o, st);
        run<4, 1>("mfma_f32", w, 4000, o, st);
        run<4, 3>("mfma_f64_4x4x4", w, 8000, o, st);
        run<16, 3>("mfma_f64_4x4x4", w, 4000, o, st);
        run<4, 2>("v_fma_f64", w, 4000, o, st);
    }
    return 0;
}
