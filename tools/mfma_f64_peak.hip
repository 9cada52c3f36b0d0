// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 (and the f32 16x16x4 form) on gfx950.
// The local guides do not list an fp64 MFMA peak; this measures the denominator of the Gram
// roofline on the box the bench runs on.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k_f64(double *out, int iters, double a0, double b0) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k_f32(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> double time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    double *o; hipMalloc(&o, 8 * 256 * 4096);
    const int iters = 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
        const int grid = 256 * wg_per_cu;  // 4 waves per WG -> 1 or 2 waves per SIMD
        double ms = time_ms([&] { hipLaunchKernelGGL(k_f64<4>, dim3(grid), dim3(256), 0, 0, o, iters, 1.0, 0.5); });
        double fl = (double)grid * 4 * iters * 4 * 2048.0;
        printf("f64 16x16x4  acc=4  waves/SIMD=%d : %.3f ms  %.1f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", wg_per_cu, ms, fl / ms * 1e-9,
               ms * 1e-3 * 2.4e9 / (iters * 4.0 * wg_per_cu));
        ms = time_ms([&] { hipLaunchKernelGGL(k_f64<16>, dim3(grid), dim3(256), 0, 0, o, iters / 4, 1.0, 0.5); });
        fl = (double)grid * 4 * (iters / 4) * 16 * 2048.0;
        printf("f64 16x16x4  acc=16 waves/SIMD=%d : %.3f ms  %.1f TFLOP/s\n", wg_per_cu, ms, fl / ms * 1e-9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(grid), dim3(256), 0, 0, (float *)o, iters, 1.0f, 0.5f); });
        fl = (double)grid * 4 * iters * 4 * 2048.0;
        printf("f32 16x16x4  acc=4  waves/SIMD=%d : %.3f ms  %.1f TFLOP/s\n", wg_per_cu, ms, fl / ms * 1e-9);
    }
    return 0;
}
