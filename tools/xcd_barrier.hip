// Micro-benchmark: cost of a grid barrier among G workgroups that all sit on ONE XCD (blockIdx % 8 == 0 of an 8 G grid: workgroups are
// dealt to the XCDs round-robin), against G workgroups spread over all XCDs.  Monotonic counter in global memory, agent-scope atomics,
// bounded spin.  Each round also passes 8 KB of data (every workgroup writes 32 doubles, reads all 32 G after the barrier).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/xcd_barrier tools/xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) bar(unsigned long long *cnt, double *buf, int G, int rounds, int stride, int *fail, double *out) {
    if ((int)blockIdx.x % stride != 0) return;
    const int g = blockIdx.x / stride;
    double acc = 0;
    for (int r = 0; r < rounds; ++r) {
        double *cur = buf + (size_t)(r & 1) * 32 * G;
        if (threadIdx.x < 32) __hip_atomic_store(cur + g * 32 + threadIdx.x, (double)(r + g + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long target = (unsigned long long)G * (r + 1);
            long spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1l << 22)) { *fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (*reinterpret_cast<volatile int *>(fail)) return;
        for (int i = threadIdx.x; i < 32 * G; i += 256) acc += __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) out[g] = acc;
}
int main() {
    unsigned long long *cnt; double *buf, *out; int *fail;
    (void)hipMalloc(&cnt, 8); (void)hipMalloc(&buf, 8 * 2 * 32 * 256); (void)hipMalloc(&out, 8 * 256); (void)hipMalloc(&fail, 4);
    const int rounds = 2000;
    for (int cfg = 0; cfg < 6; ++cfg) {
        const int G = cfg < 3 ? 32 : (cfg == 3 ? 16 : (cfg == 4 ? 64 : 256));
        const int stride = cfg == 0 ? 8 : (cfg == 1 ? 1 : (cfg == 2 ? 8 : (cfg == 3 ? 8 : (cfg == 4 ? 4 : 1))));   // 8: one XCD; 1: blocks 0..G-1 (all XCDs)
        (void)hipMemset(cnt, 0, 8); (void)hipMemset(fail, 0, 4);
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(bar, dim3(G * stride), dim3(256), 0, 0, cnt, buf, G, rounds, stride, fail, out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        int hf = 0; (void)hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
        printf("G=%3d stride=%d (%s): %.3f us per round%s\n", G, stride, stride == 8 ? "one XCD" : (stride == 4 ? "two XCDs" : "all XCDs"), ms * 1e3 / rounds, hf ? "  [SPIN LIMIT HIT]" : "");
    }
    return 0;
}
