// Pricing, before anything is built, of an ADMM iteration whose packed inverse stays ON THE CHIP (cfg3: 140.2 MB of 32-bit fixed-point tiles against
// 256 CUs x (512 KB of vector registers + 160 KB of LDS) = 172 MB): one persistent workgroup per CU, one grid barrier per iteration.  Two unknowns:
//   1. a grid barrier among 256 co-resident workgroups that also passes the iteration's vectors (every workgroup writes 32 doubles and reads all
//      8192 after the barrier): flat (one counter, everybody polls it) against two levels (16 groups of 16 arrive on their own counters, the last of
//      a group on a second-level counter, the last of all raises 16 flags, a workgroup polls its group's flag);
//   2. the product itself from registers: every lane keeps R packed dwords in vector / accumulator registers (R = 320: 80 KB per wave, 320 KB per CU
//      next to ~100 KB of LDS) and per iteration decodes each (three integer instructions + one FMA with the row's step) and multiplies it into a row
//      sum and a column sum (two more FMAs), the right-hand side coming from LDS.
// Every spin is bounded (a workgroup that is not co-resident ends the run with a flag instead of hanging the device).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/resident_price tools/resident_price.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kGroups = 16;
struct Bar { unsigned long long c1[kGroups][16]; unsigned long long c2[16]; unsigned long long flag[kGroups][16]; };   // one 128-byte line each

template <int ORDER = __ATOMIC_ACQUIRE>
__device__ __forceinline__ bool spin_until(unsigned long long *p, unsigned long long target, int *fail) {
    long spins = 0;
    while (__hip_atomic_load(p, ORDER, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > (1l << 21)) { *fail = 1; return false; }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

template <int TWO_LEVEL>   // 0: flat; 1: two levels, acquire / release on every operation; 2: two levels, relaxed operations between ONE release fence and ONE acquire fence
__global__ void __launch_bounds__(256) barrier_rounds(Bar *b, unsigned long long *flat, double *buf, int G, int rounds, int *fail, double *out, int data /* 0: none; 1: agent-scope atomic loads; 2: plain loads behind the acquire fence */) {
    const int g = blockIdx.x, grp = g % kGroups;     // (consecutive workgroups sit on different XCDs: a group spans all of them)
    const int per = G / kGroups;
    double acc = 0;
    for (int r = 0; r < rounds; ++r) {
        double *cur = buf + (size_t)(r & 1) * 32 * G;
        if (data && threadIdx.x < 32) __hip_atomic_store(cur + g * 32 + threadIdx.x, (double)(r + g + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long gen = (unsigned long long)(r + 1);
            if (TWO_LEVEL == 2) {
                __atomic_thread_fence(__ATOMIC_RELEASE);   // (agent scope is the default of the HIP fence builtin for device code: this workgroup's 32 doubles are visible)
                const unsigned long long a = __hip_atomic_fetch_add(&b->c1[grp][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a == gen * per - 1) {
                    const unsigned long long a2 = __hip_atomic_fetch_add(&b->c2[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (a2 == gen * kGroups - 1)
                        for (int q = 0; q < kGroups; ++q) __hip_atomic_store(&b->flag[q][0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                spin_until<__ATOMIC_RELAXED>(&b->flag[grp][0], gen, fail);
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
            } else if (TWO_LEVEL == 1) {
                const unsigned long long a = __hip_atomic_fetch_add(&b->c1[grp][0], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (a == gen * per - 1) {
                    const unsigned long long a2 = __hip_atomic_fetch_add(&b->c2[0], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    if (a2 == gen * kGroups - 1)
                        for (int q = 0; q < kGroups; ++q) __hip_atomic_store(&b->flag[q][0], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                spin_until(&b->flag[grp][0], gen, fail);
            } else {
                __hip_atomic_fetch_add(flat, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                spin_until(flat, gen * G, fail);
            }
        }
        __syncthreads();
        if (*reinterpret_cast<volatile int *>(fail)) return;
        if (data == 1) for (int i = threadIdx.x; i < 32 * G; i += 256) acc += __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (data == 2) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);     // (every wave: the fence of thread 0 above covered its own wave's cache view only as far as the L1 goes)
            double t[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) t[i] = cur[threadIdx.x + 256 * i];
#pragma unroll
            for (int i = 0; i < 32; ++i) acc += t[i];
        }
    }
    if (threadIdx.x == 0) out[g] = acc;
}

// ---- the product from registers
constexpr int R = 320;
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
resident_product(const unsigned int *__restrict__ packed, const float *__restrict__ steps, const double *__restrict__ rhs, int iters, double *__restrict__ out) {
    __shared__ double sI[128], sJ[128];
    unsigned int m[R];
#pragma unroll
    for (int k = 0; k < R; ++k) m[k] = packed[((size_t)blockIdx.x * R + k) * 256 + threadIdx.x];   // loaded ONCE
    const double step = (double)steps[threadIdx.x & 127], off = -(0x1p52 + 0x1p35) * step;
    double row = 0, col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x < 128) sI[threadIdx.x] = rhs[(it & 7) * 256 + threadIdx.x]; else sJ[threadIdx.x - 128] = rhs[(it & 7) * 256 + threadIdx.x];
        __syncthreads();
        double ri[8], rj[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { ri[k] = sI[(threadIdx.x + 16 * k) & 127]; rj[k] = sJ[(threadIdx.x * 8 + k) & 127]; }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            unsigned int hi = m[k];
            asm volatile("" : "+v"(hi));           // (the decode must not be hoisted out of the iteration loop: the doubles would not fit)
            const unsigned int top = __builtin_amdgcn_alignbit(0x04330000u, hi, 28);
            const unsigned int lo = hi << 4;
            const double v = fma(__hiloint2double((int)top, (int)lo), step, off);
            row = fma(v, rj[k & 7], row);
            col[k & 7] = fma(v, ri[(k >> 3) & 7], col[k & 7]);
        }
        __syncthreads();
    }
    double s = row;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += col[k];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int G = 256;
    if (cus < G) { printf("needs %d CUs (device has %d)\n", G, cus); return 1; }
    Bar *b; unsigned long long *flat; double *buf, *out; int *fail;
    (void)hipMalloc(&b, sizeof(Bar)); (void)hipMalloc(&flat, 128); (void)hipMalloc(&buf, 8 * 2 * 32 * G); (void)hipMalloc(&out, 8 * 256 * 256); (void)hipMalloc(&fail, 4);
    const int rounds = 2000;
    for (int two = 0; two < 3; ++two)
        for (int data = 0; data < 3; ++data) {
            (void)hipMemset(b, 0, sizeof(Bar)); (void)hipMemset(flat, 0, 128); (void)hipMemset(fail, 0, 4);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            if (two == 2) hipLaunchKernelGGL(barrier_rounds<2>, dim3(G), dim3(256), 0, 0, b, flat, buf, G, rounds, fail, out, data);
            else if (two == 1) hipLaunchKernelGGL(barrier_rounds<1>, dim3(G), dim3(256), 0, 0, b, flat, buf, G, rounds, fail, out, data);
            else hipLaunchKernelGGL(barrier_rounds<0>, dim3(G), dim3(256), 0, 0, b, flat, buf, G, rounds, fail, out, data);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            int hf = 0; (void)hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
            printf("grid barrier, %d workgroups, %s; %s: %.2f us per round%s\n", G, data == 0 ? "no data" : (data == 1 ? "64 KB exchanged with agent-scope atomic loads" : "64 KB exchanged with plain loads behind an acquire fence"), two == 2 ? "two levels, relaxed operations between one release and one acquire fence" : (two ? "two levels (16 x 16, a flag per group), acquire / release on every operation" : "flat (one counter)"), ms * 1e3 / rounds,
                   hf ? "  [SPIN LIMIT HIT]" : "");
        }
    unsigned int *packed; float *steps; double *rhs;
    (void)hipMalloc(&packed, sizeof(unsigned int) * (size_t)G * R * 256); (void)hipMalloc(&steps, 4 * 128); (void)hipMalloc(&rhs, 8 * 8 * 256);
    (void)hipMemset(packed, 0x5a, sizeof(unsigned int) * (size_t)G * R * 256); (void)hipMemset(steps, 0x3c, 4 * 128); (void)hipMemset(rhs, 0x3f, 8 * 8 * 256);
    for (int rep = 0; rep < 2; ++rep) {
        float ms[2];
        for (int q = 0; q < 2; ++q) {
            const int iters = q ? 2200 : 200;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(resident_product, dim3(G), dim3(256), 0, 0, packed, steps, rhs, iters, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms[q], e0, e1);
        }
        printf("product from registers: %d dwords per lane (%d KB per CU, %.1f MB on the chip), decode + row and column FMA: %.2f us per iteration (2000 more iterations: %.3f ms)\n", R,
               R * 256 * 4 / 1024, (double)R * 256 * 4 * G * 1e-6, (ms[1] - ms[0]) * 1e3 / 2000, ms[1] - ms[0]);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
    return 0;
}
