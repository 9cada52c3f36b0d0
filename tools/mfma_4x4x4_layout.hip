// Probe: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (the local guides do not list it).
// For every pair (la, lb) of lanes: a = e_la, b = e_lb -> which result lanes are non-zero.  Prints, per result lane, the
// four (la, lb) pairs that feed it, and checks the hypothesis  block = lane >> 4;  A: i = lane & 3, k = (lane >> 2) & 3;
// B: j = lane & 3, k = (lane >> 2) & 3;  D: i = (lane >> 2) & 3 ... (whatever comes out is printed).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma_4x4x4_layout tools/mfma_4x4x4_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) probe(unsigned long long *hit) {      // hit[la*64 + lb] = bitmask of result lanes
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) hit[la * 64 + lb] = m;
        }
}
int main() {
    unsigned long long *d; (void)hipMalloc(&d, 8 * 4096);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<unsigned long long> h(4096);
    (void)hipMemcpy(h.data(), d, 8 * 4096, hipMemcpyDeviceToHost);
    for (int o = 0; o < 64; ++o) {
        printf("D lane %2d <-", o);
        for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >> o & 1) printf(" (a%d,b%d)", la, lb);
        printf("\n");
    }
    return 0;
}
