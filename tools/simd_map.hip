// Which SIMD does wave w of a workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13])
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/simd_map.hip -o /tmp/simd_map && /tmp/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main() {
    for (int threads : {256, 512, 1024}) {
        const int nw = threads / 64, nb = 4;
        unsigned *d; hipMalloc(&d, sizeof(unsigned) * nw * nb);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(threads), 0, 0, d);
        std::vector<unsigned> h(nw * nb);
        hipMemcpy(h.data(), d, sizeof(unsigned) * nw * nb, hipMemcpyDeviceToHost);
        for (int b = 0; b < nb; ++b) {
            printf("block of %4d threads #%d: simd of waves 0..%d:", threads, b, nw - 1);
            for (int w = 0; w < nw; ++w) printf(" %u", (h[b * nw + w] >> 4) & 3);
            printf("   (cu %u se %u)\n", (h[b * nw] >> 8) & 15, (h[b * nw] >> 13) & 7);
        }
        hipFree(d);
    }
    return 0;
}
