// Which XCD does workgroup b of a one-workgroup-per-CU launch (512 threads, 160 KB of LDS: the shape of k_ldlt_panel) land on?
// Prints xcc(b) for the first blocks and how many b have xcc(b) == (xcc(0) + b) mod 8 (the round-robin rule).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(unsigned *out) {
    extern __shared__ double sm[];
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;   // HW_REG_XCC_ID
    long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 40000) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 1000) sm[0] = 1.0;
}
int main() {
    hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int nb : {256, 200, 137, 512}) {
        unsigned *d; hipMalloc(&d, nb * 4);
        hipLaunchKernelGGL(k_probe, dim3(nb), dim3(512), 163840, 0, d);
        std::vector<unsigned> h(nb);
        hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
        int rr = 0; for (int b = 0; b < nb; ++b) rr += (h[b] == (h[0] + b) % 8);
        printf("grid %d: xcc of blocks 0..23:", nb); for (int b = 0; b < 24 && b < nb; ++b) printf(" %u", h[b]);
        printf("  | round-robin matches %d / %d\n", rr, nb);
        hipFree(d);
    }
    return 0;
}
