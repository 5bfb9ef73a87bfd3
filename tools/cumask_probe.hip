// Which physical CUs does a CU-masked stream use?  (development probe for the two-stream look-ahead)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
__global__ void k_probe(unsigned *out) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    }
    long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 400000) __builtin_amdgcn_s_sleep(10);
}
static std::set<unsigned> run(hipStream_t s, unsigned *d, int nb) {
    hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 65536, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::set<unsigned> ids;
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[2 * b], x = h[2 * b + 1] & 0xf;
        ids.insert((x << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf));
    }
    return ids;
}
int main(int argc, char **argv) {
    const int period = argc > 1 ? atoi(argv[1]) : 32;      // clear bit i when i % period == 0
    const int nb = 1024;
    unsigned *d; hipMalloc(&d, nb * 8);
    hipStream_t s0; hipStreamCreate(&s0);
    const std::set<unsigned> all = run(s0, d, nb);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("device CUs %d, distinct CUs seen unmasked %zu\n", pr.multiProcessorCount, all.size());
    uint32_t mask[8];
    for (int w = 0; w < 8; ++w) mask[w] = 0xffffffffu;
    for (int i = 0; i < 256; ++i) if (i % period == 0) mask[i / 32] &= ~(1u << (i % 32));
    hipStream_t sm;
    hipError_t e = hipExtStreamCreateWithCUMask(&sm, 8, mask);
    printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    const std::set<unsigned> got = run(sm, d, nb);
    printf("distinct CUs seen masked %zu; missing:", got.size());
    for (unsigned id : all) if (!got.count(id)) printf(" (xcc %u se %u sh %u cu %u)", id >> 12, (id >> 8) & 7, (id >> 4) & 1, id & 0xf);
    printf("\n");
    return 0;
}
