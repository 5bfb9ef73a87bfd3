// Which (XCC, SE, SH/SA, CU) does each workgroup land on?  (development probe for CU reservation)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k_probe(unsigned *out) {
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
        unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) __builtin_amdgcn_s_sleep(10);
}
int main() {
    const int nb = 512;
    unsigned *d; hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 65536, 0, d);   // 64 KB LDS each -> 2 per CU
    std::vector<unsigned> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long, int> cnt;
    unsigned orhw = 0, orx = 0;
    for (int b = 0; b < nb; ++b) {
        unsigned hw = h[2 * b], x = h[2 * b + 1];
        orhw |= hw; orx |= x;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        cnt[((unsigned long)(x & 0xf) << 16) | (se << 8) | (sh << 4) | cu]++;
        if (b < 24) printf("b=%3d hw=%08x xcc=%08x  -> xcc %u se %u sh %u cu %u simd %u wave %u\n", b, hw, x, x & 0xf, se, sh, cu, (hw >> 4) & 3, hw & 0xf);
    }
    printf("OR of hw_id = %08x, OR of xcc = %08x, distinct (xcc,se,sh,cu) = %zu\n", orhw, orx, cnt.size());
    int mx = 0; for (auto &kv : cnt) if (kv.second > mx) mx = kv.second;
    printf("max blocks per distinct id = %d\n", mx);
    return 0;
}
