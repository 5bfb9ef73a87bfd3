// Micro-benchmark of the substitution TRSM kernel (development tool).
#include "../conicip.jl_amd/csrc/diag.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
void cip_set_error(const char *fmt, ...) {}
int main() {
    const int N = 8192;
    double *K, *W, *xm, *dinv;
    hipMalloc(&K, (size_t)N * 256 * 8); hipMalloc(&W, (size_t)N * 256 * 8); hipMalloc(&xm, 2048 * 8); hipMalloc(&dinv, 128 * 8);
    std::vector<double> h((size_t)N * 256);
    for (auto &v : h) v = (double)rand() / RAND_MAX - 0.5;
    hipMemcpy(K, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(xm, h.data(), 2048 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dinv, h.data(), 128 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rows : {8064, 4096, 1024, 128}) {
        cip_launch_trsm_subst(0, K + 128, N, rows, K, xm, dinv, W, N); hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) cip_launch_trsm_subst(0, K + 128, N, rows, K, xm, dinv, W, N);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("trsm rows=%5d : %.1f us\n", rows, ms * 1e3 / 20);
    }
    return 0;
}
