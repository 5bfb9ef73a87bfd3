// Micro-benchmark (development tool): the fp64 MFMA GEMM kernels at long K -- how far the main loop alone is from the
// 78.6 TFLOP/s peak, beside the short-K shapes of the trailing update (prologue / epilogue share).
#include "../conicip.jl_amd/csrc/gemm_f64.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
void cip_set_error(const char *fmt, ...) {}
thread_local CipGraphBuilder *cip_tl_builder = nullptr;
thread_local CipBatchCtx cip_tl_bz = {1, 0, 1ull, nullptr, nullptr};
int main(int argc, char **argv) {
    const int Nmax = 8192, Kmax = 8192;
    double *W, *L, *C;
    hipMalloc(&W, (size_t)Nmax * Kmax * 8); hipMalloc(&L, (size_t)Nmax * Kmax * 8); hipMalloc(&C, (size_t)Nmax * Nmax * 8);
    std::vector<double> h((size_t)Nmax * Kmax);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
    hipMemcpy(W, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(L, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(C, 0, (size_t)Nmax * Nmax * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int shapes[][3] = {{8192, 768, 1}, {8192, 2048, 1}, {8192, 8192, 1}, {4096, 768, 0}, {4096, 8192, 0}, {8192, 8192, 0}};
    for (auto &sh : shapes) {
        const int r = sh[0], K = sh[1], lower = sh[2];
        GemmArgs g = {};
        g.A = W; g.lda = Nmax; g.B = L; g.ldb = Nmax; g.C = C; g.ldc = Nmax; g.M = r; g.N = r; g.K = K; g.alpha = -1.0; g.lower = lower;
        for (int w = 0; w < 2; ++w) cip_launch_gemm(0, EPI_ACCUM, g);
        hipDeviceSynchronize();
        const int reps = 5;
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) cip_launch_gemm(0, EPI_ACCUM, g);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double alg = lower ? (double)r * (r + 1) * K : 2.0 * r * r * K;
        printf("r=%5d K=%4d lower=%d : %9.1f us  %.1f TF\n", r, K, lower, ms * 1e3 / reps, alg / (ms / reps) / 1e9);
    }
    return 0;
}
