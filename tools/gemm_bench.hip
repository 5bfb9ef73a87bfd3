// Micro-benchmark of the fp64 MFMA GEMM (development tool): lower-triangular trailing update shapes.
#include "../conicip.jl_amd/csrc/gemm_f64.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
void cip_set_error(const char *fmt, ...) {}
int main(int argc, char **argv) {
    const int Nmax = 8192, Kmax = 512;
    double *W, *L, *C;
    hipMalloc(&W, (size_t)Nmax * Kmax * 8); hipMalloc(&L, (size_t)Nmax * Kmax * 8); hipMalloc(&C, (size_t)Nmax * Nmax * 8);
    std::vector<double> h((size_t)Nmax * Kmax);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
    hipMemcpy(W, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
    hipMemcpy(L, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(C, 0, (size_t)Nmax * Nmax * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int dbg = argc > 1 ? atoi(argv[1]) : 0;
    printf("dbg=%d\n", dbg);
    const int shapes[][2] = {{1024, 512}, {2048, 512}, {3072, 512}, {4096, 512}, {5120, 512}, {6144, 512}, {7168, 512}, {8192, 512}, {8192, 256}, {4096, 256}, {2048, 256}};
    for (auto &sh : shapes) {
        const int r = sh[0], K = sh[1];
        GemmArgs g = {};
        unsigned *ctr = nullptr; if (dbg & 1) { static unsigned *c0 = nullptr; if (!c0) hipMalloc(&c0, 256); ctr = c0; }
        g.queue_counter = ctr; if (ctr) hipMemsetAsync(ctr, 0, 4, 0);
        g.A = W; g.lda = Nmax; g.B = L; g.ldb = Nmax; g.C = C; g.ldc = Nmax; g.M = r; g.N = r; g.K = K; g.alpha = -1.0; g.lower = 1;
        for (int w = 0; w < 2; ++w) cip_launch_gemm(0, EPI_ACCUM, g);
        hipDeviceSynchronize();
        const int reps = 10;
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) cip_launch_gemm(0, EPI_ACCUM, g);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double alg = (double)r * (r + 1) * K;
        printf("r=%5d K=%3d : %8.1f us  %.1f TF (algorithmic)\n", r, K, ms * 1e3 / reps, alg / (ms / reps) / 1e9);
    }
    // strip shape: M = r, N = 128, K = 128 (full)
    for (int r : {8192, 4096}) {
        GemmArgs g = {};
        g.A = W; g.lda = Nmax; g.B = L; g.ldb = Nmax; g.C = C; g.ldc = Nmax; g.M = r; g.N = 128; g.K = 128; g.alpha = -1.0; g.lower = 0;
        cip_launch_gemm(0, EPI_ACCUM, g); hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) cip_launch_gemm(0, EPI_ACCUM, g);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("strip r=%5d : %8.1f us  %.1f TF\n", r, ms * 100, 2.0 * r * 128 * 128 / (ms / 10) / 1e9);
    }
    return 0;
}
