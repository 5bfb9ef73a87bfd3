// Micro-benchmark of the diagonal-block kernel (development tool, not part of the library).
// Build variants with -DDIAG_SKIP=<mask>: 1 = skip micro inverse in step A, 2 = skip X phase,
// 4 = skip steps B/C, 8 = skip step A factor loop, 16 = skip global load/store loops
#include "../conicip.jl_amd/csrc/diag.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
void cip_set_error(const char *fmt, ...) {}
thread_local CipGraphBuilder *cip_tl_builder = nullptr;
thread_local CipBatchCtx cip_tl_bz = {1, 0, 1ull, nullptr, nullptr};
int main() {
    const int N = 128;
    std::vector<double> K(N * N);
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) K[i + j * N] = (i == j ? 4.0 + 0.01 * i : 0.0) + 1.0 / (1.0 + abs(i - j));
    if (getenv("DIAG_BENCH_MATRIX")) {          // 1: negative definite; 2: quasi-definite [-(SPD) C; C' SPD] (the full 3x3 KKT shape)
        const int m = atoi(getenv("DIAG_BENCH_MATRIX"));
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i) {
                if (m == 1) K[i + j * N] = -K[i + j * N];
                else if (i < 44 && j < 44) K[i + j * N] = -K[i + j * N];
                else if ((i < 44) != (j < 44)) K[i + j * N] = 0.3 * sin(1.0 + i * 0.37 + j * 0.91 + 0.01 * i * j) * 1.0;
            }
        for (int j = 0; j < N; ++j) for (int i = 0; i < j; ++i) K[i + j * N] = K[j + i * N];      // symmetric
    }
    double *dK, *dLi, *dLt, *dd, *ddi, *dK0; int *dinfo;
    hipMalloc(&dK, N * N * 8); hipMalloc(&dK0, N * N * 8); hipMalloc(&dLi, N * N * 8); hipMalloc(&dLt, N * N * 8);
    hipMalloc(&dd, N * 8); hipMalloc(&ddi, N * 8); hipMalloc(&dinfo, 4);
    hipMemcpy(dK0, K.data(), N * N * 8, hipMemcpyHostToDevice);
    hipMemset(dinfo, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    for (int w = 0; w < 3; ++w) { hipMemcpy(dK, dK0, N * N * 8, hipMemcpyDeviceToDevice); cip_launch_diag_v2(0, dK, N, dLi, dd, ddi, dinfo, 0, PivotSigns{-1, 0, 0}); }
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) cip_launch_diag_v2(0, dK, N, dLi, dd, ddi, dinfo, 0, PivotSigns{-1, 0, 0});
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    {   // correctness of one factorisation against a host LDL' of the same block
        hipMemcpy(dK, dK0, N * N * 8, hipMemcpyDeviceToDevice);
        cip_launch_diag_v2(0, dK, N, dLi, dd, ddi, dinfo, 0, PivotSigns{-1, 0, 0});
        std::vector<double> G(N * N), dv(N), H = K;
        hipMemcpy(G.data(), dK, N * N * 8, hipMemcpyDeviceToHost);
        hipMemcpy(dv.data(), dd, N * 8, hipMemcpyDeviceToHost);
        for (int j = 0; j < N; ++j) {
            const double d = H[j + j * N];
            for (int i = j + 1; i < N; ++i) H[i + j * N] /= d;
            for (int c = j + 1; c < N; ++c) for (int i = c; i < N; ++i) H[i + c * N] -= H[i + j * N] * d * H[c + j * N];
        }
        double e = 0, ed = 0;
        for (int j = 0; j < N; ++j) { ed = fmax(ed, fabs(dv[j] - H[j + j * N]) / fabs(H[j + j * N])); for (int i = j + 1; i < N; ++i) e = fmax(e, fabs(G[i + j * N] - H[i + j * N])); }
        printf("max |L - L_host| %.3e, max rel |d - d_host| %.3e\n", e, ed);
        if (getenv("DIAG_BENCH_DUMP")) {        // factor, d and the micro inverses, raw: `cmp` two builds for bit-identity
            std::vector<double> X(8 * 256);
            hipMemcpy(X.data(), dLi, 8 * 256 * 8, hipMemcpyDeviceToHost);
            for (int j = 0; j < N; ++j) for (int i = 0; i < j; ++i) G[i + j * N] = 0.0;      // the strictly upper part is not output
            FILE *f = fopen(getenv("DIAG_BENCH_DUMP"), "wb");
            fwrite(G.data(), 8, N * N, f); fwrite(dv.data(), 8, N, f); fwrite(X.data(), 8, 8 * 256, f);
            fclose(f);
        }
    }
#ifdef DIAG_TIMING
    {
        long t[64];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(g_diag_t), sizeof(t));
        printf("s_memtime ticks x10 per micro-panel: wave 0: B, wait, C11 + A | helper (wave 1): B, wait at the B barrier, panel stores, trailing tiles | iteration\n");
        for (int kb = 0; kb < 7; ++kb) {
            const long *r = t + kb * 8;
            printf("kb %d  w0 %5ld %5ld %6ld | w1 %5ld %5ld %5ld %6ld | iter %6ld\n", kb, (r[1] - r[0]) * 10, (r[2] - r[1]) * 10, (r[3] - r[2]) * 10,
                   (r[5] - r[4]) * 10, (r[2] - r[5]) * 10, (r[6] - r[2]) * 10, (r[7] - r[6]) * 10, kb < 6 ? (t[(kb + 1) * 8] - r[0]) * 10 : 0L);
        }
    }
#endif
    printf("DIAG_SKIP=%d avg %.2f us per launch\n",
#ifdef DIAG_SKIP
           DIAG_SKIP,
#else
           0,
#endif
           ms * 1e3 / reps);
    return 0;
}
