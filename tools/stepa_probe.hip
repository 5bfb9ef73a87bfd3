// Development probe: shader clocks per call of the diagonal kernel's step A (diag.hip: diag_step_a, one wave, 16 pivots of a
// 16x16 micro-block + its inverse) ALONE on a CU, and of a few dependent-instruction chains it is made of.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DDIAG_STEP_A_REF] -o tools/stepa_probe tools/stepa_probe.hip
#include "../conicip.jl_amd/csrc/diag.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
void cip_set_error(const char *fmt, ...) {}
thread_local CipGraphBuilder *cip_tl_builder = nullptr;
thread_local CipBatchCtx cip_tl_bz = {1, 0, 1ull, nullptr, nullptr};

__global__ __launch_bounds__(512) void k_stepa(const double *tile, long *ticks, double *out, int *info, int reps) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *a = sm, *xm = sm + XM_OFF;
    const int lane = threadIdx.x;
    long tot = 0;
    for (int r = 0; r < reps; ++r) {
        for (int e = lane; e < 256; e += 64) a[(e & 15) + (e >> 4) * DP] = tile[e];
        __syncthreads();
        const long t0 = __builtin_amdgcn_s_memtime();
        diag_step_a(a, xm, 0, lane, info, 0, PivotSigns{-1, 0, 0});
        __builtin_amdgcn_s_waitcnt(0);
        const long t1 = __builtin_amdgcn_s_memtime();
#ifdef STEPA_TIMING
        if (lane == 0) { g_stepa_t[18] = t1; g_stepa_t[19] = t0; }
#endif
        tot += t1 - t0;
        __syncthreads();
    }
    if (lane == 0) ticks[0] = tot / reps;
    for (int e = lane; e < 256; e += 64) {
        out[e] = ((e & 15) >= (e >> 4)) ? a[(e & 15) + (e >> 4) * DP] : 0.0;      // L / d (the strictly upper part is not output)
        out[256 + e] = xm[e];                                                       // micro inverse
    }
    if (lane < 16) { out[512 + lane] = a[128 + lane * DP]; out[528 + lane] = a[129 + lane * DP]; }   // d, 1/d
}

template <int MODE>
__global__ void k_chain(double *out, long *ticks, double seed) {
    double x = seed + threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    const int addr = ((threadIdx.x + 1) & 63) * 4;
    const long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 1024; ++it) {
        if (MODE == 0) x = fma(x, 0.999, 1e-3);
        if (MODE == 1) x = __builtin_amdgcn_rcp(x) + 1.0;
        if (MODE == 2) x = fast_rcp(x) + 1.0;
        if (MODE == 3) { const double d = rlane(x, it & 63); x = fma(d, 1e-9, 1.0); }                         // readlane -> fma
        if (MODE == 4) x = bperm_d(x, addr) + 1e-3;
        if (MODE == 5) { const double d = rlane(x, it & 63); const double t = y * fast_rcp(d); x = fma(-t, 0.5, x) + 1.0; }   // the pivot chain
        if (MODE == 6) { const double d = rlane(x, it & 63); const double w = bperm_d(x, addr); const double t = w * fast_rcp(d); x = fma(-t, 0.5, x) + 1.0; }
        if (MODE == 7) { v4d acc = {x, x, x, x}; acc = MFMA(y, x, acc); x = acc[0] * 1e-9 + 1.0; }             // MFMA -> VALU round trip
        if (MODE == 8) x = (threadIdx.x & 1) ? fma(x, 0.999, 1e-3) : x;
        if (MODE == 9) { const double d = rlane(x, it & 63); const double t = y / d; x = fma(-t, 0.5, x) + 1.0; }  // IEEE division instead
    }
    const long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + y;
    if (threadIdx.x == 0) ticks[1 + MODE] = t1 - t0;
}

int main() {
    std::vector<double> T(256);
    for (int j = 0; j < 16; ++j) for (int i = 0; i < 16; ++i) T[i + j * 16] = (i >= j) ? ((i == j ? 4.0 + 0.1 * i : 0.0) + 1.0 / (1.0 + abs(i - j))) : 0.0;
    double *dT, *dO; long *dt; int *dinfo;
    hipMalloc(&dT, 256 * 8); hipMalloc(&dO, 544 * 8); hipMalloc(&dt, 16 * 8); hipMalloc(&dinfo, 64);
    hipMemset(dinfo, 0, 64); hipMemset(dt, 0, 128);
    hipMemcpy(dT, T.data(), 256 * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_stepa, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG2_LDS_BYTES);
    for (int rep = 0; rep < 2; ++rep) {
        k_stepa<<<1, 64, DIAG2_LDS_BYTES>>>(dT, dt, dO, dinfo, 200);
        k_chain<0><<<1, 64>>>(dO, dt, 1.5); k_chain<1><<<1, 64>>>(dO, dt, 1.5); k_chain<2><<<1, 64>>>(dO, dt, 1.5); k_chain<3><<<1, 64>>>(dO, dt, 1.5);
        k_chain<4><<<1, 64>>>(dO, dt, 1.5); k_chain<5><<<1, 64>>>(dO, dt, 1.5); k_chain<6><<<1, 64>>>(dO, dt, 1.5); k_chain<7><<<1, 64>>>(dO, dt, 1.5);
        k_chain<8><<<1, 64>>>(dO, dt, 1.5); k_chain<9><<<1, 64>>>(dO, dt, 1.5);
        hipDeviceSynchronize();
    }
    long h[16]; hipMemcpy(h, dt, sizeof(h), hipMemcpyDeviceToHost);
    {
        hipFuncSetAttribute((const void *)k_stepa, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG2_LDS_BYTES);
        k_stepa<<<1, 64, DIAG2_LDS_BYTES>>>(dT, dt + 15, dO, dinfo, 1);
        std::vector<double> O(544);
        hipMemcpy(O.data(), dO, 544 * 8, hipMemcpyDeviceToHost);
        unsigned long long hsh = 1469598103934665603ull;
        const unsigned char *b = (const unsigned char *)O.data();
        for (size_t i = 0; i < 544 * 8; ++i) { hsh ^= b[i]; hsh *= 1099511628211ull; }
        if (getenv("STEPA_DUMP")) { FILE *f = fopen(getenv("STEPA_DUMP"), "wb"); fwrite(O.data(), 8, 544, f); fclose(f); }
        printf("output checksum (L, d, 1/d, micro inverse) %016llx   L[5][2] = %.17g  X[7][3] = %.17g\n", hsh, O[5 + 2 * 16], O[256 + 3 * 16 + 7]);
    }
#ifdef STEPA_TIMING
    {
        long t[32]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_stepa_t), sizeof(t));
        printf("stamps (clocks from entry): load %ld |", t[0] - t[19]);
        for (int jb = 0; jb < 4; ++jb) printf(" r%d: start %ld block done %ld operands read %ld mfmas issued %ld |", jb, t[1 + 4 * jb] - t[19], t[2 + 4 * jb] - t[19], t[3 + 4 * jb] - t[19], t[4 + 4 * jb] - t[19]);
        printf(" stores %ld end %ld\n", t[17] - t[19], t[18] - t[19]);
    }
#endif
    printf("step A (16 pivots + micro inverse), one wave alone: %ld clocks per call = %.1f per pivot\n", h[0], h[0] / 16.0);
    const char *nm[] = {"dependent fp64 FMA", "v_rcp_f64 + add", "fast_rcp (rcp + 2 Newton) + add", "readlane -> fma", "ds_bpermute 64-bit + add",
                        "readlane, fast_rcp, mul, fma, add", "same + one bpermute feeding the mul", "MFMA 16x16x4 -> VALU -> MFMA", "FMA + select",
                        "readlane, IEEE division, fma, add"};
    for (int m = 0; m < 10; ++m) printf("%-45s %7.1f clocks per iteration\n", nm[m], h[1 + m] / 1024.0);
    return 0;
}
