// fp64 MFMA issue-rate ceiling on this device (no memory traffic): waves x accumulators sweep.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_peak(double *out, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_per_cu, int threads) {
    double *out; hipMalloc(&out, 8 * 1024 * 4096);
    int iters = 20000 / NACC * 4;
    int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_peak<NACC>, dim3(grid), dim3(threads), 0, 0, out, 10, 1.0, 2.0);
    hipDeviceSynchronize();
    hipEventRecord(e0); 
    hipLaunchKernelGGL(k_peak<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * (threads / 64) * iters * NACC * 2048.0;
    printf("NACC=%2d wg/cu=%d threads=%d : %.1f TFLOP/s (%.2f ms)\n", NACC, wg_per_cu, threads, flops / ms / 1e9, ms);
    hipFree(out);
}
int main() {
    run<1>(1, 256); run<2>(1, 256); run<4>(1, 256); run<16>(1, 256); run<16>(2, 256); run<4>(2, 256); run<4>(4, 256);
    return 0;
}
