// Latency probe (development tool): how many shader cycles the dependent operations of the diagonal kernel's pivot step
// take on ONE wave -- v_rcp_f64, dependent fp64 FMAs, v_readlane + rcp, a ds_bpermute round trip.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double x, int lane) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, lane); hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bp(double x, int addr) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_ds_bpermute(addr, lo); hi = __builtin_amdgcn_ds_bpermute(addr, hi);
    return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k_probe(double *out, long *ticks, double seed) {
    double x = seed + threadIdx.x * 1e-3, y = 1.0;
    const int addr = ((threadIdx.x + 1) & 63) * 4;
    const long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 1024; ++it) {
        if (MODE == 0) { x = fma(x, 0.999, 1e-3); }                                   // one dependent fp64 FMA
        if (MODE == 1) { x = __builtin_amdgcn_rcp(x) + 1.0; }                         // rcp + add
        if (MODE == 2) {                                                              // rcp + two Newton steps (fast_rcp) + use
            double r = __builtin_amdgcn_rcp(x); double e = fma(-x, r, 1.0); r = fma(r, e, r); e = fma(-x, r, 1.0); r = fma(r, e, r);
            x = r + 1.0;
        }
        if (MODE == 3) {                                                              // readlane -> fast_rcp -> mul -> fma (the pivot chain)
            const double d = rl(x, it & 63);
            double r = __builtin_amdgcn_rcp(d); double e = fma(-d, r, 1.0); r = fma(r, e, r); e = fma(-d, r, 1.0); r = fma(r, e, r);
            const double t = y * r;
            x = fma(-t, 0.5, x) + 1.0;
        }
        if (MODE == 4) { x = bp(x, addr) + 1e-3; }                                    // one ds_bpermute round trip (two dwords)
        if (MODE == 5) {                                                              // the chain + 10 permutes issued beside it
            const double d = rl(x, it & 63);
            double p[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) p[q] = bp(x, addr + 64 * (q & 3));
            double r = __builtin_amdgcn_rcp(d); double e = fma(-d, r, 1.0); r = fma(r, e, r); e = fma(-d, r, 1.0); r = fma(r, e, r);
            const double t = p[0] * r;
            x = fma(-t, p[1], x) + 1.0 + 1e-9 * (p[2] + p[3] + p[4]);
        }
        if (MODE == 6) { x = (threadIdx.x & 1) ? fma(x, 0.999, 1e-3) : x; }            // FMA + select
    }
    const long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + y;
    if (threadIdx.x == 0) ticks[MODE] = t1 - t0;
}
int main() {
    double *out; long *ticks; hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        k_probe<0><<<1, 64>>>(out, ticks, 1.5); k_probe<1><<<1, 64>>>(out, ticks, 1.5); k_probe<2><<<1, 64>>>(out, ticks, 1.5);
        k_probe<3><<<1, 64>>>(out, ticks, 1.5); k_probe<4><<<1, 64>>>(out, ticks, 1.5); k_probe<5><<<1, 64>>>(out, ticks, 1.5);
        k_probe<6><<<1, 64>>>(out, ticks, 1.5);
        hipDeviceSynchronize();
    }
    long h[8]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    const char *name[] = {"dependent fp64 FMA", "rcp + add", "rcp + 2 Newton + add", "readlane, rcp + 2 Newton, mul, fma, add",
                          "ds_bpermute (64-bit) + add", "pivot chain + 10 permutes beside it", "FMA + select"};
    for (int m = 0; m < 7; ++m) printf("%-45s %7.1f ticks per iteration\n", name[m], h[m] / 1024.0);
    return 0;
}
