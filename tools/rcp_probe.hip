// Development probe: accuracy of v_rcp_f64 and of rcp + 1 / + 2 Newton steps against the correctly rounded reciprocal
// (host long double), in units of the result's ulp.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/rcp_probe tools/rcp_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    r0[i] = r;
    double e = fma(-d, r, 1.0); r = fma(r, e, r);
    r1[i] = r;
    e = fma(-d, r, 1.0); r = fma(r, e, r);
    r2[i] = r;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;             // mantissa in [1, 2)
        const int e = (int)((s >> 3) % 200) - 100;
        x[i] = std::ldexp(((s >> 1) & 1) ? -m : m, e);
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    std::vector<double> r0(n), r1(n), r2(n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0; long n1 = 0, n2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)x[i];
        const double ex = (double)t;
        const double ulp = std::fabs(std::nextafter(ex, INFINITY) - ex);
        const double e0 = (double)fabsl((long double)r0[i] - t) / ulp, e1 = (double)fabsl((long double)r1[i] - t) / ulp, e2 = (double)fabsl((long double)r2[i] - t) / ulp;
        if (e0 > m0) m0 = e0; if (e1 > m1) m1 = e1; if (e2 > m2) m2 = e2;
        n1 += r1[i] != ex; n2 += r2[i] != ex;
    }
    printf("max error in ulp: v_rcp_f64 %.3g, + 1 Newton step %.4f (%ld of %d not correctly rounded), + 2 steps %.4f (%ld not correctly rounded)\n", m0, m1, n1, n, m2, n2);
    return 0;
}
