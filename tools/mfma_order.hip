// Development probe: is v_mfma_f64_16x16x4_f64 bit-identical to a chain of individually rounded FMAs over k, and in which
// order?  (If acc' = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, acc)))) exactly, a rank-4 update of the diagonal kernel's
// 16x16 micro-block can replace four per-pivot FMA updates without changing a bit.)
// Convention of csrc/diag.hip: MFMA(P, R, acc): acc[q] @ lane (l15, g)  +=  sum_kk R @ lane (l15, kk) * P @ lane (g + 4q, kk).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k_mfma(const double *P, const double *R, const double *C, double *out) {
    const int lane = threadIdx.x, l15 = lane & 15, g = lane >> 4;
    v4d acc;
    for (int q = 0; q < 4; ++q) acc[q] = C[l15 + (g + 4 * q) * 16];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(P[l15 + g * 16], R[l15 + g * 16], acc, 0, 0, 0);
    for (int q = 0; q < 4; ++q) out[l15 + (g + 4 * q) * 16] = acc[q];
}
int main() {
    const int T = 2000;
    double *dP, *dR, *dC, *dO;
    hipMalloc(&dP, 64 * 8); hipMalloc(&dR, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dO, 256 * 8);
    long bad[6] = {0, 0, 0, 0, 0, 0}, total = 0;
    srand(1);
    auto rnd = [] { const double m = (rand() / (double)RAND_MAX - 0.5) * 2.0; return ldexp(m, rand() % 24 - 12); };
    for (int t = 0; t < T; ++t) {
        double P[64], R[64], C[256], O[256];
        for (int i = 0; i < 64; ++i) { P[i] = rnd(); R[i] = rnd(); }
        for (int i = 0; i < 256; ++i) C[i] = rnd();
        hipMemcpy(dP, P, sizeof(P), hipMemcpyHostToDevice); hipMemcpy(dR, R, sizeof(R), hipMemcpyHostToDevice);
        hipMemcpy(dC, C, sizeof(C), hipMemcpyHostToDevice);
        k_mfma<<<1, 64>>>(dP, dR, dC, dO);
        hipMemcpy(O, dO, sizeof(O), hipMemcpyDeviceToHost);
        for (int col = 0; col < 16; ++col)
            for (int row = 0; row < 16; ++row) {
                double a[4], b[4];
                for (int kk = 0; kk < 4; ++kk) { a[kk] = R[row + kk * 16]; b[kk] = P[col + kk * 16]; }
                const double c = C[row + col * 16], o = O[row + col * 16];
                double r0 = c; for (int kk = 0; kk < 4; ++kk) r0 = fma(a[kk], b[kk], r0);                 // forward chain from c
                double r1 = c; for (int kk = 3; kk >= 0; --kk) r1 = fma(a[kk], b[kk], r1);                // backward chain
                double r2 = fma(a[3], b[3], fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0]))) + c;          // products first, c last
                double r3 = fma(a[0], b[0], c) ; r3 = fma(a[1], b[1], r3); double r3b = fma(a[3], b[3], a[2] * b[2]); r3 += r3b;  // pairwise
                long double e = c; for (int kk = 0; kk < 4; ++kk) e += (long double)a[kk] * b[kk]; double r4 = (double)e;        // ~exact then rounded
                double r5 = c + (a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]);
                bad[0] += memcmp(&o, &r0, 8) != 0; bad[1] += memcmp(&o, &r1, 8) != 0; bad[2] += memcmp(&o, &r2, 8) != 0;
                bad[3] += memcmp(&o, &r3, 8) != 0; bad[4] += memcmp(&o, &r4, 8) != 0; bad[5] += memcmp(&o, &r5, 8) != 0;
                ++total;
            }
    }
    const char *nm[] = {"forward FMA chain from c (k = 0,1,2,3)", "backward FMA chain from c", "products first, c last", "pairwise", "exact sum rounded once", "plain sum"};
    for (int i = 0; i < 6; ++i) printf("%-42s mismatches %ld of %ld\n", nm[i], bad[i], total);
    return 0;
}
