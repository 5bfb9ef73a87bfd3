// Diagonal-block kernel of the blocked LDL' (v2): in-LDS LDL' of one 128x128 block and the
// explicit inverse of its unit-lower factor, one 256-thread workgroup, micro-blocked by 16 with
// v_mfma_f64_16x16x4_f64.
//
// This is the serial link of the factorisation chain (N/128 of these run back to back), so it is
// organised around latency, not throughput:
//   for each 16-column micro-panel kb:
//     A. wave 0: LDL' of the 16x16 diagonal micro-block + inverse of its unit-lower factor, rows held
//        one per lane, pivot rows broadcast with v_readlane (no LDS round trip in the 16-step chain)
//     B. every wave, for its row tiles below: W = U * inv(L11)' as 4 MFMAs (the accumulator layout of
//        f64 16x16x4 is also its operand layout, so W feeds step C straight from registers); L = W D^-1
//     C. every wave: trailing tiles C[it][jt] -= W[it] L[jt]'  (4 MFMAs per 16x16 tile, tiles in LDS)
//   then X = inv(L) by block rows (X[it][jt] = -(sum_kt X[it][kt] L[kt][jt]) inv(L[jt][jt])), a chain of
//   MFMAs whose running tiles stay in registers.
// Row tiles are dealt to the 4 waves as pairs (w, 7-w), which balances the triangular work.
//
// LDS image: a[row + col*144] (pitch 144 doubles: fragment reads with rows on lanes 0-15 and k on the
// lane groups are bank-conflict free), the 8 micro inverses, and d / 1/d in the pitch padding:
// 160 KiB exactly (one workgroup per CU, which is all a serial kernel needs).
#include "cip_internal.h"

#define DP 144
#define XM_OFF (CIP_NB * DP)             // 8 x 256 doubles: xm[kb][k*16 + jj] = Xm_kb[jj][k]
#define DIAG2_LDS_BYTES ((CIP_NB * DP + 8 * 256) * 8)

__device__ __forceinline__ double rlane(double x, int lane) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    return r;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// step B for one owned row tile: returns -W in wneg (operand layout), writes L into the LDS image
__device__ __forceinline__ void diag_step_b(double *a, int it, int c, int l15, int g, const double (&xa)[4],
                                            const double (&di4)[4], double (&wneg)[4]) {
    double *p = a + (it * 16 + l15) + (c + g) * DP;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = MFMA(xa[s], p[4 * s * DP], acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        p[4 * q * DP] = acc[q] * di4[q];
        wneg[q] = -acc[q];
    }
}
// step C for one tile (it, jt)
__device__ __forceinline__ void diag_step_c(double *a, int it, int jt, int c, int l15, int g, const double (&wneg)[4]) {
    double *cp = a + (it * 16 + l15) + (jt * 16 + g) * DP;
    const double *lp = a + (jt * 16 + l15) + (c + g) * DP;
    v4d acc = (v4d){cp[0], cp[4 * DP], cp[8 * DP], cp[12 * DP]};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = MFMA(lp[4 * s * DP], wneg[s], acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) cp[4 * q * DP] = acc[q];
}

// X block row `it` (runtime, wave-uniform): tiles kept in registers, statically indexed
__device__ __forceinline__ void diag_inverse_row(double *a, const double *xm, int it, int l15, int g) {
    double XT[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) XT[t][s] = 0.0;
    // X[it][it] = Xm[it]: element (row l15, col g+4s)
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t == it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) XT[t][s] = xm[t * 256 + (g + 4 * s) * 16 + l15];
        }
#pragma unroll
    for (int jt = 6; jt >= 0; --jt) {
        if (jt < it) {
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kt = 7; kt >= 1; --kt) {
                if (kt > jt && kt <= it) {
                    // Aop[cjt][k] = L[kt*16 + k][jt*16 + cjt]
                    const double *lp = a + (kt * 16 + g) + (jt * 16 + l15) * DP;
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = MFMA(lp[4 * s], XT[kt][s], acc);
                }
            }
            // X[it][jt] = -S * Xm[jt]:  Aop[c'][k] = Xm[jt][k][c'] = xm[jt][c'*16 + k]
            const double *xp = xm + jt * 256 + l15 * 16 + g;
            v4d r = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) r = MFMA(xp[4 * s], acc[s], r);
#pragma unroll
            for (int s = 0; s < 4; ++s) XT[jt][s] = -r[s];
        }
    }
    // park the finished block row in the (now free) upper triangle: X[r][cc] (r > cc) -> a[cc + r*DP]
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) a[(t * 16 + g + 4 * s) + (it * 16 + l15) * DP] = XT[t][s];
        }
}

__global__ __launch_bounds__(256) void k_ldlt_diag128_v2(double *Kb, long ld, double *Linv, double *LinvT,
                                                          double *dvec, double *dinv, int *info, int col0) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *a = sm;
    double *xm = sm + XM_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;

    for (int e = tid; e < CIP_NB * CIP_NB; e += 256) {
        const int i = e & 127, j = e >> 7;
        a[i + j * DP] = (i >= j) ? Kb[i + (long)j * ld] : 0.0;
    }
    __syncthreads();

    const int tA = wave, tB = 7 - wave;          // owned row tiles
    for (int kb = 0; kb < 8; ++kb) {
        const int c = kb * 16;
        // ------------------------------------------------------------ A: 16x16 micro-block on wave 0
        if (wave == 0) {
            double u[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) u[jj] = a[(c + l15) + (c + jj) * DP];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double d = rlane(u[j], j);
                if (lane == 0 && !(fabs(d) > 0.0 && fabs(d) < 1.7e308)) atomicCAS(info, 0, col0 + c + j + 1);
                const double di = fast_rcp(d);
                if (lane == j) {
                    a[128 + (c + j) * DP] = d;
                    a[129 + (c + j) * DP] = di;
                }
                const double wi = u[j];
#pragma unroll
                for (int jj = j + 1; jj < 16; ++jj) u[jj] -= wi * (rlane(u[j], jj) * di);
                u[j] = (l15 == j) ? d : wi * di;          // column j final: l_ij (i > j), d on the diagonal
            }
            // micro inverse: lane cc owns column cc of X = inv(L11):  x[r] = [r == cc] - sum_{k<r} L[r][k] x[k]
            double x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double s = (l15 == r) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < r; ++k) s -= rlane(u[k], r) * x[k];
                x[r] = s;
            }
            if (lane < 16) {
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    if (jj <= l15) a[(c + l15) + (c + jj) * DP] = u[jj];
                    xm[kb * 256 + l15 * 16 + jj] = x[jj];            // xm[k = cc][jj = r] = X[r][cc]
                }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ B: panel rows below
        double wA[4] = {0, 0, 0, 0}, wB[4] = {0, 0, 0, 0};
        const bool hasA = tA > kb, hasB = tB > kb;
        if (hasA || hasB) {
            double xa[4], di4[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                xa[s] = xm[kb * 256 + (g + 4 * s) * 16 + l15];       // Aop[jj = l15][k = g + 4s]
                di4[s] = a[129 + (c + g + 4 * s) * DP];
            }
            if (hasA) diag_step_b(a, tA, c, l15, g, xa, di4, wA);
            if (hasB) diag_step_b(a, tB, c, l15, g, xa, di4, wB);
        }
        __syncthreads();
        // ------------------------------------------------------------ C: trailing tiles
        if (hasA)
            for (int jt = kb + 1; jt <= tA; ++jt) diag_step_c(a, tA, jt, c, l15, g, wA);
        if (hasB)
            for (int jt = kb + 1; jt <= tB; ++jt) diag_step_c(a, tB, jt, c, l15, g, wB);
        __syncthreads();
    }

    // ---- L, d out
    for (int e = tid; e < CIP_NB * CIP_NB; e += 256) {
        const int r = e & 127, cc = e >> 7;
        if (r >= cc) Kb[r + (long)cc * ld] = a[r + cc * DP];
    }
    if (tid < CIP_NB) {
        dvec[tid] = a[128 + tid * DP];
        dinv[tid] = a[129 + tid * DP];
    }
    __syncthreads();          // everyone has read the diagonal before the upper triangle is reused

    // ---- X = inv(L), block rows (w, 7-w); results parked in the upper triangle of the LDS image
    diag_inverse_row(a, xm, tA, l15, g);
    diag_inverse_row(a, xm, tB, l15, g);
    __syncthreads();
    for (int e = tid; e < CIP_NB * CIP_NB; e += 256) {
        const int r = e & 127, cc = e >> 7;
        const int tr = r >> 4, tc = cc >> 4;
        double x, xt;
        if (tr == tc) {
            x = xm[tr * 256 + (cc & 15) * 16 + (r & 15)];            // Xm[r][cc] (zero above the diagonal)
            xt = xm[tr * 256 + (r & 15) * 16 + (cc & 15)];           // Xm[cc][r]
        } else {
            x = (tr > tc) ? a[cc + r * DP] : 0.0;                    // X[r][cc]
            xt = (tc > tr) ? a[r + cc * DP] : 0.0;                   // X[cc][r]
        }
        Linv[r + cc * CIP_NB] = x;
        LinvT[r + cc * CIP_NB] = xt;
    }
}

int cip_launch_diag_v2(hipStream_t s, double *Kb, long ld, double *Linv, double *LinvT, double *dvec, double *dinv,
                       int *info, int col0) {
    static bool attr_set = false;
    if (!attr_set) {
        CIP_HIP_CHECK(hipFuncSetAttribute((const void *)k_ldlt_diag128_v2, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          DIAG2_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_ldlt_diag128_v2, dim3(1), dim3(256), DIAG2_LDS_BYTES, s, Kb, ld, Linv, LinvT, dvec, dinv, info, col0);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
