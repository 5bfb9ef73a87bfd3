// S-cone (semidefinite) kernels: device counterparts of
//   mat / vecm                      src/ConicIP.jl:85-151
//   nestod_sdc                      src/ConicIP.jl:196-210
//   VecCongurance apply/inv/adjoint src/ConicIP.jl:35-40, :69
//   xsdc! / dsdc! (lyap)            src/ConicIP.jl:347-360
//   maxstep_sdc                     src/ConicIP.jl:272-303
//
// One workgroup per cone (256 threads up to order 48, 1024 above; or per column of A for the Schur scaling), the ONE
// matrix a serial chain works on resident in LDS (pitch padded against bank conflicts) up to order 132, r x r GEMMs on
// v_mfma_f64_16x16x4_f64 with operands straight from L2:
//   nestod_sdc    two Choleskys, G = Lz' Ls, LEFT singular vectors of G by one-sided (Hestenes) Jacobi with the parallel
//                 round-robin ordering (r/2 disjoint rotations per round, columns in registers between the dot products
//                 and the rotation), R = Lz^-T U Lambda^1/2, Rinv = Lambda^-1/2 U' Lz'
//   maxstep_sdc   lambda_max(L^-1 D L^-T) (X = L L'): Cholesky, two triangular solves, Householder tridiagonalisation,
//                 multisection on the Sturm count (one shift per thread)
//   dsdc!         element-wise when the divisor is diagonal (every division of the interior-point loop), two-sided
//                 Jacobi with vectors otherwise (sd_jacobi_core below -- the only user left of the two-sided routine)
// Orders 133 .. 512 take the chip-wide path of sdp_large.hip for NT scaling, max-step, congruences, Jordan products and
// the Schur scaling; the launchers at the end of this file split the S cones of a problem between the two.
// R is determined only up to a signed permutation of its columns; F'F, F'(F x) and every norm the driver forms
// are invariant under it.
#include "cip_internal.h"
#include "../../include/cipkkt.h"
#include <math.h>

// Threads per workgroup: 1024 (16 waves) for cones of order > 48, 256 otherwise.  The serial chains below are
// instruction-issue bound with one wave per SIMD (a wave issues ~1 instruction per 4-5 cycles; measured 11 k cycles
// per triangular-solve step at r = 128 with 4 waves for ~2000 instructions per wave), so the large cones get four
// waves per SIMD.  Device code uses the launch's own block size.
#define SD_T ((int)blockDim.x)
#define SD_TMAX 1024
#define SD_NW (SD_T / 64)
#ifdef SD_PROFILE
#include <stdio.h>
// development timers: s_memtime stamps collected in LDS, printed once at the end of the kernel
#define SD_TICK_INIT __shared__ long sd_ts[16]; int sd_nt = 0; \
    if (threadIdx.x == 0) sd_ts[0] = __builtin_amdgcn_s_memtime()
#define SD_TICK(name) do { __syncthreads(); ++sd_nt; if (threadIdx.x == 0) sd_ts[sd_nt] = __builtin_amdgcn_s_memtime(); } while (0)
#define SD_TICK_DUMP do { if (threadIdx.x == 0 && blockIdx.x == 0) { for (int i_ = 1; i_ <= sd_nt; ++i_) \
    printf("[sd] phase %d: %ld ticks\n", i_, (long)(sd_ts[i_] - sd_ts[i_ - 1])); printf("[sd] total %ld\n", (long)(sd_ts[sd_nt] - sd_ts[0])); } } while (0)
#else
#define SD_TICK(name)
#define SD_TICK_INIT
#define SD_TICK_DUMP
#endif
#define SQRT2 1.4142135623730951
#define SQRT1_2 0.7071067811865476

// Workgroup barrier.  lds_only: the data exchanged lives in LDS, so only lgkmcnt has to drain -- __syncthreads() also
// waits for every outstanding global access (vmcnt(0)), which would expose the latency of prefetched global loads at
// every step of the serial chains below.  The argument is a compile-time constant after inlining.
__device__ __forceinline__ void sd_sync(bool lds_only) {
    if (lds_only) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
}
// Newton-refined hardware reciprocal / reciprocal square root (v_rcp_f64, v_rsq_f64 give ~2^-26; two steps reach
// the last bits): a handful of instructions where the IEEE division / sqrt expansions cost 30-40 each.  Used where the
// value feeds a rotation or a multiplier of a serial chain that every lane recomputes (the chains are issue-bound).
__device__ __forceinline__ double sd_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double sd_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    return r;
}
// guarded load without a branch: the address is clamped into range (so the load can be issued unconditionally and
// the unrolled loops below stay straight-line code), the value is masked afterwards
__device__ __forceinline__ double sd_ld(const double *p, long stride, int i, int n) {
    const double x = p[(long)(i < n ? i : n - 1) * stride];
    return i < n ? x : 0.0;
}
__device__ __forceinline__ int vidx(int i, int j, int r) { return i * r - i * (i - 1) / 2 + (j - i); }   // i <= j

// X (r x r, col-major) = mat(x)
__device__ __forceinline__ void sd_mat(const double *x, long xs, double *X, int r, int ld = 0) {
    if (!ld) ld = r;
    for (int e = threadIdx.x; e < r * r; e += SD_T) {
        const int i = e % r, j = e / r;
        const int a = i < j ? i : j, b = i < j ? j : i;
        const double v = x[(long)vidx(a, b, r) * xs];
        X[i + j * ld] = (a == b) ? v : v * SQRT1_2;
    }
    __syncthreads();
}
// x = vecm(X) (upper triangle, as the reference), optionally scaled
__device__ void sd_vecm(const double *X, double *x, long xs, int r, double scale) {
    for (int e = threadIdx.x; e < r * r; e += SD_T) {
        const int i = e % r, j = e / r;
        if (i <= j) x[(long)vidx(i, j, r) * xs] = scale * ((i == j) ? X[e] : X[e] * SQRT2);
    }
    __syncthreads();
}
// C = op(A) * op(B), all r x r col-major (C must not alias A or B).  One workgroup, v_mfma_f64_16x16x4_f64 with
// the operands read straight from global memory (the matrices are L2-resident): each wave owns 32x32 super-tiles
// (2x2 MFMA tiles: two A and two B fragments per four MFMAs).  The MFMA is issued "transposed" (first operand =
// B columns, second = A rows) so that the accumulator's lane index runs along C's rows: contiguous 128-B stores.
// Any r: out-of-range rows / columns / k are fed as zeros.
__device__ void sd_gemm(double *C, const double *A, bool ta, const double *B, bool tb, int r) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nt = (r + 31) / 32;
    const long sa_i = ta ? r : 1, sa_k = ta ? 1 : r;          // opA[i][k] = A[i*sa_i + k*sa_k]
    const long sb_k = tb ? r : 1, sb_j = tb ? 1 : r;          // opB[k][j] = B[k*sb_k + j*sb_j]
    const int kfull = r & ~3;
    for (int t = wave; t < nt * nt; t += SD_T / 64) {
        const int i0 = (t % nt) * 32, j0 = (t / nt) * 32;
        const int ia = i0 + l15, ib = i0 + 16 + l15, ja = j0 + l15, jb = j0 + 16 + l15;
        const bool via = ia < r, vib = ib < r, vja = ja < r, vjb = jb < r;
        const double *pa0 = A + (via ? ia : 0) * sa_i + g * sa_k, *pa1 = A + (vib ? ib : 0) * sa_i + g * sa_k;
        const double *pb0 = B + (vja ? ja : 0) * sb_j + g * sb_k, *pb1 = B + (vjb ? jb : 0) * sb_j + g * sb_k;
        v4d acc00 = {0, 0, 0, 0}, acc01 = acc00, acc10 = acc00, acc11 = acc00;     // acc[x][y]: rows i0+16x, cols j0+16y
#pragma unroll 4
        for (int k0 = 0; k0 < kfull; k0 += 4) {
            double a0 = pa0[k0 * sa_k], a1 = pa1[k0 * sa_k], b0 = pb0[k0 * sb_k], b1 = pb1[k0 * sb_k];   // clamped rows: always valid
            a0 = via ? a0 : 0.0; a1 = vib ? a1 : 0.0; b0 = vja ? b0 : 0.0; b1 = vjb ? b1 : 0.0;
            acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc11, 0, 0, 0);
        }
        if (kfull < r) {
            const bool vk = kfull + g < r;
            const double a0 = (via && vk) ? pa0[kfull * sa_k] : 0.0, a1 = (vib && vk) ? pa1[kfull * sa_k] : 0.0;
            const double b0 = (vja && vk) ? pb0[kfull * sb_k] : 0.0, b1 = (vjb && vk) ? pb1[kfull * sb_k] : 0.0;
            acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc11, 0, 0, 0);
        }
        // D[x][y] reg q of lane (l15, g): C[i0 + 16x + l15][j0 + 16y + g + 4q]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = j0 + g + 4 * q, c1 = c0 + 16;
            if (via && c0 < r) C[ia + (long)c0 * r] = acc00[q];
            if (via && c1 < r) C[ia + (long)c1 * r] = acc01[q];
            if (vib && c0 < r) C[ib + (long)c0 * r] = acc10[q];
            if (vib && c1 < r) C[ib + (long)c1 * r] = acc11[q];
        }
    }
    __syncthreads();
}
// in-place lower Cholesky (strict upper zeroed); *flag <- (column+1) of a non-positive pivot (left untouched otherwise)
__device__ __forceinline__ void sd_chol(double *A, int r, int *flag, int ld = 0, bool lds = false) {
    if (!ld) ld = r;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = 0; j < r; ++j) {
        const double d = A[j + j * ld];
        if (!(d > 0.0)) { if (threadIdx.x == 0) *flag = j + 1; }
        const double l = sqrt(d);
        sd_sync(lds);
        for (int i = j + threadIdx.x; i < r; i += SD_T) A[i + j * ld] = (i == j) ? l : A[i + j * ld] / l;
        sd_sync(lds);
        for (int k = j + 1 + wave; k < r; k += SD_T / 64) {                 // one wave per trailing column
            const double akj = A[k + j * ld];
            for (int i = k + lane; i < r; i += 64) A[i + k * ld] -= A[i + j * ld] * akj;
        }
        sd_sync(lds);
    }
    for (int e = threadIdx.x; e < r * r; e += SD_T) if (e % r < e / r) A[e % r + (e / r) * ld] = 0.0;
    sd_sync(lds);
}
// X <- L^-T X  (L lower), column per thread
__device__ void sd_solve_LT(const double *L, double *X, int r) {
    for (int c = threadIdx.x; c < r; c += SD_T) {
        double *x = X + c * r;
        for (int i = r - 1; i >= 0; --i) {
            double s = x[i];
            for (int k = i + 1; k < r; ++k) s -= L[k + i * r] * x[k];
            x[i] = s / L[i + i * r];
        }
    }
    __syncthreads();
}
// Two-sided Jacobi, parallel (round-robin) ordering.  A (symmetric) is destroyed: eigenvalues end on its
// diagonal; V (may be NULL) receives the eigenvectors as columns: A_in = V diag V'.
__device__ void sd_jacobi_core(double *A, double *V, int r, double *sh /* >= 4*(r/2+1) + 8 doubles */) {
    const int tid = threadIdx.x;
    if (V) {
        for (int e = tid; e < r * r; e += SD_T) V[e] = (e % r == e / r) ? 1.0 : 0.0;
    }
    __syncthreads();
    const int m = (r + 1) & ~1;               // players of the tournament (one dummy when r is odd)
    const int np = m / 2;
    double *cs = sh, *sn = sh + np;
    int *pp = (int *)(sh + 2 * np), *qq = pp + np;
    double *red = sh + 4 * np;
    for (int sweep = 0; sweep < 30; ++sweep) {
        // convergence: off(A)^2 <= (1e-15)^2 * ||A||_F^2
        double off = 0.0, tot = 0.0;
        for (int e = tid; e < r * r; e += SD_T) {
            const double a = A[e];
            tot += a * a;
            if (e % r != e / r) off += a * a;
        }
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o); tot += __shfl_xor(tot, o); }
        __syncthreads();
        if ((tid & 63) == 0) { red[tid >> 6] = off; red[16 + (tid >> 6)] = tot; }
        __syncthreads();
        off = 0.0; tot = 0.0;
        for (int q = 0; q < SD_NW; ++q) { off += red[q]; tot += red[16 + q]; }
        if (off <= 1e-30 * tot || tot == 0.0) break;
        for (int t = 0; t < m - 1; ++t) {
            for (int k = tid; k < np; k += SD_T) {
                int p, q;
                if (k == 0) { p = m - 1; q = t; }
                else { p = (t + k) % (m - 1); q = (t - k + (m - 1)) % (m - 1); }
                if (p > q) { const int x = p; p = q; q = x; }
                double c = 1.0, s = 0.0;
                if (q < r) {
                    const double apq = A[p + q * r];
                    if (apq != 0.0) {
                        const double tau = (A[q + q * r] - A[p + p * r]) / (2.0 * apq);
                        const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + tt * tt);
                        s = tt * c;
                    }
                } else { p = -1; }
                cs[k] = c; sn[k] = s; pp[k] = p; qq[k] = q;
            }
            __syncthreads();
            // rows: A <- J' A
            for (int e = tid; e < np * r; e += SD_T) {
                const int k = e / r, j = e % r;
                const int p = pp[k], q = qq[k];
                if (p < 0) continue;
                const double c = cs[k], s = sn[k];
                const double ap = A[p + j * r], aq = A[q + j * r];
                A[p + j * r] = c * ap - s * aq;
                A[q + j * r] = s * ap + c * aq;
            }
            __syncthreads();
            // columns: A <- A J, V <- V J
            for (int e = tid; e < np * r; e += SD_T) {
                const int k = e / r, i = e % r;
                const int p = pp[k], q = qq[k];
                if (p < 0) continue;
                const double c = cs[k], s = sn[k];
                const double ap = A[i + p * r], aq = A[i + q * r];
                A[i + p * r] = c * ap - s * aq;
                A[i + q * r] = s * ap + c * aq;
                if (V) {
                    const double vp = V[i + p * r], vq = V[i + q * r];
                    V[i + p * r] = c * vp - s * vq;
                    V[i + q * r] = s * vp + c * vq;
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
}

// Jacobi with the matrices staged in LDS when they fit (`cap` doubles behind the rotation scratch; A alone when no
// vectors are wanted, A and V otherwise): the sweep is a chain of ~3 (r-1) barrier-separated passes per sweep,
// each a handful of dependent accesses per thread, so LDS latency instead of global-memory latency is a ~10x
// difference (r = 128, values only: 80 ms -> ~1 ms).
#define SD_SCRATCH(r) (4 * (((r) + 2) / 2 + 1) + 40)
#define SD_LDS_CAPMAX ((160 * 1024 - 1024) / 8)          // doubles of dynamic LDS a kernel may ask for
__device__ void sd_jacobi(double *A, double *V, int r, double *sh, int cap) {
    if ((V ? 2 : 1) * r * r > cap) { sd_jacobi_core(A, V, r, sh); return; }
    double *la = sh + SD_SCRATCH(r);
    double *lv = V ? la + r * r : nullptr;
    for (int e = threadIdx.x; e < r * r; e += SD_T) la[e] = A[e];
    __syncthreads();
    sd_jacobi_core(la, lv, r, sh);
    for (int e = threadIdx.x; e < r * r; e += SD_T) {
        A[e] = la[e];
        if (V) V[e] = lv[e];
    }
    __syncthreads();
}

// B <- L^-1 B  (L lower r x r pitch ldl, B r x r pitch ldb): right-looking by rows.  Lanes own rows, waves own
// columns: a lane fetches its entries of column j of L once per step (L may sit in global memory while B is in LDS)
// and reuses them for every column of B; the next step's entries are fetched before the barrier.
#define SD_RPL 8                                             // rows per lane kept in registers: r <= 512
__device__ __forceinline__ void sd_trsm_l(const double *L, double *B, int r, int ldl = 0, int ldb = 0, bool lds = false) {
    if (!ldl) ldl = r;
    if (!ldb) ldb = r;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double lcol[SD_RPL];
#pragma unroll
    for (int q = 0; q < SD_RPL; ++q) lcol[q] = sd_ld(L, 1, lane + 64 * q, r);      // column 0
    for (int j = 0; j < r; ++j) {
        double ljj = 0.0;
#pragma unroll
        for (int q = 0; q < SD_RPL; ++q) if (lane + 64 * q == j) ljj = lcol[q];
        ljj = __shfl(ljj, j & 63);                           // the diagonal entry sits in lane j % 64
        const double inv = sd_rcp(ljj);
        double lcur[SD_RPL];
#pragma unroll
        for (int q = 0; q < SD_RPL; ++q) lcur[q] = lcol[q];
        if (j + 1 < r) {
#pragma unroll
            for (int q = 0; q < SD_RPL; ++q) lcol[q] = sd_ld(L + (long)(j + 1) * ldl, 1, lane + 64 * q, r);
        }
        // four columns of B per pass: their LDS reads are issued together (the loop is latency-, not bandwidth-bound)
        for (int c0 = wave * 4; c0 < r; c0 += (SD_T / 64) * 4) {
            double bj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bj[u] = sd_ld(B + j, ldb, c0 + u, r) * inv;
#pragma unroll
            for (int q = 0; q < SD_RPL; ++q) {
                const int i = lane + 64 * q;
                if (i > j && i < r) {
                    double t[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) t[u] = sd_ld(B + i, ldb, c0 + u, r);
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (c0 + u < r) B[i + (long)(c0 + u) * ldb] = t[u] - lcur[q] * bj[u];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (c0 + u < r) B[j + (long)(c0 + u) * ldb] = bj[u];
            }
        }
        sd_sync(lds);
    }
}
// B <- L^-T B: the same from the last row upwards; row j of L' is column j of L, so a lane's multipliers L[j][i]
// (i < j) are a strided row of L: fetched once per step as well
__device__ __forceinline__ void sd_trsm_lt(const double *L, double *B, int r, int ldl = 0, int ldb = 0, bool lds = false) {
    if (!ldl) ldl = r;
    if (!ldb) ldb = r;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double lrow[SD_RPL];
#pragma unroll
    for (int q = 0; q < SD_RPL; ++q) lrow[q] = sd_ld(L + (r - 1), ldl, lane + 64 * q, r);
    for (int j = r - 1; j >= 0; --j) {
        double ljj = 0.0;
#pragma unroll
        for (int q = 0; q < SD_RPL; ++q) if (lane + 64 * q == j) ljj = lrow[q];
        ljj = __shfl(ljj, j & 63);
        const double inv = sd_rcp(ljj);
        double lcur[SD_RPL];
#pragma unroll
        for (int q = 0; q < SD_RPL; ++q) lcur[q] = lrow[q];
        if (j > 0) {
#pragma unroll
            for (int q = 0; q < SD_RPL; ++q) lrow[q] = sd_ld(L + (j - 1), ldl, lane + 64 * q, j);
        }
        // four columns of B per pass: their LDS reads are issued together (the loop is latency-, not bandwidth-bound)
        for (int c0 = wave * 4; c0 < r; c0 += (SD_T / 64) * 4) {
            double bj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bj[u] = sd_ld(B + j, ldb, c0 + u, r) * inv;
#pragma unroll
            for (int q = 0; q < SD_RPL; ++q) {
                const int i = lane + 64 * q;
                if (i < j) {
                    double t[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) t[u] = sd_ld(B + i, ldb, c0 + u, r);
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (c0 + u < r) B[i + (long)(c0 + u) * ldb] = t[u] - lcur[q] * bj[u];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (c0 + u < r) B[j + (long)(c0 + u) * ldb] = bj[u];
            }
        }
        sd_sync(lds);
    }
}
__device__ __forceinline__ void sd_transpose(double *A, int r, int ld = 0, bool lds = false) {
    if (!ld) ld = r;
    for (int e = threadIdx.x; e < r * r; e += SD_T) {
        const int i = e % r, j = e / r;
        if (i < j) { const double a = A[i + j * ld], b = A[j + i * ld]; A[i + j * ld] = b; A[j + i * ld] = a; }
    }
    sd_sync(lds);
}

// ---- extreme eigenvalue of a symmetric matrix: Householder tridiagonalisation + multisection on the Sturm count.
// maxstep_sdc needs only the largest eigenvalue of X^-1/2 D X^-1/2 (or the smallest of X) (:272-303): r^3 4/3 flops
// and r barrier-separated steps instead of ~10 Jacobi sweeps of 3 (r - 1) steps each.
#define SD_SCRATCH_TRI(r) (4 * (r) + SD_TMAX + 48)
__device__ __forceinline__ double sd_block_sum(double x, double *red) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    sd_sync(true);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    sd_sync(true);
    double sum = 0.0;
    for (int q = 0; q < SD_NW; ++q) sum += red[q];
    return sum;
}
// A (full symmetric storage, pitch ld, destroyed).  sc: SD_SCRATCH_TRI(r) doubles of LDS.  Result returned to all threads.
__device__ __forceinline__ double sd_extreme_eig(double *A, int r, int ld, double *sc, bool want_max, bool lds = false) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *dg = sc, *of = sc + r, *v = sc + 2 * r, *w = sc + 3 * r, *pb = sc + 4 * r, *red = sc + 4 * r + SD_TMAX;
    for (int k = 0; k + 1 < r; ++k) {
        const int m = r - k - 1;
        double *x = A + (k + 1) + (long)k * ld;                 // column k below the diagonal
        double *A22 = A + (k + 1) + (long)(k + 1) * ld;
        double part = 0.0;
        for (int i = tid; i < m; i += SD_T) part += x[i] * x[i];
        const double sigma = sd_block_sum(part, red);
        const double x0 = x[0];
        if (tid == 0) dg[k] = A[k + (long)k * ld];
        if (m == 1 || !(sigma - x0 * x0 > 0.0)) {               // already tridiagonal in this column (uniform branch)
            if (tid == 0) of[k] = x0;
            sd_sync(lds);
            continue;
        }
        const double alpha = -copysign(sqrt(sigma), x0);
        const double beta = 1.0 / (sigma - x0 * alpha);        // 2 / ||v||^2 with v = x - alpha e1
        for (int i = tid; i < m; i += SD_T) v[i] = x[i] - (i == 0 ? alpha : 0.0);
        if (tid == 0) of[k] = alpha;
        sd_sync(lds);
        // p = beta A22 v, as a sum of columns (rows on consecutive lanes); the column range is split over the
        // workgroup's spare threads and combined through pb
        int mp = 64;
        while (mp < m && mp < SD_T) mp *= 2;
        const int nparts = SD_T / mp, jpart = tid / mp;
        for (int i0 = 0; i0 < m; i0 += mp) {
            const int i = i0 + tid % mp;
            double acc = 0.0;
            if (i < m) {
                for (int j0 = jpart; j0 < m; j0 += 8 * nparts) {
                    double av[8], vv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int j = j0 + u * nparts; av[u] = sd_ld(A22 + i, ld, j, m); vv[u] = sd_ld(v, 1, j, m); }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += av[u] * vv[u];
                }
            }
            pb[tid] = acc;
            sd_sync(lds);
            if (tid < mp && i < m) {
                double sum = 0.0;
                for (int q = 0; q < nparts; ++q) sum += pb[q * mp + tid];
                w[i] = beta * sum;
            }
            sd_sync(lds);
        }
        part = 0.0;
        for (int i = tid; i < m; i += SD_T) part += v[i] * w[i];
        const double kk = 0.5 * beta * sd_block_sum(part, red);
        for (int i = tid; i < m; i += SD_T) w[i] -= kk * v[i];
        sd_sync(lds);
        {                                                       // A22 -= v w' + w v': lanes own rows, four columns per pass
            double vi[SD_RPL], wi[SD_RPL];
#pragma unroll
            for (int q = 0; q < SD_RPL; ++q) { const int i = lane + 64 * q; vi[q] = sd_ld(v, 1, i, m); wi[q] = sd_ld(w, 1, i, m); }
            for (int j0 = wave * 4; j0 < m; j0 += (SD_T / 64) * 4) {
                double vj[4], wj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { vj[u] = sd_ld(v, 1, j0 + u, m); wj[u] = sd_ld(w, 1, j0 + u, m); }
#pragma unroll
                for (int q = 0; q < SD_RPL; ++q) {
                    const int i = lane + 64 * q;
                    if (i < m) {
                        double t[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) t[u] = sd_ld(A22 + i, ld, j0 + u, m);
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (j0 + u < m) A22[i + (long)(j0 + u) * ld] = t[u] - (vi[q] * wj[u] + wi[q] * vj[u]);
                    }
                }
            }
        }
        sd_sync(lds);
    }
    if (tid == 0) { dg[r - 1] = A[(r - 1) + (long)(r - 1) * ld]; of[r - 1] = 0.0; }
    sd_sync(lds);
    // Gershgorin interval
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (int i = tid; i < r; i += SD_T) {
        const double rad = (i > 0 ? fabs(of[i - 1]) : 0.0) + (i + 1 < r ? fabs(of[i]) : 0.0);
        lo = fmin(lo, dg[i] - rad);
        hi = fmax(hi, dg[i] + rad);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
    sd_sync(lds);
    if (lane == 0) { red[wave] = lo; red[16 + wave] = hi; }
    sd_sync(lds);
    lo = red[0]; hi = red[16];
    for (int q = 1; q < SD_NW; ++q) { lo = fmin(lo, red[q]); hi = fmax(hi, red[16 + q]); }
    const double span = fmax(fabs(lo), fabs(hi));
    hi += 1e-15 * span + 1e-300;                               // the count at hi must be r, at lo 0
    lo -= 1e-15 * span + 1e-300;
    int *first = (int *)(red + 40);
    for (int round = 0; round < 12; ++round) {
        if (!(hi - lo > 4.4e-16 * fmax(fabs(lo), fabs(hi)))) break;
        const double step = (hi - lo) / (SD_T + 1);
        const double xs = lo + step * (tid + 1);
        // Sturm count: number of eigenvalues below xs
        int cnt = 0;
        double q = dg[0] - xs;
        if (q < 0.0) ++cnt;
        for (int i = 1; i < r; ++i) {
            if (q == 0.0) q = 1e-300;
            q = (dg[i] - xs) - of[i - 1] * of[i - 1] / q;
            if (q < 0.0) ++cnt;
        }
        const bool hit = want_max ? (cnt >= r) : (cnt >= 1);   // monotone in tid
        if (tid == 0) *first = SD_T;
        sd_sync(lds);
        if (hit) atomicMin(first, tid);
        sd_sync(lds);
        const int f = *first;
        sd_sync(lds);
        const double nlo = (f == 0) ? lo : lo + step * f;       // x_{f-1}
        const double nhi = (f == SD_T) ? hi : lo + step * (f + 1);
        lo = nlo; hi = nhi;
    }
    return 0.5 * (lo + hi);
}

// LDS pitch of the one-sided Jacobi's matrix: == 4 (mod 32) doubles, so that the 8 column pairs x 4 row-interleaved
// lanes of a half-wave (columns p, p+1, ... of consecutive pairs) fall on 32 different 8-byte bank pairs
__host__ __device__ __forceinline__ int sd_pitch(int r) { return r + ((4 - r % 32) + 32) % 32; }
// One-sided (Hestenes) Jacobi: right rotations until the columns of G (r x r, ld) are mutually orthogonal:
// G_in V = U diag(sigma), i.e. column i ends as sigma_i u_i -- the left singular vectors and singular values of
// G_in, which is all nestod_sdc needs of svd(Lz' Ls) (src/ConicIP.jl:204-208).  Only ONE matrix is live, so r = 128
// (padded pitch 129: the 64 column pairs of a round then fall on different LDS banks) stays LDS-resident.  A round
// rotates r/2 disjoint column pairs at once, `tpp` lanes per pair, one barrier per round.
__device__ __forceinline__ void sd_jacobi_onesided(double *G, int r, int ld, int *sflag, bool lds = false) {
    const int tid = threadIdx.x;
    const int m = (r + 1) & ~1, np = m / 2;
    int tpp = 1;
    while (tpp < 64 && 2 * tpp * np <= SD_T) tpp *= 2;
    const int ngroups = SD_T / tpp;
    const int part = tid % tpp;
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (tid == 0) *sflag = 0;
        sd_sync(lds);
        for (int t = 0; t < m - 1; ++t) {
            for (int k = tid / tpp; k < np; k += ngroups) {
                int p, q;
                if (k == 0) { p = m - 1; q = t; }
                else { p = (t + k) % (m - 1); q = (t - k + (m - 1)) % (m - 1); }
                const bool live = p < r && q < r;              // dummy player (index r) when r is odd
                double *gp = G + (long)(live ? p : 0) * ld, *gq = G + (long)(live ? q : 0) * ld;
                double a = 0.0, b = 0.0, c = 0.0;
                if (r <= 8 * tpp) {
                    // both columns stay in registers between the dot products and the rotation
                    double xv[8], yv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = part + u * tpp;
                        xv[u] = sd_ld(gp, 1, i, r);
                        yv[u] = sd_ld(gq, 1, i, r);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) { a += xv[u] * xv[u]; b += yv[u] * yv[u]; c += xv[u] * yv[u]; }
                    for (int o = tpp >> 1; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
                    if (live && c * c > 1e-30 * (a * b) && c != 0.0) {
                        const double zeta = (b - a) * 0.5 * sd_rcp(c);
                        const double h2 = 1.0 + zeta * zeta;
                        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) * sd_rcp(fabs(zeta) + h2 * sd_rsqrt(h2));
                        const double cs = sd_rsqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = part + u * tpp;
                            if (i < r) { gp[i] = cs * xv[u] - sn * yv[u]; gq[i] = sn * xv[u] + cs * yv[u]; }
                        }
                        if (part == 0) *sflag = 1;
                    }
                    continue;
                }
                if (live) {
                    for (int i0 = part; i0 < r; i0 += 8 * tpp) {       // 16 LDS reads in flight per pass
                        double xv[8], yv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { const int i = i0 + u * tpp; xv[u] = sd_ld(gp, 1, i, r); yv[u] = sd_ld(gq, 1, i, r); }
#pragma unroll
                        for (int u = 0; u < 8; ++u) { a += xv[u] * xv[u]; b += yv[u] * yv[u]; c += xv[u] * yv[u]; }
                    }
                }
                for (int o = tpp >> 1; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
                if (live && c * c > 1e-30 * (a * b) && c != 0.0) {
                    const double zeta = (b - a) * 0.5 * sd_rcp(c);
                    const double h2 = 1.0 + zeta * zeta;
                    const double tt = (zeta >= 0.0 ? 1.0 : -1.0) * sd_rcp(fabs(zeta) + h2 * sd_rsqrt(h2));
                    const double cs = sd_rsqrt(1.0 + tt * tt), sn = cs * tt;
                    for (int i0 = part; i0 < r; i0 += 8 * tpp) {
                        double xv[8], yv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { const int i = i0 + u * tpp; xv[u] = sd_ld(gp, 1, i, r); yv[u] = sd_ld(gq, 1, i, r); }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = i0 + u * tpp;
                            if (i < r) { gp[i] = cs * xv[u] - sn * yv[u]; gq[i] = sn * xv[u] + cs * yv[u]; }
                        }
                    }
                    if (part == 0) *sflag = 1;
                }
            }
            sd_sync(lds);
        }
        const int again = *sflag;
        sd_sync(lds);
        if (!again) break;
    }
}

// workspace of one workgroup: NW r x r matrices
#define SD_NW 6
__device__ __forceinline__ double *sd_ws(double *base, int slot, int r, int which) {
    return base + ((size_t)slot * SD_NW + which) * (size_t)r * r;
}

// ---------------------------------------------------------------------------------- NT scaling
// The body is inlined twice (live matrix W in LDS / in global memory) so that the LDS instance is compiled to ds_*
// instructions: a pointer selected at run time between the two address spaces makes every access a FLAT one
// (measured: 6.6 ms instead of sub-millisecond for the one-sided Jacobi at r = 128).
__device__ __forceinline__ void sd_nt_body(const ConeDesc &cd, const double *v, const double *s, double *R, double *Ri,
                                           double *lambda, double *Z, double *S, double *T, double *U, double *W, int ld,
                                           int *flag, int *sflagp, bool lds) {
    const int r = cd.r;
    SD_TICK_INIT;
    sd_mat(v + cd.off, 1, W, r, ld);
    SD_TICK("nt mat");
    sd_chol(W, r, flag, ld, lds);                                       // Lz     (src/ConicIP.jl:202-203)
    SD_TICK("nt chol");
    for (int e = threadIdx.x; e < r * r; e += SD_T) Z[e] = W[e % r + (e / r) * ld];
    __syncthreads();
    sd_mat(s + cd.off, 1, W, r, ld);
    sd_chol(W, r, flag, ld, lds);                                       // Ls
    for (int e = threadIdx.x; e < r * r; e += SD_T) S[e] = W[e % r + (e / r) * ld];
    __syncthreads();
    SD_TICK("nt chol2+copy");
    sd_gemm(U, Z, true, S, false, r);                              // G = Lz' Ls = U Lambda V'   (:204)
    SD_TICK("nt gemm");
    for (int e = threadIdx.x; e < r * r; e += SD_T) W[e % r + (e / r) * ld] = U[e];
    __syncthreads();
    sd_jacobi_onesided(W, r, ld, sflagp, lds);                          // column i = Lambda_i u_i
    SD_TICK("nt jacobi1");
    double *lam = S;                                               // Ls is no longer needed: first r entries hold Lambda
    for (int i = threadIdx.x; i < r; i += SD_T) {
        double n2 = 0.0;
        for (int k = 0; k < r; ++k) { const double g = W[k + i * ld]; n2 += g * g; }
        lam[i] = sqrt(n2);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < r * r; e += SD_T) {
        const int i = e % r, j = e / r;
        const double u = W[i + j * ld] / lam[j];
        U[e] = u;
        W[i + j * ld] = u;
    }
    __syncthreads();
    // R = Lz^-T U Lambda^(1/2) (:206-208);  Rinv = Lambda^(-1/2) U' Lz'
    SD_TICK("nt lam/U");
    sd_trsm_lt(Z, W, r, r, ld, lds);                                    // Lz^-T U
    SD_TICK("nt trsm_lt");
    for (int e = threadIdx.x; e < r * r; e += SD_T) R[e] = W[e % r + (e / r) * ld] * sqrt(lam[e / r]);
    __syncthreads();
    sd_gemm(T, U, true, Z, true, r);                               // U' Lz'
    for (int e = threadIdx.x; e < r * r; e += SD_T) Ri[e] = T[e] / sqrt(lam[e % r]);
    SD_TICK("nt gemm2+Ri");
    SD_TICK_DUMP;
    if (lambda) {
        // lambda = F v = vecm(R' Z R) = vecm(diag(Lambda))
        for (int e = threadIdx.x; e < cd.dim; e += SD_T) lambda[cd.off + e] = 0.0;
        __syncthreads();
        for (int i = threadIdx.x; i < r; i += SD_T) lambda[cd.off + vidx(i, i, r)] = lam[i];
    }
}
__global__ __launch_bounds__(SD_TMAX) void k_sdp_nt_scaling(const ConeDesc *cones, const int *sidx, const double *v,
                                                          const double *s, double *scal, double *lambda, double *wsb,
                                                          int *flag, int cap, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO6(cb, v, s, scal, lambda, wsb, flag);
    extern __shared__ double sh[];
    __shared__ int sflag;
    const ConeDesc cd = cones[sidx[blockIdx.x]];
    const int r = cd.r;
    double *Z = sd_ws(wsb, blockIdx.x, r, 0), *S = sd_ws(wsb, blockIdx.x, r, 1), *T = sd_ws(wsb, blockIdx.x, r, 2),
           *U = sd_ws(wsb, blockIdx.x, r, 4);
    double *R = scal + cd.soff, *Ri = R + (size_t)r * r;
    // Every O(r^3) stage with a serial chain works on ONE matrix at a time, LDS-resident (pitch r + 1) when
    // r (r + 1) doubles fit; the finished factor is parked in global memory for the next stage.
    if (r * sd_pitch(r) <= cap) sd_nt_body(cd, v, s, R, Ri, lambda, Z, S, T, U, sh, sd_pitch(r), flag, &sflag, true);
    else sd_nt_body(cd, v, s, R, Ri, lambda, Z, S, T, U, T, r, flag, &sflag, false);
}

// out = vecm(P' X P) with P = R (F), R' (F'), Rinv (F^-1), Rinv' (F^-T); x / out strided (xs, os)
__device__ void sd_congruence(const double *R, const double *Ri, int mode, const double *x, long xs, double *out, long os,
                              int r, double *X, double *T, double *Y) {
    const double *P = (mode == CIP_OP_F || mode == CIP_OP_FT) ? R : Ri;
    const bool tr = (mode == CIP_OP_FT || mode == CIP_OP_FINVT);     // use P' in place of P
    sd_mat(x, xs, X, r);
    sd_gemm(T, X, false, P, tr, r);            // X P  (or X P')
    sd_gemm(Y, P, !tr, T, false, r);           // P' X P (or P X P')
    sd_vecm(Y, out, os, r, 1.0);
}

__global__ __launch_bounds__(SD_TMAX) void k_sdp_apply(const ConeDesc *cones, const int *sidx, const double *scal, int mode,
                                                     const double *x, double *out, double *wsb, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, scal, x, out, wsb);
    const ConeDesc cd = cones[sidx[blockIdx.x]];
    const int r = cd.r;
    const double *R = scal + cd.soff;
    sd_congruence(R, R + (size_t)r * r, mode, x + cd.off, 1, out + cd.off, 1, r, sd_ws(wsb, blockIdx.x, r, 0),
                  sd_ws(wsb, blockIdx.x, r, 1), sd_ws(wsb, blockIdx.x, r, 2));
}

// Wt[i, off+e] = (F^-T a_i)_e for rows i of At (grid.x loops over i, grid.y = S cone)
__global__ __launch_bounds__(SD_TMAX) void k_sdp_scale_At(const ConeDesc *cones, const int *sidx, const double *scal, int n,
                                                        const double *At, long ldat, double *Wt, long ldwt, double *wsb, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, scal, At, Wt, wsb);
    const ConeDesc cd = cones[sidx[blockIdx.y]];
    const int r = cd.r;
    const double *R = scal + cd.soff;
    const int slot = blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = blockIdx.x; i < n; i += gridDim.x)
        sd_congruence(R, R + (size_t)r * r, CIP_OP_FINVT, At + i + (long)cd.aoff * ldat, ldat, Wt + i + (long)cd.aoff * ldwt,
                      ldwt, r, sd_ws(wsb, slot, r, 0), sd_ws(wsb, slot, r, 1), sd_ws(wsb, slot, r, 2));
}

// column c of -(F'F) for the literal 3x3 assembly: K[off+e, off+c] = -(F'(F e_c))_e  (lower part)
__global__ __launch_bounds__(SD_TMAX) void k_sdp_fill_ftf(const ConeDesc *cones, const int *sidx, const double *scal, double *K,
                                                        long ldk, double *wsb, double *vtmp, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, scal, K, wsb, vtmp);
    const ConeDesc cd = cones[sidx[blockIdx.y]];
    const int r = cd.r, k = cd.dim;
    const double *R = scal + cd.soff;
    const int slot = blockIdx.y * gridDim.x + blockIdx.x;
    double *u = vtmp + (size_t)slot * 2 * k, *w = u + k;
    for (int c = blockIdx.x; c < k; c += gridDim.x) {
        for (int e = threadIdx.x; e < k; e += SD_T) u[e] = (e == c) ? 1.0 : 0.0;
        __syncthreads();
        sd_congruence(R, R + (size_t)r * r, CIP_OP_F, u, 1, w, 1, r, sd_ws(wsb, slot, r, 0), sd_ws(wsb, slot, r, 1),
                      sd_ws(wsb, slot, r, 2));
        sd_congruence(R, R + (size_t)r * r, CIP_OP_FT, w, 1, u, 1, r, sd_ws(wsb, slot, r, 0), sd_ws(wsb, slot, r, 1),
                      sd_ws(wsb, slot, r, 2));
        for (int e = c + threadIdx.x; e < k; e += SD_T) K[(cd.off + e) + (long)(cd.off + c) * ldk] = -u[e];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------- Jordan product / division
__global__ __launch_bounds__(SD_TMAX) void k_sdp_prod(const ConeDesc *cones, const int *sidx, const double *x, const double *y,
                                                    double *out, double *wsb, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, x, y, out, wsb);
    const ConeDesc cd = cones[sidx[blockIdx.x]];
    const int r = cd.r;
    double *X = sd_ws(wsb, blockIdx.x, r, 0), *Y = sd_ws(wsb, blockIdx.x, r, 1), *T = sd_ws(wsb, blockIdx.x, r, 2);
    sd_mat(x + cd.off, 1, X, r);
    sd_mat(y + cd.off, 1, Y, r);
    sd_gemm(T, X, false, Y, false, r);                       // XY ; XY + YX = T + T'
    for (int e = threadIdx.x; e < r * r; e += SD_T) { const int i = e % r, j = e / r; X[e] = T[e] + T[j + i * r]; }
    __syncthreads();
    sd_vecm(X, out + cd.off, 1, r, 1.0);                     // xsdc! src/ConicIP.jl:355-360
}

// out: Y O + O Y = X  (dsdc! = vecm(lyap(Y, -X)) src/ConicIP.jl:347-353)
// gate (large cones, sdp_large.hip: cip_sdp_large_div): the chip-wide element-wise kernels have already written the quotient
// unless *gate is up (the divisor is not diagonal); the word is taken down again here
__global__ __launch_bounds__(SD_TMAX) void k_sdp_div(const ConeDesc *cones, const int *sidx, const double *x, const double *y,
                                                   double *out, double *wsb, int cap, int *gate, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, x, y, out, wsb);
    extern __shared__ double sh[];
    if (gate) {
        const int up = *gate;
        __syncthreads();
        if (threadIdx.x == 0) *gate = 0;
        if (!up) return;
    }
    const ConeDesc cd = cones[sidx[blockIdx.x]];
    const int r = cd.r;
    double *X = sd_ws(wsb, blockIdx.x, r, 0), *Y = sd_ws(wsb, blockIdx.x, r, 1), *V = sd_ws(wsb, blockIdx.x, r, 2),
           *T = sd_ws(wsb, blockIdx.x, r, 3), *W = sd_ws(wsb, blockIdx.x, r, 4);
    sd_mat(x + cd.off, 1, X, r);
    sd_mat(y + cd.off, 1, Y, r);
    {
        // Y diagonal (every division of the interior-point loop is by lambda = vecm(diag(Lambda)), src/ConicIP.jl:686):
        // the Lyapunov solve is an element-wise quotient -- same numbers as the general path with V = I, none of its
        // four r^3 products
        __shared__ double dred[32];
        double off = 0.0, tot = 0.0;
        for (int e = threadIdx.x; e < r * r; e += SD_T) { const double a = Y[e]; tot += a * a; if (e % r != e / r) off += a * a; }
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o); tot += __shfl_xor(tot, o); }
        if ((threadIdx.x & 63) == 0) { dred[threadIdx.x >> 6] = off; dred[16 + (threadIdx.x >> 6)] = tot; }
        __syncthreads();
        off = 0.0; tot = 0.0;
        for (int q = 0; q < SD_T / 64; ++q) { off += dred[q]; tot += dred[16 + q]; }
        if (off == 0.0 && tot > 0.0) {
            for (int e = threadIdx.x; e < r * r; e += SD_T) { const int i = e % r, j = e / r; X[e] /= (Y[i + i * r] + Y[j + j * r]); }
            __syncthreads();
            sd_vecm(X, out + cd.off, 1, r, 1.0);
            return;
        }
    }
    sd_jacobi(Y, V, r, sh, cap);                             // Y = V diag V'
    sd_gemm(T, X, false, V, false, r);
    sd_gemm(W, V, true, T, false, r);                        // V' X V
    for (int e = threadIdx.x; e < r * r; e += SD_T) { const int i = e % r, j = e / r; W[e] /= (Y[i + i * r] + Y[j + j * r]); }
    __syncthreads();
    sd_gemm(T, W, false, V, true, r);
    sd_gemm(X, V, false, T, false, r);                       // V O' V'
    sd_vecm(X, out + cd.off, 1, r, 1.0);
}

// ---------------------------------------------------------------------------------- max step
// eigvals(X^-1/2 D X^-1/2) (:272-293) are the eigenvalues of L^-1 D L^-T with X = L L' (the two matrices are
// similar through the orthogonal X^-1/2 L), and only the largest is used: one Cholesky, two triangular solves, one
// tridiagonalisation + bisection -- with the ONE live matrix LDS-resident (pitch r + 1) up to r = 139 -- instead of
// an eigendecomposition with vectors, three GEMMs and a second Jacobi, all in global memory at r = 128 (80 ms per
// call -> see DESIGN.md).  `cap` = doubles of dynamic LDS behind the scratch.
__device__ __forceinline__ void sd_maxstep_body(const ConeDesc &cd, const double *x, const double *d, double scale,
                                                double *partial, double *Lg, double *M, int ld, double *sc, int *sflagp, bool lds) {
    const int r = cd.r;
    const double INF = __builtin_inf();
    SD_TICK_INIT;
    sd_mat(x + cd.off, 1, M, r, ld);
    if (!d) {                                                // maxstep_sdc(x, nothing) :295-303
        const double mn = sd_extreme_eig(M, r, ld, sc, false, lds);
        if (threadIdx.x == 0) partial[cd.item] = (mn > 0.0) ? 0.0 : -1.0 + mn;
        return;
    }
    if (threadIdx.x == 0) *sflagp = 0;
    __syncthreads();
    sd_chol(M, r, sflagp, ld, lds);                               // M <- L
    if (*sflagp) {                                           // X not PD -> Inf (:277-280)
        if (threadIdx.x == 0) partial[cd.item] = INF;
        return;
    }
    for (int e = threadIdx.x; e < r * r; e += SD_T) Lg[e] = M[e % r + (e / r) * ld];
    __syncthreads();
    sd_mat(d + cd.off, 1, M, r, ld);
    SD_TICK("ms chol+copy+mat");
    sd_trsm_l(Lg, M, r, r, ld, lds);                              // L^-1 D
    SD_TICK("ms trsm1");
    sd_transpose(M, r, ld, lds);                                  // D L^-T
    sd_trsm_l(Lg, M, r, r, ld, lds);                              // L^-1 D L^-T
    SD_TICK("ms trsm2");
    for (int e = threadIdx.x; e < r * r; e += SD_T) {
        const int i = e % r, j = e / r;
        if (i > j) { const double a = 0.5 * (M[i + j * ld] + M[j + i * ld]); M[i + j * ld] = a; M[j + i * ld] = a; }
    }
    __syncthreads();
    const double mx = sd_extreme_eig(M, r, ld, sc, true, lds) * scale;
    SD_TICK("ms tridiag+bis");
    SD_TICK_DUMP;
    if (threadIdx.x == 0) partial[cd.item] = (mx < 0.0) ? INF : 1.0 / mx;
}
__global__ __launch_bounds__(SD_TMAX) void k_sdp_maxstep(const ConeDesc *cones, const int *sidx, const double *x, const double *d,
                                                       double scale, double *partial, double *wsb, int cap, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, x, d, partial, wsb);
    extern __shared__ double sh[];
    __shared__ int sflag;
    const ConeDesc cd = cones[sidx[blockIdx.x]];
    const int r = cd.r;
    double *Lg = sd_ws(wsb, blockIdx.x, r, 0);               // L in global memory (pitch r)
    if (r * (r + 1) <= cap) sd_maxstep_body(cd, x, d, scale, partial, Lg, sh + SD_SCRATCH_TRI(r), r + 1, sh, &sflag, true);
    else sd_maxstep_body(cd, x, d, scale, partial, Lg, sd_ws(wsb, blockIdx.x, r, 2), r, sh, &sflag, false);
}

// ---------------------------------------------------------------------------------- host launchers
static int sd_threads(int rmax) { return rmax > 48 ? 1024 : 256; }
// dynamic LDS: rotation scratch + up to `nmat` matrices of the largest cone (pitch r+1 allowed for), capped
static int sd_cap(int rmax, int nmat) {
    const int pitch = sd_pitch(rmax) > rmax + 1 ? sd_pitch(rmax) : rmax + 1;
    const long want = (long)nmat * rmax * pitch;
    const long room = SD_LDS_CAPMAX - SD_SCRATCH(rmax);
    return (int)(want < room ? want : room);
}
static size_t sd_shmem(int rmax, int nmat) { return ((size_t)SD_SCRATCH(rmax) + sd_cap(rmax, nmat)) * sizeof(double); }
static int sd_set_lds_attr(const void *fn, int rmax, int nmat) {
    CIP_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sd_shmem(rmax, nmat)));
    return 0;
}

int cip_sdp_nt_scaling(hipStream_t s, const ConeSet &cs, const double *v, const double *sv, double *lambda) {
    if (cs.ns_small > 0) {
        if (sd_set_lds_attr((const void *)k_sdp_nt_scaling, cs.rmax, 1)) return -3;
        cip_launch_b(k_sdp_nt_scaling, dim3(cs.ns_small), dim3(sd_threads(cs.rmax)), sd_shmem(cs.rmax, 1), s, cs.d_cones,
                           cs.d_sidx_small, v, sv, cs.d_scal, lambda, cs.d_sdpws, cs.d_sdpflag, sd_cap(cs.rmax, 1) + SD_SCRATCH(cs.rmax));
        CIP_HIP_CHECK(hipGetLastError());
    }
    for (int li = 0; li < cs.nlarge; ++li) {                 // orders 133 .. 512: chip-wide pieces (sdp_large.hip)
        const int rc = cip_sdp_large_nt(s, cs.lg, cs.h_cones[cs.large_cone[li]], li, v, sv, cs.d_scal, lambda, cs.d_sdpflag);
        if (rc) return rc;
    }
    return 0;
}
// the packed scaling was replaced from outside (cip_set_scaling_packed / identity): refresh what the large path derives from it
int cip_sdp_scaling_changed(hipStream_t s, const ConeSet &cs) {
    for (int li = 0; li < cs.nlarge; ++li) {
        const int rc = cip_sdp_large_refresh(s, cs.lg, cs.h_cones[cs.large_cone[li]], li, cs.d_scal);
        if (rc) return rc;
    }
    return 0;
}
int cip_sdp_apply(hipStream_t s, const ConeSet &cs, int mode, const double *x, double *out) {
    if (cs.ns_small > 0) {
        cip_launch_b(k_sdp_apply, dim3(cs.ns_small), dim3(sd_threads(cs.rmax)), 0, s, cs.d_cones, cs.d_sidx_small, cs.d_scal, mode, x,
                           out, cs.d_sdpws);
        CIP_HIP_CHECK(hipGetLastError());
    }
    for (int li = 0; li < cs.nlarge; ++li) {
        const int rc = cip_sdp_large_apply(s, cs.lg, cs.h_cones[cs.large_cone[li]], li, mode, x, out);
        if (rc) return rc;
    }
    return 0;
}
int cip_sdp_prod(hipStream_t s, const ConeSet &cs, const double *x, const double *y, double *out) {
    if (cs.ns_small > 0) {
        cip_launch_b(k_sdp_prod, dim3(cs.ns_small), dim3(sd_threads(cs.rmax)), 0, s, cs.d_cones, cs.d_sidx_small, x, y, out, cs.d_sdpws);
        CIP_HIP_CHECK(hipGetLastError());
    }
    for (int li = 0; li < cs.nlarge; ++li) {
        const int rc = cip_sdp_large_prod(s, cs.lg, cs.h_cones[cs.large_cone[li]], x, y, out);
        if (rc) return rc;
    }
    return 0;
}
int cip_sdp_div(hipStream_t s, const ConeSet &cs, const double *x, const double *y, double *out) {
    if (sd_set_lds_attr((const void *)k_sdp_div, cs.rmax, 2)) return -3;
    if (cs.nlarge == 0 || cip_in_batch()) {
        cip_launch_b(k_sdp_div, dim3(cs.ns), dim3(sd_threads(cs.rmax)), sd_shmem(cs.rmax, 2), s, cs.d_cones, cs.d_sidx, x, y, out, cs.d_sdpws,
                           sd_cap(cs.rmax, 2), (int *)nullptr);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (cs.ns_small > 0)
        cip_launch_b(k_sdp_div, dim3(cs.ns_small), dim3(sd_threads(cs.rmax)), sd_shmem(cs.rmax, 2), s, cs.d_cones, cs.d_sidx_small, x, y, out,
                           cs.d_sdpws, sd_cap(cs.rmax, 2), (int *)nullptr);
    for (int li = 0; li < cs.nlarge; ++li) {
        // a large cone: element-wise and chip-wide when the divisor is diagonal (it is, in the interior-point loop); the general
        // one-workgroup kernel behind it only runs when the check raised the gate
        const int c = cs.large_cone[li];
        int pos = 0;
        for (int q = 0; q < c; ++q) pos += cs.h_cones[q].type == CIP_CONE_S;
        int *gate = cs.d_sdpflag + 4 + li;
        const int rc = cip_sdp_large_div(s, cs.lg, cs.h_cones[c], x, y, out, gate);
        if (rc) return rc;
        cip_launch_b(k_sdp_div, dim3(1), dim3(sd_threads(cs.rmax)), sd_shmem(cs.rmax, 2), s, cs.d_cones, (const int *)(cs.d_sidx + pos), x, y, out,
                           cs.d_sdpws, sd_cap(cs.rmax, 2), gate);
    }
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_sdp_maxstep(hipStream_t s, const ConeSet &cs, const double *x, const double *d, double scale, double *partial) {
    // scratch of the tridiagonalisation + one matrix of pitch r + 1 when it fits
    const long want = (long)cs.rmax * (cs.rmax + 1), room = SD_LDS_CAPMAX - SD_SCRATCH_TRI(cs.rmax);
    const int cap = (int)(want < room ? want : (room > 0 ? room : 0));
    const size_t shm = ((size_t)SD_SCRATCH_TRI(cs.rmax) + cap) * sizeof(double);
    if (cs.ns_small > 0) {
        CIP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sdp_maxstep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        cip_launch_b(k_sdp_maxstep, dim3(cs.ns_small), dim3(sd_threads(cs.rmax)), shm, s, cs.d_cones, cs.d_sidx_small, x, d, scale,
                           partial, cs.d_sdpws, cap);
        CIP_HIP_CHECK(hipGetLastError());
    }
    for (int li = 0; li < cs.nlarge; ++li) {
        const int rc = cip_sdp_large_maxstep(s, cs.lg, cs.h_cones[cs.large_cone[li]], x, d, scale, partial);
        if (rc) return rc;
    }
    return 0;
}
// both sides of a pair: the small cones one side after the other (they share their scratch), the large ones side by side
int cip_sdp_maxstep2(hipStream_t s, const ConeSet &cs, const double *x1, const double *d1, double *p1, const double *x2,
                     const double *d2, double *p2, double scale) {
    int rc;
    if (cs.nlarge == 0 || !cip_sdp_large_pairable(cs.lg)) {
        if ((rc = cip_sdp_maxstep(s, cs, x1, d1, scale, p1))) return rc;
        return cip_sdp_maxstep(s, cs, x2, d2, scale, p2);
    }
    ConeSet small = cs;                                            // (a view: same buffers, no large cones)
    small.nlarge = 0;
    if ((rc = cip_sdp_maxstep(s, small, x1, d1, scale, p1))) return rc;
    if ((rc = cip_sdp_maxstep(s, small, x2, d2, scale, p2))) return rc;
    hipStream_t s2;
    if ((rc = cip_sdp_large_fork(s, cs.lg, &s2))) return rc;
    // (a failure between fork and join still joins: the forked stream must not be left running ahead of s)
    for (int li = 0; li < cs.nlarge && rc == 0; ++li)
        rc = cip_sdp_large_maxstep(s, cs.lg, cs.h_cones[cs.large_cone[li]], x1, d1, scale, p1, 0);
    for (int li = 0; li < cs.nlarge && rc == 0; ++li)
        rc = cip_sdp_large_maxstep(s2, cs.lg, cs.h_cones[cs.large_cone[li]], x2, d2, scale, p2, 1);
    const int rj = cip_sdp_large_join(s, cs.lg);
    return rc ? rc : rj;
}
int cip_sdp_scale_At(hipStream_t s, const ConeSet &cs, int n, const double *At, long ldat, double *Wt, long ldwt) {
    if (cs.ns_small > 0) {
        const int gx = n < cs.sdp_slots / cs.ns ? n : cs.sdp_slots / cs.ns;
        cip_launch_b(k_sdp_scale_At, dim3(gx, cs.ns_small), dim3(sd_threads(cs.rmax)), 0, s, cs.d_cones, cs.d_sidx_small, cs.d_scal, n,
                           At, ldat, Wt, ldwt, cs.d_sdpws);
        CIP_HIP_CHECK(hipGetLastError());
    }
    for (int li = 0; li < cs.nlarge; ++li) {
        const int rc = cip_sdp_large_scale_At(s, cs.lg, cs.h_cones[cs.large_cone[li]], li, n, At, ldat, Wt, ldwt);
        if (rc) return rc;
    }
    return 0;
}
int cip_sdp_fill_ftf(hipStream_t s, const ConeSet &cs, double *K, long ldk) {
    const int gx = cs.sdp_slots / cs.ns;
    cip_launch_b(k_sdp_fill_ftf, dim3(gx, cs.ns), dim3(sd_threads(cs.rmax)), 0, s, cs.d_cones, cs.d_sidx, cs.d_scal, K, ldk, cs.d_sdpws,
                       cs.d_sdpvec);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
