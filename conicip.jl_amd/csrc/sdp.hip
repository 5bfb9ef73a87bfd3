// S-cone (semidefinite) kernels: device counterparts of
//   mat / vecm                      src/ConicIP.jl:85-151
//   nestod_sdc                      src/ConicIP.jl:196-210
//   VecCongurance apply/inv/adjoint src/ConicIP.jl:35-40, :69
//   xsdc! / dsdc! (lyap)            src/ConicIP.jl:347-360
//   maxstep_sdc                     src/ConicIP.jl:272-303
//
// One 256-thread workgroup per cone (or per column of A for the Schur scaling); the r x r matrices live in
// a global workspace (r is small in every reference test, <= 30; the layout scales to a few hundred).  The
// dense symmetric eigenproblems behind nestod_sdc (the reference uses chol + SVD; here chol + symmetric
// eigendecomposition of Lz' S Lz, which has the same invariant subspaces), maxstep_sdc (eigvals, X^-1/2) and
// dsdc! (Lyapunov solve) all go through one two-sided Jacobi routine with the round-robin parallel ordering:
// r/2 disjoint rotations per round, applied as row pass + column pass across the whole workgroup.
// R is determined only up to a signed permutation of its columns; F'F, F'(F x) and every norm the driver forms
// are invariant under it.
#include "cip_internal.h"
#include "../../include/cipkkt.h"
#include <math.h>

#define SD_T 256
#define SQRT2 1.4142135623730951
#define SQRT1_2 0.7071067811865476

__device__ __forceinline__ int vidx(int i, int j, int r) { return i * r - i * (i - 1) / 2 + (j - i); }   // i <= j

// X (r x r, col-major) = mat(x)
__device__ void sd_mat(const double *x, long xs, double *X, int r) {
    for (int e = threadIdx.x; e < r * r; e += SD_T) {
        const int i = e % r, j = e / r;
        const int a = i < j ? i : j, b = i < j ? j : i;
        const double v = x[(long)vidx(a, b, r) * xs];
        X[e] = (a == b) ? v : v * SQRT1_2;
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
// C = op(A) * op(B), all r x r col-major
__device__ void sd_gemm(double *C, const double *A, bool ta, const double *B, bool tb, int r) {
    for (int e = threadIdx.x; e < r * r; e += SD_T) {
        const int i = e % r, j = e / r;
        double s = 0.0;
        for (int k = 0; k < r; ++k) s += (ta ? A[k + i * r] : A[i + k * r]) * (tb ? B[j + k * r] : B[k + j * r]);
        C[e] = s;
    }
    __syncthreads();
}
// in-place lower Cholesky (strict upper zeroed); returns 0 or (column+1) of a non-positive pivot in *flag
__device__ void sd_chol(double *A, int r, int *flag) {
    for (int j = 0; j < r; ++j) {
        const double d = A[j + j * r];
        if (!(d > 0.0)) { if (threadIdx.x == 0) *flag = j + 1; }
        const double l = sqrt(d);
        __syncthreads();
        for (int i = j + threadIdx.x; i < r; i += SD_T) A[i + j * r] = (i == j) ? l : A[i + j * r] / l;
        __syncthreads();
        const int m = r - j - 1;
        for (int e = threadIdx.x; e < m * m; e += SD_T) {
            const int i = j + 1 + e % m, k = j + 1 + e / m;
            if (i >= k) A[i + k * r] -= A[i + j * r] * A[k + j * r];
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < r * r; e += SD_T) if (e % r < e / r) A[e] = 0.0;
    __syncthreads();
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
        if ((tid & 63) == 0) { red[tid >> 6] = off; red[4 + (tid >> 6)] = tot; }
        __syncthreads();
        off = red[0] + red[1] + red[2] + red[3];
        tot = red[4] + red[5] + red[6] + red[7];
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

// Jacobi with the matrices staged in LDS when they fit (r <= SD_LDS_RMAX: A and V, 2 r^2 doubles): the sweep is a
// chain of ~3 (r-1) barrier-separated passes per sweep, each a handful of dependent accesses per thread, so LDS
// latency instead of global-memory latency is a ~10x difference.  `sh` = rotation scratch followed by the staging area.
#define SD_LDS_RMAX 88
__device__ void sd_jacobi(double *A, double *V, int r, double *sh) {
    if (r > SD_LDS_RMAX) { sd_jacobi_core(A, V, r, sh); return; }
    double *la = sh + 4 * ((r + 2) / 2 + 1) + 16;
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

// workspace of one workgroup: NW r x r matrices
#define SD_NW 6
__device__ __forceinline__ double *sd_ws(double *base, int slot, int r, int which) {
    return base + ((size_t)slot * SD_NW + which) * (size_t)r * r;
}

// ---------------------------------------------------------------------------------- NT scaling
__global__ __launch_bounds__(SD_T) void k_sdp_nt_scaling(const ConeDesc *cones, const int *sidx, const double *v,
                                                          const double *s, double *scal, double *lambda, double *wsb,
                                                          int *flag) {
    extern __shared__ double sh[];
    const ConeDesc cd = cones[sidx[blockIdx.x]];
    const int r = cd.r;
    double *Z = sd_ws(wsb, blockIdx.x, r, 0), *S = sd_ws(wsb, blockIdx.x, r, 1), *T = sd_ws(wsb, blockIdx.x, r, 2),
           *M = sd_ws(wsb, blockIdx.x, r, 3), *U = sd_ws(wsb, blockIdx.x, r, 4);
    double *R = scal + cd.soff, *Ri = R + (size_t)r * r;
    sd_mat(v + cd.off, 1, Z, r);
    sd_mat(s + cd.off, 1, S, r);
    sd_chol(Z, r, flag);                       // Z <- Lz
    sd_gemm(T, S, false, Z, false, r);         // S Lz
    sd_gemm(M, Z, true, T, false, r);          // Lz' S Lz  = U Lambda^2 U'
    // symmetrise against rounding
    for (int e = threadIdx.x; e < r * r; e += SD_T) { const int i = e % r, j = e / r; if (i > j) { const double a = 0.5 * (M[e] + M[j + i * r]); M[e] = a; M[j + i * r] = a; } }
    __syncthreads();
    sd_jacobi(M, U, r, sh);
    // R = Lz^-T U Lambda^(1/2);  Rinv = Lambda^(-1/2) U' Lz'
    for (int e = threadIdx.x; e < r * r; e += SD_T) T[e] = U[e];
    __syncthreads();
    sd_solve_LT(Z, T, r);                      // T = Lz^-T U
    for (int e = threadIdx.x; e < r * r; e += SD_T) {
        const int j = e / r;
        R[e] = T[e] * sqrt(sqrt(fmax(M[j + j * r], 0.0)));       // Lambda_j = sqrt(eig_j)
    }
    __syncthreads();
    sd_gemm(T, U, true, Z, true, r);           // U' Lz'
    for (int e = threadIdx.x; e < r * r; e += SD_T) {
        const int i = e % r;
        Ri[e] = T[e] / sqrt(sqrt(fmax(M[i + i * r], 0.0)));
    }
    if (lambda) {
        // lambda = F v = vecm(R' Z R) = vecm(diag(Lambda))
        for (int e = threadIdx.x; e < cd.dim; e += SD_T) lambda[cd.off + e] = 0.0;
        __syncthreads();
        for (int i = threadIdx.x; i < r; i += SD_T) lambda[cd.off + vidx(i, i, r)] = sqrt(fmax(M[i + i * r], 0.0));
    }
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

__global__ __launch_bounds__(SD_T) void k_sdp_apply(const ConeDesc *cones, const int *sidx, const double *scal, int mode,
                                                     const double *x, double *out, double *wsb) {
    const ConeDesc cd = cones[sidx[blockIdx.x]];
    const int r = cd.r;
    const double *R = scal + cd.soff;
    sd_congruence(R, R + (size_t)r * r, mode, x + cd.off, 1, out + cd.off, 1, r, sd_ws(wsb, blockIdx.x, r, 0),
                  sd_ws(wsb, blockIdx.x, r, 1), sd_ws(wsb, blockIdx.x, r, 2));
}

// Wt[i, off+e] = (F^-T a_i)_e for rows i of At (grid.x loops over i, grid.y = S cone)
__global__ __launch_bounds__(SD_T) void k_sdp_scale_At(const ConeDesc *cones, const int *sidx, const double *scal, int n,
                                                        const double *At, long ldat, double *Wt, long ldwt, double *wsb) {
    const ConeDesc cd = cones[sidx[blockIdx.y]];
    const int r = cd.r;
    const double *R = scal + cd.soff;
    const int slot = blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = blockIdx.x; i < n; i += gridDim.x)
        sd_congruence(R, R + (size_t)r * r, CIP_OP_FINVT, At + i + (long)cd.off * ldat, ldat, Wt + i + (long)cd.off * ldwt,
                      ldwt, r, sd_ws(wsb, slot, r, 0), sd_ws(wsb, slot, r, 1), sd_ws(wsb, slot, r, 2));
}

// column c of -(F'F) for the literal 3x3 assembly: K[off+e, off+c] = -(F'(F e_c))_e  (lower part)
__global__ __launch_bounds__(SD_T) void k_sdp_fill_ftf(const ConeDesc *cones, const int *sidx, const double *scal, double *K,
                                                        long ldk, double *wsb, double *vtmp) {
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
__global__ __launch_bounds__(SD_T) void k_sdp_prod(const ConeDesc *cones, const int *sidx, const double *x, const double *y,
                                                    double *out, double *wsb) {
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
__global__ __launch_bounds__(SD_T) void k_sdp_div(const ConeDesc *cones, const int *sidx, const double *x, const double *y,
                                                   double *out, double *wsb) {
    extern __shared__ double sh[];
    const ConeDesc cd = cones[sidx[blockIdx.x]];
    const int r = cd.r;
    double *X = sd_ws(wsb, blockIdx.x, r, 0), *Y = sd_ws(wsb, blockIdx.x, r, 1), *V = sd_ws(wsb, blockIdx.x, r, 2),
           *T = sd_ws(wsb, blockIdx.x, r, 3), *W = sd_ws(wsb, blockIdx.x, r, 4);
    sd_mat(x + cd.off, 1, X, r);
    sd_mat(y + cd.off, 1, Y, r);
    sd_jacobi(Y, V, r, sh);                                  // Y = V diag V'
    sd_gemm(T, X, false, V, false, r);
    sd_gemm(W, V, true, T, false, r);                        // V' X V
    for (int e = threadIdx.x; e < r * r; e += SD_T) { const int i = e % r, j = e / r; W[e] /= (Y[i + i * r] + Y[j + j * r]); }
    __syncthreads();
    sd_gemm(T, W, false, V, true, r);
    sd_gemm(X, V, false, T, false, r);                       // V O' V'
    sd_vecm(X, out + cd.off, 1, r, 1.0);
}

// ---------------------------------------------------------------------------------- max step
__global__ __launch_bounds__(SD_T) void k_sdp_maxstep(const ConeDesc *cones, const int *sidx, const double *x, const double *d,
                                                       double scale, double *partial, double *wsb) {
    extern __shared__ double sh[];
    __shared__ double sres;
    const ConeDesc cd = cones[sidx[blockIdx.x]];
    const int r = cd.r;
    const double INF = __builtin_inf();
    double *X = sd_ws(wsb, blockIdx.x, r, 0), *V = sd_ws(wsb, blockIdx.x, r, 1), *D = sd_ws(wsb, blockIdx.x, r, 2),
           *T = sd_ws(wsb, blockIdx.x, r, 3), *W = sd_ws(wsb, blockIdx.x, r, 4);
    sd_mat(x + cd.off, 1, X, r);
    if (!d) {                                                // maxstep_sdc(x, nothing) :295-303
        sd_jacobi(X, nullptr, r, sh);
        if (threadIdx.x == 0) {
            double mn = INF;
            for (int i = 0; i < r; ++i) mn = fmin(mn, X[i + i * r]);
            partial[cd.item] = (mn > 0.0) ? 0.0 : -1.0 + mn;
        }
        return;
    }
    sd_jacobi(X, V, r, sh);                                  // X = V diag V'
    if (threadIdx.x == 0) {
        double mn = INF;
        for (int i = 0; i < r; ++i) mn = fmin(mn, X[i + i * r]);
        sres = mn;
    }
    __syncthreads();
    if (!(sres > 0.0)) {                                     // X not PD -> Inf (:277-280)
        if (threadIdx.x == 0) partial[cd.item] = INF;
        return;
    }
    // Xih = V diag^-1/2 V'
    for (int e = threadIdx.x; e < r * r; e += SD_T) { const int j = e / r; T[e] = V[e] / sqrt(X[j + j * r]); }
    __syncthreads();
    sd_gemm(W, T, false, V, true, r);                        // W = Xih
    sd_mat(d + cd.off, 1, D, r);
    sd_gemm(T, D, false, W, false, r);
    sd_gemm(X, W, false, T, false, r);                       // Xih D Xih
    for (int e = threadIdx.x; e < r * r; e += SD_T) { const int i = e % r, j = e / r; if (i > j) { const double a = 0.5 * (X[e] + X[j + i * r]); X[e] = a; X[j + i * r] = a; } }
    __syncthreads();
    sd_jacobi(X, nullptr, r, sh);
    if (threadIdx.x == 0) {
        double mx = -INF;
        bool allneg = true;
        for (int i = 0; i < r; ++i) {
            const double l = X[i + i * r] * scale;
            if (!(l < 0.0)) { allneg = false; mx = fmax(mx, l); }
        }
        partial[cd.item] = allneg ? INF : 1.0 / mx;
    }
}

// ---------------------------------------------------------------------------------- host launchers
static size_t sd_shmem(int rmax) {
    size_t d = 4 * ((size_t)(rmax + 2) / 2 + 1) + 16;
    if (rmax <= SD_LDS_RMAX) d += 2 * (size_t)rmax * rmax;          // LDS-resident Jacobi
    return d * sizeof(double);
}
static int sd_set_lds_attr(const void *fn, int rmax) {
    CIP_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sd_shmem(rmax)));
    return 0;
}

int cip_sdp_nt_scaling(hipStream_t s, const ConeSet &cs, const double *v, const double *sv, double *lambda) {
    if (sd_set_lds_attr((const void *)k_sdp_nt_scaling, cs.rmax)) return -3;
    hipLaunchKernelGGL(k_sdp_nt_scaling, dim3(cs.ns), dim3(SD_T), sd_shmem(cs.rmax), s, cs.d_cones, cs.d_sidx, v, sv, cs.d_scal,
                       lambda, cs.d_sdpws, cs.d_sdpflag);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_sdp_apply(hipStream_t s, const ConeSet &cs, int mode, const double *x, double *out) {
    hipLaunchKernelGGL(k_sdp_apply, dim3(cs.ns), dim3(SD_T), 0, s, cs.d_cones, cs.d_sidx, cs.d_scal, mode, x, out, cs.d_sdpws);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_sdp_prod(hipStream_t s, const ConeSet &cs, const double *x, const double *y, double *out) {
    hipLaunchKernelGGL(k_sdp_prod, dim3(cs.ns), dim3(SD_T), 0, s, cs.d_cones, cs.d_sidx, x, y, out, cs.d_sdpws);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_sdp_div(hipStream_t s, const ConeSet &cs, const double *x, const double *y, double *out) {
    if (sd_set_lds_attr((const void *)k_sdp_div, cs.rmax)) return -3;
    hipLaunchKernelGGL(k_sdp_div, dim3(cs.ns), dim3(SD_T), sd_shmem(cs.rmax), s, cs.d_cones, cs.d_sidx, x, y, out, cs.d_sdpws);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_sdp_maxstep(hipStream_t s, const ConeSet &cs, const double *x, const double *d, double scale, double *partial) {
    if (sd_set_lds_attr((const void *)k_sdp_maxstep, cs.rmax)) return -3;
    hipLaunchKernelGGL(k_sdp_maxstep, dim3(cs.ns), dim3(SD_T), sd_shmem(cs.rmax), s, cs.d_cones, cs.d_sidx, x, d, scale, partial,
                       cs.d_sdpws);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_sdp_scale_At(hipStream_t s, const ConeSet &cs, int n, const double *At, long ldat, double *Wt, long ldwt) {
    const int gx = n < cs.sdp_slots / cs.ns ? n : cs.sdp_slots / cs.ns;
    hipLaunchKernelGGL(k_sdp_scale_At, dim3(gx, cs.ns), dim3(SD_T), 0, s, cs.d_cones, cs.d_sidx, cs.d_scal, n, At, ldat, Wt, ldwt,
                       cs.d_sdpws);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_sdp_fill_ftf(hipStream_t s, const ConeSet &cs, double *K, long ldk) {
    const int gx = cs.sdp_slots / cs.ns;
    hipLaunchKernelGGL(k_sdp_fill_ftf, dim3(gx, cs.ns), dim3(SD_T), 0, s, cs.d_cones, cs.d_sidx, cs.d_scal, K, ldk, cs.d_sdpws,
                       cs.d_sdpvec);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
