// S cones of matrix order 133 .. 512: the stages that do not fit one workgroup's LDS.
//
// The workgroup-per-cone kernels of sdp.hip keep ONE r x r matrix LDS-resident up to r = 132; above that they ran the
// same serial chains from global memory (BASELINE config 4 at its literal size, r = 256: 67 ms per NT scaling, 10.9 ms
// per max-step, 10 ms per Schur scaling).  Here the three expensive entry points are rebuilt from chip-wide pieces:
//
//   nestod_sdc  (src/ConicIP.jl:196-210)
//       chol(mat(z)), chol(mat(s))        the library's own blocked LDL' (ldlt.hip: MFMA trailing updates) on the matrix
//                                         padded to 256 / 512 with an identity block; L_chol = L D^1/2
//       G = Lz' Ls                        one fp64-MFMA GEMM (64x64 tiles over the chip)
//       svd(G): U, Lambda                 BLOCK one-sided Jacobi: column blocks of 32 (16 above r = 256), disjoint block
//                                         pairs on different workgroups (a pair = 128 KB of LDS), a full inner sweep per
//                                         pair, rounds separated by a grid barrier inside ONE persistent launch
//       R = Lz^-T U Lambda^1/2, R^-1      GEMMs with the explicit inverse of the unit-lower factor (the block inverses the
//                                         LDL' builds for its solves: one block = the whole matrix here)
//   maxstep_sdc (src/ConicIP.jl:272-303)
//       lambda_max(L^-1 D L^-T)           LDL' + two GEMMs with inv(L), then a COOPERATIVE Householder tridiagonalisation:
//                                         each of r/32 workgroups keeps a 32-column slab of the matrix in LDS for the whole
//                                         reduction; per column two grid barriers and two r-vectors through global memory
//                                         (coherent 8-byte accesses); Sturm multisection as before
//   A' F^-1 for the Schur complement (Block * matrix, src/blockmatrices.jl:176-177; src/kktsolvers.jl:289-290)
//       vecm(Rinv mat(a_i) Rinv') for all columns i: two BATCHED MFMA GEMMs per chunk of 64 columns
//
// apply / product / division stay on the workgroup-per-cone kernels (one or two r^3 GEMMs each, 0.1-0.3 ms at r = 256).
#include "cip_handle.h"
#include "../../include/cipkkt.h"
#include <math.h>
#include <vector>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>

#define LG_T 1024
#define LG_SQRT2 1.4142135623730951
#define LG_SQRT1_2 0.7071067811865476

__device__ int g_lz_hist[40];           // k_lg_lanczos1: histogram of step counts (bins of 8)
#ifdef LZ_TIMING
__device__ long g_lz_t[8];              // development: clocks per phase, summed over all calls (thread 0)
#define LZ_T0() long lz_t = __builtin_amdgcn_s_memtime()
#define LZ_T(i) do { if (threadIdx.x == 0) { const long t_ = __builtin_amdgcn_s_memtime(); g_lz_t[i] += t_ - lz_t; lz_t = t_; } } while (0)
#else
#define LZ_T0() do { } while (0)
#define LZ_T(i) do { } while (0)
#endif
#define LG_NPAD 4
struct LargeWs {
    int rp = 0;                    // padded order (256 or 512)
    int chunk = 64;                // columns of A per batched congruence
    double *base = nullptr;        // one allocation
    double *Kz, *Ks, *Tz, *Ts, *G, *M1, *M2, *M3;      // rp x rp
    double *Rip;                   // nlarge x LG_NPAD x rp x rp: Rinv, Rinv', R, R' of every large cone, zero padded
    double *vec;                   // 12 x rp: lam, dg, of, xbuf[2], pbuf, ...
    double *batchX, *batchT;       // chunk x rp x rp each
    // mat(a_i) of every column of A, per large cone, built once per upload (A does not change between the factorisations of a
    // problem): round 4 -- the strided pass over A' was 0.38 ms of every factorisation at order 256, n = 1024
    double *amat = nullptr;        // nlarge x ncols x rp x rp (null when that exceeds 4 GB or the columns do not fit one batch)
    int ncols = 0;
    const double *amat_src[CIP_MAX_LARGE_S] = {};   // the A' the images were built from (null: not built)
    unsigned *ctr;                 // barrier counters / sweep flags (256 words)
    // round 5: warm start of the one-sided Jacobi (see cip_sdp_large_nt): the right singular vectors of the previous NT scaling of
    // every large cone, and three work matrices
    double *Vw = nullptr;          // nlarge x rp x rp
    double *G0 = nullptr, *W1 = nullptr, *W2 = nullptr;
    int have_v[CIP_MAX_LARGE_S] = {};
    unsigned long long *vstate = nullptr;   // device, 2 words per large cone: [0] = 1: the stored V may be used; [1]: max |V'V - I| (bit pattern) of the current call
    int *hflag = nullptr, *hflag_dev = nullptr;   // host-mapped: {sweep flag, Cholesky flags, sequence number} of the Jacobi's read-backs
    int hseq = 0;
    void *ldl_z = nullptr, *ldl_s = nullptr;
    LdltWorkspace wz, ws;
    int xz_z = 0, xz_s = 0;        // the upper triangles of their block inverses have been zeroed (they stay zero)
    hipStream_t s2 = nullptr;      // the s-side max-step of a pair runs here, beside the v-side one on the handle's stream
    hipEvent_t efork = nullptr, ejoin = nullptr;
};

__device__ __forceinline__ int lg_vidx(int i, int j, int r) { return i * r - i * (i - 1) / 2 + (j - i); }   // i <= j
__device__ __forceinline__ long lg_rowoff(int i, int r) { return (long)i * r - (long)i * (i - 1) / 2; }       // vecm offset of (i, i)
__device__ __forceinline__ double lg_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lg_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// grid barrier for the few workgroups of a cooperative launch (all resident: <= 16 workgroups): monotonic counter,
// every thread drains its stores, one lane arrives and polls (`sc1` loads).  Bounded spin -> *err.
__device__ __forceinline__ void lg_grid_barrier(unsigned *ctr, unsigned nwg, unsigned &phase, int *err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++phase;
    if (threadIdx.x == 0) {
        atomicAdd(ctr, 1u);
        const long t0 = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg * phase) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memtime() - t0 > 2000000000L) { atomicExch(err, -9); break; }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ double lg_block_sum(double x, double *red) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    double s = 0.0;
    for (int q = 0; q < LG_T / 64; ++q) s += red[q];
    return s;
}

// ------------------------------------------------------------------------------------------ element-wise pieces
// X[b] (rp x rp) = mat(x_b) in the leading r x r block, pad: `padval` on the diagonal, 0 elsewhere.
// x_b = x + b * xb, element e at stride xs (a column block of A' is strided by its leading dimension).
__global__ __launch_bounds__(256) void k_lg_mat(const double *x, long xs, long xb, double *X, int r, int rp, double padval) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rp * rp) return;
    const int i = (int)(e % rp), j = (int)(e / rp);
    double v;
    if (i < r && j < r) {
        const int a = i < j ? i : j, b = i < j ? j : i;
        v = x[(long)blockIdx.y * xb + (long)lg_vidx(a, b, r) * xs];
        if (a != b) v *= LG_SQRT1_2;
    } else {
        v = (i == j) ? padval : 0.0;
    }
    X[(size_t)blockIdx.y * rp * rp + e] = v;
}
// out_b = vecm(Y_b leading r x r), strided like the input of k_lg_mat
__global__ __launch_bounds__(256) void k_lg_vecm(const double *Y, double *out, long os, long ob, int r, int rp) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)r * r) return;
    const int i = (int)(e % r), j = (int)(e / r);
    if (i > j) return;
    const double y = Y[(size_t)blockIdx.y * rp * rp + i + (long)j * rp];
    out[(long)blockIdx.y * ob + (long)lg_vidx(i, j, r) * os] = (i == j) ? y : y * LG_SQRT2;
}
// The same for a batch of nb matrices whose vecm images are the COLUMNS i0.. of a matrix with leading dimension `os` (W' = A'F^-1:
// element e of matrix i goes to out[i + e os]): 64 entries x 64 matrices per workgroup through LDS, so that both sides move whole
// cache lines -- entry (a, b) is read from the LOWER triangle, Y[b + a rp] (consecutive entries of a vecm row are consecutive
// there; the batched GEMM in front computes the lower tiles), and 64 consecutive i are written per entry.  (One thread per entry
// and matrix wrote 8 bytes every `os` doubles: 0.35 ms per factorisation at order 256, n = 1024.)
__global__ __launch_bounds__(256) void k_lg_vecm_cols(const double *Y, int nb, double *out, long os, int r, int rp) {
    __shared__ double tile[64][65];
    const long dim = (long)r * (r + 1) / 2;
    const long e0 = (long)blockIdx.x * 64;
    const int i0 = blockIdx.y * 64;
    {
        const int el = threadIdx.x & 63, cq = threadIdx.x >> 6;
        const long e = e0 + el;
        if (e < dim) {
            int a = (int)(((double)(2 * r + 1) - sqrt((double)(2 * r + 1) * (2 * r + 1) - 8.0 * (double)e)) * 0.5);
            while (a > 0 && lg_rowoff(a, r) > e) --a;
            while (lg_rowoff(a + 1, r) <= e) ++a;
            const int b = a + (int)(e - lg_rowoff(a, r));
            const double sc = (a == b) ? 1.0 : LG_SQRT2;
            for (int k = 0; k < 16; ++k) {
                const int il = cq + 4 * k;
                if (i0 + il < nb) tile[el][il] = Y[(size_t)(i0 + il) * rp * rp + b + (long)a * rp] * sc;
            }
        }
    }
    __syncthreads();
    {
        const int il = threadIdx.x & 63, eg = threadIdx.x >> 6;
        if (i0 + il < nb)
            for (int k = 0; k < 16; ++k) {
                const int el = eg + 4 * k;
                if (e0 + el < dim) out[(i0 + il) + (e0 + el) * os] = tile[el][il];
            }
    }
}
// out = vecm(Y + Y') of the leading r x r block (Jordan product X o Y = XY + YX from the one product XY)
__global__ __launch_bounds__(256) void k_lg_vecm_sym(const double *Y, double *out, int r, int rp) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)r * r) return;
    const int i = (int)(e % r), j = (int)(e / r);
    if (i > j) return;
    const double y = Y[i + (long)j * rp] + Y[j + (long)i * rp];
    out[lg_vidx(i, j, r)] = (i == j) ? y : y * LG_SQRT2;
}
// T = (L D^1/2)' as a dense matrix, from the factored K (unit L strictly below the diagonal, d separately):
//   T[i, k] = L[k, i] sqrt(d_i)  (k > i),  sqrt(d_i) on the diagonal, 0 below
// L is read from the LOWER triangle (round 5: the factorisation no longer mirrors the 128 x 128 diagonal blocks into the upper
// triangle -- the solves never read them there -- and this kernel was the one reader; a strided read of a <= 8 MB matrix)
__global__ __launch_bounds__(256) void k_lg_tfac(const double *K, const double *d, double *T, int rp) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rp * rp) return;
    const int i = (int)(e % rp), k = (int)(e / rp);
    const double sd = sqrt(d[i]);
    T[e] = (i < k) ? K[k + (long)i * rp] * sd : (i == k ? sd : 0.0);
}
// lam[j] = || G[:, j] ||   (one wave per column)
__global__ __launch_bounds__(256) void k_lg_colnorm(const double *G, double *lam, int rp) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= rp) return;
    double s = 0.0;
    for (int i = lane; i < rp; i += 64) { const double g = G[i + (long)j * rp]; s += g * g; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) lam[j] = sqrt(s);
}
// operands of the two closing GEMMs of nestod_sdc, from G = U diag(lam) (columns), d_z and the factored K_z:
//   U2t[j, k] = U[k, j] sqrt(lam_j) / sqrt(dz_k)        R    = inv(Lz_unit)' (U2t)'
//   U3t[i, k] = U[k, i] sqrt(dz_k) / sqrt(lam_i)        Rinv = U3t Lz_unit'
//   Lu        = Lz_unit as a dense lower-triangular matrix
__global__ __launch_bounds__(256) void k_lg_build(const double *G, const double *lam, const double *dz, const double *Kz,
                                                   double *U2t, double *U3t, double *Lu, int rp) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rp * rp) return;
    const int k = (int)(e % rp), j = (int)(e / rp);          // element (k, j) of G
    const double u = G[e] / lam[j];
    const double sl = sqrt(lam[j]), sd = sqrt(dz[k]);
    U2t[j + (long)k * rp] = u * sl / sd;
    U3t[j + (long)k * rp] = u * sd / sl;
    Lu[e] = (k > j) ? Kz[e] : (k == j ? 1.0 : 0.0);
}
// compact copies into the packed scaling (R, Rinv: r x r, pitch r) + zero-padded Rinv, Rinv', R, R' (pad[0..3]) for the
// chip-wide congruences; lambda = vecm(diag(lam)) (src/ConicIP.jl:735: lambda = F v = vecm(R' Z R))
__global__ __launch_bounds__(256) void k_lg_store(const double *Rp, const double *Rip_full, double *R, double *Ri, double *pad,
                                                   const double *lam, double *lambda, int r, int rp, int kdim) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < (long)rp * rp) {
        const int i = (int)(e % rp), j = (int)(e / rp);
        const bool in = i < r && j < r;
        const double rv = in ? Rp[e] : 0.0, iv = in ? Rip_full[e] : 0.0;
        if (in) { R[i + (long)j * r] = rv; Ri[i + (long)j * r] = iv; }
        const long n2 = (long)rp * rp, et = j + (long)i * rp;
        pad[e] = iv; pad[n2 + et] = iv; pad[2 * n2 + e] = rv; pad[3 * n2 + et] = rv;
    }
    if (lambda && e < kdim) {
        // diagonal positions of the row-major upper triangle: i r - i (i - 1) / 2
        lambda[e] = 0.0;
    }
}
__global__ __launch_bounds__(256) void k_lg_lambda_diag(const double *lam, double *lambda, int r) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < r) lambda[lg_vidx(i, i, r)] = lam[i];
}
// the four padded matrices from a packed scaling handed over by the host (cip_set_scaling_packed / identity)
__global__ __launch_bounds__(256) void k_lg_pad(const double *R, const double *Ri, double *pad, int r, int rp) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rp * rp) return;
    const int i = (int)(e % rp), j = (int)(e / rp);
    const bool in = i < r && j < r;
    const double rv = in ? R[i + (long)j * r] : 0.0, iv = in ? Ri[i + (long)j * r] : 0.0;
    const long n2 = (long)rp * rp, et = j + (long)i * rp;
    pad[e] = iv; pad[n2 + et] = iv; pad[2 * n2 + e] = rv; pad[3 * n2 + et] = rv;
}
// a non-positive pivot of either Cholesky = the iterate has left the cone: same flag the small-cone kernels raise
__global__ void k_lg_flag(const int *info_a, const int *info_b, int *flag) {
    if (threadIdx.x == 0) {
        if (info_a[0]) *flag = info_a[0];
        else if (info_b && info_b[0]) *flag = info_b[0];
    }
}

// sum over a group of tpp = 32 or 64 consecutive lanes (aligned), in every lane: DPP + lane swaps, no LDS permutes
__device__ __forceinline__ double lg_sum_group(double x, int tpp) {
    x = lz_sum16(x);
    if (tpp == 16) return x;
    return tpp == 64 ? lz_sum_rows(x) : lz_sum_row_pair(x);
}
// ------------------------------------------------------------------------------------------ block one-sided Jacobi
__host__ __device__ inline int lg_pitch(int rows) { return rows + ((4 - rows % 32) + 32) % 32; }   // == 4 (mod 32) doubles
__device__ __forceinline__ double lg_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double lg_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    return r;
}
// nbk / 2 workgroups, every column pair rotated exactly once per sweep (cyclic Jacobi), ONE PHASE PER LAUNCH:
//   step 0              workgroup k holds blocks 2k, 2k+1 in LDS: b - 1 rounds of the round-robin tournament inside
//                       each block (b / 2 + b / 2 disjoint pairs per round)
//   step t + 1          round t of the tournament over blocks (nbk - 1 rounds): the workgroup holds its block pair (I, J) and
//                       rotates the b x b cross pairs in b rounds of b disjoint pairs (p in I with p + it in J)
// The launch boundary orders the phases: 255 rotation rounds per sweep at r = 256 -- the depth of the plain parallel ordering.
// (Rounds 3-5 also had ONE persistent launch per NT scaling with a grid barrier per phase and the blocks handed from workgroup to
//  workgroup through `sc1` stores + a counter: about one scaling in 800 / 4000 / 40000 at order 1024 / 512 / 256 came out with other
//  bits and the cause was never found -- DESIGN_LOG.md, round 5.  Round 6 removed that form and its switch: a library must not offer
//  a mode whose documented effect is other bits once in a while.)
// rp / 8 lanes per pair, both columns in registers between the dot products and the rotation.  A sweep behind one that found no
// rotation with cos^2 >= 1e-16 returns at once (the next sweep would find nothing above the 1e-30 threshold: quadratic convergence).
// G_in V = U diag(sigma): column i ends as sigma_i u_i, all of svd(Lz' Ls) that nestod_sdc uses (src/ConicIP.jl:204-208).
// EPL: elements of a column per lane (rp == EPL * tpp): 8 up to order 512; 16 at order 1024 (round 4), where 64 lanes hold a column
template <int NT, int EPL = 8>
__global__ __launch_bounds__(NT) void k_lg_jacobi(double *G, int rp, int bflags, unsigned *ctr) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int b = bflags & 0xff;
    const int step_sweep = (bflags >> 16) & 0x3f, step = (bflags >> 22) & 0x1ff;
    __shared__ int s_rot;
    __shared__ double s_nrm[32];
    const int tid = threadIdx.x;
    const int nbk = rp / b, m = nbk;
    const int ld = lg_pitch(rp);
    const int tpp = NT / b;                              // lanes per column pair (32 or 64); rp == EPL * tpp
    const int part = tid % tpp, pair = tid / tpp;
    unsigned *sweepflag = ctr + 8;
    // the host enqueues sweeps ahead of reading their flags; a sweep behind one that found nothing to rotate is a no-op
    if (step_sweep > 0 && __hip_atomic_load(sweepflag + step_sweep - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    // rotation of LDS columns p, q; returns 0 / 1 (rotated, small) / 2 (rotated, cos^2 >= 1e-16)
    auto rotate = [&](int p, int q) -> int {
        double *gp = sh + p * ld, *gq = sh + q * ld;
        double xv[EPL], yv[EPL], a = 0.0, bb = 0.0, c = 0.0;
#pragma unroll
        for (int u = 0; u < EPL; ++u) { xv[u] = gp[part + u * tpp]; yv[u] = gq[part + u * tpp]; }
#pragma unroll
        for (int u = 0; u < EPL; ++u) { a += xv[u] * xv[u]; bb += yv[u] * yv[u]; c += xv[u] * yv[u]; }
        a = lg_sum_group(a, tpp); bb = lg_sum_group(bb, tpp); c = lg_sum_group(c, tpp);
        if (!(c * c > 1e-30 * (a * bb) && c != 0.0)) return 0;
        const double zeta = (bb - a) * 0.5 * lg_rcp(c);
        const double h2 = 1.0 + zeta * zeta;
        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) * lg_rcp(fabs(zeta) + h2 * lg_rsqrt(h2));
        const double cs = lg_rsqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            gp[part + u * tpp] = cs * xv[u] - sn * yv[u];
            gq[part + u * tpp] = sn * xv[u] + cs * yv[u];
        }
        return (c * c > 1e-16 * (a * bb)) ? 2 : 1;
    };
    // a block pair is 2 b rp = 16384 doubles whatever (b, rp): 16 per thread.  All loads of a thread are issued before the first
    // LDS store (one at a time, store after load, each global round trip was exposed: 16 x ~1.5 us per outer round -- half of a
    // sweep's time).  16-byte buffer accesses with the `sc1` policy bit: a relic of the in-launch block exchange (the blocks now
    // change hands across a launch boundary, where plain accesses would do); kept because the one-launch-per-phase form was
    // measured, repeated 11 500 times and pinned bit for bit with exactly these instructions.
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)G, 0, 0x7fffffff, 0x00020000);
    auto load = [&](int bp, int bq) {                      // coherent loads: other workgroups wrote these blocks
        v4i_t t[EPL];
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int e = 2 * (tid + u * NT), i = e % rp, c = e / rp;
            t[u] = __builtin_amdgcn_raw_buffer_load_b128(grs, (int)((i + (long)(c < b ? bp * b + c : bq * b + (c - b)) * rp) * 8), 0, 16);
        }
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int e = 2 * (tid + u * NT);
            *(v2d *)(sh + e % rp + (e / rp) * ld) = __builtin_bit_cast(v2d, t[u]);
        }
        if (tid == 0) s_rot = 0;
        __syncthreads();
    };
    auto store = [&](int bp, int bq) {
        v2d t[EPL];
#pragma unroll
        for (int u = 0; u < EPL; ++u) { const int e = 2 * (tid + u * NT); t[u] = *(const v2d *)(sh + e % rp + (e / rp) * ld); }
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int e = 2 * (tid + u * NT), i = e % rp, c = e / rp;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, t[u]), grs, (int)((i + (long)(c < b ? bp * b + c : bq * b + (c - b)) * rp) * 8), 0, 16);
        }
    };
    int rotated = 0;
    if (step == 0) {
        // ---- pairs inside the blocks
        const int bp = 2 * (int)blockIdx.x, bq = bp + 1;
        load(bp, bq);
        const int hb = b / 2, blk = pair / hb, kk = pair % hb;
        for (int it = 0; it < b - 1; ++it) {
            int p, q;
            if (kk == 0) { p = b - 1; q = it; }
            else { p = (it + kk) % (b - 1); q = (it - kk + (b - 1)) % (b - 1); }
            const int rr = rotate(blk * b + p, blk * b + q);
            if (rr && part == 0) atomicMax(&s_rot, rr);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        rotated = s_rot > rotated ? s_rot : rotated;
        store(bp, bq);
    } else {
        // ---- pairs across blocks: round t = step - 1 of the tournament
        const int t = step - 1;
        int bp, bq;
        const int k = blockIdx.x;
        if (k == 0) { bp = m - 1; bq = t; }
        else { bp = (t + k) % (m - 1); bq = (t - k + (m - 1)) % (m - 1); }
        load(bp, bq);
        {
            // column p = `pair` of block I stays with this lane group for all b rounds: it lives in registers, its
            // squared norm `a` and the partners' (s_nrm, in LDS) follow the rotations (a' = a - t c, b' = b + t c)
            // instead of being re-summed -- a round is then one dot product, 8 LDS loads and 8 stores per lane
            double *gp = sh + pair * ld;
            double xv[EPL], a = 0.0;
#pragma unroll
            for (int u = 0; u < EPL; ++u) { xv[u] = gp[part + u * tpp]; a += xv[u] * xv[u]; }
            {
                double *gq0 = sh + (b + pair) * ld;
                double bq2 = 0.0;
#pragma unroll
                for (int u = 0; u < EPL; ++u) { const double y = gq0[part + u * tpp]; bq2 += y * y; }
                a = lg_sum_group(a, tpp); bq2 = lg_sum_group(bq2, tpp);
                if (part == 0) s_nrm[pair] = bq2;
            }
            __syncthreads();
            for (int it = 0; it < b; ++it) {
                const int qi = (pair + it) % b;
                double *gq = sh + (b + qi) * ld;
                double yv[EPL], c = 0.0;
#pragma unroll
                for (int u = 0; u < EPL; ++u) { yv[u] = gq[part + u * tpp]; c += xv[u] * yv[u]; }
                c = lg_sum_group(c, tpp);
                const double bb = s_nrm[qi];
                if (c * c > 1e-30 * (a * bb) && c != 0.0) {
                    const double zeta = (bb - a) * 0.5 * lg_rcp(c);
                    const double h2 = 1.0 + zeta * zeta;
                    const double tt = (zeta >= 0.0 ? 1.0 : -1.0) * lg_rcp(fabs(zeta) + h2 * lg_rsqrt(h2));
                    const double cs = lg_rsqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
                    for (int u = 0; u < EPL; ++u) {
                        const double x = xv[u];
                        xv[u] = cs * x - sn * yv[u];
                        gq[part + u * tpp] = sn * x + cs * yv[u];
                    }
                    if (part == 0) { s_nrm[qi] = bb + tt * c; atomicMax(&s_rot, (c * c > 1e-16 * (a * bb)) ? 2 : 1); }
                    a -= tt * c;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
#pragma unroll
            for (int u = 0; u < EPL; ++u) gp[part + u * tpp] = xv[u];
            __syncthreads();
        }
        rotated = s_rot > rotated ? s_rot : rotated;
        store(bp, bq);
    }
    if (tid == 0 && rotated == 2) atomicAdd(sweepflag + step_sweep, 1u);
}

// ------------------------------------------------------------------------------------------ cooperative tridiagonalisation
// A = the symmetric r x r matrix  0.5 (M + M') .* (rs rs')  (rs == NULL: no scaling), column slab [32 w, 32 w + 32) in the
// LDS of workgroup w for the whole reduction.  Output: dg[0..r), of[0..r-1) (sub-diagonal).  Householder by columns
// exactly as sd_extreme_eig (sdp.hip) does it in one workgroup.
#define LG_SLAB 32
// ONE grid barrier per column: beside its entries of p = beta A v every step also publishes the NOT YET UPDATED next
// column (by its owner), so that after the barrier every workgroup can form the updated next column -- the Householder
// vector of the following step -- by itself:  a[:, k+1] - v w_0 - w v_0.
__global__ __launch_bounds__(LG_T) void k_lg_tridiag(const double *M, int ldm, const double *dscale, int r, double *dg, double *of,
                                                      double *xbuf /* 2 x ldm */, double *pbuf /* 2 x ldm */, unsigned *ctr, int *err, int slab) {
    extern __shared__ double sh[];
    const int tid = threadIdx.x;
    const int nwg = gridDim.x, w = blockIdx.x;
    const int c0 = w * slab, ncol = (r - c0 < slab) ? (r - c0) : slab;
    const int ld = r | 1;
    double *a = sh;                                  // slab: a[i + c * ld]
    double *v = sh + (size_t)slab * ld, *wv = v + r, *vn = wv + r, *red = vn + r;
    unsigned phase = 0;
    for (int e = tid; e < ncol * r; e += LG_T) {
        const int i = e % r, c = e / r, j = c0 + c;
        double x = 0.5 * (M[i + (long)j * ldm] + M[j + (long)i * ldm]);
        if (dscale) x *= rsqrt(dscale[i]) * rsqrt(dscale[j]);
        a[i + c * ld] = x;
    }
    __syncthreads();
    // column `col` below its diagonal, as it stands in the owner's slab -> xb
    auto publish = [&](int col, double *xb) {
        if (col / slab == w) {
            const int cc = col - c0, mm = r - col - 1;
            for (int i = tid; i < mm; i += LG_T) lg_st(xb + i, a[(col + 1 + i) + cc * ld]);
        }
    };
    publish(0, xbuf);
    lg_grid_barrier(ctr, nwg, phase, err);
    for (int i = tid; i < r - 1; i += LG_T) v[i] = lg_ld(xbuf + i);
    __syncthreads();
    for (int k = 0; k + 1 < r; ++k) {
        const int m = r - k - 1;
        const int owner = k / slab, kc = k - owner * slab;
        double *xb = xbuf + (size_t)((k + 1) & 1) * ldm, *pb = pbuf + (size_t)(k & 1) * ldm;
        if (w == owner && tid == 0) dg[k] = a[k + kc * ld];
        double part = 0.0;
        for (int i = tid; i < m; i += LG_T) part += v[i] * v[i];
        const double sigma = lg_block_sum(part, red);
        const double x0 = v[0];
        if (m == 1) { if (w == owner && tid == 0) of[k] = x0; break; }
        const bool flat = !(sigma - x0 * x0 > 0.0);          // already tridiagonal in this column (uniform over the grid)
        double alpha = x0, beta = 0.0;
        if (!flat) {
            alpha = -copysign(sqrt(sigma), x0);
            beta = 1.0 / (sigma - x0 * alpha);               // 2 / ||v||^2 with v = x - alpha e1
        }
        __syncthreads();
        if (tid == 0) { if (!flat) v[0] = x0 - alpha; if (w == owner) of[k] = alpha; }
        __syncthreads();
        if (!flat) {
            // p_j = beta sum_i A[i, j] v_i for the slab's columns j > k: 32 lanes per column
            const int c = tid >> 5, l32 = tid & 31, j = c0 + c;
            if (c < ncol && j > k) {
                double acc = 0.0;
                const double *col = a + (k + 1) + c * ld;
                for (int i = l32; i < m; i += 32) acc += col[i] * v[i];
                for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                if (l32 == 0) lg_st(pb + (j - k - 1), beta * acc);
            }
        }
        publish(k + 1, xb);                                  // the next column BEFORE this step's update
        lg_grid_barrier(ctr, nwg, phase, err);
        if (flat) {
            for (int i = tid; i < m - 1; i += LG_T) v[i] = lg_ld(xb + i);
            __syncthreads();
            continue;
        }
        part = 0.0;
        for (int i = tid; i < m; i += LG_T) { const double p = lg_ld(pb + i); wv[i] = p; part += p * v[i]; }
        const double kk = 0.5 * beta * lg_block_sum(part, red);
        for (int i = tid; i < m; i += LG_T) wv[i] -= kk * v[i];
        __syncthreads();
        // A22 -= v w' + w v' on the slab's columns j > k
        for (int e = tid; e < ncol * m; e += LG_T) {
            const int i = e % m, c = e / m, j = c0 + c;
            if (j > k) a[(k + 1 + i) + c * ld] -= v[i] * wv[j - k - 1] + wv[i] * v[j - k - 1];
        }
        // the updated column k+1 below its diagonal = next step's x, formed by everybody
        for (int i = tid; i < m - 1; i += LG_T) vn[i] = lg_ld(xb + i) - (v[i + 1] * wv[0] + wv[i + 1] * v[0]);
        __syncthreads();
        double *t = v; v = vn; vn = t;
    }
    if ((r - 1) / slab == w && tid == 0) { dg[r - 1] = a[(r - 1) + ((r - 1) - c0) * ld]; of[r - 1] = 0.0; }
}

// extreme eigenvalue of the tridiagonal (dg, of) by multisection on the Sturm count (one workgroup), then the max-step
// verdict of maxstep_sdc (src/ConicIP.jl:272-303): partial[item] <- Inf / 1/(scale lambda_max) / the `nothing` variant
__global__ __launch_bounds__(LG_T) void k_lg_sturm(const double *dgg, const double *ofg, int r, int want_max, double scale,
                                                    const int *info, double *partial, int item, const int *gate = nullptr) {
    if (gate && !gate[0]) return;
    __shared__ double dg[2048], of[2048], red[64];
    __shared__ int first;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double INF = __builtin_inf();
    if (info && info[0]) {                                   // X not positive definite -> Inf (:277-280)
        if (tid == 0) partial[item] = INF;
        return;
    }
    for (int i = tid; i < r; i += LG_T) { dg[i] = dgg[i]; of[i] = ofg[i]; }
    __syncthreads();
    double lo = INF, hi = -INF;
    for (int i = tid; i < r; i += LG_T) {
        const double rad = (i > 0 ? fabs(of[i - 1]) : 0.0) + (i + 1 < r ? fabs(of[i]) : 0.0);
        lo = fmin(lo, dg[i] - rad);
        hi = fmax(hi, dg[i] + rad);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
    if (lane == 0) { red[wave] = lo; red[16 + wave] = hi; }
    __syncthreads();
    lo = red[0]; hi = red[16];
    for (int q = 1; q < LG_T / 64; ++q) { lo = fmin(lo, red[q]); hi = fmax(hi, red[16 + q]); }
    const double span = fmax(fabs(lo), fabs(hi));
    hi += 1e-15 * span + 1e-300;
    lo -= 1e-15 * span + 1e-300;
    for (int round = 0; round < 12; ++round) {
        if (!(hi - lo > 4.4e-16 * fmax(fabs(lo), fabs(hi)))) break;
        const double step = (hi - lo) / (LG_T + 1);
        const double xs = lo + step * (tid + 1);
        int cnt = 0;
        double q = dg[0] - xs;
        if (q < 0.0) ++cnt;
        for (int i = 1; i < r; ++i) {
            if (q == 0.0) q = 1e-300;
            q = (dg[i] - xs) - of[i - 1] * of[i - 1] / q;
            if (q < 0.0) ++cnt;
        }
        const bool hit = want_max ? (cnt >= r) : (cnt >= 1);
        if (tid == 0) first = LG_T;
        __syncthreads();
        if (hit) atomicMin(&first, tid);
        __syncthreads();
        const int f = first;
        __syncthreads();
        const double nlo = (f == 0) ? lo : lo + step * f;
        const double nhi = (f == LG_T) ? hi : lo + step * (f + 1);
        lo = nlo; hi = nhi;
    }
    const double ev = 0.5 * (lo + hi);
    if (tid == 0) {
        if (want_max) { const double mx = ev * scale; partial[item] = (mx < 0.0) ? INF : 1.0 / mx; }
        else partial[item] = (ev > 0.0) ? 0.0 : -1.0 + ev;
    }
}

// ------------------------------------------------------------------------------------------ host side
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
int cip_sdp_large_padded(int r) { return r <= 256 ? 256 : (r <= 512 ? 512 : (r <= 1024 ? 1024 : 2048)); }

int cip_sdp_large_create(int rmax_large, int nlarge, int ncols, LargeWs **out) {
    LargeWs *w = new LargeWs();
    const int rp = cip_sdp_large_padded(rmax_large);
    w->rp = rp;
    const size_t m2 = al256((size_t)rp * rp * 8);
    // columns of A per batched congruence: all n of them when the two images fit 2 GB each (round 4: at order 256, n = 1024 the
    // 16 chunks of 64 columns were 64 launches of 512 tiles -- two workgroups per CU, 40 TFLOP/s -- now two launches), at least 16
    {
        const size_t cap = ((size_t)2 << 30) / m2;
        int c = ncols > 0 ? ncols : 64;
        if ((size_t)c > cap) c = (int)cap;
        if (c < 16) c = 16;
        if (const char *e = getenv("CIP_LG_CHUNK")) if (atoi(e) >= 1) c = atoi(e);
        w->chunk = c;
    }
    w->ncols = ncols;
    const bool cache_mat = ncols > 0 && w->chunk >= ncols && (size_t)nlarge * ncols * m2 <= ((size_t)4 << 30) && !(getenv("CIP_LG_AMAT") && atoi(getenv("CIP_LG_AMAT")) == 0);
    size_t bytes = (cache_mat ? (size_t)nlarge * ncols * m2 : 0) + 8 * m2 + (3 + (size_t)nlarge) * m2 + LG_NPAD * (size_t)nlarge * m2 + al256(12 * (size_t)rp * 8) + 2 * (size_t)w->chunk * m2 + al256(1024) +
                   al256(16 * (size_t)nlarge);
    // the two LDL' workspaces: solve block = the whole padded matrix (X = inv(L_unit) in one piece: the "triangular solves" here are
    // GEMMs with it), also at order 2048 (the calling thread's solve-block limit for the sizing and the carve)
    struct SolveBlockOverride {          // restores the thread's limit on every way out
        int saved;
        explicit SolveBlockOverride(int b) : saved(cip_tl_solve_block_max) { cip_tl_solve_block_max = b; }
        ~SolveBlockOverride() { cip_tl_solve_block_max = saved; }
    } whole_matrix_block(rp);
    bytes += 2 * al256(cip_ldlt_ws_bytes(rp, 0));
    if (hipMalloc((void **)&w->base, bytes) != hipSuccess) { cip_set_error("large S cone workspace: hipMalloc of %zu bytes failed", bytes); delete w; return -3; }
    char *p = (char *)w->base;
    double **mats[8] = {&w->Kz, &w->Ks, &w->Tz, &w->Ts, &w->G, &w->M1, &w->M2, &w->M3};
    for (auto m : mats) { *m = (double *)p; p += m2; }
    w->G0 = (double *)p; p += m2; w->W1 = (double *)p; p += m2; w->W2 = (double *)p; p += m2;
    w->Vw = (double *)p; p += (size_t)nlarge * m2;
    w->Rip = (double *)p; p += LG_NPAD * (size_t)nlarge * m2;
    w->vec = (double *)p; p += al256(12 * (size_t)rp * 8);
    w->batchX = (double *)p; p += (size_t)w->chunk * m2;
    w->batchT = (double *)p; p += (size_t)w->chunk * m2;
    if (cache_mat) { w->amat = (double *)p; p += (size_t)nlarge * ncols * m2; }
    w->ctr = (unsigned *)p; p += al256(1024);
    CIP_HIP_CHECK(hipMemset(w->ctr, 0, 1024));
    w->vstate = (unsigned long long *)p; p += al256(16 * (size_t)nlarge);
    CIP_HIP_CHECK(hipMemset(w->vstate, 0, 16 * (size_t)nlarge));
    if (hipHostMalloc((void **)&w->hflag, 8 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void **)&w->hflag_dev, w->hflag, 0) != hipSuccess) {
        cip_set_error("large S cone workspace: host-mapped flag words"); if (w->hflag) (void)hipHostFree(w->hflag); (void)hipFree(w->base); delete w; return -3;
    }
    memset(w->hflag, 0, 8 * sizeof(int));
    CIP_HIP_CHECK(hipMemset(w->vec, 0, 12 * (size_t)rp * 8));
    w->ldl_z = p; p += al256(cip_ldlt_ws_bytes(rp, 0));
    w->ldl_s = p;
    cip_ldlt_ws_carve(w->ldl_z, rp, &w->wz, 0);
    cip_ldlt_ws_carve(w->ldl_s, rp, &w->ws, 0);
    w->wz.signs = w->ws.signs = PivotSigns{0, rp, rp};       // a Cholesky in disguise: every pivot must be positive
    w->wz.x_zeroed = &w->xz_z; w->ws.x_zeroed = &w->xz_s;
    *out = w;
    return 0;
}
void cip_lg_cks_dump(void);
void cip_sdp_large_destroy(LargeWs *w) {
    if (!w) return;
    cip_lg_cks_dump();
    if (w->hflag) (void)hipHostFree(w->hflag);
    if (w->s2) (void)hipStreamDestroy(w->s2);
    if (w->efork) (void)hipEventDestroy(w->efork);
    if (w->ejoin) (void)hipEventDestroy(w->ejoin);
    if (const char *e = getenv("CIP_LG_LANCZOS_STATS")) {
        if (atoi(e)) {
            int st[40];
            if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_lz_hist), sizeof(st)) == hipSuccess) {
                fprintf(stderr, "lanczos max-step, steps per call (bins of 8):");
                for (int q = 0; q <= 32; ++q) if (st[q]) fprintf(stderr, " [%d-%d]: %d", 8 * q, 8 * q + 7, st[q]);
                fprintf(stderr, "\n");
            }
#ifdef LZ_TIMING
            long tt[8];
            if (hipMemcpyFromSymbol(tt, HIP_SYMBOL(g_lz_t), sizeof(tt)) == hipSuccess)
                fprintf(stderr, "lanczos clocks: load+v1 %ld | A v %ld | alpha %ld | dots %ld | update %ld | beta %ld | multisection %ld | rest of check %ld\n", tt[0], tt[1], tt[2], tt[3], tt[4], tt[5], tt[7], tt[6]);
#endif
        }
    }
    if (w->base) (void)hipFree(w->base);
    delete w;
}

static dim3 lg_grid(long n) { return dim3((unsigned)((n + 255) / 256)); }
// C_b = A_b B_b'  (rp x rp each, 64x64 fp64-MFMA tiles); stride 0 = operand shared by the batch
// ONE product C = A B' of order rp <= 256 (the congruences of apply / max-step / NT scaling: ~60 per iteration of config 4).  The
// 64x64-tile kernel has 16 workgroups for it and walks K = 256 in each: 14 us on 16 of 256 CUs.  Here a workgroup owns one 16x16
// tile of C (256 workgroups at order 256) and its four waves split the k range; operands go from global memory (L2: 0.5 MB per
// matrix) straight into the MFMA lanes -- lane l supplies row l % 16, k = l / 16 of its operand tile -- no LDS staging; the three
// partial accumulators of waves 1..3 are added to wave 0's in a fixed order.  Register q of lane l holds C[i0 + l % 16, j0 + l / 16 + 4 q]
// (the operand order of gemm_tile_64: B's rows first).
// BVEC: B is mat(xv) of a vecm vector (symmetric; zero outside the leading r x r block), read straight from the vector -- no k_lg_mat
// pass; CVEC: the result goes out as vecm (entries i <= j < r, off-diagonal ones times sqrt 2; tiles below the diagonal are not
// computed) -- no k_lg_vecm pass.  Same MFMAs on the same operands in the same order as the plain form: same bits.
template <bool BVEC, bool CVEC>
__global__ __launch_bounds__(256) void k_gemm_nt_small(const double *A, long lda, const double *B, long ldb, double *C, long ldc, int K, int r) {
    __shared__ double red[3][4][64];
    if (CVEC && blockIdx.x > blockIdx.y) return;               // (16 i0 > 16 j0 + 15 for every entry of the tile)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int kq = K >> 2;
    const double *a = A + (long)blockIdx.x * 16 + l15 + (long)(wave * kq + l4) * lda;
    const double *b = B + (long)blockIdx.y * 16 + l15 + (long)(wave * kq + l4) * ldb;
    const int jb = blockIdx.y * 16 + l15, kb0 = wave * kq + l4;      // BVEC: this lane's row of mat(xv) and its first k
    v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
    for (int k = 0; k < kq; k += 32) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = a[(long)(k + 4 * u) * lda];
            if (BVEC) {
                const int kk = kb0 + k + 4 * u;
                double v = 0.0;
                if (jb < r && kk < r) {
                    const int lo = jb < kk ? jb : kk, hi = jb < kk ? kk : jb;
                    v = B[lg_vidx(lo, hi, r)];
                    if (lo != hi) v *= LG_SQRT1_2;
                }
                bv[u] = v;
            } else {
                bv[u] = b[(long)(k + 4 * u) * ldb];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[u], av[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[u + 1], av[u + 1], acc1, 0, 0, 0);
        }
    }
    acc0 += acc1;
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave - 1][q][lane] = acc0[q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double c = ((acc0[q] + red[0][q][lane]) + red[1][q][lane]) + red[2][q][lane];
            const int i = blockIdx.x * 16 + l15, j = blockIdx.y * 16 + l4 + 4 * q;
            if (CVEC) { if (i <= j && j < r) C[lg_vidx(i, j, r)] = (i == j) ? c : c * LG_SQRT2; }
            else C[i + (long)j * ldc] = c;
        }
    }
}
// part 2: only the 64-tiles that touch i <= j are computed (the others keep what C held); 3: only those that touch i >= j
static int lg_small_gemm(void) {
    static const int small = [] { const char *e = getenv("CIP_LG_SMALLGEMM"); return e ? atoi(e) : 1; }();
    return small;
}
static int lg_gemm(hipStream_t s, double *C, long sC, const double *A, long sA, const double *B, long sB, int rp, int batch, int part = 0) {
    if (batch == 1 && part == 0 && rp <= 256 && lg_small_gemm()) {
        hipLaunchKernelGGL((k_gemm_nt_small<false, false>), dim3(rp / 16, rp / 16), dim3(256), 0, s, A, (long)rp, B, (long)rp, C, (long)rp, rp, rp);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    GemmArgs g = {};
    g.A = A; g.lda = rp; g.B = B; g.ldb = rp; g.C = C; g.ldc = rp;
    g.M = g.N = g.K = rp; g.alpha = 1.0; g.overwrite = 1; g.by = batch; g.bz = 1;
    g.lower = part;
    g.sAy = sA; g.sBy = sB; g.sCy = sC;
    return cip_launch_gemm(s, EPI_ACCUM, g);
}
static int lg_set_attr(const void *fn, size_t bytes) {
    CIP_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
// ---- Householder tridiagonalisation of a symmetric matrix of order r <= 256 by ONE workgroup, the matrix in REGISTERS.
// The cooperative kernel above pays a grid barrier and two coherent global round trips per column (5.7 us x 255 columns
// at r = 256: 76 % of a max-step).  A CU's register file holds what its LDS cannot: 512 threads (two waves per SIMD: 256
// registers each) as a 32 x 16 grid, thread (tr, tc) owning the entries (i, j) with i = tr (mod 32), j = tc (mod 16) of the
// blocks on or below the block diagonal -- 72 of the 128 blocks of 32 x 16, the 32 x 32 diagonal blocks whole -- in 72
// doubles.  A Householder step is then
//   column k -> LDS, norm, v (zero outside the active rows), p = A v as 8 row + 16 column partial sums per thread reduced
//   through two LDS images (row parts over tc, column parts over tr), w = p - (beta v'p / 2) v, rank-2 update from LDS
//   copies of v and w (zero-padded: finished rows and columns take no part, no index tests in the inner loops)
// with workgroup barriers only.  Same reflectors as k_lg_tridiag; d and e go to the same Sturm kernel.
// Measured at r = 256: 1.17 ms against 1.45 ms for the cooperative kernel (max-step 1.45 against 1.75 ms).  Phases switched off
// one at a time: barriers + norm + scalars 1.3 us per column, rank-2 update 1.4, A v 1.0, the reduction of its partial sums 0.7,
// column extraction 0.15 -- the arithmetic of a column (2.6e5 flop) is now one CU's (0.85 + 0.43 us at its fp64 rate), which
// is the floor of this form.  (A 1024-thread version with 36 doubles per thread was built first: at four waves per SIMD the
// budget is 128 registers, 23 of the 36 spilled, 22 us per column; and `break` / `continue` inside the column loop made the
// compiler spill the register matrix around the exits even at 256 registers: 16 us per column.)
#define T1_LDS_DOUBLES (256 * 17 + 32 * 256 + 4 * 256 + 64)     // row parts, column parts, 2 column buffers, v, p, reduction scratch
template <int KB>
__device__ __forceinline__ void t1_extract(const double (&a)[8][16], double *xs, int tr) {      // column block KB (16 wide)
#pragma unroll
    for (int ai = KB / 2; ai < 8; ++ai) xs[32 * ai + tr] = a[ai][KB];
}
__global__ __launch_bounds__(512) void k_lg_tridiag1(const double *M, int ldm, const double *dscale, int r, double *dg, double *of, const int *gate) {
    extern __shared__ double sh[];
    if (gate && !gate[0]) return;                              // (the fallback behind a passed inertia certificate)
    double *rowp = sh, *colp = sh + 256 * 17;                 // partial sums: [i][tc] (pitch 17) and [tr][j]
    double *xs0 = colp + 32 * 256, *vs = xs0 + 512, *red = vs + 512;
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    double a[8][16];                                           // blocks bj <= 2 ai + 1 only
#pragma unroll
    for (int ai = 0; ai < 8; ++ai)
#pragma unroll
        for (int bj = 0; bj < 16; ++bj) {
            if (bj <= 2 * ai + 1) {
                a[ai][bj] = 0.0;
                const int i = 32 * ai + tr, j = 16 * bj + tc;
                if (i < r && j < r) {
                    double x = 0.5 * (M[i + (long)j * ldm] + M[j + (long)i * ldm]);
                    if (dscale) x *= rsqrt(dscale[i]) * rsqrt(dscale[j]);
                    a[ai][bj] = x;
                }
            }
        }
    double *ps = vs + 256;                                     // p, read back by everybody
    // columns 0 .. r-2 (the last one only hands out d and e: its "reflector" is 1 x 1).  Five barriers per column; the column
    // buffer alternates, so the next column's extraction may start while slower waves still read this one's
    for (int k = 0; k + 1 < r; ++k) {
        const int kb = k >> 4, kc = k & 15;
        double *xs = xs0 + (k & 1) * 256;
        // ---- column k -> xs (rows of the stored blocks: everything below the diagonal, and d_k); readers mask by index
        if (tc == kc) {
            switch (kb) {
                case 0: t1_extract<0>(a, xs, tr); break;
                case 1: t1_extract<1>(a, xs, tr); break;
                case 2: t1_extract<2>(a, xs, tr); break;
                case 3: t1_extract<3>(a, xs, tr); break;
                case 4: t1_extract<4>(a, xs, tr); break;
                case 5: t1_extract<5>(a, xs, tr); break;
                case 6: t1_extract<6>(a, xs, tr); break;
                case 7: t1_extract<7>(a, xs, tr); break;
                case 8: t1_extract<8>(a, xs, tr); break;
                case 9: t1_extract<9>(a, xs, tr); break;
                case 10: t1_extract<10>(a, xs, tr); break;
                case 11: t1_extract<11>(a, xs, tr); break;
                case 12: t1_extract<12>(a, xs, tr); break;
                case 13: t1_extract<13>(a, xs, tr); break;
                case 14: t1_extract<14>(a, xs, tr); break;
                default: t1_extract<15>(a, xs, tr); break;
            }
        }
        __syncthreads();                                                                        // (1)
        if (tid == 0) dg[k] = xs[k];
        // ---- sigma = |x(k+1:)|^2
        double part = 0.0;
        if (tid < 256 && tid > k && tid < r) part = xs[tid] * xs[tid];
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (tid < 256 && (tid & 63) == 0) red[(k & 1) * 4 + (tid >> 6)] = part;
        __syncthreads();                                                                        // (2)
        const double *rs = red + (k & 1) * 4;
        const double sigma = (rs[0] + rs[1]) + (rs[2] + rs[3]);
        const double x0 = xs[k + 1];
        // flat: already tridiagonal in this column (always so for the last one); the reflector is then the identity:
        // beta = 0 makes p, w and the update vanish -- one code path, no divergent loop exits around the register matrix
        const bool flat = !(sigma - x0 * x0 > 0.0);
        const double alpha = flat ? x0 : -copysign(sqrt(sigma), x0);
        const double beta = flat ? 0.0 : 1.0 / (sigma - x0 * alpha);        // 2 / |v|^2 with v = x - alpha e1
        if (tid == 0) of[k] = alpha;
        if (tid < 256) vs[tid] = (tid > k && tid < r && !flat) ? (tid == k + 1 ? x0 - alpha : xs[tid]) : 0.0;
        __syncthreads();                                                                        // (3)
        // ---- p = A v: row parts (all stored blocks) and column parts (blocks strictly below the 32 x 32 diagonal blocks)
        {
            double t16[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) t16[q] = vs[16 * q + tc];
#pragma unroll
            for (int ai = 0; ai < 8; ++ai) {
                double pr = 0.0;
#pragma unroll
                for (int bj = 0; bj <= 2 * ai + 1; ++bj) pr = fma(a[ai][bj], t16[bj], pr);
                rowp[(32 * ai + tr) * 17 + tc] = pr;
            }
            double t8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t8[q] = vs[32 * q + tr];
#pragma unroll
            for (int bj = 0; bj < 16; ++bj) {
                double pc = 0.0;
#pragma unroll
                for (int ai = bj / 2 + 1; ai < 8; ++ai) pc = fma(a[ai][bj], t8[ai], pc);
                colp[tr * 256 + 16 * bj + tc] = pc;
            }
        }
        __syncthreads();                                                                        // (4)
        {
            double pj = 0.0;
            if (tid < 256) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) s0 += rowp[tid * 17 + q];
#pragma unroll 8
                for (int q = 0; q < 32; ++q) s1 += colp[q * 256 + tid];
                pj = (tid > k && tid < r) ? beta * (s0 + s1) : 0.0;
                ps[tid] = pj;
            }
            part = (tid < 256) ? pj * vs[tid] : 0.0;
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
            if (tid < 256 && (tid & 63) == 0) red[8 + (k & 1) * 4 + (tid >> 6)] = part;
        }
        __syncthreads();                                                                        // (5)
        const double *rk = red + 8 + (k & 1) * 4;
        const double kk = 0.5 * beta * ((rk[0] + rk[1]) + (rk[2] + rk[3]));
        // ---- A -= v w' + w v',  w = p - kk v
        {
            double t16[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { const int j = 16 * q + tc; t16[q] = ps[j] - kk * vs[j]; }
#pragma unroll
            for (int ai = 0; ai < 8; ++ai) {                    // - v w'
                const double vi = vs[32 * ai + tr];
#pragma unroll
                for (int bj = 0; bj <= 2 * ai + 1; ++bj) a[ai][bj] = fma(-vi, t16[bj], a[ai][bj]);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) t16[q] = vs[16 * q + tc];
#pragma unroll
            for (int ai = 0; ai < 8; ++ai) {                    // - w v'
                const int i = 32 * ai + tr;
                const double wi = ps[i] - kk * vs[i];
#pragma unroll
                for (int bj = 0; bj <= 2 * ai + 1; ++bj) a[ai][bj] = fma(-wi, t16[bj], a[ai][bj]);
            }
        }
        // (no barrier here: the next column's extraction writes the OTHER column buffer; vs, ps and red are next written
        //  behind barriers (2), (4) and (1) of the next column, which every reader of this column's values has passed by then)
    }
    __syncthreads();
    // the last diagonal entry
    if (tr == ((r - 1) & 31) && tc == ((r - 1) & 15)) {
        double dlast = 0.0;
#pragma unroll
        for (int ai = 0; ai < 8; ++ai)
#pragma unroll
            for (int bj = 2 * ai; bj <= 2 * ai + 1; ++bj)
                if (ai == ((r - 1) >> 5) && bj == ((r - 1) >> 4)) dlast = a[ai][bj];
        dg[r - 1] = dlast; of[r - 1] = 0.0;
    }
}

// ---- The max-step needs ONE eigenvalue -- the largest (or smallest) -- of its symmetric r x r matrix, not the tridiagonal
// form: Lanczos (round 6: the plain three-term recurrence by default, the Gram-Schmidt sweeps below behind CIP_LG_LANCZOS_REORTH=1:
// lz_reorth), the matrix in registers as in k_lg_tridiag1 (same load, same p = A v), and the
// verdict of maxstep_sdc (src/ConicIP.jl:272-303) in the same launch (round 3; replaces k_lg_tridiag1 + k_lg_sturm for
// r <= 256: 1.15 + 0.28 ms per max-step, six max-steps per iteration, half of config 4's time).
//   v_1 fixed (every component non-zero inside the r x r block, zero in the padding: the padded rows never enter);
//   step j:  w = A v_j, alpha_j = w'v_j, w orthogonalised against v_1..v_j twice (classical Gram-Schmidt, "twice is
//   enough": it also removes alpha_j v_j and beta_{j-1} v_{j-1}), beta_j = |w|, v_{j+1} = w / beta_j;
//   at j = 8, 12, .., 32, 40, .., 64, 80, ..: theta = the wanted extreme eigenvalue of T_j by multisection on the Sturm
//   count (512 shifts per round, down to the last bit), s = its eigenvector by one inverse iteration with the shift at the
//   OUTER end of the final bracket (T - sigma I is then definite: the LDL' sweep needs no pivoting), and the classical
//   bound: A has an eigenvalue within beta_j |s_j| of theta.  Stop at beta_j |s_j| <= 1e-11 |T|; at j = r the recurrence is
//   a complete tridiagonalisation and theta is exact -- the fallback is the loop's own end.
// theta <= lambda_max always (Ritz values lie inside the spectrum); that the largest Ritz value converges to lambda_max
// and not to a smaller eigenvalue rests on v_1 not being orthogonal to the extreme eigenvector, as in every Krylov eigensolver.
// On the max-step matrices of config 4's family the stop comes after 9-90 steps, 30 on average; the eigenvalue then agrees
// with LAPACK's to 1e-15 (the bound is quadratic in the residual when the eigenvalue is isolated).  Deterministic: fixed
// start, fixed order of every sum.
#define LZ_LDSV 64                      // Lanczos vectors kept in LDS; the later ones in a global scratch (L2)
#define LZ_LDS_DOUBLES (256 + 8 * 256 + 256 * 6 + 64 + LZ_LDSV * 256)
// 1/x to ~1 ulp (hardware estimate + one Newton step): the Sturm counts only look at signs, the inverse iteration only
// feeds a convergence test
__device__ __forceinline__ double lz_rcp(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r0, 1.0), r0, r0);
}
__device__ __forceinline__ double lz_sum256(double x, double *red, int slot) {      // sum over threads 0..255 (4 waves), all 512 threads call
    const int tid = threadIdx.x;
    x = lz_sum_rows(lz_sum16(x));
    if (tid < 256 && (tid & 63) == 0) red[slot * 4 + (tid >> 6)] = x;
    __syncthreads();
    return (red[slot * 4] + red[slot * 4 + 1]) + (red[slot * 4 + 2] + red[slot * 4 + 3]);
}
__global__ __launch_bounds__(512) void k_lg_lanczos1(const double *M, int ldm, const double *dscale, int r, int want_max, double scale,
                                                      const int *info, double *partial, int item, double *Vg, int *stat, double *cert, double cert_tol, int reorth) {
    extern __shared__ double sh[];
    double *rowp = sh, *colp = sh + 256;                      // p = A v: row sums [i], column sums per wave [wave][j]
    double *vs = colp + 8 * 256, *wsv = vs + 256, *al = wsv + 256, *be = al + 256, *hb = be + 256, *zz = hb + 256;
    double *red = zz + 256, *Vl = red + 64;
    __shared__ int s_first;
    __shared__ double s_res[4];
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    const double INF = __builtin_inf();
    if (info && info[0]) {                                     // X not positive definite -> Inf (:277-280)
        if (tid == 0) { partial[item] = INF; if (stat) stat[0] = 0; if (cert) { cert[0] = 0.0; cert[1] = 1.0; } }
        return;
    }
    LZ_T0();
    double a[8][16];                                           // blocks bj <= 2 ai + 1 only (k_lg_tridiag1's layout)
#pragma unroll
    for (int ai = 0; ai < 8; ++ai)
#pragma unroll
        for (int bj = 0; bj < 16; ++bj) {
            if (bj <= 2 * ai + 1) {
                a[ai][bj] = 0.0;
                const int i = 32 * ai + tr, j = 16 * bj + tc;
                if (i < r && j < r) {
                    // M[j + i ldm] is the coalesced one of the pair (tc runs along a column); an unscaled M is mat(x): symmetric
                    double x = M[j + (long)i * ldm];
                    if (dscale) x = 0.5 * (M[i + (long)j * ldm] + x) * (rsqrt(dscale[i]) * rsqrt(dscale[j]));
                    a[ai][bj] = x;
                }
            }
        }
    // LDS copy of v_k: element i at k 256 + (i ^ 8 (k & 7)) -- the eight vectors a wave reads together in the dot products
    // (eight lanes each, eight consecutive elements) then sit in different banks; a whole vector read by 256 threads is a permutation
    auto vput = [&](int k, int i, double x) { if (k < LZ_LDSV) Vl[k * 256 + (i ^ ((k & 7) << 3))] = x; else __builtin_nontemporal_store(x, Vg + (size_t)(k - LZ_LDSV) * 256 + i); };
    // v_1
    double vi = 0.0;
    if (tid < r) vi = cos(0.7 * tid + 0.3) + 1.0 / (1.0 + tid);
    {
        const double n2 = lz_sum256(tid < 256 ? vi * vi : 0.0, red, 0);
        vi *= 1.0 / sqrt(n2);
    }
    if (tid < 256) { vs[tid] = vi; vput(0, tid, vi); }
    __syncthreads();
    LZ_T(0);
    int m = 0;
    double th_lo = 0.0, th_hi = 0.0, ascale = 0.0, prev_lo = 0.0;
    bool done = false, have_prev = false;
    for (int j = 0; j < r && !done; ++j) {
        // ---- w = A v_j: row parts (all stored blocks) summed over the 16 lanes of a row of the thread grid, column parts (blocks
        // strictly below the 32 x 32 diagonal blocks) over the wave's four grid rows, then over the eight waves through LDS
        {
            double t16[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) t16[q] = vs[16 * q + tc];
#pragma unroll
            for (int ai = 0; ai < 8; ++ai) {
                double pr = 0.0;
#pragma unroll
                for (int bj = 0; bj <= 2 * ai + 1; ++bj) pr = fma(a[ai][bj], t16[bj], pr);
                pr = lz_sum16(pr);
                if (tc == 0) rowp[32 * ai + tr] = pr;
            }
            double t8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t8[q] = vs[32 * q + tr];
#pragma unroll
            for (int bj = 0; bj < 16; ++bj) {
                double pc = 0.0;
#pragma unroll
                for (int ai = bj / 2 + 1; ai < 8; ++ai) pc = fma(a[ai][bj], t8[ai], pc);
                pc = lz_sum_rows(pc);
                if ((tid & 63) < 16) colp[(tid >> 6) * 256 + 16 * bj + tc] = pc;
            }
        }
        __syncthreads();
        double wi = 0.0;
        if (tid < 256) {
            double s1 = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) s1 += colp[q * 256 + tid];
            wi = rowp[tid] + s1;
        }
        LZ_T(1);
        // ---- the recurrence's own terms first: w -= beta_{j-1} v_{j-1}, alpha_j = v_j'w, w -= alpha_j v_j.  (Taking them out
        // in the Gram-Schmidt sweep with everything else needs its second pass at EVERY step: they are the large components,
        // and with a basis orthogonal to delta the sweep then puts delta |alpha| / beta of them back along the old vectors.)
        if (tid < 256 && j > 0) {
            const int k = j - 1;
            const double vp = k < LZ_LDSV ? Vl[k * 256 + (tid ^ ((k & 7) << 3))] : __builtin_nontemporal_load(Vg + (size_t)(k - LZ_LDSV) * 256 + tid);
            wi = fma(-be[k], vp, wi);
        }
        double alpha = lz_sum256(tid < 256 ? wi * vs[tid] : 0.0, red, 2);
        if (tid < 256) { wi = fma(-alpha, vs[tid], wi); wsv[tid] = wi; }
        __syncthreads();
        LZ_T(2);
        // ---- then w orthogonal to v_0 .. v_j by classical Gram-Schmidt: what it finds is rounding noise while the basis is
        // orthogonal; a second sweep when the first one took out a sizeable part of w ("twice is enough")
        double nrm2 = 0.0;
        if (!reorth) nrm2 = lz_sum256(tid < 256 ? wi * wi : 0.0, red, 0);       // plain three-term Lanczos (the default since round 6: see lz_reorth)
        for (int pass = 0; pass < (reorth ? 2 : 0); ++pass) {
            {                                                  // h_k = v_k'w: eight threads per vector
                const int k = tid >> 3, l8 = tid & 7;
                double h = 0.0;
                if (k <= j) {
                    const double *vk = Vl + k * 256;
                    const int sw = (k & 7) << 3;
#pragma unroll 8
                    for (int q = 0; q < 32; ++q) { const int i = 8 * q + l8; h = fma(vk[i ^ sw], wsv[i], h); }
                }
                h += __shfl_xor(h, 1); h += __shfl_xor(h, 2); h += __shfl_xor(h, 4);
                if (k <= j && l8 == 0) hb[k] = h;
            }
            for (int k0 = LZ_LDSV; k0 <= j; k0 += 64) {         // the vectors beyond the LDS copy
                const int k = k0 + (tid >> 3), l8 = tid & 7;
                double h = 0.0;
                if (k <= j) {
                    const double *vk = Vg + (size_t)(k - LZ_LDSV) * 256;
#pragma unroll 8
                    for (int q = 0; q < 32; ++q) { const int i = 8 * q + l8; h = fma(__builtin_nontemporal_load(vk + i), wsv[i], h); }
                }
                h += __shfl_xor(h, 1); h += __shfl_xor(h, 2); h += __shfl_xor(h, 4);
                if (k <= j && l8 == 0) hb[k] = h;
            }
            __syncthreads();
            LZ_T(3);
            alpha += hb[j];
            double h2 = 0.0;                                   // |h|^2: what this sweep takes out of w
            {
                // w -= V h on all 512 threads (round 5, second session): the even-k chain (acc0) on threads 0..255, the odd-k chain and
                // the vectors beyond the LDS copy (acc1) on threads 256..511 for the same element i, handed over through `zz` (free
                // outside the convergence test); same two chains, same final acc0 + acc1: same bits as the 256-thread form
                const int i = tid & 255;
                const int jl = j < LZ_LDSV ? j : LZ_LDSV - 1;
                if (tid >= 256) {
                    double acc1 = 0.0;
                    int k = 0;
                    for (; k + 1 <= jl; k += 2) acc1 = fma(hb[k + 1], Vl[(k + 1) * 256 + (i ^ (((k + 1) & 7) << 3))], acc1);
#pragma unroll 8
                    for (int kk = LZ_LDSV; kk <= j; ++kk) acc1 = fma(hb[kk], __builtin_nontemporal_load(Vg + (size_t)(kk - LZ_LDSV) * 256 + i), acc1);
                    zz[i] = acc1;
                }
                double acc0 = 0.0;
                if (tid < 256) {
                    int k = 0;
                    for (; k + 1 <= jl; k += 2) {
                        const double h0 = hb[k], h1 = hb[k + 1];
                        acc0 = fma(h0, Vl[k * 256 + (i ^ ((k & 7) << 3))], acc0);
                        h2 = fma(h0, h0, fma(h1, h1, h2));
                    }
                    if (k <= jl) { const double h0 = hb[k]; acc0 = fma(h0, Vl[k * 256 + (i ^ ((k & 7) << 3))], acc0); h2 = fma(h0, h0, h2); }
                    for (int kk = LZ_LDSV; kk <= j; ++kk) { const double h0 = hb[kk]; h2 = fma(h0, h0, h2); }
                }
                __syncthreads();
                if (tid < 256) {
                    wi -= acc0 + zz[i];
                    wsv[tid] = wi;
                }
            }
            nrm2 = lz_sum256(tid < 256 ? wi * wi : 0.0, red, pass);        // (its barrier also publishes wsv)
            LZ_T(4);
            // (Daniel, Gragg, Kaufman, Stewart.)  h2 is the same in every thread below 256; thread 0 decides
            if (pass == 0) {
                if (tid == 0) s_first = (nrm2 >= 0.5 * (nrm2 + h2)) ? 1 : 0;
                __syncthreads();
                const int skip = s_first;
                __syncthreads();
                if (skip) break;
            }
        }
        const double beta = sqrt(nrm2);
        if (tid == 0) { al[j] = alpha; be[j] = beta; }
        m = j + 1;
        LZ_T(5);
        // ---- convergence test
        ascale = fmax(ascale, fabs(alpha) + beta);
        const bool check = m == r || !(beta > 1e-13 * ascale) || (m >= 24 && ((m <= 64 && (m & 7) == 0) || (m & 15) == 0));
        if (check) {
            __syncthreads();                                   // al / be of this step
            double lo = INF, hi = -INF;
            for (int i = tid; i < m; i += 512) {
                const double rad = (i > 0 ? fabs(be[i - 1]) : 0.0) + (i + 1 < m ? fabs(be[i]) : 0.0);
                lo = fmin(lo, al[i] - rad);
                hi = fmax(hi, al[i] + rad);
            }
            for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
            if ((tid & 63) == 0) { red[16 + (tid >> 6)] = lo; red[24 + (tid >> 6)] = hi; }
            __syncthreads();
            lo = red[16]; hi = red[24];
            for (int q = 1; q < 8; ++q) { lo = fmin(lo, red[16 + q]); hi = fmax(hi, red[24 + q]); }
            const double span = fmax(fabs(lo), fabs(hi));
            const double isp = 1.0 / fmax(span, 1e-300);
            hi += 1e-15 * span + 1e-300;
            lo -= 1e-15 * span + 1e-300;
            // the extreme Ritz value only moves outwards as T grows (interlacing): the last test's inner bracket end still bounds it
            // on the inside, and it usually moves very little -- the first round's shifts crowd towards that end (geometrically,
            // eight per octave), the others are uniform
            // scaled copies for the Sturm counts: a_i / |T| in zz, (b_i / |T|)^2 in wsv (both free here)
            for (int i = tid; i < m; i += 512) { zz[i] = al[i] * isp; const double bsc = be[i] * isp; wsv[i] = bsc * bsc; }
            __syncthreads();
            bool geo = false;
            if (have_prev) { if (want_max) { if (prev_lo > lo) { lo = prev_lo; geo = true; } } else if (prev_lo < hi) { hi = prev_lo; geo = true; } }
            for (int round = 0; round < 9; ++round) {
                if (!(hi - lo > 1e-15 * fmax(fabs(lo), fabs(hi)))) break;      // a few ulps: the midpoint is good to 5e-16
                // shift number t (1 .. 512) at distance u_t (hi - lo) from the inner end: u_t = t / 513, or 2^((t - 513) / 8) in a
                // warm-started first round (eight per octave, crowding towards the inner end)
                const bool g0 = geo && round == 0;
                const double wdt = hi - lo;
                auto u_of = [&](int t) -> double {
                    if (!g0) return (double)t * (1.0 / 513.0);
                    const double c8[8] = {1.0, 1.0905077326652577, 1.189207115002721, 1.2968395546510096, 1.4142135623730951,
                                          1.5422108254079407, 1.681792830507429, 1.8340080864093424};       // 2^(q / 8)
                    const int e = t - 513;
                    return ldexp(c8[e & 7], e >> 3);
                };
                auto shift_at = [&](int t) -> double {         // t = 0 .. 513: inner end .. outer end
                    if (t <= 0) return want_max ? lo : hi;
                    if (t >= 513) return want_max ? hi : lo;
                    return want_max ? lo + u_of(t) * wdt : hi - u_of(t) * wdt;
                };
                const double xs = want_max ? lo + u_of(tid + 1) * wdt : hi - u_of(tid + 1) * wdt;
                // Sturm count without divisions: p_i = (a_i - x) p_{i-1} - b_{i-1}^2 p_{i-2} on entries scaled by 1 / |T| (factors
                // of modulus <= 2; the pair is rescaled every eight steps); an eigenvalue below x for every i where p_i and
                // p_{i-1} differ in sign -- the q_i = p_i / p_{i-1} < 0 of the quotient form (a zero takes its predecessor's sign)
                int cnt = 0;
                const double xsc = xs * isp;
                double p0 = 1.0, p1 = zz[0] - xsc;
                if (p1 < 0.0) ++cnt;
#pragma unroll 4
                for (int i = 1; i < m; ++i) {
                    const double p2 = fma(zz[i] - xsc, p1, -wsv[i - 1] * p0);
                    // q_i = p2 / p1 < 0  <=>  signs differ (a zero p1 counts as positive, as the 1e-300 of the quotient form)
                    if ((p2 < 0.0) != (p1 < 0.0 || (p1 == 0.0 && p0 < 0.0))) ++cnt;
                    p0 = p1; p1 = p2;
                    if ((i & 7) == 7) {
                        const double mg = fmax(fabs(p0), fabs(p1));
                        const double sc = mg > 1e100 ? 1e-100 : (mg < 1e-100 ? 1e100 : 1.0);
                        p0 *= sc; p1 *= sc;
                    }
                }
                // outside the spectrum of T (on the wanted side)?  the first such shift, counted from the inner end
                const bool hit = want_max ? (cnt >= m) : (cnt == 0);
                if (tid == 0) s_first = 512;
                __syncthreads();
                if (hit) atomicMin(&s_first, tid);
                __syncthreads();
                const int f = s_first;
                __syncthreads();
                const double inner = shift_at(f), outer = shift_at(f + 1);      // theta lies between shifts f and f + 1
                if (want_max) { lo = inner; hi = outer; } else { hi = inner; lo = outer; }
            }
            th_lo = lo; th_hi = hi;
            prev_lo = want_max ? lo : hi; have_prev = true;
            LZ_T(7);
            if (tid == 0) {
                // the Ritz vector's last component: (T - theta I) s = 0 solved from the BOTTOM with s_m = 1 --
                //   s_{i-1} = ((theta - a_i) s_i - b_i s_{i+1}) / b_{i-1}
                // -- the growing direction of the recurrence when the Ritz pair has converged (its weight sits in the early
                // Lanczos vectors), hence the stable one; before convergence the estimate is rough and the answer is "go on" anyway
                const double th = 0.5 * (lo + hi);
                double s1 = 1.0, s2 = 0.0, n2 = 1.0;               // s_i, s_{i+1}
                int rescaled = 0;
                for (int i = m - 1; i >= 1; --i) {
                    double bb = be[i - 1];
                    if (fabs(bb) < 1e-300) bb = 1e-300;
                    const double s0 = ((th - al[i]) * s1 - (i + 1 < m ? be[i] : 0.0) * s2) * lz_rcp(bb);
                    s2 = s1; s1 = s0;
                    n2 += s0 * s0;
                    if (n2 > 1e200) { s1 *= 1e-100; s2 *= 1e-100; n2 *= 1e-200; rescaled += 1; }
                }
                // |s_m| / |s| = 10^(-100 rescaled) / sqrt(n2)
                s_res[0] = rescaled ? 0.0 : beta / sqrt(n2);
                s_res[1] = span;
            }
            __syncthreads();
            done = (s_res[0] <= 1e-11 * s_res[1]) || !(beta > 1e-14 * s_res[1]) || m == r;
        }
        LZ_T(6);
        if (!done) {
            const double vn = wi / beta;
            if (tid < 256) { vs[tid] = vn; vput(j + 1, tid, vn); }
            __syncthreads();
        }
    }
    if (tid == 0) {
        const double ev = 0.5 * (th_lo + th_hi);
        if (want_max) { const double mx = ev * scale; partial[item] = (mx < 0.0) ? INF : 1.0 / mx; }
        else partial[item] = (ev > 0.0) ? 0.0 : -1.0 + ev;
        if (stat) stat[0] = m;
        // the bound the inertia certificate checks: no eigenvalue beyond theta by more than 1e-9 |T| (the stop was at 1e-11 |T|)
        if (cert) { const double tol = cert_tol * s_res[1] + 1e-300; cert[0] = want_max ? ev + tol : ev - tol; cert[1] = 0.0; }
        atomicAdd(g_lz_hist + (m >> 3 < 32 ? m >> 3 : 32), 1);       // histogram of step counts (CIP_LG_LANCZOS_STATS=1 prints it at destroy)
    }
}

// ---- Orders above 1024: the slabs of the cooperative kernel no longer fit the LDS of 256 resident workgroups.  The matrix lives in
// global memory (A, full symmetric storage, both triangles kept up to date) and a Householder step is TWO launches, the launch
// boundary in place of the grid barrier:
//   k_tg_symv   every workgroup forms the reflector of column k for itself (x = A[k+1:, k], at most 2047 entries: sigma, alpha, beta, v --
//               exactly k_lg_tridiag's), workgroup 0 stores v, beta and d_k, e_k; then p_j = beta A22[:, j] . v, one wave per column j
//   k_tg_rank2  every workgroup forms kk = beta v'p / 2 for itself; A22 -= v w' + w v' with w = p - kk v, one 64 x 64 tile each
// 2 (r - 1) launches, 8 r^3 bytes in all: ~35 ms at order 2048.  Same reflectors as the other two forms; d and e go to k_lg_sturm.
__global__ __launch_bounds__(256) void k_tg_init(const double *M, int ldm, const double *dscale, int r, double *A, int lda) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)r * r) return;
    const int i = (int)(e % r), j = (int)(e / r);
    double x = 0.5 * (M[i + (long)j * ldm] + M[j + (long)i * ldm]);
    if (dscale) x *= rsqrt(dscale[i]) * rsqrt(dscale[j]);
    A[i + (long)j * lda] = x;
}
__device__ __forceinline__ double tg_block_sum(double x, double *red) {          // 256 threads; every thread gets the sum
    x = cip_wave_sum(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// scal: [0] beta, [1] flat (1.0: nothing to do in this column)
__global__ __launch_bounds__(256) void k_tg_symv(const double *A, int lda, int r, int k, double *vbuf, double *pbuf, double *dg, double *of, double *scal) {
    __shared__ double red[4];
    __shared__ double vs[2048];
    const int tid = threadIdx.x, m = r - k - 1;
    const double *x = A + (k + 1) + (long)k * lda;
    double part = 0.0;
    for (int i = tid; i < m; i += 256) { const double xi = x[i]; vs[i] = xi; part += xi * xi; }
    const double sigma = tg_block_sum(part, red);
    const double x0 = vs[0];
    const bool flat = (m == 1) || !(sigma - x0 * x0 > 0.0);
    double alpha = x0, beta = 0.0;
    if (!flat) { alpha = -copysign(sqrt(sigma), x0); beta = 1.0 / (sigma - x0 * alpha); }
    if (blockIdx.x == 0 && tid == 0) {
        dg[k] = A[k + (long)k * lda];
        of[k] = alpha;
        if (m == 1) { dg[r - 1] = A[(r - 1) + (long)(r - 1) * lda]; of[r - 1] = 0.0; }
        scal[0] = beta; scal[1] = flat ? 1.0 : 0.0;
    }
    if (flat) return;
    if (tid == 0) vs[0] = x0 - alpha;
    __syncthreads();
    if (blockIdx.x == 0) for (int i = tid; i < m; i += 256) vbuf[i] = vs[i];
    const int lane = tid & 63, j = blockIdx.x * 4 + (tid >> 6);
    if (j >= m) return;
    const double *col = A + (k + 1) + (long)(k + 1 + j) * lda;
    double acc = 0.0;
    for (int i = lane; i < m; i += 64) acc = fma(col[i], vs[i], acc);
    acc = cip_wave_sum(acc);
    if (lane == 0) pbuf[j] = beta * acc;
}
__global__ __launch_bounds__(256) void k_tg_rank2(double *A, int lda, int r, int k, const double *vbuf, const double *pbuf, const double *scal) {
    __shared__ double red[4];
    __shared__ double vi[64], wi[64], vj[64], wj[64];
    if (scal[1] != 0.0) return;
    const int tid = threadIdx.x, m = r - k - 1;
    const double beta = scal[0];
    double part = 0.0;
    for (int i = tid; i < m; i += 256) part += pbuf[i] * vbuf[i];
    const double kk = 0.5 * beta * tg_block_sum(part, red);
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    if (tid < 64) {
        const int i = i0 + tid;
        const double v = i < m ? vbuf[i] : 0.0, p = i < m ? pbuf[i] : 0.0;
        vi[tid] = v; wi[tid] = p - kk * v;
    } else if (tid < 128) {
        const int j = j0 + tid - 64;
        const double v = j < m ? vbuf[j] : 0.0, p = j < m ? pbuf[j] : 0.0;
        vj[tid - 64] = v; wj[tid - 64] = p - kk * v;
    }
    __syncthreads();
    const int ti = tid & 63, tq = tid >> 6;
    if (i0 + ti >= m) return;
    double *a = A + (k + 1 + i0 + ti) + (long)(k + 1 + j0) * lda;
    const double v_i = vi[ti], w_i = wi[ti];
#pragma unroll 4
    for (int c = tq; c < 64; c += 4)
        if (j0 + c < m) a[(long)c * lda] -= v_i * wj[c] + w_i * vj[c];
}
static int lg_tridiag_stepped(hipStream_t s, LargeWs *w, const double *M, const double *dscale, int r, double *work) {
    const int rp = w->rp;
    double *dg = w->vec + 1 * rp, *of = w->vec + 2 * rp, *vbuf = w->vec + 3 * rp, *pbuf = w->vec + 5 * rp, *scal = w->vec + 7 * rp;
    hipLaunchKernelGGL(k_tg_init, lg_grid((long)r * r), dim3(256), 0, s, M, rp, dscale, r, work, rp);
    for (int k = 0; k + 1 < r; ++k) {
        const int m = r - k - 1;
        hipLaunchKernelGGL(k_tg_symv, dim3((m + 3) / 4), dim3(256), 0, s, (const double *)work, rp, r, k, vbuf, pbuf, dg, of, scal);
        if (m > 1) hipLaunchKernelGGL(k_tg_rank2, dim3((m + 63) / 64, (m + 63) / 64), dim3(256), 0, s, work, rp, r, k, (const double *)vbuf, (const double *)pbuf, (const double *)scal);
    }
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

static int lg_tridiag(hipStream_t s, LargeWs *w, const double *M, const double *dscale, int r, double *work = nullptr) {
    int rc;
    if (r > 1024) {
        if (!work) { cip_set_error("large S cone: the tridiagonalisation above order 1024 needs a work matrix"); return CIP_E_INVALID; }
        return lg_tridiag_stepped(s, w, M, dscale, r, work);
    }
    // CIP_LG_TRIDIAG1=0: the cooperative kernel at every order (A/B runs)
    static const int one_wg = [] { const char *e = getenv("CIP_LG_TRIDIAG1"); return (e && atoi(e) == 0) ? 0 : 1; }();
    if (one_wg && r <= 256) {
        const size_t shm1 = T1_LDS_DOUBLES * sizeof(double);
        if ((rc = lg_set_attr((const void *)k_lg_tridiag1, shm1))) return rc;
        hipLaunchKernelGGL(k_lg_tridiag1, dim3(1), dim3(512), shm1, s, M, w->rp, dscale, r, w->vec + 1 * w->rp, w->vec + 2 * w->rp, (const int *)nullptr);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // a workgroup keeps a slab of 32 columns in LDS for the whole reduction; above order ~600 that no longer fits: 16 columns
    const int slab = ((size_t)LG_SLAB * (r | 1) + 3 * (size_t)r + 64) * sizeof(double) <= 160 * 1024 ? LG_SLAB : LG_SLAB / 2;
    const int nwg = (r + slab - 1) / slab;
    const size_t shm = ((size_t)slab * (r | 1) + 3 * (size_t)r + 64) * sizeof(double);
    if ((rc = lg_set_attr((const void *)k_lg_tridiag, shm))) return rc;
    CIP_HIP_CHECK(hipMemsetAsync(w->ctr, 0, 1024, s));
    double *dg = w->vec + 1 * w->rp, *of = w->vec + 2 * w->rp, *xbuf = w->vec + 3 * w->rp, *pbuf = w->vec + 5 * w->rp;   // 2 x rp each
    hipLaunchKernelGGL(k_lg_tridiag, dim3(nwg), dim3(LG_T), shm, s, M, w->rp, dscale, r, dg, of, xbuf, pbuf, w->ctr,
                       (int *)(w->ctr + 128), slab);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// dst = src' (optionally row j of dst scaled by 1 / sc[j]^2), rp x rp through a 32 x 33 LDS tile
__global__ __launch_bounds__(256) void k_lg_transpose(const double *src, double *dst, int rp, const double *sc) {
    __shared__ double t[32][33];
    const int bi = blockIdx.x * 32, bj = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int q = 0; q < 32; q += 8) t[ty + q][tx] = src[(bi + tx) + (long)(bj + ty + q) * rp];        // t[j][i] = src[i, j]
    __syncthreads();
    for (int q = 0; q < 32; q += 8) {
        const int j = bj + tx, i = bi + ty + q;                                                      // dst[j, i] = src[i, j]
        double v = t[tx][ty + q];
        if (sc) { const double s1 = sc[j]; v = v / (s1 * s1); }
        dst[j + (long)i * rp] = v;
    }
}
// M <- 1.5 I - 0.5 M   (the Newton-Schulz factor of V <- V (3 I - V'V) / 2)
// M = V'V  ->  1.5 I - 0.5 M (the Newton-Schulz factor); vst[1] = max |V'V - I| as the bit pattern of a non-negative double
// (monotone for finite values; NaN / inf compare as larger than every finite tolerance): what the gate below looks at
__global__ __launch_bounds__(256) void k_lg_ns(double *M, int rp, unsigned long long *vst) {
    __shared__ unsigned long long smax;
    if (threadIdx.x == 0) smax = 0ull;
    __syncthreads();
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < (long)rp * rp) {
        const double d = ((e % rp) == (e / rp)) ? 1.0 : 0.0, mv = M[e];
        M[e] = 1.5 * d - 0.5 * mv;
        atomicMax(&smax, (unsigned long long)__double_as_longlong(fabs(mv - d)));
    }
    __syncthreads();
    if (threadIdx.x == 0 && smax) atomicMax(vst + 1, smax);
}
// Round 6 (advisor): the warm start must not be taken from a V that is not a (near-)orthogonal matrix -- the previous scaling of
// this cone came from an iterate outside the cone (sqrt of a negative pivot: G, lam, V all NaN), or from a G so ill-conditioned
// that V = G'A_f Sigma^-2 is orthogonal to less than what ONE Newton-Schulz step repairs to rounding (e <= 1e-6 -> 4e-13).  Decided
// on the device, no read-back: vst[0] = 1 when the previous call ended with clean Cholesky flags and finite positive singular
// values (k_lg_vok), vst[1] from k_lg_ns; otherwise G <- G0, i.e. exactly the cold start.
#define LG_WARM_TOL 1e-6
__global__ __launch_bounds__(256) void k_lg_warm_gate(double *G, const double *G0, long n2, const unsigned long long *vst) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n2) return;
    if (vst[0] != 1ull || vst[1] > (unsigned long long)__double_as_longlong(LG_WARM_TOL)) G[e] = G0[e];
}
__global__ __launch_bounds__(256) void k_lg_vok(const double *lam, int rp, const int *info_a, const int *info_b, unsigned long long *vst) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = (info_a[0] != 0 || info_b[0] != 0) ? 1 : 0;
    __syncthreads();
    for (int j = threadIdx.x; j < rp; j += 256) { const double l = lam[j]; if (!(l > 0.0 && l < 1.7e308)) bad = 1; }
    __syncthreads();
    if (threadIdx.x == 0) { vst[0] = bad ? 0ull : 1ull; vst[1] = 0ull; }
}
// a sweep's flag and the two Cholesky flags -> host-mapped memory, the sequence word behind them with system-scope release (the
// pattern of api.hip: k_publish_info): the host polls the word instead of a 4-byte copy + stream synchronisation per checked sweep
__global__ void k_lg_pubflag(const unsigned *sweepflag, const int *info_a, const int *info_b, int *host, int seq) {
    if (threadIdx.x == 0) {
        host[0] = (int)(*sweepflag != 0u);
        host[1] = (info_a[0] != 0 || info_b[0] != 0) ? 1 : 0;
        __hip_atomic_store(host + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static int lg_warm(void) {
    static const int on = [] { const char *e = getenv("CIP_LG_WARM"); return e ? atoi(e) : 1; }();
    return on;
}
// ---- development aid (CIP_LG_CHECKSUM=1): order-independent checksums (integer sums of the bit patterns) of the NT scaling's
// intermediates, per call and stage, printed when the workspace is destroyed -- two runs of one program must print the same table
// (tools/sdp640_repeat.py: a difference names the first racy stage)
__global__ __launch_bounds__(256) void k_lg_checksum(const double *x, long n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) acc += (unsigned long long)__double_as_longlong(x[e]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
__global__ __launch_bounds__(256) void k_lg_sumsq(const double *x, long n, unsigned long long *out) {     // sum of squares as a double in a 64-bit slot (atomic adds: good to rounding)
    double acc = 0.0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) acc += x[e] * x[e];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd((double *)out, acc);
}
static unsigned long long *g_lg_cks = nullptr;      // LG_CKS_CALLS calls x 16 stages
#define LG_CKS_CALLS 4096
static int g_lg_cks_call = 0;
static int lg_cks_on(void) { static const int on = [] { const char *e = getenv("CIP_LG_CHECKSUM"); return e ? atoi(e) : 0; }(); return on; }
static void lg_cks(hipStream_t s, int stage, const double *x, long n) {
    if (!lg_cks_on() || g_lg_cks_call >= LG_CKS_CALLS) return;
    if (!g_lg_cks) { (void)hipMalloc((void **)&g_lg_cks, LG_CKS_CALLS * 16 * 8); (void)hipMemset(g_lg_cks, 0, LG_CKS_CALLS * 16 * 8); }
    if (stage >= 12) hipLaunchKernelGGL(k_lg_sumsq, dim3(256), dim3(256), 0, s, x, n, g_lg_cks + g_lg_cks_call * 16 + stage);
    else hipLaunchKernelGGL(k_lg_checksum, dim3(256), dim3(256), 0, s, x, n, g_lg_cks + g_lg_cks_call * 16 + stage);
}
void cip_lg_cks_dump(void) {
    if (!g_lg_cks) return;
    static unsigned long long h[LG_CKS_CALLS * 16];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, g_lg_cks, sizeof(h), hipMemcpyDeviceToHost);
    for (int c = 0; c < g_lg_cks_call && c < LG_CKS_CALLS; ++c) {
        fprintf(stderr, "cks call %d:", c);
        for (int q = 0; q < 12; ++q) fprintf(stderr, " %016llx", h[c * 16 + q]);
        { double a, b; __builtin_memcpy(&a, &h[c * 16 + 12], 8); __builtin_memcpy(&b, &h[c * 16 + 13], 8); fprintf(stderr, " | |G_in|^2 %.17g |G_out|^2 %.17g rel %.3e", a, b, (b - a) / a); }
        fprintf(stderr, "\n");
    }
    (void)hipMemset(g_lg_cks, 0, LG_CKS_CALLS * 16 * 8);
    g_lg_cks_call = 0;
}
// nestod_sdc for one large cone (index li among the large cones)
int cip_sdp_large_nt(hipStream_t s, LargeWs *w, const ConeDesc &cd, int li, const double *v, const double *sv, double *scal,
                     double *lambda, int *flag) {
    const int r = cd.r, rp = w->rp;
    const long n2 = (long)rp * rp;
    int rc;
    hipLaunchKernelGGL(k_lg_mat, lg_grid(n2), dim3(256), 0, s, v + cd.off, 1L, 0L, w->Kz, r, rp, 1.0);
    hipLaunchKernelGGL(k_lg_mat, lg_grid(n2), dim3(256), 0, s, sv + cd.off, 1L, 0L, w->Ks, r, rp, 1.0);
    if ((rc = cip_ldlt_factor(s, w->Kz, rp, rp, w->wz))) return rc;                 // Lz (unit) and d_z  (:202-203)
    if ((rc = cip_ldlt_factor(s, w->Ks, rp, rp, w->ws))) return rc;
    lg_cks(s, 0, w->Kz, n2); lg_cks(s, 1, w->Ks, n2);
    hipLaunchKernelGGL(k_lg_flag, dim3(1), dim3(64), 0, s, w->wz.info, w->ws.info, flag);
    hipLaunchKernelGGL(k_lg_tfac, lg_grid(n2), dim3(256), 0, s, w->Kz, w->wz.dvec, w->Tz, rp);
    hipLaunchKernelGGL(k_lg_tfac, lg_grid(n2), dim3(256), 0, s, w->Ks, w->ws.dvec, w->Ts, rp);
    if ((rc = lg_gemm(s, w->G, 0, w->Tz, 0, w->Ts, 0, rp, 1))) return rc;           // G = Lz' Ls          (:204)
    lg_cks(s, 2, w->G, n2);
    // Round 5: WARM START of the one-sided Jacobi.  svd(G) = U Sigma V' is needed for U and Sigma only (:204-208), and the
    // Jacobi below may start from G W for ANY orthogonal W: the left singular vectors and the singular values are those of G.
    // Between two interior-point iterations the NT scaling moves little, so with W = the right singular vectors V of the previous
    // scaling of this cone the columns of G W are already nearly orthogonal and the sweeps drop from 8 to 2-4 (each sweep of order
    // 256 is 255 latency-bound rotation rounds, 0.38 ms: the NT scaling was 30 % of config 4's iteration).  V is not tracked
    // through the rotations: afterwards, with the columns A_f = U Sigma in hand, V = G' A_f Sigma^-2 is one GEMM; before it is
    // used again one Newton-Schulz step V <- V (3 I - V'V) / 2 restores its orthogonality to rounding (G' U Sigma^-1 is orthogonal
    // only to cond(G) eps, and a W that is not orthogonal would change the answer by that much).  Four extra 256^3 products per
    // scaling (~5 us each).  The first scaling after the packed scaling was replaced from outside (cip_set_scaling_identity at the
    // start of every interior-point solve, cip_set_scaling_packed) starts cold: a solve's results do not depend on what the handle
    // did before.  CIP_LG_WARM=0 switches it off.
    const dim3 tg(rp / 32, rp / 32);
    double *Vw = w->Vw + (size_t)li * n2;
    unsigned long long *vst = w->vstate + 2 * (size_t)li;
    const bool keep_v = lg_warm() != 0;
    const bool jacobi_warm = keep_v && w->have_v[li];
    if (keep_v) CIP_HIP_CHECK(hipMemcpyAsync(w->G0, w->G, sizeof(double) * n2, hipMemcpyDeviceToDevice, s));
    if (jacobi_warm) {
        hipLaunchKernelGGL(k_lg_transpose, tg, dim3(256), 0, s, (const double *)Vw, w->W1, rp, (const double *)nullptr);      // V'
        if ((rc = lg_gemm(s, w->W2, 0, w->W1, 0, w->W1, 0, rp, 1))) return rc;                                              // V'V
        hipLaunchKernelGGL(k_lg_ns, lg_grid(n2), dim3(256), 0, s, w->W2, rp, vst);                                          // 1.5 I - 0.5 V'V (symmetric); max |V'V - I|
        if ((rc = lg_gemm(s, w->W1, 0, Vw, 0, w->W2, 0, rp, 1))) return rc;                                                 // Vn = V (1.5 I - 0.5 V'V)
        hipLaunchKernelGGL(k_lg_transpose, tg, dim3(256), 0, s, (const double *)w->W1, w->W2, rp, (const double *)nullptr);  // Vn'
        lg_cks(s, 3, w->W2, n2);
        if ((rc = lg_gemm(s, w->G, 0, w->G0, 0, w->W2, 0, rp, 1))) return rc;                                               // G <- G Vn
        hipLaunchKernelGGL(k_lg_warm_gate, lg_grid(n2), dim3(256), 0, s, w->G, (const double *)w->G0, n2, (const unsigned long long *)vst);   // ... unless V is not fit for it: G <- G0 (cold)
        lg_cks(s, 4, w->G, n2);
    }
    bool left_cone = false;                                 // either Cholesky met a non-positive pivot (read back with the sweep flags)
    {
        // column blocks of 8 at order 256 (16 workgroups of 256 threads; round 4, with the DPP sums: 4 / 8 / 16 / 32 wide ->
        // 3.95 / 3.11 / 3.25 / 4.48 ms per NT scaling -- a rotation round is bound by the hand-over between the wave's lane groups
        // (LDS + workgroup barrier: fewer waves per workgroup, shorter rounds), an outer round costs ~3 us (block exchange through
        // L2 + launch boundary) and their number doubles as the blocks halve), 16 at order 512 (16 workgroups), 8 at order 1024.
        // CIP_LG_JACOBI_B = 4 / 8 / 16 / 32 overrides at order 256.
        static const int bforce = [] { const char *e = getenv("CIP_LG_JACOBI_B"); return e ? atoi(e) : 0; }();
        const int b = rp > 1024 ? 4 : rp > 512 ? 8 : (rp <= 256 ? ((bforce == 32 || bforce == 16 || bforce == 4) ? bforce : 8) : 16);       // order 2048: two blocks of 4 columns are a CU's LDS
        const int nt = rp > 512 ? b * 64 : b * (rp / 8);           // 512 (order 256, b = 16), 256 (b = 8), 1024 (order 512) or 512 (order 1024: 64 lanes x 16 elements per column)
        const size_t shm = (size_t)2 * b * lg_pitch(rp) * sizeof(double);
        CIP_HIP_CHECK(hipMemsetAsync(w->ctr, 0, 1024, s));
        lg_cks(s, 12, w->G, n2);
        if ((rc = cip_prof_slot_begin(CIP_PROF_JACOBI, s, 0.0))) return rc;
        // ONE LAUNCH PER PHASE (round 5, second session; the only form since round 6): the pairs inside the blocks, then each round of
        // the tournament over blocks, the launch boundary in place of a grid barrier.  0 differing results in 11 500 repeated scalings
        // at orders 256 ... 1024 and in 40 000 at order 256 (profiles/r5/jacobi_repeatability.txt).  The host enqueues whole sweeps; a
        // sweep behind a converged one is a no-op (the kernel's gate), so the sweep flags are read only from the sweep on that is
        // usually the last but one (warm start: 6-8 sweeps, cold: 9-11) -- through host-mapped memory (k_lg_pubflag: a one-wave kernel
        // and a polled word; until round 6 a 4-byte copy + stream synchronisation per checked sweep).
        bool converged = false;
        auto run = [&](auto kern, int nthreads, int bb, size_t lds) -> int {
            int rc2;
            if ((rc2 = lg_set_attr((const void *)kern, lds))) return rc2;
            const int m = rp / bb;
            const int first_check = jacobi_warm ? 4 : 7;
            for (int sweep = 0; sweep < 40; ++sweep) {
                for (int step = 0; step < m; ++step)
                    hipLaunchKernelGGL(kern, dim3(m / 2), dim3(nthreads), lds, s, w->G, rp, bb | (sweep << 16) | (step << 22), w->ctr);
                if (sweep < first_check) continue;
                w->hseq = (w->hseq == 0x7fffffff) ? 1 : w->hseq + 1;
                hipLaunchKernelGGL(k_lg_pubflag, dim3(1), dim3(64), 0, s, (const unsigned *)(w->ctr + 8 + sweep), (const int *)w->wz.info, (const int *)w->ws.info, w->hflag_dev, w->hseq);
                CIP_HIP_CHECK(hipGetLastError());
                volatile int *seqp = w->hflag + 2;
                for (long spin = 0; __atomic_load_n(seqp, __ATOMIC_ACQUIRE) != w->hseq; ++spin) {
                    if ((spin & 0xfff) != 0xfff) continue;
                    std::this_thread::yield();
                    const hipError_t q = hipStreamQuery(s);              // (a failed launch must not hang the host)
                    if (q == hipSuccess && __atomic_load_n(seqp, __ATOMIC_ACQUIRE) != w->hseq) { cip_set_error("NT scaling of an S cone: the Jacobi's sweep flag never arrived"); return CIP_E_HIP; }
                    if (q != hipSuccess && q != hipErrorNotReady) { cip_set_error("hipStreamQuery failed: %s", hipGetErrorString(q)); return CIP_E_HIP; }
                }
                left_cone = w->hflag[1] != 0;
                if (!w->hflag[0]) { converged = true; break; }
            }
            return 0;
        };
        if (rp == 256 && bforce == 116) {                  // A/B form (round 5): 16-column blocks, 16 lanes x 16 elements per column, 4 waves, 8 workgroups -- half the
                                                           // outer rounds, shorter lane sums, and SLOWER: config 4 8.41 against 8.03 ms per iteration
            rc = run(k_lg_jacobi<256, 16>, 256, 16, (size_t)2 * 16 * lg_pitch(rp) * sizeof(double));
        } else if (rp > 1024) rc = run(k_lg_jacobi<256, 32>, 256, b, shm);
        else if (rp > 512) rc = run(k_lg_jacobi<512, 16>, 512, b, shm);
        else if (nt == 128) rc = run(k_lg_jacobi<128>, 128, b, shm);
        else if (nt == 256) rc = run(k_lg_jacobi<256>, 256, b, shm);
        else if (nt == 512) rc = run(k_lg_jacobi<512>, 512, b, shm);
        else rc = run(k_lg_jacobi<1024>, 1024, b, shm);
        const int rce = cip_prof_slot_end(CIP_PROF_JACOBI, s);      // (also on the error paths: an unpaired event would mis-pair every later collect)
        if (rc) return rc;
        if (rce) return rce;
        if (!converged) {
            // (never seen: the cyclic one-sided Jacobi converges quadratically; a scaling built on unconverged columns would be silently wrong)
            w->have_v[li] = 0;
            cip_set_error("NT scaling of an S cone of order %d: the one-sided Jacobi still rotated after 40 sweeps", r);
            return CIP_E_SINGULAR;
        }
    }
    if (getenv("CIP_LG_DEBUG")) {
        unsigned hc[64];
        CIP_HIP_CHECK(hipMemcpyAsync(hc, w->ctr, sizeof(hc), hipMemcpyDeviceToHost, s));
        CIP_HIP_CHECK(hipStreamSynchronize(s));
        int sweeps = 0;
        for (int q = 8; q < 48; ++q) if (hc[q]) ++sweeps;
        fprintf(stderr, "[lg] jacobi: %d sweeps with a rotation above cos^2 = 1e-16%s\n", sweeps, left_cone ? " (iterate outside the cone)" : "");
    }
    double *lam = w->vec;
    lg_cks(s, 5, w->G, n2); lg_cks(s, 13, w->G, n2);
    hipLaunchKernelGGL(k_lg_colnorm, dim3((rp + 3) / 4), dim3(256), 0, s, w->G, lam, rp);
    lg_cks(s, 6, lam, rp);
    if (keep_v) {                                                                                                           // V = G0' (A_f Sigma^-2) for the next call
        hipLaunchKernelGGL(k_lg_transpose, tg, dim3(256), 0, s, (const double *)w->G, w->W1, rp, (const double *)lam);        // (A_f Sigma^-2)'
        hipLaunchKernelGGL(k_lg_transpose, tg, dim3(256), 0, s, (const double *)w->G0, w->W2, rp, (const double *)nullptr);   // G0'
        if ((rc = lg_gemm(s, Vw, 0, w->W2, 0, w->W1, 0, rp, 1))) return rc;
        lg_cks(s, 7, Vw, n2);
        // fit for the next call's warm start?  The host knows the Cholesky flags from the sweep read-back (an iterate outside the
        // cone leaves NaN in G, lam and V: round 5 kept such a V, and the next scaling of a perfectly interior iterate came out NaN
        // with a clean flag); the device-side word also covers non-finite / non-positive singular values (k_lg_warm_gate reads it)
        hipLaunchKernelGGL(k_lg_vok, dim3(1), dim3(256), 0, s, (const double *)lam, rp, (const int *)w->wz.info, (const int *)w->ws.info, vst);
        w->have_v[li] = left_cone ? 0 : 1;
    }
    hipLaunchKernelGGL(k_lg_build, lg_grid(n2), dim3(256), 0, s, w->G, lam, w->wz.dvec, w->Kz, w->M1, w->M2, w->M3, rp);
    const double *XTz = (w->wz.Bs == CIP_NB) ? w->wz.LinvT : w->wz.XT;             // inv(Lz_unit)' (one block = the matrix)
    if ((rc = lg_gemm(s, w->Tz, 0, XTz, 0, w->M1, 0, rp, 1))) return rc;            // R    = Lz^-T U Lambda^1/2   (:206-208)
    if ((rc = lg_gemm(s, w->Ts, 0, w->M2, 0, w->M3, 0, rp, 1))) return rc;          // Rinv = Lambda^-1/2 U' Lz'
    double *R = scal + cd.soff, *Ri = R + (size_t)r * r;
    hipLaunchKernelGGL(k_lg_store, lg_grid(n2 > cd.dim ? n2 : cd.dim), dim3(256), 0, s, w->Tz, w->Ts, R, Ri, w->Rip + LG_NPAD * (size_t)li * n2,
                       lam, lambda ? lambda + cd.off : nullptr, r, rp, cd.dim);
    if (lambda) hipLaunchKernelGGL(k_lg_lambda_diag, lg_grid(r), dim3(256), 0, s, lam, lambda + cd.off, r);
    lg_cks(s, 8, w->Tz, n2); lg_cks(s, 9, w->Ts, n2);
    if (lg_cks_on()) g_lg_cks_call += 1;
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// padded Rinv after the host replaced the packed scaling
int cip_sdp_large_refresh(hipStream_t s, LargeWs *w, const ConeDesc &cd, int li, const double *scal) {
    const int r = cd.r, rp = w->rp;
    w->have_v[li] = 0;                                     // the next NT scaling of this cone starts its Jacobi cold
    hipLaunchKernelGGL(k_lg_pad, lg_grid((long)rp * rp), dim3(256), 0, s, scal + cd.soff, scal + cd.soff + (size_t)r * r,
                       w->Rip + LG_NPAD * (size_t)li * rp * rp, r, rp);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Round 6: the max-step's Lanczos WITHOUT the Gram-Schmidt sweeps by default (CIP_LG_LANCZOS_REORTH=1 restores them).  The recurrence
// stops at the FIRST convergence of the extreme Ritz value, and a Lanczos basis loses its orthogonality only when a Ritz value
// converges (Paige): up to the stop the plain three-term recurrence is the reorthogonalised one to working precision, unless the
// other end of the spectrum converges first -- then the wanted end takes more steps (config 4's max-steps: the longest runs 144-151 ->
// 176-183 steps, the typical 40-47 unchanged).  What the returned value claims is checked by the inertia certificate either way.
// Measured: config 4 8.17-8.22 -> 7.88-8.00 ms per iteration; tests/test_gpu_sdp.py (LAPACK on hard spectra, the certificate's
// self-test, five trajectories) green in both modes.
static int lz_reorth(void) {
    static const int on = [] { const char *e = getenv("CIP_LG_LANCZOS_REORTH"); return e ? atoi(e) : 0; }();
    return on;
}
// 2 (default; CIP_LG_LANCZOS changes it): the max-step's extreme eigenvalue by k_lg_lanczos1 at orders <= 256 + the inertia certificate
// (lg_certify); 1: Lanczos alone; 3: certificate self-test; 0: full
// tridiagonalisation + Sturm multisection at every order (A/B runs, tests).  on < 0 only reads; returns the previous setting
#include <atomic>
int cip_sdp_large_lanczos(int on) {
    static std::atomic<int> mode{[] { const char *e = getenv("CIP_LG_LANCZOS"); const int v = e ? atoi(e) : 2; return (v >= 0 && v <= 3) ? v : 2; }()};
    const int prev = mode.load();
    if (on >= 0 && on <= 3) mode.store(on);
    return prev;
}
// The two max-steps of a pair (v side, s side: src/ConicIP.jl:708-709, :881-882, :927-928) are independent: side 1 works in the
// NT scaling's s-side buffers (Ks, its LDL' workspace, Tz / Ts / G as M1 / M2 / M3 -- all idle between scalings) on a second
// stream.  Orders <= 256 only (the cooperative tridiagonalisation above that shares `vec` and `ctr`).
bool cip_sdp_large_pairable(const LargeWs *w) { return w && w->rp <= 256 && cip_sdp_large_lanczos(-1); }
int cip_sdp_large_fork(hipStream_t s, LargeWs *w, hipStream_t *s2) {
    if (!w->s2) {
        CIP_HIP_CHECK(hipStreamCreateWithFlags(&w->s2, hipStreamNonBlocking));
        CIP_HIP_CHECK(hipEventCreateWithFlags(&w->efork, hipEventDisableTiming));
        CIP_HIP_CHECK(hipEventCreateWithFlags(&w->ejoin, hipEventDisableTiming));
    }
    CIP_HIP_CHECK(hipEventRecord(w->efork, s));
    CIP_HIP_CHECK(hipStreamWaitEvent(w->s2, w->efork, 0));
    *s2 = w->s2;
    return 0;
}
int cip_sdp_large_join(hipStream_t s, LargeWs *w) {
    CIP_HIP_CHECK(hipEventRecord(w->ejoin, w->s2));
    CIP_HIP_CHECK(hipStreamWaitEvent(s, w->ejoin, 0));
    return 0;
}
// maxstep_sdc for one large cone: partial[cd.item]
// ---- inertia certificate of the Lanczos max-step (mode 2; ADVICE r3: a Ritz value can settle on an interior eigenvalue when the
// fixed start vector has no component along the extreme eigenvector).  With theta' = theta +- 1e-9 |T| from the kernel:
//   B = theta' I - A (largest eigenvalue wanted)  or  A - theta' I (smallest)  must be positive definite
// -- checked by the library's own LDL' with all-positive prescribed pivot signs (a wrong-sign or dead pivot raises info).  If it is
// not, `gate` goes up and the tridiagonalisation + Sturm path behind it recomputes the verdict from the same scaled matrix
// (both kernels return at once on a lowered gate).  cert[1] != 0: the Lanczos kernel answered Inf for an X outside the cone --
// nothing to certify (B = I).  Asym: the matrix the Lanczos kernel saw (scaled, symmetrised), for the fallback.
__global__ __launch_bounds__(256) void k_lg_cert_build(const double *M, int ldm, const double *dscale, int r, int rp, int want_max,
                                                        const double *cert, double *B, double *Asym) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rp * rp) return;
    const int i = (int)(e % rp), j = (int)(e / rp);
    double a = 0.0, b = (i == j) ? 1.0 : 0.0;
    if (i < r && j < r) {
        a = M[j + (long)i * ldm];
        if (dscale) a = 0.5 * (M[i + (long)j * ldm] + a) * (rsqrt(dscale[i]) * rsqrt(dscale[j]));
        if (cert[1] == 0.0) {
            const double th = (i == j) ? cert[0] : 0.0;
            b = want_max ? th - a : a - th;
        }
    }
    B[e] = b;
    Asym[e] = a;
}
__global__ void k_lg_cert_check(const int *info, int *gate, unsigned *count) {
    if (threadIdx.x == 0) {
        const int up = (info[0] != 0 || info[2] != 0) ? 1 : 0;
        gate[0] = up;
        if (up) atomicAdd(count, 1u);
    }
}
static int lg_certify(hipStream_t s, LargeWs *w, const double *M3, const double *dscale, int r, int want_max, double scale,
                      double *M1, double *M2, const LdltWorkspace &wx, double *partial, int item, int side) {
    const int rp = w->rp;
    const long n2 = (long)rp * rp;
    int rc;
    const double *cert = w->vec + 11 * rp + 2 * side;
    int *gate = (int *)(w->ctr + 210 + side);
    hipLaunchKernelGGL(k_lg_cert_build, lg_grid(n2), dim3(256), 0, s, M3, rp, dscale, r, rp, want_max, cert, M2, M1);
    LdltWorkspace wc = wx;
    wc.no_prep = 1; wc.lazyC = nullptr; wc.prof = nullptr;
    if ((rc = cip_ldlt_factor(s, M2, rp, rp, wc))) return rc;
    hipLaunchKernelGGL(k_lg_cert_check, dim3(1), dim3(64), 0, s, (const int *)wc.info, gate, (unsigned *)(w->vec + 11 * rp + 8));      // (w->ctr is cleared by every NT scaling)
    // the fallback (one workgroup each, at once back unless the gate is up): side 1 of a pair has its own d / e vectors
    double *dg = w->vec + (side ? 9 : 1) * rp, *of = w->vec + (side ? 10 : 2) * rp;
    const size_t shm1 = T1_LDS_DOUBLES * sizeof(double);
    if ((rc = lg_set_attr((const void *)k_lg_tridiag1, shm1))) return rc;
    hipLaunchKernelGGL(k_lg_tridiag1, dim3(1), dim3(512), shm1, s, (const double *)M1, rp, (const double *)nullptr, r, dg, of, (const int *)gate);
    hipLaunchKernelGGL(k_lg_sturm, dim3(1), dim3(LG_T), 0, s, dg, of, r, want_max, scale, (const int *)nullptr, partial, item, (const int *)gate);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_sdp_large_cert_stats(hipStream_t s, LargeWs *w, int *out2) {
    unsigned c = 0;
    if (w) { CIP_HIP_CHECK(hipMemcpyAsync(&c, w->vec + 11 * w->rp + 8, sizeof(c), hipMemcpyDeviceToHost, s)); CIP_HIP_CHECK(hipStreamSynchronize(s)); }
    out2[0] = (int)c; out2[1] = 0;
    return 0;
}

int cip_sdp_large_maxstep(hipStream_t s, LargeWs *w, const ConeDesc &cd, const double *x, const double *d, double scale,
                          double *partial, int side) {
    const int r = cd.r, rp = w->rp;
    const long n2 = (long)rp * rp;
    int rc;
    double *dg = w->vec + 1 * rp, *of = w->vec + 2 * rp;
    double *const Kx = side ? w->Ks : w->Kz, *const M1 = side ? w->Tz : w->M1, *const M2 = side ? w->Ts : w->M2, *const M3 = side ? w->G : w->M3;
    LdltWorkspace &wx = side ? w->ws : w->wz;
    int *const stat = (int *)(w->ctr + 200 + side);
    // CIP_LG_LANCZOS=0: the full tridiagonalisation + Sturm multisection at every order (A/B runs, tests)
    const int lzmode = cip_sdp_large_lanczos(-1);
    const bool lz = lzmode && r <= 256;
    double *const cert = (lz && lzmode >= 2) ? w->vec + 11 * rp + 2 * side : nullptr;          // mode 2: + inertia certificate
    const double cert_tol = lzmode == 3 ? -1e-3 : 1e-9;      // mode 3 (self-test): the bound on the WRONG side of theta -- every certificate fails, every verdict comes from the fallback
    if (lz && (rc = lg_set_attr((const void *)k_lg_lanczos1, LZ_LDS_DOUBLES * sizeof(double)))) return rc;
    if (!d) {                                                                       // maxstep_sdc(x, nothing) :295-303
        hipLaunchKernelGGL(k_lg_mat, lg_grid(n2), dim3(256), 0, s, x + cd.off, 1L, 0L, M3, r, rp, 0.0);
        if (lz) {
            hipLaunchKernelGGL(k_lg_lanczos1, dim3(1), dim3(512), LZ_LDS_DOUBLES * sizeof(double), s, M3, rp, (const double *)nullptr, r, 0,
                               1.0, (const int *)nullptr, partial, cd.item, M1, stat, cert, cert_tol, lz_reorth());
            CIP_HIP_CHECK(hipGetLastError());
            if (cert) return lg_certify(s, w, M3, nullptr, r, 0, 1.0, M1, M2, wx, partial, cd.item, side);
            return 0;
        }
        if ((rc = lg_tridiag(s, w, M3, nullptr, r, M2))) return rc;
        hipLaunchKernelGGL(k_lg_sturm, dim3(1), dim3(LG_T), 0, s, dg, of, r, 0, 1.0, (const int *)nullptr, partial, cd.item);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_lg_mat, lg_grid(n2), dim3(256), 0, s, x + cd.off, 1L, 0L, Kx, r, rp, 1.0);
    if ((rc = cip_ldlt_factor(s, Kx, rp, rp, wx))) return rc;                 // X = L D L'
    const double *Xi = (wx.Bs == CIP_NB) ? wx.Linv : wx.X;                // inv(L_unit)
    if (rp <= 256 && lg_small_gemm()) {                                             // inv(L) D, D = mat(d) read from the vector
        hipLaunchKernelGGL((k_gemm_nt_small<true, false>), dim3(rp / 16, rp / 16), dim3(256), 0, s, Xi, (long)rp, d + cd.off, 0L, M2, (long)rp, rp, r);
    } else {
        hipLaunchKernelGGL(k_lg_mat, lg_grid(n2), dim3(256), 0, s, d + cd.off, 1L, 0L, M1, r, rp, 0.0);
        if ((rc = lg_gemm(s, M2, 0, Xi, 0, M1, 0, rp, 1))) return rc;             // inv(L) D        (D symmetric)
    }
    if ((rc = lg_gemm(s, M3, 0, M2, 0, Xi, 0, rp, 1))) return rc;             // inv(L) D inv(L)'
    if (lz) {                                                                       // ... scaled by d^-1/2 on both sides
        hipLaunchKernelGGL(k_lg_lanczos1, dim3(1), dim3(512), LZ_LDS_DOUBLES * sizeof(double), s, M3, rp, (const double *)wx.dvec, r, 1,
                           scale, (const int *)wx.info, partial, cd.item, M1, stat, cert, cert_tol, lz_reorth());
        CIP_HIP_CHECK(hipGetLastError());
        if (cert) return lg_certify(s, w, M3, wx.dvec, r, 1, scale, M1, M2, wx, partial, cd.item, side);
        return 0;
    }
    if ((rc = lg_tridiag(s, w, M3, wx.dvec, r, M2))) return rc;
    hipLaunchKernelGGL(k_lg_sturm, dim3(1), dim3(LG_T), 0, s, dg, of, r, 1, scale, (const int *)wx.info, partial, cd.item);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Wt[i, off + e] = (F^-T a_i)_e = vecm(Rinv mat(a_i) Rinv')_e for every row i of At (= column of A): two batched GEMMs over all
// columns (chunks when they do not fit), the second on the lower tiles only, then the tiled vecm pass; mat(a_i) comes from the
// per-upload cache when the workspace has one
void cip_sdp_large_invalidate(LargeWs *w) {
    if (w) for (auto &p : w->amat_src) p = nullptr;
}
int cip_sdp_large_scale_At(hipStream_t s, LargeWs *w, const ConeDesc &cd, int li, int n, const double *At, long ldat, double *Wt,
                           long ldwt) {
    const int r = cd.r, rp = w->rp;
    const long n2 = (long)rp * rp;
    const double *Rip = w->Rip + LG_NPAD * (size_t)li * n2;
    const long dim = (long)r * (r + 1) / 2;
    int rc;
    const bool cached = w->amat && n == w->ncols && n <= w->chunk;
    double *amat = cached ? w->amat + (size_t)li * n * n2 : nullptr;
    if (cached && w->amat_src[li] != At) {
        dim3 gm((unsigned)((n2 + 255) / 256), n);
        hipLaunchKernelGGL(k_lg_mat, gm, dim3(256), 0, s, At + (long)cd.aoff * ldat, ldat, 1L, amat, r, rp, 0.0);
        w->amat_src[li] = At;
    }
    for (int i0 = 0; i0 < n; i0 += w->chunk) {
        const int nb = (n - i0 < w->chunk) ? (n - i0) : w->chunk;
        const double *X = amat;
        if (!cached) {
            dim3 gm((unsigned)((n2 + 255) / 256), nb);
            hipLaunchKernelGGL(k_lg_mat, gm, dim3(256), 0, s, At + i0 + (long)cd.aoff * ldat, ldat, 1L, w->batchX, r, rp, 0.0);
            X = w->batchX;
        }
        if ((rc = lg_gemm(s, w->batchT, n2, Rip, 0, X, n2, rp, nb))) return rc;                  // Rinv X      (X symmetric)
        if ((rc = lg_gemm(s, w->batchX, n2, w->batchT, n2, Rip, 0, rp, nb, 3))) return rc;     // (Rinv X) Rinv': lower tiles, all k_lg_vecm_cols reads
        dim3 gv((unsigned)((dim + 63) / 64), (unsigned)((nb + 63) / 64));
        hipLaunchKernelGGL(k_lg_vecm_cols, gv, dim3(256), 0, s, w->batchX, nb, Wt + i0 + (long)cd.aoff * ldwt, ldwt, r, rp);
    }
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// out = vecm(P' mat(x) P), P = R (F), R' (F'), Rinv (F^-1), Rinv' (F^-T)  (VecCongurance, src/ConicIP.jl:35-40, :69):
// two chip-wide GEMMs  Y = Q X Q'  with Q = P' taken from the padded copies
int cip_sdp_large_apply(hipStream_t s, LargeWs *w, const ConeDesc &cd, int li, int mode, const double *x, double *out) {
    const int r = cd.r, rp = w->rp;
    const long n2 = (long)rp * rp;
    // pad[]: 0 Rinv, 1 Rinv', 2 R, 3 R'
    const int which = (mode == CIP_OP_F) ? 3 : (mode == CIP_OP_FT) ? 2 : (mode == CIP_OP_FINV) ? 1 : 0;
    const double *Q = w->Rip + (LG_NPAD * (size_t)li + which) * n2;
    int rc;
    if (rp <= 256 && lg_small_gemm()) {
        // two launches: mat(x) is read from the vector by the first product, the second writes vecm (k_gemm_nt_small<BVEC / CVEC>)
        hipLaunchKernelGGL((k_gemm_nt_small<true, false>), dim3(rp / 16, rp / 16), dim3(256), 0, s, Q, (long)rp, x + cd.off, 0L, w->M2, (long)rp, rp, r);
        hipLaunchKernelGGL((k_gemm_nt_small<false, true>), dim3(rp / 16, rp / 16), dim3(256), 0, s, (const double *)w->M2, (long)rp, Q, (long)rp, out + cd.off, 0L, rp, r);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_lg_mat, lg_grid(n2), dim3(256), 0, s, x + cd.off, 1L, 0L, w->M1, r, rp, 0.0);
    if ((rc = lg_gemm(s, w->M2, 0, Q, 0, w->M1, 0, rp, 1))) return rc;              // Q X      (X symmetric)
    if ((rc = lg_gemm(s, w->M3, 0, w->M2, 0, Q, 0, rp, 1))) return rc;              // (Q X) Q'
    hipLaunchKernelGGL(k_lg_vecm, lg_grid((long)r * r), dim3(256), 0, s, w->M3, out + cd.off, 1L, 0L, r, rp);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Jordan division by a DIAGONAL element (dsdc!, src/ConicIP.jl:347-353, with Y = diag(y): every division of the interior-point loop
// is by lambda = vecm(diag(Lambda)), :686): O_ij = X_ij / (y_i + y_j), element-wise on the vecm layout (the sqrt(2) of the off-diagonal
// entries cancels) -- chip-wide, one workgroup per matrix row, instead of one workgroup walking mat / divide / vecm over r^2
// entries (225 us at order 256).  k_lg_div_check raises flag[0] when y has a non-zero off-diagonal entry; k_lg_div_diag writes the
// quotient when it has none; the general kernel (sdp.hip: k_sdp_div) runs behind them and returns at once unless the flag is up.
__global__ __launch_bounds__(256) void k_lg_div_check(const double *y, int r, int *flag) {
    const int i = blockIdx.x;
    const double *row = y + lg_rowoff(i, r);
    int bad = 0;
    for (int j = 1 + threadIdx.x; j < r - i; j += 256) bad |= (row[j] != 0.0);
    if (threadIdx.x == 0 && !(row[0] == row[0])) bad = 1;                  // NaN on the diagonal: let the general path propagate it
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(flag, 1);
}
__global__ __launch_bounds__(256) void k_lg_div_diag(const double *x, const double *y, double *out, int r, const int *flag) {
    if (*flag) return;
    const int i = blockIdx.x;
    const long o = lg_rowoff(i, r);
    const double yi = y[o];
    for (int j = threadIdx.x; j < r - i; j += 256) out[o + j] = x[o + j] / (yi + y[lg_rowoff(i + j, r)]);
}
// returns with flag[0] = 1 when the general path must run for this cone (the caller launches it; it resets the flag)
int cip_sdp_large_div(hipStream_t s, LargeWs *w, const ConeDesc &cd, const double *x, const double *y, double *out, int *flag) {
    (void)w;
    hipLaunchKernelGGL(k_lg_div_check, dim3(cd.r), dim3(256), 0, s, y + cd.off, cd.r, flag);
    hipLaunchKernelGGL(k_lg_div_diag, dim3(cd.r), dim3(256), 0, s, x + cd.off, y + cd.off, out + cd.off, cd.r, (const int *)flag);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// out = x o y = vecm(XY + YX)  (xsdc!, src/ConicIP.jl:355-360): one chip-wide GEMM
int cip_sdp_large_prod(hipStream_t s, LargeWs *w, const ConeDesc &cd, const double *x, const double *y, double *out) {
    const int r = cd.r, rp = w->rp;
    const long n2 = (long)rp * rp;
    int rc;
    hipLaunchKernelGGL(k_lg_mat, lg_grid(n2), dim3(256), 0, s, x + cd.off, 1L, 0L, w->M1, r, rp, 0.0);
    hipLaunchKernelGGL(k_lg_mat, lg_grid(n2), dim3(256), 0, s, y + cd.off, 1L, 0L, w->M2, r, rp, 0.0);
    if ((rc = lg_gemm(s, w->M3, 0, w->M1, 0, w->M2, 0, rp, 1))) return rc;          // X Y'  = X Y  (Y symmetric)
    hipLaunchKernelGGL(k_lg_vecm_sym, lg_grid((long)r * r), dim3(256), 0, s, w->M3, out + cd.off, r, rp);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
