// KKT assembly on the device (level 2 of the plugin, before the factorisation).
//
//  CIP_ROUTE_SCHUR    K = [ Q + A'(F'F)^-1 A   G' ]   -- the elimination order of `pivot`
//                         [ G                  0  ]      (src/kktsolvers.jl:289-293, :316-338)
//  CIP_ROUTE_FULL3X3  K = [ -F'F  -A   0  ]            -- the literal 3x3 block assembly
//                         [ -A'    Q   G' ]               [Q G' -A'; G 0 0; A 0 F'F]
//                         [  0     G   0  ]               (src/kktsolvers.jl:254-256), symmetrised and
//                                                         permuted to the static pivot order (3,1,2)
// Only the lower triangle is written / referenced; K is padded to a multiple of 128 with an
// identity block.
#include "cip_handle.h"
#include <atomic>
#include "../../include/cipkkt.h"

// rows >= n of the Schur-route matrix: G block, zero block, identity padding (lower part)
__global__ __launch_bounds__(256) void k_fill_rest(double *K, long ldk, int n, int p, int Npad, const double *G, long ldg, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, K, G);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Npad) return;
    for (int i = n + blockIdx.y; i < Npad; i += gridDim.y) {
        if (j > i) continue;
        double v;
        if (i < n + p) v = (j < n) ? G[(i - n) + (long)j * ldg] : 0.0;
        else v = (i == j) ? 1.0 : 0.0;
        K[i + (long)j * ldk] = v;
    }
}

// K[r0 + i, c0 + j] = sign * M[i + j*ldm]   (i < rows, j < cols); coalesced along i
__global__ __launch_bounds__(256) void k_copy_block(double *K, long ldk, int r0, int c0, const double *M, long ldm,
                                                     int rows, int cols, double sign, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, K, M);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    for (int j = blockIdx.y; j < cols; j += gridDim.y)
        K[(r0 + i) + (long)(c0 + j) * ldk] = sign * M[i + (long)j * ldm];
}

// The same for a square block on the diagonal (r0 == c0), restricted to the 128x128 tiles of K on or below the
// diagonal: the factorisation never references a tile above it, and Q is half of the assembly's HBM traffic
// (n = 8192: 1.07 GB -> 0.56 GB per factorisation).
__global__ __launch_bounds__(256) void k_copy_block_lower(double *K, long ldk, int r0, const double *M, long ldm, int n, double sign, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, K, M);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int jend = min(n, (((r0 + i) >> 7) + 1) * 128 - r0);      // columns up to the end of this row's diagonal tile
    for (int j = blockIdx.y; j < jend; j += gridDim.y)
        K[(r0 + i) + (long)(r0 + j) * ldk] = sign * M[i + (long)j * ldm];
}
// The 16-byte form for even n, r0 and leading dimensions and 16-byte-aligned bases: a workgroup moves 512 rows x 8 columns
// per step, a thread's eight loads in flight before its first store (one column of 512 rows per workgroup -- the first
// 16-byte version, 262144 workgroups at n = 8192 -- ran at 2.8 TB/s: 0.19 ms of a 6.1-ms step).
#define COPY_COLS 8
__global__ __launch_bounds__(256) void k_copy_block_lower_v(double *K, long ldk, int r0, const double *M, long ldm, int n, double sign, int jmax, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, K, M);
    const int i = 2 * (blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    const int jend = min(min(n, jmax), (((r0 + i) >> 7) + 1) * 128 - r0);      // columns up to the end of this row pair's diagonal tile (and < jmax)
    for (int j0 = blockIdx.y * COPY_COLS; j0 < jend; j0 += gridDim.y * COPY_COLS) {
        v2d v[COPY_COLS];
#pragma unroll
        for (int u = 0; u < COPY_COLS; ++u)
            if (j0 + u < jend) v[u] = *(const v2d *)(M + i + (long)(j0 + u) * ldm);
#pragma unroll
        for (int u = 0; u < COPY_COLS; ++u)
            if (j0 + u < jend) *(v2d *)(K + (r0 + i) + (long)(r0 + j0 + u) * ldk) = sign * v[u];
    }
}
static bool copy_lower_vectorisable(const double *K, long ldk, int r0, const double *M, long ldm, int n) {
    return !(n & 1) && !(r0 & 1) && !(ldk & 1) && !(ldm & 1) && !(((uintptr_t)K | (uintptr_t)M) & 15);
}
// jmax < n: only the columns [0, jmax) (the lazy copy of the Schur route)
static void launch_copy_block_lower(hipStream_t s, double *K, long ldk, int r0, const double *M, long ldm, int n, double sign, int jmax = 1 << 30) {
    if (copy_lower_vectorisable(K, ldk, r0, M, ldm, n)) {
        const int nc = n < jmax ? n : jmax;
        const int gy = (nc + COPY_COLS - 1) / COPY_COLS;
        cip_launch_b(k_copy_block_lower_v, dim3((n + 511) / 512, gy < 32768 ? gy : 32768), dim3(256), 0, s, K, ldk, r0, M, ldm, n, sign, jmax);
    } else {
        cip_launch_b(k_copy_block_lower, dim3((n + 255) / 256, n < 32768 ? n : 32768), dim3(256), 0, s, K, ldk, r0, M, ldm, n, sign);
    }
}

// ---------------------------------------------------------------- sparse-A Schur terms (R and Q cones)
// K[i, j] += sum_r w_r a_ri a_rj  (i >= j), w_r = 1/d_r^2 (R), -J_rr/beta^2 (Q).  One thread per variable i OWNS row i
// of the lower triangle: it walks column i of A (= row i of the CSR of A', rows r ascending) and, for every r, row r
// of A.  No atomics, fixed summation order: bit-reproducible (the first version scattered with unsafeAtomicAdd per
// row of A).
__device__ __forceinline__ double schur_row_weight(const ConeDesc &cd, const double *scal, int r) {
    if (cd.type == CIP_CONE_R) { const double d = scal[cd.soff + (r - cd.off)]; return 1.0 / (d * d); }
    if (cd.type == CIP_CONE_S) return 0.0;                  // (never used: the callers skip S rows, whose (F'F)^-1 is not diagonal)
    const double beta = scal[cd.soff];
    return ((r == cd.off) ? -1.0 : 1.0) / (beta * beta);
}
__global__ __launch_bounds__(256) void k_schur_rows(int n, const int *trp, const int *tci, const double *tv, const int *rp,
                                                     const int *ci, const double *av, const int *row_cone,
                                                     const ConeDesc *cones, const double *scal, double *K, long ldk, int base, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO8(cb, trp, tci, tv, rp, ci, av, scal, K);
    // Round 4: one WAVE per variable i (it was one thread: n = 500 with a 10 %-dense A -- the reference's "many small SOCs"
    // benchmark, benchmark/profile.jl:54-69 -- put 3750 dependent read-modify-writes on each of 500 threads, 2.3 ms beside a
    // 0.19-ms factorisation).  The rows r of column i are still taken one after the other, ascending; the entries of row r
    // (distinct columns j) go to the lanes.  Entry (i, j) therefore receives the same additions in the same order as
    // before: bit-identical, no atomics.  Two rows may hit the same (i, j) from different lanes: the wave's accesses to K go
    // to L2 (agent scope) and one row's stores are waited for before the next row's loads.
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    double *Ki = K + (base + i) + (long)base * ldk;
    for (int q = trp[i]; q < trp[i + 1]; ++q) {
        const int r = tci[q];
        const ConeDesc rc = cones[row_cone[r]];
        if (rc.type == CIP_CONE_S) continue;                // S rows: dense congruences + a GEMM (assemble_schur)
        const double wa = schur_row_weight(rc, scal, r) * tv[q];
        const int b1 = rp[r + 1];
        bool wrote = false;
        for (int b = rp[r] + lane; b < b1; b += 64) {
            const int j = ci[b];
            if (j <= i) {
                double *kp = Ki + (long)j * ldk;
                double x = __hip_atomic_load(kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x += wa * av[b];
                __hip_atomic_store(kp, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wrote = true;
            }
        }
        (void)wrote;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
// The same for an A with at most one entry per row and R cones only -- the Schur terms are then on the diagonal -- when the
// copy of Q is lazy: rows below `nb0` deliver K_ii = Q_ii + sum_r w_r a_ri^2 (the same fused multiply-adds in the same order
// on the same start value as k_schur_rows on the copied Q_ii) into kdiag[i] instead of K
__global__ __launch_bounds__(256) void k_schur_diag_lazy(int n, int nb0, const int *trp, const int *tci, const double *tv, const int *rp,
                                                          const int *ci, const double *av, const int *row_cone, const ConeDesc *cones,
                                                          const double *scal, const double *Q, long ldq, double *K, long ldk, double *kdiag,
                                                          CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO8(cb, trp, tci, tv, rp, ci, av, scal, Q);
    CIP_BO2(cb, K, kdiag);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double d = i < nb0 ? K[i + (long)i * ldk] : Q[i + (long)i * ldq];
    for (int q = trp[i]; q < trp[i + 1]; ++q) {
        const int r = tci[q];
        const double wa = schur_row_weight(cones[row_cone[r]], scal, r) * tv[q];
        for (int b = rp[r]; b < rp[r + 1]; ++b)
            if (ci[b] == i) d += wa * av[b];
    }
    if (i < nb0) K[i + (long)i * ldk] = d;
    else kdiag[i] = d;
}
// Gm[i, qidx] = sum over the rows r of Q cone qidx of (sqrt2/beta) (J wbar)_r A[r, i]: same ownership (thread i, column i of
// A in ascending r; the rows of one cone are consecutive)
__global__ __launch_bounds__(256) void k_schur_qcols(int n, const int *trp, const int *tci, const double *tv,
                                                      const int *row_cone, const ConeDesc *cones, const double *scal,
                                                      double *Gm, long ldgm, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO5(cb, trp, tci, tv, scal, Gm);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cur = -1;
    double acc = 0.0;
    for (int q = trp[i]; q < trp[i + 1]; ++q) {
        const int r = tci[q];
        const ConeDesc cd = cones[row_cone[r]];
        if (cd.type != CIP_CONE_Q) continue;
        if (cd.qidx != cur) {
            if (cur >= 0) Gm[i + (long)cur * ldgm] = acc;
            cur = cd.qidx; acc = 0.0;
        }
        const double beta = scal[cd.soff];
        const double *w = scal + cd.soff + 1;
        const int e = r - cd.off;
        // wbar_1 = w1^2/beta - 1 ; wbar_t = w1 w_t / beta ; (J wbar)_t = -wbar_t
        const double jw = (e == 0) ? (w[0] * w[0] / beta - 1.0) : -(w[0] * w[e] / beta);
        acc += (jw * 1.4142135623730951 / beta) * tv[q];
    }
    if (cur >= 0) Gm[i + (long)cur * ldgm] = acc;
}

// CSR A with S cones: AtS[i, aoff_c + (r - off_c)] = A[r, i] for the rows r of every S cone c (thread i walks column i of A)
__global__ __launch_bounds__(256) void k_scatter_AtS(int n, const int *trp, const int *tci, const double *tv, const int *row_cone,
                                                      const ConeDesc *cones, double *AtS, long ld, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, trp, tci, tv, AtS);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int q = trp[i]; q < trp[i + 1]; ++q) {
        const int r = tci[q];
        const ConeDesc cd = cones[row_cone[r]];
        if (cd.type == CIP_CONE_S) AtS[i + (long)(cd.aoff + (r - cd.off)) * ld] = tv[q];
    }
}
int cip_scatter_AtS(cip_handle *h) {
    if (!h->AtS || h->n == 0) return 0;
    CIP_HIP_CHECK(hipMemsetAsync(h->AtS, 0, sizeof(double) * (size_t)h->npad * h->mSpad, h->stream));      // (cip_update_problem: new values)
    hipLaunchKernelGGL(k_scatter_AtS, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->n, h->T_rp, h->T_ci, h->T_v, h->row_cone,
                       h->cs.d_cones, h->AtS, (long)h->npad, CipBatch{0, 1ull});
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- full 3x3 route: -F'F block
__global__ __launch_bounds__(256) void k_fill_ftf(const ConeDesc *cones, const WorkItem *items, const double *scal,
                                                   double *K, long ldk, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, scal, K);
    const WorkItem it = items[blockIdx.x];
    const ConeDesc cd = cones[it.cone];
    const int tid = threadIdx.x;
    if (cd.type == CIP_CONE_R) {
        for (int e = it.start + tid; e < it.start + it.len; e += 256) {
            const double d = scal[cd.soff + e];
            const long g = cd.off + e;
            K[g + g * ldk] = -d * d;
        }
    } else if (cd.type == CIP_CONE_Q) {
        // F'F = F^2 = beta^2 (2 wbar wbar' - J); a pack of small cones: the workgroup walks through its cones
        if (!it.width) return;                            // a large cone: k_fill_ftf_qbig
        const int ncone = it.len;
        for (int q = 0; q < ncone; ++q) {
            const ConeDesc qc = cones[it.cone + q];
            const int k = qc.dim;
            const double beta = scal[qc.soff];
            const double *w = scal + qc.soff + 1;
            const double b2 = beta * beta, w0 = w[0];
            for (long e = tid; e < (long)k * k; e += 256) {
                const int i = (int)(e % k), j = (int)(e / k);
                if (i < j) continue;
                const double wi = (i == 0) ? (w0 * w0 / beta - 1.0) : (w0 * w[i] / beta);
                const double wj = (j == 0) ? (w0 * w0 / beta - 1.0) : (w0 * w[j] / beta);
                double v = 2.0 * wi * wj;
                if (i == j) v -= (i == 0) ? 1.0 : -1.0;
                K[(qc.off + i) + (long)(qc.off + j) * ldk] = -b2 * v;
            }
        }
    }
}
// -F'F of a Q cone of dimension > 64: the k (k + 1) / 2 entries of its lower triangle by column, many workgroups (round 4: one
// workgroup walked all k^2 of them, 19 ms at k = 4097)
__global__ __launch_bounds__(256) void k_fill_ftf_qbig(const ConeDesc *cones, const WorkItem *items, const int *bigq, const double *scal,
                                                        double *K, long ldk, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, scal, K);
    const WorkItem it = items[bigq[blockIdx.x]];
    const ConeDesc qc = cones[it.cone];
    const int k = qc.dim;
    const double beta = scal[qc.soff];
    const double *w = scal + qc.soff + 1;
    const double b2 = beta * beta, w0 = w[0];
    for (int j = blockIdx.y; j < k; j += gridDim.y) {
        const double wj = (j == 0) ? (w0 * w0 / beta - 1.0) : (w0 * w[j] / beta);
        for (int i = j + threadIdx.x; i < k; i += 256) {
            const double wi = (i == 0) ? (w0 * w0 / beta - 1.0) : (w0 * w[i] / beta);
            double v = 2.0 * wi * wj;
            if (i == j) v -= (i == 0) ? 1.0 : -1.0;
            K[(qc.off + i) + (long)(qc.off + j) * ldk] = -b2 * v;
        }
    }
}
__global__ __launch_bounds__(256) void k_scatter_negA(int nrowsT, const int *trp, const int *tci, const double *tv,
                                                       double *K, long ldk, int r0, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, trp, tci, tv, K);
    // CSR of A' (row i = variable, column = constraint r):  K[r0 + i, r] = -A[r, i]
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nrowsT) return;
    for (int q = trp[i]; q < trp[i + 1]; ++q) K[(r0 + i) + (long)tci[q] * ldk] = -tv[q];
}
__global__ __launch_bounds__(256) void k_pad_identity(double *K, long ldk, int N, int Npad, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO1(cb, K);
    const int i = N + blockIdx.x * 256 + threadIdx.x;
    if (i < Npad) K[i + (long)i * ldk] = 1.0;
}

// lazy_ok (the caller runs cip_ldlt_factor on the result right away): with a CSR A that has one entry per row, R cones only,
// no equality block and no padding, K = Q + diag(..) -- and 3/4 (order 2048) to 9/10 (order 8192) of the copy of Q into K
// is overwritten by the FIRST trailing update without anybody else having looked at it.  Then only the first outer block's
// columns are copied; the first trailing update takes its C operand from Q itself with the Schur diagonal from h->kdiag
// (EPI_LAZYC) and writes K.  Every entry of K goes through the same operations as with the full copy (bit-identical
// factor): 115 -> 25 us per factorisation at n = 8192, 7 of the 86 ms of a config-5 pass.
int cip_ldlt_outer_block_for(int Npad);
static std::atomic<int> g_lazy_copy{-1};  // CIP_LAZY_COPY / cip_set_lazy_copy: 1 (default) on, 0 off; read by worker threads
int cip_lazy_copy_set(int on) {
    int prev = g_lazy_copy.load(std::memory_order_relaxed);
    if (prev < 0) {                                                  // first use: the environment decides (racing threads agree)
        const char *e = getenv("CIP_LAZY_COPY");
        const int env = e ? (atoi(e) != 0) : 1;
        int expect = -1;
        g_lazy_copy.compare_exchange_strong(expect, env);
        prev = g_lazy_copy.load(std::memory_order_relaxed);
    }
    if (on == 0 || on == 1) g_lazy_copy.store(on);
    return prev;
}
static int assemble_schur(cip_handle *h, bool lazy_ok) {
    hipStream_t s = h->stream;
    int rc;
    const int n = h->n, p = h->p;
    h->ws.lazyC = nullptr;
    if (!h->A_sparse) {
        if ((rc = cip_cones_scale_At(s, h->cs, n, h->At, h->npad, h->Wt, h->npad))) return rc;
        GemmArgs g = {};
        g.A = h->Wt; g.lda = h->npad; g.B = h->Wt; g.ldb = h->npad;
        g.C = h->K; g.ldc = h->ldk; g.M = h->npad; g.N = h->npad; g.K = h->mpad;
        g.alpha = 1.0; g.lower = 1; g.Qin = h->Q; g.ldq = n; g.nvalid = n;
        g.ksplit_ws = h->syrk_ws; g.ksplit_n = h->syrk_n; g.ksplit_len = h->syrk_len;
        // (algorithmic work of the Schur formation: m n^2 flop on the lower half)
        if ((rc = cip_prof_slot_begin(CIP_PROF_SYRK, s, (double)h->m * (double)n * (double)n * (cip_in_batch() ? (double)__builtin_popcountll(cip_tl_bz.mask) : 1.0)))) return rc;
        if ((rc = cip_launch_gemm(s, EPI_SYRKQ, g))) return rc;
        if ((rc = cip_prof_slot_end(CIP_PROF_SYRK, s))) return rc;
    } else {
        const int lazy_now = g_lazy_copy.load(std::memory_order_relaxed);
        const int lazy_on = lazy_now < 0 ? cip_lazy_copy_set(-1) : lazy_now;
        bool all_r = h->m > 0 && h->nq == 0 && !h->cs.has_S;
        const int nb0 = cip_ldlt_outer_block_for(h->Npad);
        if (lazy_on && lazy_ok && all_r && h->A_one_per_row && p == 0 && h->Npad == n && n > nb0 && h->reg_rel <= 0.0 && h->kdiag &&
            copy_lower_vectorisable(h->K, h->ldk, 0, h->Q, (long)n, n)) {
            launch_copy_block_lower(s, h->K, h->ldk, 0, h->Q, (long)n, n, 1.0, nb0);
            cip_launch_b(k_schur_diag_lazy, dim3((n + 255) / 256), dim3(256), 0, s, n, nb0, h->T_rp, h->T_ci, h->T_v, h->A_rp, h->A_ci, h->A_v,
                         h->row_cone, h->cs.d_cones, h->cs.d_scal, h->Q, (long)n, h->K, h->ldk, h->kdiag);
            CIP_HIP_CHECK(hipGetLastError());
            h->ws.lazyC = h->Q; h->ws.lazy_ld = n; h->ws.lazy_diag = h->kdiag;
            return 0;
        }
        if (n > 0) {
            launch_copy_block_lower(s, h->K, h->ldk, 0, h->Q, (long)n, n, 1.0);
        }
        if (h->m > 0) {
            cip_launch_b(k_schur_rows, dim3((n + 3) / 4), dim3(256), 0, s, n, h->T_rp, h->T_ci, h->T_v, h->A_rp, h->A_ci,
                               h->A_v, h->row_cone, h->cs.d_cones, h->cs.d_scal, h->K, h->ldk, 0);
            if (h->nq > 0) {
                if ((rc = cip_zero(s, (long)h->npad * h->nqpad, h->Gm))) return rc;
                cip_launch_b(k_schur_qcols, dim3((n + 255) / 256), dim3(256), 0, s, n, h->T_rp, h->T_ci, h->T_v,
                                   h->row_cone, h->cs.d_cones, h->cs.d_scal, h->Gm, (long)h->npad);
            }
        }
    }
    if (h->Npad > n) {
        cip_launch_b(k_fill_rest, dim3((h->Npad + 255) / 256, (h->Npad - n) < 32768 ? (h->Npad - n) : 32768), dim3(256), 0, s, h->K, h->ldk, n, p,
                           h->Npad, h->G, (long)p);
    }
    if (h->A_sparse && h->AtS) {
        // the S cones' part of A'(F'F)^-1 A: W = A_S' F^-1 by congruences on the dense block of their rows, then K += W W'
        if ((rc = cip_sdp_scale_At(s, h->cs, n, h->AtS, (long)h->npad, h->WtS, (long)h->npad))) return rc;
        GemmArgs g = {};
        g.A = h->WtS; g.lda = h->npad; g.B = h->WtS; g.ldb = h->npad;
        g.C = h->K; g.ldc = h->ldk; g.M = h->npad; g.N = h->npad; g.K = h->mSpad; g.alpha = 1.0; g.lower = 1;
        if ((rc = cip_launch_gemm(s, EPI_ACCUM, g))) return rc;
    }
    if (h->A_sparse && h->nq > 0 && h->m > 0) {
        // after k_fill_rest: the rank-nq update touches whole 128-tiles (adds exact zeros outside [0,n)^2)
        GemmArgs g = {};
        g.A = h->Gm; g.lda = h->npad; g.B = h->Gm; g.ldb = h->npad;
        g.C = h->K; g.ldc = h->ldk; g.M = h->npad; g.N = h->npad; g.K = h->nqpad; g.alpha = 1.0; g.lower = 1;
        if ((rc = cip_launch_gemm(s, EPI_ACCUM, g))) return rc;
    }
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cip_sdp_fill_ftf(hipStream_t s, const ConeSet &cs, double *K, long ldk);     // sdp.hip

static int assemble_full(cip_handle *h) {
    hipStream_t s = h->stream;
    const int n = h->n, m = h->m, p = h->p;
    { int rcz = cip_zero(s, (long)h->ldk * h->Npad, h->K); if (rcz) return rcz; }
    if (h->cs.nitems > 0)
        cip_launch_b(k_fill_ftf, dim3(h->cs.nitems), dim3(256), 0, s, h->cs.d_cones, h->cs.d_items, h->cs.d_scal,
                           h->K, h->ldk);
    if (h->cs.nbigq > 0)
        cip_launch_b(k_fill_ftf_qbig, dim3(h->cs.nbigq, 128), dim3(256), 0, s, h->cs.d_cones, h->cs.d_items, (const int *)h->cs.d_bigq, h->cs.d_scal, h->K, h->ldk);
    if (h->cs.has_S) { int rc = cip_sdp_fill_ftf(s, h->cs, h->K, h->ldk); if (rc) return rc; }
    if (m > 0 && n > 0) {
        if (!h->A_sparse)
            cip_launch_b(k_copy_block, dim3((n + 255) / 256, m < 32768 ? m : 32768), dim3(256), 0, s, h->K, h->ldk, m, 0, h->At,
                               (long)h->npad, n, m, -1.0);
        else
            cip_launch_b(k_scatter_negA, dim3((n + 255) / 256), dim3(256), 0, s, n, h->T_rp, h->T_ci, h->T_v,
                               h->K, h->ldk, m);
    }
    if (n > 0)
        launch_copy_block_lower(s, h->K, h->ldk, m, h->Q, (long)n, n, 1.0);
    if (p > 0)
        cip_launch_b(k_copy_block, dim3((p + 255) / 256, n < 32768 ? n : 32768), dim3(256), 0, s, h->K, h->ldk, m + n, m, h->G,
                           (long)p, p, n, 1.0);
    if (h->Npad > h->N)
        cip_launch_b(k_pad_identity, dim3((h->Npad - h->N + 255) / 256), dim3(256), 0, s, h->K, h->ldk, h->N, h->Npad);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Static regularisation, scaled per row: K_ii += s_i * rel * max_j |K_ij| (the symmetric matrix' full row i: its stored
// row part j <= i and its column part below the diagonal), s_i = +1 on the positive-pivot block [p0, p1), -1 elsewhere.
// Late interior-point iterates spread the diagonal of S over 20 orders of magnitude; one global delta either drowns
// the small rows or does nothing for the large ones.
__global__ __launch_bounds__(256) void k_rowmax_lower(const double *K, long ldk, int N, double *rowmax, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, K, rowmax);
    // thread <-> row i: the row part K[i, 0..i] (coalesced across the threads of a workgroup)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    double mx = 0.0;
    for (int j = 0; j <= i; ++j) mx = fmax(mx, fabs(K[i + (long)j * ldk]));
    rowmax[i] = mx;
}
__global__ __launch_bounds__(256) void k_regularize_rows(double *K, long ldk, int N, int p0, int p1, double rel, const double *rowmax, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, K, rowmax);
    // workgroup <-> column i: the column part K[i+1.., i], then the diagonal update
    __shared__ double red[4];
    const int i = blockIdx.x;
    double mx = 0.0;
    for (int r = i + 1 + threadIdx.x; r < N; r += 256) mx = fmax(mx, fabs(K[r + (long)i * ldk]));
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double delta = rel * fmax(fmax(fmax(red[0], red[1]), fmax(red[2], red[3])), rowmax[i]);
        K[i + (long)i * ldk] += (i >= p0 && i < p1) ? delta : -delta;
    }
}

int cip_assemble(cip_handle *h, bool lazy_ok) {
    h->ws.lazyC = nullptr;
    // K is about to be overwritten: the side stream may still be reading the previous factor (a factorisation without solves)
    { const int rj = cip_ldlt_side_join(h->stream, h->ws, -1); if (rj) return rj; }
    int rc = (h->route == CIP_ROUTE_SCHUR) ? assemble_schur(h, lazy_ok) : assemble_full(h);
    if (rc == 0 && h->reg_rel > 0.0) {
        // (the diagonal is modified only after every row / column maximum has been read: the second kernel reads
        //  column i strictly below the diagonal, the first one has finished before it starts)
        cip_launch_b(k_rowmax_lower, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->K, h->ldk, h->N, h->rhs);
        cip_launch_b(k_regularize_rows, dim3(h->N), dim3(256), 0, h->stream, h->K, h->ldk, h->N, h->ws.signs.p0,
                           h->ws.signs.p1, h->reg_rel, h->rhs);
        CIP_HIP_CHECK(hipGetLastError());
    }
    if (rc == 0) { h->assembled = true; h->factored = false; }
    return rc;
}
