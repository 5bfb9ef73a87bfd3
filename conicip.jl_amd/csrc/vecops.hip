// HBM-bound vector helpers for the device-resident Newton step: gemv with the
// problem matrices (the residual mat-vecs of src/ConicIP.jl:747-749, :912-914 and the
// A / A' products of src/kktsolvers.jl:326-328), multi-dot reductions, axpby.
//
// Every dense product is expressed as a column-dot "T" gemv, y[j] = sum_i M[i + j*ld] x[i]
// (the handle keeps both orientations of A and G; Q is symmetric), so that lanes read
// consecutive addresses and results are deterministic (no atomics).
#include "cip_internal.h"

__device__ __forceinline__ double wsum(double v) { return cip_wave_sum(v); }      // (round 4: DPP / lane swaps instead of six ds_bpermute round trips)

// one wave per column, 4 columns per workgroup; 4 x 16-byte loads of the matrix in flight per lane (the first version
// had one: the triangular solves ran at 2.1 TB/s)
__global__ __launch_bounds__(256) void k_gemv_t(int rows, int cols, double alpha, const double *A, long lda,
                                                 const double *x, double beta, double *y, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO3(cb, A, x, y);
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= cols) return;
    const double *a = A + (long)j * lda;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = lane * 2;
    // rows even and 16-byte aligned columns are the common case (padded leading dimensions)
    if (((lda & 1) == 0) && ((((uintptr_t)A) & 15) == 0) && ((((uintptr_t)x) & 15) == 0)) {
        // (round 6) 1024 rows per trip, EIGHT 16-byte loads of the matrix in flight per lane: the block steps of the triangular sweeps are
        // column dots over exactly 1024 rows (the solve block), i.e. two dependent memory round trips per wave with four loads in flight --
        // and those launches are latency-bound (a 58-MB step at 3.6 TB/s, an 8-MB one in 4.6 us).  Same fused multiply-adds on the same
        // accumulators in the same order as two trips of the 512-row loop below: same bits.  solve4x4 at n = 8192 0.203 -> 0.199 ms.
        // (NOT kept: skipping the 128-row pieces of the triangular block inverses that are exact zeros -- half the bytes of every
        //  block product, same bits -- made the predicated loads issue one behind the other: 0.199 -> 0.220 ms.)
        for (; i + 897 < rows; i += 1024) {
            v2d av[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = *(const v2d *)(a + i + 128 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = *(const v2d *)(x + i + 128 * u);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                s0 = fma(av[4 * h].x, xv[4 * h].x, s0); s1 = fma(av[4 * h].y, xv[4 * h].y, s1);
                s2 = fma(av[4 * h + 1].x, xv[4 * h + 1].x, s2); s3 = fma(av[4 * h + 1].y, xv[4 * h + 1].y, s3);
                s0 = fma(av[4 * h + 2].x, xv[4 * h + 2].x, s0); s1 = fma(av[4 * h + 2].y, xv[4 * h + 2].y, s1);
                s2 = fma(av[4 * h + 3].x, xv[4 * h + 3].x, s2); s3 = fma(av[4 * h + 3].y, xv[4 * h + 3].y, s3);
            }
        }
        for (; i + 385 < rows; i += 512) {
            const v2d a0 = *(const v2d *)(a + i), a1 = *(const v2d *)(a + i + 128), a2 = *(const v2d *)(a + i + 256),
                      a3 = *(const v2d *)(a + i + 384);
            const v2d x0 = *(const v2d *)(x + i), x1 = *(const v2d *)(x + i + 128), x2 = *(const v2d *)(x + i + 256),
                      x3 = *(const v2d *)(x + i + 384);
            s0 = fma(a0.x, x0.x, s0); s1 = fma(a0.y, x0.y, s1);
            s2 = fma(a1.x, x1.x, s2); s3 = fma(a1.y, x1.y, s3);
            s0 = fma(a2.x, x2.x, s0); s1 = fma(a2.y, x2.y, s1);
            s2 = fma(a3.x, x3.x, s2); s3 = fma(a3.y, x3.y, s3);
        }
        for (; i + 1 < rows; i += 128) {
            const v2d av = *(const v2d *)(a + i);
            const v2d xv = *(const v2d *)(x + i);
            s0 = fma(av.x, xv.x, s0);
            s1 = fma(av.y, xv.y, s1);
        }
        if (i < rows) s0 += a[i] * x[i];
    } else {
        for (i = lane; i < rows; i += 64) s0 += a[i] * x[i];
    }
    const double s = wsum((s0 + s1) + (s2 + s3));
    if (lane == 0) y[j] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[j]);
}

int cip_gemv_t(hipStream_t s, int rows, int cols, double alpha, const double *A, long lda, const double *x,
               double beta, double *y) {
    if (cols <= 0) return 0;
    cip_launch_b(k_gemv_t, dim3((cols + 3) / 4), dim3(256), 0, s, rows, cols, alpha, A, lda, x, beta, y);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Symmetric mat-vec from the tiles on and below the diagonal only: y = alpha Q x + beta y for a dense symmetric Q of order
// n (n a multiple of 128, even leading dimension, 16-byte aligned) with HALF the HBM traffic of the column-dot gemv -- Q x is
// the residual product of every interior-point iteration (src/ConicIP.jl:747, :912), 0.54 GB at n = 8192 and three passes
// over all 64 matrices per iteration of a config-5 batch.  Deterministic (no atomics), two launches:
//   k_symv_tiles   one workgroup per 128 x 128 tile (I >= J), read once: the row sums  T x_J  (a piece of y_I) and, below the
//                  diagonal, the column dots  T' x_I  (a piece of y_J) go to two partial-sum tables P1[J][i], P2[I][j]
//   k_symv_reduce  y_i = alpha (sum_{J <= I(i)} P1[J][i] + sum_{I > I(i)} P2[I][i]) + beta y_i, fixed order
__global__ __launch_bounds__(256) void k_symv_tiles(int n, const double *Q, long ldq, const double *x, double *P1, double *P2, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, Q, x, P1, P2);
    __shared__ double rs[4][128];
    // tile index -> (I, J), row-major over the lower triangle
    const int t = blockIdx.x;
    int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)I * (I + 1) / 2 > t) --I;
    while ((long)(I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - (int)((long)I * (I + 1) / 2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *q = Q + (long)I * 128 + 2 * lane + (long)(J * 128 + wave * 32) * ldq;      // rows 2 lane, 2 lane + 1; the wave's 32 columns
    const v2d xi = *(const v2d *)(x + I * 128 + 2 * lane);
    const double *xj = x + J * 128 + wave * 32;
    v2d ra = {0.0, 0.0};
    double cp[32];                                // this lane's two-row share of the 32 column dots
    {
        v2d v[32];                                // the wave's whole 128 x 32 slice in flight at once
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = *(const v2d *)(q + (long)u * ldq);
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const double xc = xj[u];
            ra.x = fma(v[u].x, xc, ra.x);
            ra.y = fma(v[u].y, xc, ra.y);
            cp[u] = fma(v[u].x, xi.x, v[u].y * xi.y);
        }
    }
    rs[wave][2 * lane] = ra.x;
    rs[wave][2 * lane + 1] = ra.y;
    if (I != J) {
        // 32 sums over 64 lanes as ONE transposing butterfly (32 shuffles instead of 32 x 6): at offset o a lane keeps the
        // half of its values that its bit o selects, hands the other half to lane ^ o and adds what it receives
#pragma unroll
        for (int o = 32, m = 32; o >= 2; o >>= 1, m >>= 1) {
            const bool hi = (lane & o) != 0;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < m / 2) {
                    const double give = hi ? cp[k] : cp[k + m / 2];
                    const double keep = hi ? cp[k + m / 2] : cp[k];
                    cp[k] = keep + __shfl_xor(give, o);
                }
        }
        const double tot = cp[0] + __shfl_xor(cp[0], 1);
        const int col = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        if (!(lane & 1)) P2[(long)I * n + J * 128 + wave * 32 + col] = tot;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int r = threadIdx.x;
        P1[(long)J * n + I * 128 + r] = ((rs[0][r] + rs[1][r]) + rs[2][r]) + rs[3][r];
    }
}
__global__ __launch_bounds__(64) void k_symv_reduce(int n, const double *P1, const double *P2, double alpha, double beta, double *y, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO3(cb, P1, P2, y);
    const int i = blockIdx.x * 64 + threadIdx.x;                   // one wave per 64 outputs: n / 64 workgroups
    if (i >= n) return;
    const int Ii = i >> 7, nb = n >> 7;
    // term k of y_i: P1[k][i] for k <= I(i), P2[k][i] above -- eight loads in flight, added in index order
    double s = 0.0;
    for (int k0 = 0; k0 < nb; k0 += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u;
            t[u] = k < nb ? (k <= Ii ? P1 : P2)[(long)k * n + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k0 + u < nb) s += t[u];
    }
    y[i] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[i]);
}
// ws: 2 (n / 128) n doubles
int cip_symv_lower(hipStream_t s, int n, double alpha, const double *Q, long ldq, const double *x, double beta, double *y, double *ws) {
    const int nb = n / 128;
    double *P1 = ws, *P2 = ws + (size_t)nb * n;
    cip_launch_b(k_symv_tiles, dim3((unsigned)(nb * (nb + 1) / 2)), dim3(256), 0, s, n, Q, ldq, x, P1, P2);
    cip_launch_b(k_symv_reduce, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, n, (const double *)P1, (const double *)P2, alpha, beta, y);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// CSR spmv, one wave per row group: rows are short in the KKT use (identity-like A), so use
// thread-per-row for simplicity (coalescing over rows of rowptr; gathers from x).
__global__ __launch_bounds__(256) void k_spmv_csr(int rows, const int *rowptr, const int *colind, const double *val,
                                                   double alpha, const double *x, double beta, double *y, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO5(cb, rowptr, colind, val, x, y);
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    double s = 0;
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) s += val[q] * x[colind[q]];
    y[r] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[r]);
}

int cip_spmv_csr(hipStream_t s, int rows, const int *rowptr, const int *colind, const double *val, double alpha,
                 const double *x, double beta, double *y) {
    if (rows <= 0) return 0;
    cip_launch_b(k_spmv_csr, dim3((rows + 255) / 256), dim3(256), 0, s, rows, rowptr, colind, val, alpha, x, beta, y);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- multi-dot: stage 1 = (count x DOT_NB) partial sums, stage 2 = one block per dot
#define DOT_NB 32
struct DotPtrs { const double *x; const double *y; int len; int pad; };
#define DOT_MAX 32
// The pointer table travels BY VALUE in the kernel arguments (768 bytes; round 6, second session): until then every call
// uploaded it from the caller's stack with a hipMemcpyAsync -- a staged copy and a blit kernel in front of every dot-product group,
// three to four per interior-point iteration.
struct DotTable { DotPtrs p[DOT_MAX]; };

// In a batch the pointer table is problem 0's (shared): the vectors it names and the partial sums are shifted.
__global__ __launch_bounds__(256) void k_dots1(const DotTable tab, double *partial, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    __shared__ double sh[4];
    DotPtrs d = tab.p[blockIdx.y];
    CIP_BO3(cb, d.x, d.y, partial);
    double s = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < d.len; i += (long)DOT_NB * 256) s += d.x[i] * d.y[i];
    s = wsum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.y * DOT_NB + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// gather != NULL (batch): the result of problem z also goes to gather[z * CIP_GATHER + dot] (one read-back for all)
__global__ __launch_bounds__(64) void k_dots2(const double *partial, double *out, double *gather, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, partial, out);
    double s = (threadIdx.x < DOT_NB) ? partial[blockIdx.x * DOT_NB + threadIdx.x] : 0.0;
    s = wsum(s);
    if (threadIdx.x == 0) {
        out[blockIdx.x] = s;
        if (gather) gather[blockIdx.z * CIP_GATHER + blockIdx.x] = s;
    }
}

// Round 5: both stages in ONE launch.  The block that completes a dot's DOT_NB partial sums (a counter per dot, left at zero again
// for the next call) adds them up exactly as k_dots2 does -- the partials read with agent-scope loads: they were written by other
// CUs, and this CU's vector cache may still hold the previous call's values at the same addresses.  Same bits, one dependent
// launch (~4 us) less per dot-product group, three or four groups per interior-point iteration -- and NOT faster: the in-launch
// hand-off (write-through store, wait, atomic, coherent re-load) costs what the launch boundary does (8 / 64 problems of order 2048
// in lock-step: 16.7 / 79.8 ms per pass against 15.9 / 78.3 with two launches).  Off by default; CIP_DOTS_FUSED=1 selects it.
__global__ __launch_bounds__(256) void k_dots(const DotTable tab, double *partial, unsigned *cnt, double *out, double *gather, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    __shared__ double sh[4];
    __shared__ unsigned last;
    DotPtrs d = tab.p[blockIdx.y];
    CIP_BO3(cb, d.x, d.y, partial);
    CIP_BO2(cb, cnt, out);
    double s = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < d.len; i += (long)DOT_NB * 256) s += d.x[i] * d.y[i];
    s = wsum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        // written through (agent-scope store), waited for, then counted: NO release / acquire fences -- on this chip an agent-scope
        // release is an L2 write-back, and tens of thousands of blocks doing one each made a 64-problem lock-step pass 8 % slower
        // than the two-launch form (the library's other in-launch hand-offs are built the same way: diag.hip st_pub / ld_pub)
        __hip_atomic_store(&partial[blockIdx.y * DOT_NB + blockIdx.x], (sh[0] + sh[1]) + (sh[2] + sh[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(&cnt[blockIdx.y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == DOT_NB - 1;
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {
        asm volatile("" ::: "memory");
        double t = (threadIdx.x < DOT_NB) ? __hip_atomic_load(&partial[blockIdx.y * DOT_NB + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        t = wsum(t);
        if (threadIdx.x == 0) {
            out[blockIdx.y] = t;
            if (gather) gather[blockIdx.z * CIP_GATHER + blockIdx.y] = t;
            __hip_atomic_store(&cnt[blockIdx.y], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int cip_dots(hipStream_t s, int count, const double *const *x_host, const double *const *y_host, const int *len_host,
             double *scratch_dev, void *ptrs_dev, double *out_host) {
    if (count <= 0) return 0;
    if (count > DOT_MAX) { cip_set_error("dots: count > %d", DOT_MAX); return -1; }
    DotTable tab = {};
    for (int i = 0; i < count; ++i) { tab.p[i].x = x_host[i]; tab.p[i].y = y_host[i]; tab.p[i].len = len_host[i]; tab.p[i].pad = 0; }
    (void)ptrs_dev;                                  // (the device copy of the table: unused since the table is a kernel argument)
    double *partial = scratch_dev;
    double *out = scratch_dev + DOT_MAX * DOT_NB;
    const CipBatchCtx &bc = cip_tl_bz;
    CipHostScratch hs;
    int rc;
    if ((rc = cip_host_scratch(&hs))) return rc;
    const bool direct = bc.B <= 1 && count <= 512;                 // one problem: the sums go straight to the host
    static const int fused = [] { const char *e = getenv("CIP_DOTS_FUSED"); return e ? atoi(e) : 0; }();
    if (fused) {
        unsigned *cnt = (unsigned *)(scratch_dev + DOT_MAX * DOT_NB + DOT_MAX);         // zeroed when the scratch was allocated; every call leaves it zero
        cip_launch_b(k_dots, dim3(DOT_NB, count), dim3(256), 0, s, tab, partial, cnt, direct ? hs.dev : out,
                     bc.B > 1 ? bc.gather_dev : (double *)nullptr);
    } else {
        cip_launch_b(k_dots1, dim3(DOT_NB, count), dim3(256), 0, s, tab, partial);
        cip_launch_b(k_dots2, dim3(count), dim3(64), 0, s, (const double *)partial, direct ? hs.dev : out, bc.B > 1 ? bc.gather_dev : (double *)nullptr);
    }
    CIP_HIP_CHECK(hipGetLastError());
    if (bc.B > 1) {
        // out_host: B x count, problem-major; masked-off problems keep whatever the gather buffer held (callers ignore them)
        CIP_HIP_CHECK(hipMemcpyAsync(bc.gather_host, bc.gather_dev, sizeof(double) * bc.B * CIP_GATHER, hipMemcpyDeviceToHost, s));
        if ((rc = cip_wait(s))) return rc;
        for (int z = 0; z < bc.B; ++z)
            for (int i = 0; i < count; ++i) out_host[z * count + i] = bc.gather_host[z * CIP_GATHER + i];
        return 0;
    }
    if (!direct) CIP_HIP_CHECK(hipMemcpyAsync(out_host, out, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    if ((rc = cip_wait(s))) return rc;
    if (direct) for (int i = 0; i < count; ++i) out_host[i] = hs.host[i];
    return 0;
}

// ---- small results back to the host (cip_internal.h)
#include <stdlib.h>
struct HostScratchTL {
    double *host = nullptr, *dev = nullptr;
    hipEvent_t ev = nullptr;
    ~HostScratchTL() { if (ev) (void)hipEventDestroy(ev); if (host) (void)hipHostFree(host); }
};
static thread_local HostScratchTL g_tl_scratch;
static int scratch_init(void) {
    HostScratchTL &t = g_tl_scratch;
    if (t.host) return 0;
    void *p = nullptr, *d = nullptr;
    CIP_HIP_CHECK(hipHostMalloc(&p, 512 * sizeof(double), hipHostMallocMapped));
    CIP_HIP_CHECK(hipHostGetDevicePointer(&d, p, 0));
    CIP_HIP_CHECK(hipEventCreateWithFlags(&t.ev, hipEventDisableTiming));
    t.host = (double *)p; t.dev = (double *)d;
    return 0;
}
int cip_host_scratch(CipHostScratch *out) {
    const int rc = scratch_init();
    if (rc) return rc;
    out->host = g_tl_scratch.host; out->dev = g_tl_scratch.dev;
    return 0;
}
int cip_wait(hipStream_t s) {
    static const int spin = [] { const char *e = getenv("CIP_SPIN_WAIT"); return (e && atoi(e) == 0) ? 0 : 1; }();
    if (!spin) { CIP_HIP_CHECK(hipStreamSynchronize(s)); return 0; }
    const int rc = scratch_init();
    if (rc) return rc;
    CIP_HIP_CHECK(hipEventRecord(g_tl_scratch.ev, s));
    for (;;) {
        const hipError_t e = hipEventQuery(g_tl_scratch.ev);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) { CIP_HIP_CHECK(e); }
        __builtin_ia32_pause();
    }
}

// y = alpha x + beta y as ONE explicit operation: the product beta y rounded, then a fused multiply-add -- written out so that the
// fused kernels of the interior-point loop below (round 5) reproduce a chain of k_axpby launches bit for bit
__device__ __forceinline__ double axpby_op(double alpha, double x, double beta, double y) {
    return fma(alpha, x, beta == 0.0 ? 0.0 : beta * y);
}
__global__ __launch_bounds__(256) void k_axpby(int len, double alpha, const double *x, double beta, double *y, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, x, y);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long)gridDim.x * 256)
        y[i] = axpby_op(alpha, x[i], beta, y[i]);
}
int cip_axpby(hipStream_t s, int len, double alpha, const double *x, double beta, double *y) {
    if (len <= 0) return 0;
    int nb = (len + 255) / 256;
    if (nb > 2048) nb = 2048;
    cip_launch_b(k_axpby, dim3(nb), dim3(256), 0, s, len, alpha, x, beta, y);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_copy_neg(hipStream_t s, int len, const double *x, double *y, double scale) {
    return cip_axpby(s, len, scale, x, 0.0, y);
}

// y = alpha[z] x + beta y with one alpha per problem of a lock-step batch (step lengths, sigma mu)
__global__ __launch_bounds__(256) void k_axpby_ps(int len, CipScal64 alpha, const double *x, double beta, double *y, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, x, y);
    const double a = alpha.v[blockIdx.z];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long)gridDim.x * 256)
        y[i] = axpby_op(a, x[i], beta, y[i]);
}
int cip_axpby_ps(hipStream_t s, int len, const double *alpha_host, const double *x, double beta, double *y) {
    if (len <= 0) return 0;
    CipScal64 a;
    for (int z = 0; z < CIP_BATCH_MAX; ++z) a.v[z] = z < cip_tl_bz.B ? alpha_host[z] : 0.0;
    int nb = (len + 255) / 256;
    if (nb > 2048) nb = 2048;
    cip_launch_b(k_axpby_ps, dim3(nb), dim3(256), 0, s, len, a, x, beta, y);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// memset / device-to-device copy that follow the batch dimension (hipMemsetAsync / hipMemcpyAsync would touch problem 0 only)
__global__ __launch_bounds__(256) void k_fill(long len, double value, double *y, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO1(cb, y);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long)gridDim.x * 256) y[i] = value;
}
int cip_zero(hipStream_t s, long len, double *y) {
    if (len <= 0) return 0;
    if (!cip_in_batch() && !cip_tl_builder) { CIP_HIP_CHECK(hipMemsetAsync(y, 0, sizeof(double) * len, s)); return 0; }
    long nb = (len + 255) / 256;
    if (nb > 2048) nb = 2048;
    cip_launch_b(k_fill, dim3((unsigned)nb), dim3(256), 0, s, len, 0.0, y);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_copy(hipStream_t s, long len, const double *x, double *y) {
    if (len <= 0) return 0;
    if (!cip_in_batch() && !cip_tl_builder) { CIP_HIP_CHECK(hipMemcpyAsync(y, x, sizeof(double) * len, hipMemcpyDeviceToDevice, s)); return 0; }
    return cip_axpby(s, (int)len, 1.0, x, 0.0, y);
}

// ---------------------------------------------------------------------------------------------------------------------
// solve4x4 around the triangular sweeps when EVERY cone is an R cone (F = diag(f): configs 1, 2 and 5) on the Schur route:
// the eight element-wise launches in front of the sweeps (cone division, F', two axpby, F^-T, F^-1, the copy of x, A't for
// a CSR A) become ONE, the nine behind them (copies, A a, F^-T, F^-1, two axpby, F, F', axpby) ONE -- each small launch is
// ~3 us of dependency latency in a 0.27-ms call.  Every element goes through the same operations in the same order as in
// the separate kernels (cones.hip: k_cone_div / k_apply, k_axpby, k_spmv_csr; src/ConicIP.jl:684-692 and
// src/kktsolvers.jl:324-330): bit-identical results (tests/test_gpu_kernels.py::test_solve4x4_fused_r_path_bitwise).
//   pre:   q = r.s / lambda ; t1 = f q ; z = t1 + r.v ; t = (z / f) / f ;  rhs = [r.y + A't ; r.w ; 0]
//   post:  a = rhs ; u = A a ; c = t - (u / f) / f ; dv = c ; ds = t1 - f (f c)
#pragma clang fp contract(off)
// No contraction in these three functions (the pragma holds to the end of the file): the separate kernels round every product before the next kernel adds to it
// (explicit fma() where k_spmv_csr's accumulation is a fused multiply-add).
__device__ __forceinline__ double s4_t_of(const double *rs, const double *lam, const double *f, const double *rv, int j) {
    const double fj = f[j];
    const double t1 = (rs[j] / lam[j]) * fj;
    return ((t1 + rv[j]) / fj) / fj;
}
__global__ __launch_bounds__(256) void k_s4_pre_r(int m, int n, int p, int Npad, const double *f, const double *rs, const double *lam,
                                                   const double *rv, const double *ry, const double *rw, double *t1_out, double *t_out,
                                                   double *rhs, const int *T_rp, const int *T_ci, const double *T_v, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO8(cb, f, rs, lam, rv, ry, rw, t1_out, t_out);
    CIP_BO4(cb, rhs, T_rp, T_ci, T_v);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) {
        const double fi = f[i];
        const double t1 = (rs[i] / lam[i]) * fi;
        t1_out[i] = t1;
        t_out[i] = ((t1 + rv[i]) / fi) / fi;
    }
    if (i < Npad) {
        double r = 0.0;
        if (i < n) {
            r = ry[i];
            if (T_rp) {                                            // CSR of A': row i of A' . t, t recomputed per entry (same operations)
                double s = 0;
                for (int q = T_rp[i]; q < T_rp[i + 1]; ++q) s = fma(T_v[q], s4_t_of(rs, lam, f, rv, T_ci[q]), s);
                r = s + r;
            }
        } else if (i < n + p) r = rw[i - n];
        rhs[i] = r;
    }
}
__global__ __launch_bounds__(256) void k_s4_post_r(int m, int n, int p, const double *f, const double *t, const double *rhs, const double *u_dense,
                                                    const int *A_rp, const int *A_ci, const double *A_v, double *dy, double *dw, double *dv,
                                                    double *ds, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO8(cb, f, t, rhs, u_dense, A_rp, A_ci, A_v, dy);
    CIP_BO3(cb, dw, dv, ds);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dy[i] = rhs[i];
    if (i < p) dw[i] = rhs[n + i];
    if (i < m) {
        double u;
        if (A_rp) {
            double s = 0;
            for (int q = A_rp[i]; q < A_rp[i + 1]; ++q) s = fma(A_v[q], rhs[A_ci[q]], s);
            u = s + 0.0;                                     // k_spmv_csr's alpha s + (beta == 0 ? 0 : ..): -0 becomes +0 there too
        } else u = u_dense[i];
        const double fi = f[i];
        const double c = (t[i] + 0.0) - (u / fi) / fi;       // c = t (axpby, beta = 0), then c -= (F'F)^-1 A a
        dv[i] = c;
        ds[i] = ds[i] - (c * fi) * fi;                             // ds = t1 - F'(F dv)
    }
}
int cip_s4_pre_r(hipStream_t s, int m, int n, int p, int Npad, const double *f, const double *rs, const double *lam, const double *rv,
                 const double *ry, const double *rw, double *t1_out, double *t_out, double *rhs, const int *T_rp, const int *T_ci, const double *T_v) {
    const int len = m > Npad ? m : Npad;
    cip_launch_b(k_s4_pre_r, dim3((len + 255) / 256), dim3(256), 0, s, m, n, p, Npad, f, rs, lam, rv, ry, rw, t1_out, t_out, rhs, T_rp, T_ci, T_v);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_s4_post_r(hipStream_t s, int m, int n, int p, const double *f, const double *t, const double *rhs, const double *u_dense,
                  const int *A_rp, const int *A_ci, const double *A_v, double *dy, double *dw, double *dv, double *ds) {
    int len = m > n ? m : n;
    if (p > len) len = p;
    cip_launch_b(k_s4_post_r, dim3((len + 255) / 256), dim3(256), 0, s, m, n, p, f, t, rhs, u_dense, A_rp, A_ci, A_v, dy, dw, dv, ds);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Round 5: the element-wise chains of the interior-point loop as ONE kernel each (src/ConicIP.jl:746-753, :893-901, :912-917).
// An iteration was 117 launches of which ~45 were 3-6 us element-wise kernels whose work is a fraction of their launch slot
// (profiles/r4/c5_b8_iter_timeline.txt).  Each kernel below applies, per element, exactly the operations of the launches it
// replaces in their order (axpby_op for every axpby / copy -- a copy is axpby(1, x, 0, .) --, the R-cone formulas of cones.hip:
// product x * y, F x = x * f, F^-T x = x / f), so the per-operation loop of cipkkt/driver.py, which still issues the separate
// launches through the C ABI, and the two native loops that use these kernels walk the same bits (tests/test_gpu_driver.py).
// `f` != NULL: every cone is an R cone (f = the packed scaling = diag F) and the cone operations of the chain are fused in as well;
// f == NULL: the caller has run the cone kernels and passes their results.
// (No contraction here either: this file's `fp contract(off)` pragma above holds to its end; fma() is explicit.)
struct LoopDims { int n, m, p; };
// residual chain.  In: rl = (Q y + G'w - A'v, G y, A y, [lam o lam]) with rl.s filled only when f == NULL; z.s; c, d, b; lam.
// Out: rl.v = A y - s, [rl.s = lam o lam], Gy = copy(rl.w), Ays = copy(rl.v), r0 = copy(rl) - (c, d, b, 0)
__global__ __launch_bounds__(256) void k_loop_resid(LoopDims D, double *rl, const double *zs, const double *c, const double *d, const double *b,
                                                     const double *lam, const double *f, double *r0, double *Gy, double *Ays, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO8(cb, rl, zs, c, d, b, lam, r0, Gy);
    CIP_BO1(cb, Ays);
    if (f) CIP_BO1(cb, f);
    const long NT = (long)D.n + D.p + 2L * D.m;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < NT; i += (long)gridDim.x * 256) {
        if (i < D.n) {
            r0[i] = axpby_op(-1.0, c[i], 1.0, axpby_op(1.0, rl[i], 0.0, 0.0));
        } else if (i < D.n + D.p) {
            const long j = i - D.n;
            const double w = rl[i];
            Gy[j] = axpby_op(1.0, w, 0.0, 0.0);
            r0[i] = axpby_op(-1.0, d[j], 1.0, axpby_op(1.0, w, 0.0, 0.0));
        } else if (i < D.n + D.p + D.m) {
            const long j = i - D.n - D.p;
            const double v = axpby_op(-1.0, zs[j], 1.0, rl[i]);              // A y - s
            rl[i] = v;
            Ays[j] = axpby_op(1.0, v, 0.0, 0.0);
            r0[i] = axpby_op(-1.0, b[j], 1.0, axpby_op(1.0, v, 0.0, 0.0));
        } else {
            const long j = i - D.n - D.p - D.m;
            double sv;
            if (f) { sv = lam[j] * lam[j]; rl[i] = sv; } else sv = rl[i];
            r0[i] = axpby_op(1.0, sv, 0.0, 0.0);
        }
    }
}
// corrector right-hand side: r = copy(r0); r.s += mb3; r.s -= sigma mu e, with mb3 = (F^-T daff.s) o (F daff.v) given
// (f == NULL) or formed here (all R cones)
__global__ __launch_bounds__(256) void k_loop_corr(LoopDims D, const double *r0, const double *daff, const double *mb3, const double *e,
                                                    const double *f, CipScal64 sigmu, double *r, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO5(cb, r0, daff, mb3, e, r);
    if (f) CIP_BO1(cb, f);
    const long NT = (long)D.n + D.p + 2L * D.m, so = (long)D.n + D.p + D.m, vo = (long)D.n + D.p;
    const double sm = sigmu.v[blockIdx.z];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < NT; i += (long)gridDim.x * 256) {
        double x = axpby_op(1.0, r0[i], 0.0, 0.0);
        if (i >= so) {
            const long j = i - so;
            double t3;
            if (f) { const double fj = f[j]; const double t1 = daff[so + j] / fj, t2 = daff[vo + j] * fj; t3 = t1 * t2; }
            else t3 = mb3[j];
            x = axpby_op(1.0, t3, 1.0, x);
            x = axpby_op(-sm, e[j], 1.0, x);
        }
        r[i] = x;
    }
}
// refinement residual: rk = (Q dy + G'dw - A'dv, G dy, A dy, .) in; rk.v -= dz.s; rk.s = lam o (F dz.v) + lam o (F^-T dz.s) (given as
// mb2, mb3 when f == NULL); rIr = copy(r) - rk
__global__ __launch_bounds__(256) void k_loop_refine(LoopDims D, double *rk, const double *dz, const double *r, const double *lam,
                                                      const double *mb2, const double *mb3, const double *f, double *rIr, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO7(cb, rk, dz, r, lam, mb2, mb3, rIr);
    if (f) CIP_BO1(cb, f);
    const long NT = (long)D.n + D.p + 2L * D.m, so = (long)D.n + D.p + D.m, vo = (long)D.n + D.p;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < NT; i += (long)gridDim.x * 256) {
        double k = rk[i];
        if (i >= so) {
            const long j = i - so;
            double t2, t3;
            if (f) { const double fj = f[j], lj = lam[j]; t2 = lj * (dz[vo + j] * fj); t3 = lj * (dz[so + j] / fj); }
            else { t2 = mb2[j]; t3 = mb3[j]; }
            k = axpby_op(1.0, t3, 1.0, axpby_op(1.0, t2, 0.0, 0.0));
            rk[i] = k;
        } else if (i >= vo) {
            k = axpby_op(-1.0, dz[so + (i - vo)], 1.0, k);
            rk[i] = k;
        }
        rIr[i] = axpby_op(-1.0, k, 1.0, axpby_op(1.0, r[i], 0.0, 0.0));
    }
}
static int loop_grid(long NT) { long nb = (NT + 255) / 256; return (int)(nb > 2048 ? 2048 : (nb < 1 ? 1 : nb)); }
int cip_loop_resid(hipStream_t s, int n, int m, int p, double *rl, const double *zs, const double *c, const double *d, const double *b,
                   const double *lam, const double *f, double *r0, double *Gy, double *Ays) {
    cip_launch_b(k_loop_resid, dim3(loop_grid((long)n + p + 2L * m)), dim3(256), 0, s, LoopDims{n, m, p}, rl, zs, c, d, b, lam, f, r0, Gy, Ays);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
// sigmu_host: one value, or B values in a lock-step batch
int cip_loop_corr(hipStream_t s, int n, int m, int p, const double *r0, const double *daff, const double *mb3, const double *e,
                  const double *f, const double *sigmu_host, double *r) {
    CipScal64 a;
    const int B = cip_tl_bz.B > 1 ? cip_tl_bz.B : 1;
    for (int z = 0; z < CIP_BATCH_MAX; ++z) a.v[z] = z < B ? sigmu_host[z] : 0.0;
    cip_launch_b(k_loop_corr, dim3(loop_grid((long)n + p + 2L * m)), dim3(256), 0, s, LoopDims{n, m, p}, r0, daff, mb3, e, f, a, r);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_loop_refine(hipStream_t s, int n, int m, int p, double *rk, const double *dz, const double *r, const double *lam,
                    const double *mb2, const double *mb3, const double *f, double *rIr) {
    cip_launch_b(k_loop_refine, dim3(loop_grid((long)n + p + 2L * m)), dim3(256), 0, s, LoopDims{n, m, p}, rk, dz, r, lam, mb2, mb3, f, rIr);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
