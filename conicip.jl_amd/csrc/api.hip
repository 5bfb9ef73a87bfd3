// C ABI of libcipkkt (see include/cipkkt.h for the contract and the reference lines
// each entry point replaces).
#include "cip_handle.h"
#include <thread>
#include <stdlib.h>
#include "../../include/cipkkt.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <new>

static thread_local char g_err[512] = "";
void cip_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *cip_last_error(void) { return g_err; }

// ------------------------------------------------------------------ profiler ranges (roctx)
// "cip:scaling", "cip:assemble", "cip:ldlt", "cip:solve3x3", "cip:iteration" ranges for rocprofv3 --marker-trace.  The
// marker library is looked up at run time (no link-time dependency: the library links libamdhip64 only); absent or
// CIP_ROCTX=0 -> no-ops.
#include <dlfcn.h>
#include <mutex>
static int (*g_roctx_push)(const char *) = nullptr;
static int (*g_roctx_pop)(void) = nullptr;
static std::once_flag g_roctx_once;
static void roctx_init(void) {
    const char *e = getenv("CIP_ROCTX");
    if (e && atoi(e) == 0) return;
    for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
        if (void *lib = dlopen(name, RTLD_LAZY | RTLD_LOCAL)) {
            g_roctx_push = (int (*)(const char *))dlsym(lib, "roctxRangePushA");
            g_roctx_pop = (int (*)(void))dlsym(lib, "roctxRangePop");
            if (g_roctx_push && g_roctx_pop) return;
            g_roctx_push = nullptr; g_roctx_pop = nullptr;
        }
    }
}
void cip_range_push(const char *name) {
    std::call_once(g_roctx_once, roctx_init);
    if (g_roctx_push) (void)g_roctx_push(name);
}
void cip_range_pop(void) { if (g_roctx_pop) (void)g_roctx_pop(); }
struct CipRange { explicit CipRange(const char *n) { cip_range_push(n); } ~CipRange() { cip_range_pop(); } };

static int rup(int x, int q) { return ((x + q - 1) / q) * q; }

// TEST SWITCH (CIP_DEBUG_POISON=<byte 1..255>): every buffer of a handle starts out filled with that byte (255: NaNs and -1s; 63: doubles of
// 4.8e-4 and huge ints) instead of whatever the allocator hands out -- fresh memory of a fresh process is zero, recycled memory of a long
// one is not, and nothing may depend on either.  tests/test_gpu_poison.py runs trajectories under it and compares the bits.
static int debug_poison(void *p, size_t bytes) {
    static const int v = [] { const char *e = getenv("CIP_DEBUG_POISON"); return e ? atoi(e) & 255 : 0; }();
    if (!v) return 0;
    CIP_HIP_CHECK(hipDeviceSynchronize());
    CIP_HIP_CHECK(hipMemset(p, v, bytes));
    CIP_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}
int cip_handle_alloc(cip_handle *h, void **out, size_t bytes) {
    if (bytes == 0) bytes = 256;
    const size_t b = (bytes + 255) & ~(size_t)255;
    h->alloc_bytes += b;
    if (h->arena) {
        if (h->arena_used + b <= h->arena_cap) { *out = h->arena + h->arena_used; h->arena_used += b; return debug_poison(*out, b); }
        h->arena_overflow = true;                  // falls back to its own allocation: the slab layout is broken
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        // the lock-step arena of an earlier batch (several GB, kept for the next one) may be what is in the way: give it back, once
        (void)hipGetLastError();
        (void)cip_release_cached_memory();
        e = hipMalloc(out, bytes);
    }
    CIP_HIP_CHECK(e);
    return debug_poison(*out, bytes);
}
#define DMALLOC(ptr, bytes)                                                        \
    do {                                                                           \
        int rc__ = cip_handle_alloc(h, (void **)&(ptr), (size_t)(bytes));          \
        if (rc__) return rc__;                                                     \
    } while (0)
static bool in_arena(const cip_handle *h, const void *p) {
    return h->arena && (const char *)p >= h->arena && (const char *)p < h->arena + h->arena_cap;
}

static void free_all(cip_handle *h) {
    if (h->gx_factor) { (void)hipGraphExecDestroy(h->gx_factor); h->gx_factor = nullptr; }
    if (h->ldlt_side) { cip_ldlt_side_destroy(h->ldlt_side); h->ldlt_side = nullptr; }
    if (h->gx_solve) { (void)hipGraphExecDestroy(h->gx_solve); h->gx_solve = nullptr; }
    if (h->cs.lg) { cip_sdp_large_destroy(h->cs.lg); h->cs.lg = nullptr; }
    void *ptrs[] = {h->cs.d_bigq, h->cs.d_ritems, h->cs.d_packq, h->cs.d_sidx_small, h->cs.d_sidx, h->cs.d_sdpws, h->cs.d_sdpvec, h->cs.d_sdpflag, h->Q, h->symv_ws, h->A, h->At, h->A_rp, h->A_ci, h->A_v, h->T_rp, h->T_ci, h->T_v, h->kdiag, h->row_cone, h->G, h->Gt,
                    h->cs.d_cones, h->cs.d_items, h->cs.d_scal, h->cs.d_partial, h->cs.d_scalar, h->K, h->Wt, h->syrk_ws, h->Gm, h->AtS, h->WtS,
                    h->ws_base, h->rhs, h->mt1, h->mt2, h->mt3, h->nt1, h->pt1, h->dot_scratch, h->dot_ptrs, h->stage, h->drv, h->ref, h->c2x2};
    for (void *p : ptrs)
        if (p && !in_arena(h, p)) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->info_host) (void)hipHostFree(h->info_host);
    if (h->ws.prof) { cip_ldlt_profile_destroy(h->ws.prof); h->ws.prof = nullptr; }
}

// copy a (rows x cols, ld) column-major matrix from src (host or device) into a tight device buffer
static int upload_matrix(double *dst, long ld_dst, const double *src, long ld_src, int rows, int cols, bool src_dev,
                         hipStream_t s) {
    if (rows == 0 || cols == 0) return 0;
    CIP_HIP_CHECK(hipMemcpy2DAsync(dst, ld_dst * sizeof(double), src, ld_src * sizeof(double), rows * sizeof(double), cols,
                                   src_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    return 0;
}

// dst (cols x rows, ld ld_dst) = src' ; tiled through LDS
__global__ __launch_bounds__(256) void k_transpose(const double *src, long ld_src, int rows, int cols, double *dst, long ld_dst) {
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int q = 0; q < 32; q += 8) {
        const int r = r0 + tx, c = c0 + ty + q;
        if (r < rows && c < cols) t[ty + q][tx] = src[r + (long)c * ld_src];
    }
    __syncthreads();
    for (int q = 0; q < 32; q += 8) {
        const int c = c0 + tx, r = r0 + ty + q;
        if (r < rows && c < cols) dst[c + (long)r * ld_dst] = t[tx][ty + q];
    }
}
static int transpose_dev(hipStream_t s, const double *src, long ld_src, int rows, int cols, double *dst, long ld_dst) {
    if (rows == 0 || cols == 0) return 0;
    hipLaunchKernelGGL(k_transpose, dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, s, src, ld_src, rows, cols, dst, ld_dst);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Contents of Q, G (+G'), A (+A', or the CSR of A and of A') into the handle's buffers, on its stream.
static int upload_problem(cip_handle *h, const cip_problem *pr) {
    const int n = h->n, m = h->m, p = h->p;
    const bool dev = (pr->flags & CIP_FLAG_DEVICE_PTRS) != 0;
    hipStream_t s = h->stream;
    int rc;
    cip_sdp_large_invalidate(h->cs.lg);                     // (the mat(a_i) images of the large S cones follow A)
    if ((rc = upload_matrix(h->Q, n, pr->Q, pr->ldq > 0 ? pr->ldq : n, n, n, dev, s))) return rc;
    if (p > 0) {
        if ((rc = upload_matrix(h->G, p, pr->G, pr->ldg > 0 ? pr->ldg : p, p, n, dev, s))) return rc;
        if ((rc = transpose_dev(s, h->G, p, p, n, h->Gt, n))) return rc;
    }
    if (!h->A_sparse) {
        if (m > 0) {
            if ((rc = upload_matrix(h->A, m, pr->A, pr->lda > 0 ? pr->lda : m, m, n, dev, s))) return rc;
            if ((rc = transpose_dev(s, h->A, m, m, n, h->At, h->npad))) return rc;
        }
        return 0;
    }
    // CSR of A (given) and of A' (built here on the host).  The host arrays live in the handle: the uploads below are
    // asynchronous and nothing here waits for them (a previous upload from the same vectors is waited for first).
    if (h->staging_live) CIP_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<int> &rp = h->st_rp, &ci = h->st_ci, &trp = h->st_trp, &tci = h->st_tci;
    std::vector<double> &av = h->st_av, &tv = h->st_tv;
    rp.assign(m + 1, 0);
    const bool csr_dev = dev && !(pr->flags & CIP_FLAG_CSR_HOST);
    if (csr_dev) CIP_HIP_CHECK(hipMemcpy(rp.data(), pr->A_rowptr, sizeof(int) * (m + 1), hipMemcpyDeviceToHost));
    else memcpy(rp.data(), pr->A_rowptr, sizeof(int) * (m + 1));
    const int nnz = rp[m];
    if (rp[0] != 0 || nnz != h->A_nnz) { cip_set_error("bad CSR row pointer (nnz %d, handle holds %d)", nnz, h->A_nnz); return CIP_E_INVALID; }
    ci.assign(nnz > 0 ? nnz : 1, 0);
    av.assign(nnz > 0 ? nnz : 1, 0.0);
    if (nnz > 0) {
        if (csr_dev) {
            CIP_HIP_CHECK(hipMemcpy(ci.data(), pr->A_colind, sizeof(int) * nnz, hipMemcpyDeviceToHost));
            CIP_HIP_CHECK(hipMemcpy(av.data(), pr->A_val, sizeof(double) * nnz, hipMemcpyDeviceToHost));
        } else {
            memcpy(ci.data(), pr->A_colind, sizeof(int) * nnz);
            memcpy(av.data(), pr->A_val, sizeof(double) * nnz);
        }
    }
    for (int q = 0; q < nnz; ++q)
        if (ci[q] < 0 || ci[q] >= n) { cip_set_error("CSR column index out of range"); return CIP_E_INVALID; }
    h->A_one_per_row = true;
    for (int r = 0; r < m; ++r) if (rp[r + 1] - rp[r] > 1) { h->A_one_per_row = false; break; }
    trp.assign(n + 1, 0); tci.assign(nnz > 0 ? nnz : 1, 0);
    tv.assign(nnz > 0 ? nnz : 1, 0.0);
    for (int q = 0; q < nnz; ++q) trp[ci[q] + 1]++;
    for (int i = 0; i < n; ++i) trp[i + 1] += trp[i];
    {
        std::vector<int> fill(trp.begin(), trp.end() - 1);
        for (int r = 0; r < m; ++r)
            for (int q = rp[r]; q < rp[r + 1]; ++q) { const int d = fill[ci[q]]++; tci[d] = r; tv[d] = av[q]; }
    }
    CIP_HIP_CHECK(hipMemcpyAsync(h->A_rp, rp.data(), sizeof(int) * (m + 1), hipMemcpyHostToDevice, s));
    CIP_HIP_CHECK(hipMemcpyAsync(h->T_rp, trp.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice, s));
    if (nnz > 0) {
        CIP_HIP_CHECK(hipMemcpyAsync(h->A_ci, ci.data(), sizeof(int) * nnz, hipMemcpyHostToDevice, s));
        CIP_HIP_CHECK(hipMemcpyAsync(h->A_v, av.data(), sizeof(double) * nnz, hipMemcpyHostToDevice, s));
        CIP_HIP_CHECK(hipMemcpyAsync(h->T_ci, tci.data(), sizeof(int) * nnz, hipMemcpyHostToDevice, s));
        CIP_HIP_CHECK(hipMemcpyAsync(h->T_v, tv.data(), sizeof(double) * nnz, hipMemcpyHostToDevice, s));
    }
    h->staging_live = true;
    if (h->AtS) return cip_scatter_AtS(h);                  // assemble.hip: the S cones' rows of A' as a dense block
    return 0;
}

static int create_impl(const cip_problem *pr, cip_handle *h, bool final_sync = true) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        cip_set_error("no HIP device available (libcipkkt has no CPU fallback)");
        return CIP_E_NODEVICE;
    }
    CIP_HIP_CHECK(hipGetDevice(&h->device));
    if (cip_kernels_init()) return CIP_E_HIP;
    const int n = pr->n, m = pr->m, p = pr->p;
    if (n <= 0 || m < 0 || p < 0 || pr->ncones < 0) { cip_set_error("bad dimensions n=%d m=%d p=%d", n, m, p); return CIP_E_INVALID; }
    if (!pr->Q) { cip_set_error("Q is NULL"); return CIP_E_INVALID; }
    if (m > 0 && !pr->A && !(pr->A_rowptr && pr->A_colind && pr->A_val)) { cip_set_error("A is NULL"); return CIP_E_INVALID; }
    if (p > 0 && !pr->G) { cip_set_error("G is NULL"); return CIP_E_INVALID; }
    if (pr->route != CIP_ROUTE_SCHUR && pr->route != CIP_ROUTE_FULL3X3) { cip_set_error("bad route"); return CIP_E_INVALID; }
    h->n = n; h->m = m; h->p = p; h->ncones = pr->ncones; h->route = pr->route;
    const bool dev = (pr->flags & CIP_FLAG_DEVICE_PTRS) != 0;
    hipStream_t s = h->stream;   // default (null) stream until cip_set_stream

    // ---- cones
    int off = 0, nq = 0;
    size_t soff = 0;
    bool has_S = false;
    std::vector<int> sidx;
    int rmax = 0, kmax = 0;
    for (int c = 0; c < pr->ncones; ++c) {
        ConeDesc cd = {};
        cd.type = pr->cone_type[c]; cd.dim = pr->cone_dim[c]; cd.off = off; cd.soff = (int)soff; cd.r = 0; cd.qidx = -1; cd.aoff = off;
        if (cd.dim <= 0) { cip_set_error("cone %d has dimension %d", c, cd.dim); return CIP_E_INVALID; }
        if (cd.type == CIP_CONE_R) { soff += cd.dim; }
        else if (cd.type == CIP_CONE_Q) { soff += 1 + cd.dim; cd.qidx = nq++; }
        else if (cd.type == CIP_CONE_S) {
            const int r = (int)llround((sqrt(1.0 + 8.0 * cd.dim) - 1.0) / 2.0);     // ord() src/ConicIP.jl:85
            if (r * (r + 1) / 2 != cd.dim) { cip_set_error("S cone %d: %d is not a triangular number", c, cd.dim); return CIP_E_INVALID; }
            if (r > 2048) { cip_set_error("S cone %d: matrix order %d > 2048 is not supported", c, r); return CIP_E_UNSUPPORTED; }
            cd.r = r; soff += 2 * (size_t)r * r; has_S = true;
            sidx.push_back(c);
            if (r > rmax) rmax = r;
            if (cd.dim > kmax) kmax = cd.dim;
        } else { cip_set_error("cone %d: unknown type %d", c, cd.type); return CIP_E_INVALID; }
        if (soff > 0x7fffffffULL) { cip_set_error("scaling storage too large"); return CIP_E_INVALID; }
        h->h_cones.push_back(cd);
        off += cd.dim;
    }
    // work items (one workgroup each) and partial-result slots.  Consecutive Q cones of dimension <= 64 are packed:
    // lane segments of width W = next power of two of the run's largest cone, 256 / W cones per workgroup.
    int nslots = 0;
    for (int c = 0; c < pr->ncones;) {
        ConeDesc &cd = h->h_cones[c];
        if (cd.type == CIP_CONE_R) {
            cd.item = nslots;
            for (int st = 0; st < cd.dim; st += 2048) h->h_items.push_back(WorkItem{c, st, (cd.dim - st < 2048) ? cd.dim - st : 2048, nslots++, 0});
            ++c;
        } else if (cd.type == CIP_CONE_Q && cd.dim <= 64) {
            int e = c, dmax = 0;
            while (e < pr->ncones && h->h_cones[e].type == CIP_CONE_Q && h->h_cones[e].dim <= 64) { if (h->h_cones[e].dim > dmax) dmax = h->h_cones[e].dim; ++e; }
            int W = 1;
            while (W < dmax) W *= 2;
            const int per = 256 / W;
            for (int q = c; q < e; q += per) {
                const int cnt = (e - q < per) ? (e - q) : per;
                for (int u = 0; u < cnt; ++u) h->h_cones[q + u].item = nslots + u;
                h->h_items.push_back(WorkItem{q, 0, cnt, nslots, W});
                nslots += cnt;
            }
            c = e;
        } else {
            cd.item = nslots;
            h->h_items.push_back(WorkItem{c, 0, cd.dim, nslots++, 0});      // a large Q cone or an S cone: one workgroup
            ++c;
        }
    }
    if (off != m) { cip_set_error("cone_dims cover %d rows but A has %d", off, m); return CIP_E_INVALID; }
    if (has_S && pr->A == NULL && m > 0) {
        // CSR A with S cones (round 4; the reference builds its Schur system from a sparse A whatever the cones,
        // src/kktsolvers.jl:289-293): the S cones' rows are expanded, once, into a dense transposed block of their own
        // (upload_problem) for the congruences of the Schur scaling; everything else stays CSR
        int ao = 0;
        for (ConeDesc &cd : h->h_cones) if (cd.type == CIP_CONE_S) { cd.aoff = ao; ao += cd.dim; }
        h->mS = ao; h->mSpad = rup(ao, CIP_KT);
    }
    h->nq = nq; h->nqpad = rup(nq > 0 ? nq : 1, CIP_KT);
    h->cs.nbigq = 0; h->cs.d_bigq = nullptr;
    h->st_bigq.clear();
    for (size_t it = 0; it < h->h_items.size(); ++it) {
        const ConeDesc &cd = h->h_cones[h->h_items[it].cone];
        if (cd.type == CIP_CONE_Q && h->h_items[it].width == 0) h->st_bigq.push_back((int)it);
    }
    h->cs.nbigq = (int)h->st_bigq.size();
    h->st_ritems.clear();
    for (size_t it = 0; it < h->h_items.size(); ++it)
        if (h->h_cones[h->h_items[it].cone].type == CIP_CONE_R) h->st_ritems.push_back((int)it);
    h->cs.nritems = (int)h->st_ritems.size(); h->cs.d_ritems = nullptr;
    h->st_packq.clear();
    for (size_t it = 0; it < h->h_items.size(); ++it)
        if (h->h_cones[h->h_items[it].cone].type == CIP_CONE_Q && h->h_items[it].width != 0) h->st_packq.push_back((int)it);
    h->cs.npackq = (int)h->st_packq.size(); h->cs.d_packq = nullptr;
    if (h->cs.npackq > 0) {
        DMALLOC(h->cs.d_packq, sizeof(int) * h->st_packq.size());
        CIP_HIP_CHECK(hipMemcpyAsync(h->cs.d_packq, h->st_packq.data(), sizeof(int) * h->st_packq.size(), hipMemcpyHostToDevice, s));
    }
    if (h->cs.nritems > 0) {
        DMALLOC(h->cs.d_ritems, sizeof(int) * h->st_ritems.size());
        CIP_HIP_CHECK(hipMemcpyAsync(h->cs.d_ritems, h->st_ritems.data(), sizeof(int) * h->st_ritems.size(), hipMemcpyHostToDevice, s));
    }
    if (h->cs.nbigq > 0) {
        DMALLOC(h->cs.d_bigq, sizeof(int) * h->st_bigq.size());
        CIP_HIP_CHECK(hipMemcpyAsync(h->cs.d_bigq, h->st_bigq.data(), sizeof(int) * h->st_bigq.size(), hipMemcpyHostToDevice, s));
    }
    h->cs.ncones = pr->ncones; h->cs.nitems = (int)h->h_items.size(); h->cs.nslots = nslots; h->cs.m = m; h->cs.scal_len = soff; h->cs.has_S = has_S;
    DMALLOC(h->cs.d_cones, sizeof(ConeDesc) * h->h_cones.size());
    DMALLOC(h->cs.d_items, sizeof(WorkItem) * h->h_items.size());
    DMALLOC(h->cs.d_scal, sizeof(double) * soff);
    DMALLOC(h->cs.d_partial, sizeof(double) * 2 * (nslots + 1));      // two sets: the pair of max-steps (cip_cones_maxstep2)
    DMALLOC(h->cs.d_scalar, sizeof(double) * 8);
    h->cs.ns = (int)sidx.size(); h->cs.rmax = rmax; h->cs.kmax = kmax;
    h->cs.ns_small = 0; h->cs.nlarge = 0; h->cs.lg = nullptr; h->cs.d_sidx_small = nullptr;
    h->cs.h_cones = h->h_cones.data();
    if (has_S) {
        std::vector<int> &small = h->st_small;
        small.clear();
        int rmax_large = 0;
        for (int c : sidx) {
            if (h->h_cones[c].r >= CIP_LARGE_S_MIN) {
                if (h->cs.nlarge == CIP_MAX_LARGE_S) { cip_set_error("more than %d S cones of order >= %d", CIP_MAX_LARGE_S, CIP_LARGE_S_MIN); return CIP_E_UNSUPPORTED; }
                h->cs.large_cone[h->cs.nlarge++] = c;
                if (h->h_cones[c].r > rmax_large) rmax_large = h->h_cones[c].r;
            } else small.push_back(c);
        }
        h->cs.ns_small = (int)small.size();
        DMALLOC(h->cs.d_sidx_small, sizeof(int) * (small.size() + 1));
        if (!small.empty()) CIP_HIP_CHECK(hipMemcpyAsync(h->cs.d_sidx_small, small.data(), sizeof(int) * small.size(), hipMemcpyHostToDevice, s));
        if (h->cs.nlarge > 0) { int rcl = cip_sdp_large_create(rmax_large, h->cs.nlarge, n, &h->cs.lg); if (rcl) return rcl; }
    }
    if (has_S) {
        const int per = n < 64 ? n : 64;
        h->cs.sdp_slots = h->cs.ns * (per > 0 ? per : 1);
        DMALLOC(h->cs.d_sidx, sizeof(int) * sidx.size());
        h->st_sidx = sidx;
        CIP_HIP_CHECK(hipMemcpyAsync(h->cs.d_sidx, h->st_sidx.data(), sizeof(int) * sidx.size(), hipMemcpyHostToDevice, s));
        DMALLOC(h->cs.d_sdpws, sizeof(double) * (size_t)h->cs.sdp_slots * 6 * rmax * rmax);
        DMALLOC(h->cs.d_sdpvec, sizeof(double) * (size_t)h->cs.sdp_slots * 2 * kmax);
        const size_t nflag = (size_t)rup(4 + (h->cs.nlarge > 12 ? h->cs.nlarge : 12), 16);   // [0]: an iterate left the cone; [4 + li]: gate of large cone li's general division (li < nlarge <= CIP_MAX_LARGE_S)
        DMALLOC(h->cs.d_sdpflag, sizeof(int) * nflag);
        CIP_HIP_CHECK(hipMemsetAsync(h->cs.d_sdpflag, 0, sizeof(int) * nflag, s));
    }
    if (!h->h_cones.empty())
        CIP_HIP_CHECK(hipMemcpyAsync(h->cs.d_cones, h->h_cones.data(), sizeof(ConeDesc) * h->h_cones.size(), hipMemcpyHostToDevice, s));
    if (!h->h_items.empty())
        CIP_HIP_CHECK(hipMemcpyAsync(h->cs.d_items, h->h_items.data(), sizeof(WorkItem) * h->h_items.size(), hipMemcpyHostToDevice, s));

    // ---- sizes
    h->npad = rup(n, CIP_NB);
    h->mpad = rup(m > 0 ? m : 1, CIP_KT);
    h->N = (h->route == CIP_ROUTE_SCHUR) ? n + p : n + p + m;
    h->Npad = rup(h->N, CIP_NB);
    h->ldk = h->Npad;

    // ---- Q, G, A: buffers here, contents by upload_problem (also used by cip_update_problem)
    DMALLOC(h->Q, sizeof(double) * (size_t)n * n);
    if (n >= 2048 && n % 128 == 0) DMALLOC(h->symv_ws, sizeof(double) * 2 * (size_t)(n / 128) * n);
    DMALLOC(h->G, sizeof(double) * (size_t)p * n);
    DMALLOC(h->Gt, sizeof(double) * (size_t)p * n);
    h->A_sparse = (pr->A == NULL && m > 0);
    int rc;
    if (!h->A_sparse) {
        DMALLOC(h->A, sizeof(double) * (size_t)m * n);
        DMALLOC(h->At, sizeof(double) * (size_t)h->npad * h->mpad);
        CIP_HIP_CHECK(hipMemsetAsync(h->At, 0, sizeof(double) * (size_t)h->npad * h->mpad, s));
        if (h->route == CIP_ROUTE_SCHUR) {
            DMALLOC(h->Wt, sizeof(double) * (size_t)h->npad * h->mpad);
            CIP_HIP_CHECK(hipMemsetAsync(h->Wt, 0, sizeof(double) * (size_t)h->npad * h->mpad, s));
            h->syrk_n = cip_syrk_split(h->npad, h->mpad, &h->syrk_len);
            if (h->syrk_n > 1) DMALLOC(h->syrk_ws, sizeof(double) * (size_t)h->syrk_n * h->npad * h->npad);
        }
    } else {
        int nnz = 0;
        if (dev && !(pr->flags & CIP_FLAG_CSR_HOST)) CIP_HIP_CHECK(hipMemcpy(&nnz, pr->A_rowptr + m, sizeof(int), hipMemcpyDeviceToHost));
        else nnz = pr->A_rowptr[m];
        if (nnz < 0) { cip_set_error("bad CSR row pointer"); return CIP_E_INVALID; }
        h->A_nnz = nnz;
        DMALLOC(h->A_rp, sizeof(int) * (m + 1)); DMALLOC(h->A_ci, sizeof(int) * nnz); DMALLOC(h->A_v, sizeof(double) * nnz);
        DMALLOC(h->T_rp, sizeof(int) * (n + 1)); DMALLOC(h->T_ci, sizeof(int) * nnz); DMALLOC(h->T_v, sizeof(double) * nnz);
        DMALLOC(h->kdiag, sizeof(double) * (n > 0 ? n : 1));
        std::vector<int> &rc_ = h->st_rowcone;
        rc_.assign(m > 0 ? m : 1, 0);
        for (size_t c = 0; c < h->h_cones.size(); ++c)
            for (int e = 0; e < h->h_cones[c].dim; ++e) rc_[h->h_cones[c].off + e] = (int)c;
        DMALLOC(h->row_cone, sizeof(int) * m);
        if (m > 0) CIP_HIP_CHECK(hipMemcpyAsync(h->row_cone, rc_.data(), sizeof(int) * m, hipMemcpyHostToDevice, s));
        if (h->mS > 0 && h->route == CIP_ROUTE_SCHUR) {
            DMALLOC(h->AtS, sizeof(double) * (size_t)h->npad * h->mSpad);
            DMALLOC(h->WtS, sizeof(double) * (size_t)h->npad * h->mSpad);
            CIP_HIP_CHECK(hipMemsetAsync(h->AtS, 0, sizeof(double) * (size_t)h->npad * h->mSpad, s));
            CIP_HIP_CHECK(hipMemsetAsync(h->WtS, 0, sizeof(double) * (size_t)h->npad * h->mSpad, s));
        }
        if (h->route == CIP_ROUTE_SCHUR) DMALLOC(h->Gm, sizeof(double) * (size_t)h->npad * h->nqpad);
    }
    if ((rc = upload_problem(h, pr))) return rc;

    // ---- KKT matrix, workspace, scratch
    DMALLOC(h->K, sizeof(double) * (size_t)h->ldk * h->Npad);
    {
        const int fz = cip_ldlt_fused_for(h->Npad);             // read ONCE: sizing and carve must agree
        DMALLOC(h->ws_base, cip_ldlt_ws_bytes(h->Npad, fz));
        cip_ldlt_ws_carve(h->ws_base, h->Npad, &h->ws, fz);
    }
    h->ws.x_zeroed = &h->x_zeroed;
    h->ws.side = h->arena ? nullptr : &h->ldlt_side;      // (handles of a lock-step arena factor inside batched launches)
    // pivot signs of the quasi-definite order: Schur route [S G'; G 0] = n positive, p negative; literal 3x3 in the
    // order (3,1,2) = m negative (-F'F), n positive, p negative
    h->ws.signs = (h->route == CIP_ROUTE_SCHUR) ? PivotSigns{0, n, n + p} : PivotSigns{m, m + n, m + n + p};
    DMALLOC(h->rhs, sizeof(double) * h->Npad);
    DMALLOC(h->mt1, sizeof(double) * m); DMALLOC(h->mt2, sizeof(double) * m); DMALLOC(h->mt3, sizeof(double) * m);
    DMALLOC(h->nt1, sizeof(double) * n); DMALLOC(h->pt1, sizeof(double) * p);
    DMALLOC(h->dot_scratch, sizeof(double) * (32 * 32 + 64));
    CIP_HIP_CHECK(hipMemsetAsync(h->dot_scratch + 32 * 32 + 32, 0, sizeof(double) * 32, s));      // the completion counters of vecops.hip: k_dots
    DMALLOC(h->dot_ptrs, 32 * 32);
    DMALLOC(h->stage, sizeof(double) * 2 * (size_t)(n + p + m));
    CIP_HIP_CHECK(hipEventCreate(&h->ev0)); CIP_HIP_CHECK(hipEventCreate(&h->ev1)); CIP_HIP_CHECK(hipEventCreate(&h->ev2));
    // (the pinned pivot-flag words are allocated by the first factorisation: pinned allocations cost ~0.1 ms each, and the
    //  64 handles of a lock-step group never use theirs)
    if ((rc = cip_cones_identity_scaling(s, h->cs))) return rc;
    if ((rc = cip_sdp_scaling_changed(s, h->cs))) return rc;
    // the caller's Q / A / G may go away as soon as a public create returns: wait for the uploads.  A lock-step group
    // keeps its problems alive for the whole call and creates its 64 handles without a single host wait.
    if (final_sync) CIP_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
}

extern "C" int cip_create_ex(const cip_problem *prob, cip_handle **out) {
    if (!prob || !out) { cip_set_error("NULL argument"); return CIP_E_INVALID; }
    *out = NULL;
    cip_handle *h = new (std::nothrow) cip_handle();
    if (!h) { cip_set_error("out of host memory"); return CIP_E_INVALID; }
    const int rc = create_impl(prob, h);
    if (rc) { free_all(h); delete h; return rc; }
    *out = h;
    return 0;
}

// A handle whose creation-time buffers are carved out of `slab` (lock-step batches, driver.hip); the slab belongs to the
// caller and outlives the handle.
int cip_create_in_arena(const cip_problem *pr, char *slab, size_t cap, hipStream_t stream, cip_handle **out) {
    *out = nullptr;
    cip_handle *h = new (std::nothrow) cip_handle();
    if (!h) { cip_set_error("out of host memory"); return CIP_E_INVALID; }
    h->arena = slab; h->arena_cap = cap; h->arena_used = 0;
    h->stream = stream;
    const int rc = create_impl(pr, h, false);
    if (rc) { free_all(h); delete h; return rc; }
    *out = h;
    return 0;
}

extern "C" int cip_create(int n, int m, int p, int ncones, const int *cone_type, const int *cone_dim, const double *Q,
                          const double *A, const double *G, int route, cip_handle **out) {
    cip_problem pr;
    memset(&pr, 0, sizeof(pr));
    pr.n = n; pr.m = m; pr.p = p; pr.ncones = ncones; pr.cone_type = cone_type; pr.cone_dim = cone_dim;
    pr.Q = Q; pr.ldq = n; pr.A = A; pr.lda = m; pr.G = G; pr.ldg = p; pr.route = route; pr.flags = 0;
    if (m > 0 && !A) { cip_set_error("A is NULL"); return CIP_E_INVALID; }
    return cip_create_ex(&pr, out);
}

// Level 1 again on an existing handle: new Q / A / G of the SAME shape (n, m, p, cones, route, dense-or-CSR A with the same
// number of non-zeros).  Keeps every allocation -- hipMalloc / hipFree synchronise the whole device, which is what a
// batch of small problems on several streams must avoid (csrc/batch.hip reuses one handle per worker this way).
extern "C" int cip_update_problem(cip_handle *h, const cip_problem *pr) {
    if (!h || !pr) { cip_set_error("NULL argument"); return CIP_E_INVALID; }
    if (pr->n != h->n || pr->m != h->m || pr->p != h->p || pr->ncones != h->ncones || pr->route != h->route ||
        (pr->A == NULL && pr->m > 0) != h->A_sparse) { cip_set_error("cip_update_problem: shape differs from the handle's"); return CIP_E_INVALID; }
    for (int c = 0; c < pr->ncones; ++c)
        if (pr->cone_type[c] != h->h_cones[c].type || pr->cone_dim[c] != h->h_cones[c].dim) { cip_set_error("cip_update_problem: cone %d differs", c); return CIP_E_INVALID; }
    if (!pr->Q || (h->p > 0 && !pr->G)) { cip_set_error("NULL matrix"); return CIP_E_INVALID; }
    int rc;
    if ((rc = upload_problem(h, pr))) return rc;
    h->assembled = h->factored = false;
    h->info_pending = false;
    h->reg_rel = 0.0; h->n_regularized = 0;          // a fresh problem starts unregularised, as a fresh handle does
    if ((rc = cip_cones_identity_scaling(h->stream, h->cs))) return rc;
    return cip_sdp_scaling_changed(h->stream, h->cs);
}

extern "C" int cip_destroy(cip_handle *h) {
    if (!h) return 0;
    (void)hipStreamSynchronize(h->stream);
    free_all(h);
    delete h;
    return 0;
}

extern "C" int cip_set_stream(cip_handle *h, void *stream) {
    if (!h) return CIP_E_INVALID;
    CIP_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)stream;
    h->graph_state = 0;              // graphs are decided per stream (no capture on the null stream); existing ones stay valid
    return 0;
}

// ------------------------------------------------------------------ level 2
extern "C" size_t cip_scaling_packed_len(const cip_handle *h) { return h ? h->cs.scal_len : 0; }

extern "C" int cip_set_scaling_packed(cip_handle *h, const double *packedF) {
    if (!h || !packedF) { cip_set_error("NULL argument"); return CIP_E_INVALID; }
    CIP_HIP_CHECK(hipMemcpyAsync(h->cs.d_scal, packedF, sizeof(double) * h->cs.scal_len, hipMemcpyHostToDevice, h->stream));
    CIP_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->assembled = h->factored = false;
    return cip_sdp_scaling_changed(h->stream, h->cs);
}
extern "C" int cip_get_scaling_packed(cip_handle *h, double *packedF) {
    if (!h || !packedF) { cip_set_error("NULL argument"); return CIP_E_INVALID; }
    CIP_HIP_CHECK(hipMemcpyAsync(packedF, h->cs.d_scal, sizeof(double) * h->cs.scal_len, hipMemcpyDeviceToHost, h->stream));
    CIP_HIP_CHECK(hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int cip_set_scaling_identity(cip_handle *h) {
    if (!h) return CIP_E_INVALID;
    h->assembled = h->factored = false;
    int rc = cip_cones_identity_scaling(h->stream, h->cs);
    return rc ? rc : cip_sdp_scaling_changed(h->stream, h->cs);
}
extern "C" int cip_set_scaling_from_iterate_dev(cip_handle *h, const double *v, const double *s, double *lambda_out) {
    if (!h || (h->m > 0 && (!v || !s))) { cip_set_error("NULL argument"); return CIP_E_INVALID; }
    h->assembled = h->factored = false;
    CipRange rg("cip:scaling");
    return cip_cones_nt_scaling(h->stream, h->cs, v, s, lambda_out);
}

extern "C" int cip_assemble_only(cip_handle *h) {
    if (!h) return CIP_E_INVALID;
    return cip_assemble(h);
}

// Relative size of the automatic regularisation.  Miles problem 3 under the reference's 10 scalings (test/runtests.jl:
// 618-637), iterations to :Optimal (oracle: 15 16 16 15 21 29 30 | 20 29 32): 1e-9 -> one false :Unbounded; 1e-11 -> 11 16 16
// 15 21 26 24 | 20 28 27; 1e-13 -> 16 16 16 15 21 29 29 | 21 29 32.  The refinement inside solve3x3 is what makes the
// small value work.  CIP_AUTO_REG overrides.
#define CIP_AUTO_REG 1e-13

// ---- hipGraph replay of the LDL' factorisation / the triangular solves of SMALL systems (opt-in: CIP_GRAPH=1).  The
// launch sequence of a handle never changes (same pointers, same sizes), so it is recorded once as an explicit graph
// (cip_launch, cip_internal.h) and replayed: ~60 launches per factorisation and ~10-33 per solve become one graph launch.
// Built for config 5 (64 x n = 2048, 8 problems in flight) on the hypothesis that the batch is bound by the host's launch
// rate (95 k launches per pass).  It is not: 1936 KKT solves/s with graphs, 1995 without.  A kernel trace of a pass
// (profiles/r2/c5_*) shows 16.8 ms of serial kernel time per problem and an average of 3.4 kernels executing at once with
// 8 problems in flight (more hardware queues make it worse): the ceiling is the number of concurrently progressing
// launch-latency-bound streams, which only a lock-step batch (one launch for many problems) would lift.  Off by default
// (rocprofv3 also crashed on the graph path with several host threads).
static bool graph_wanted(cip_handle *h) {
    if (h->graph_state == 0) {
        const char *e = getenv("CIP_GRAPH");
        const int npmax = getenv("CIP_GRAPH_NMAX") ? atoi(getenv("CIP_GRAPH_NMAX")) : 4096;
        h->graph_state = (h->Npad <= npmax && e && atoi(e) == 1 && !h->timing && !h->ws.prof) ? 1 : -1;
    }
    return h->graph_state == 1;
}
thread_local CipGraphBuilder *cip_tl_builder = nullptr;
template <class F>
static int graph_run(cip_handle *h, hipGraphExec_t *exec, F &&enqueue) {
    if (cip_in_batch() || !graph_wanted(h) || h->timing || h->ws.prof) return enqueue();
    if (!*exec) {
        CipGraphBuilder b = {nullptr, nullptr, false, true};
        if (hipGraphCreate(&b.graph, 0) != hipSuccess) { (void)hipGetLastError(); h->graph_state = -1; return enqueue(); }
        cip_tl_builder = &b;
        const int rc = enqueue();                // records kernel nodes, launches nothing
        cip_tl_builder = nullptr;
        hipError_t ei = hipErrorUnknown;
        if (rc == 0 && b.ok && b.have_last) ei = hipGraphInstantiate(exec, b.graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(b.graph);
        if (ei != hipSuccess) { (void)hipGetLastError(); *exec = nullptr; h->graph_state = -1; return enqueue(); }
    }
    CIP_HIP_CHECK(hipGraphLaunch(*exec, h->stream));
    return 0;
}

// the factorisation's four flag words -> host-mapped pinned memory, by a store from the device.  (Until round 5 a 16-byte
// hipMemcpyAsync: a blit kernel between two barriers -- in the kernel trace the main queue stood idle for 24 us between the last
// panel launch and the first solve's first kernel, of which the copy itself was 5.)
// No event behind it either (an event record is a barrier packet with a system-scope release: 11 us in the same trace): the fifth
// word is the factorisation's sequence number, stored with system-scope release behind the four flags; the host polls it.
__global__ void k_publish_info(const int *info, int *host, int seq) {
    if (threadIdx.x == 0) {
        host[0] = info[0]; host[1] = info[1]; host[2] = info[2]; host[3] = info[3];
        __hip_atomic_store(host + 4, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// has the flag read-back of factorisation `info_seq` landed?  wait: spin (the stream is polled too: a failed launch must not hang the host)
static int info_landed(cip_handle *h, bool wait, bool *landed) {
    volatile int *seqp = h->info_host + 4;
    *landed = __atomic_load_n(seqp, __ATOMIC_ACQUIRE) == h->info_seq;
    if (*landed || !wait) return 0;
    for (long spin = 0;; ++spin) {
        if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == h->info_seq) { *landed = true; return 0; }
        if ((spin & 0xfff) == 0xfff) {
            std::this_thread::yield();
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) {                                   // everything enqueued has run: the word must be there
                *landed = __atomic_load_n(seqp, __ATOMIC_ACQUIRE) == h->info_seq;
                if (*landed) return 0;
                cip_set_error("LDL': the pivot flags of factorisation %d never arrived", h->info_seq);
                return CIP_E_HIP;
            }
            if (q != hipErrorNotReady) { cip_set_error("hipStreamQuery failed: %s", hipGetErrorString(q)); return CIP_E_HIP; }
        }
    }
}
// assembly + LDL' + an asynchronous read-back of the pivot flag into pinned host memory; nothing here waits for the GPU
static int factor_enqueue(cip_handle *h) {
    int rc;
    { CipRange rg("cip:assemble"); if ((rc = cip_assemble(h, true))) return rc; }
    if (h->timing) CIP_HIP_CHECK(hipEventRecord(h->ev1, h->stream));
    {
        CipRange rg("cip:ldlt");
        if (h->gx_factor && h->gx_factor_lazyC != h->ws.lazyC) {     // recorded under another lazy-copy state: record again
            (void)hipGraphExecDestroy(h->gx_factor); h->gx_factor = nullptr;
        }
        h->gx_factor_lazyC = h->ws.lazyC;
        if ((rc = graph_run(h, &h->gx_factor, [&]() { return cip_ldlt_factor(h->stream, h->K, h->Npad, h->ldk, h->ws); }))) return rc;
    }
    h->n_factor += 1;
    if (!h->info_host) {
        CIP_HIP_CHECK(hipHostMalloc((void **)&h->info_host, 8 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
        memset(h->info_host, 0, 8 * sizeof(int));
    }
    {
        int *info_dev = nullptr;
        CIP_HIP_CHECK(hipHostGetDevicePointer((void **)&info_dev, h->info_host, 0));
        h->info_seq = (h->info_seq == 0x7fffffff) ? 1 : h->info_seq + 1;
        hipLaunchKernelGGL(k_publish_info, dim3(1), dim3(64), 0, h->stream, (const int *)h->ws.info, info_dev, h->info_seq);
        CIP_HIP_CHECK(hipGetLastError());
    }
    h->info_pending = true;
    h->spec_solves = 0;
    h->factored = true;
    return 0;
}

// Resolve the pivot flag of the last factorisation.  wait = false: only if the read-back has already landed (no host
// wait; the caller goes ahead speculatively otherwise).  A zero / non-finite / wrong-sign pivot means [S G'; G 0] is not
// quasi-definite in the static order (typically an LP or a QP with singular Q and free variables: S is singular although
// the KKT matrix is not).  The reference's pivoting LU / QR does not care; the static-order LDL' switches to a
// regularised factorisation, once, for good -- and if that one fails too the error is reported (CIP_E_SINGULAR).
static int factor_resolve(cip_handle *h, bool wait);
int cip_factor_resolve(cip_handle *h, int wait) { return factor_resolve(h, wait != 0); }
static int factor_resolve(cip_handle *h, bool wait) {
    if (!h->info_pending) return 0;
    {
        bool landed = false;
        const int rc = info_landed(h, wait, &landed);
        if (rc) return rc;
        if (!landed) return 0;
    }
    h->info_pending = false;
    // info_host: [0] first bad pivot of any kind (1-based column), [1] bail-out flag of the sweep kernels,
    //            [2] first zero / non-finite pivot
    if (h->info_host[3] != 0 && h->info_host[1] == 0 && !h->ws.unfused) {
        // An in-launch wait of the fused panel chain gave up.  Its waits are for workgroups of the same launch and end within
        // microseconds when the launch has the GPU's attention; on a GPU SHARED WITH OTHER PROCESSES the hardware scheduler can take
        // a launch's workgroups off the chip for longer than the bound (seen with eight processes on one MI355X: round 6).  The
        // three-launch chain has no in-launch wait and produces the same bits: this handle keeps it from now on, and the
        // factorisation is redone (assembly included: K holds a partial factor).
        const int spec0 = h->spec_solves;
        h->ws.unfused = 1;
        h->n_chain_fallbacks += 1;
        int rc;
        if ((rc = factor_enqueue(h))) return rc;
        { bool landed = false; if ((rc = info_landed(h, true, &landed))) return rc; }
        h->info_pending = false;
        if (spec0 > 0) {
            cip_set_error("LDL': %d solve(s) were enqueued on a factorisation whose panel chain gave up an in-launch wait; the handle has "
                          "switched to the three-launch chain -- repeat them", spec0);
            return CIP_E_SINGULAR;
        }
    }
    if (h->info_host[3] != 0 || h->info_host[1] != 0) {
        cip_set_error("LDL': in-launch wait of the panel chain gave up (%d)", h->info_host[3]);
        h->factored = false;
        return CIP_E_HIP;
    }
    int info = h->info_host[0];
    if (info == 0) { h->pivots_verified = true; return 0; }
    const int spec = h->spec_solves;
    if (h->reg_rel > 0.0) {
        // regularised factor: a wrong-sign pivot (|d| ~ delta, rounding decides its sign) is harmless -- the refinement in
        // solve3x3 works against the true operator -- but a zero / non-finite one is not
        if (h->info_host[2] == 0) { h->pivots_verified = true; return 0; }
        info = h->info_host[2];
    } else if (h->auto_reg) {
        h->reg_rel = getenv("CIP_AUTO_REG") ? atof(getenv("CIP_AUTO_REG")) : CIP_AUTO_REG;
        h->n_regularized += 1;
        int rc;
        if ((rc = factor_enqueue(h))) return rc;
        { bool landed = false; if ((rc = info_landed(h, true, &landed))) return rc; }
        h->info_pending = false;
        if (h->info_host[2] == 0) {
            h->pivots_verified = true;
            if (spec > 0) {
                cip_set_error("LDL': %d solve(s) were enqueued on a factorisation that met a bad pivot; the handle has switched "
                              "to the regularised factorisation -- repeat them", spec);
                return CIP_E_SINGULAR;
            }
            return 0;
        }
        info = h->info_host[2];
    }
    h->factored = false;
    h->pivots_verified = false;      // keep waiting for the flag after a factorisation that failed
    cip_set_error("LDL': zero, non-finite or wrong-sign pivot at column %d%s", info,
                  h->reg_rel > 0.0 ? " (regularised factorisation)" : "");
    return CIP_E_SINGULAR;
}

extern "C" int cip_factor(cip_handle *h) {
    if (!h) return CIP_E_INVALID;
    int rc;
    if (h->timing) CIP_HIP_CHECK(hipEventRecord(h->ev0, h->stream));
    if ((rc = factor_enqueue(h))) return rc;
    h->flops_ldlt = (double)h->N * h->N * h->N / 3.0;
    h->factored = true;
    if (h->timing) {
        // (timed factorisations include the solve preparation that otherwise runs beside the first solve)
        if ((rc = cip_ldlt_side_join(h->stream, h->ws, -1))) return rc;
        CIP_HIP_CHECK(hipEventRecord(h->ev2, h->stream));
        CIP_HIP_CHECK(hipEventSynchronize(h->ev2));
        float a = 0, b = 0;
        CIP_HIP_CHECK(hipEventElapsedTime(&a, h->ev0, h->ev1));
        CIP_HIP_CHECK(hipEventElapsedTime(&b, h->ev1, h->ev2));
        h->ms_assemble = a; h->ms_ldlt = b;
        return factor_resolve(h, true);
    }
    return 0;
}
extern "C" int cip_set_regularization(cip_handle *h, double rel, int automatic) {
    if (!h || !(rel >= 0.0)) { cip_set_error("cip_set_regularization: bad argument"); return CIP_E_INVALID; }
    h->reg_rel = rel;
    h->auto_reg = automatic != 0;
    h->pivots_verified = false;      // a different matrix is factored from now on: the next *_dev solve waits for its flag again
    return 0;
}
extern "C" int cip_get_regularization(cip_handle *h, double *rel, int *times_switched_on) {
    if (!h) return CIP_E_INVALID;
    if (rel) *rel = h->reg_rel;
    if (times_switched_on) *times_switched_on = h->n_regularized;
    return 0;
}

extern "C" int cip_check_factor(cip_handle *h) {
    if (!h) return CIP_E_INVALID;
    if (!h->factored) { cip_set_error("no factorisation"); return CIP_E_NOTFACTORED; }
    int rc;
    if ((rc = factor_resolve(h, true))) return rc;
    CIP_HIP_CHECK(hipStreamSynchronize(h->stream));
    return 0;
}

// ------------------------------------------------------------------ products with A, A'
static int mul_A(cip_handle *h, double alpha, const double *x, double beta, double *y) {      // y = alpha A x + beta y  (m)
    if (h->m == 0) return 0;
    if (h->A_sparse) return cip_spmv_csr(h->stream, h->m, h->A_rp, h->A_ci, h->A_v, alpha, x, beta, y);
    return cip_gemv_t(h->stream, h->n, h->m, alpha, h->At, h->npad, x, beta, y);
}
static int mul_At(cip_handle *h, double alpha, const double *x, double beta, double *y) {     // y = alpha A' x + beta y (n)
    if (h->m == 0) { if (beta == 0.0) { int rcc = cip_zero(h->stream, h->n, y); if (rcc) return rcc; } return 0; }
    if (h->A_sparse) return cip_spmv_csr(h->stream, h->n, h->T_rp, h->T_ci, h->T_v, alpha, x, beta, y);
    return cip_gemv_t(h->stream, h->m, h->n, alpha, h->A, h->m, x, beta, y);
}

// (F'F)^-1 z = F^-1 F^-T z
static int apply_FtF_inv(cip_handle *h, const double *z, double *tmp, double *out) {
    int rc;
    if ((rc = cip_cones_apply(h->stream, h->cs, CIP_OP_FINVT, z, tmp))) return rc;
    return cip_cones_apply(h->stream, h->cs, CIP_OP_FINV, tmp, out);
}

// ------------------------------------------------------------------ level 3
// z == NULL (Schur route only): the 2x2 form [S G'; G 0][a; b] = [x; y] (src/kktsolvers.jl:297-302), c is not written
static int solve3x3_once(cip_handle *h, const double *x, const double *y, const double *z, double *a, double *b, double *c) {
    hipStream_t s = h->stream;
    const int n = h->n, p = h->p;
    const int m = z ? h->m : 0;
    int rc;
    if (h->route == CIP_ROUTE_SCHUR) {
        // algebra of pivotgen, src/kktsolvers.jl:324-330, with the exact (F'F)^-1 = F^-1 F^-T:
        //   t = (F'F)^-1 z ; [S G'; G 0][a; b] = [x + A't; y] ; c = t - (F'F)^-1 A a
        double *t = h->mt1, *tmp = h->mt2, *u = h->mt3;
        if (m > 0 && (rc = apply_FtF_inv(h, z, tmp, t))) return rc;
        { int rcc = cip_copy(s, n, x, h->rhs); if (rcc) return rcc; }
        if (m > 0 && (rc = mul_At(h, 1.0, t, 1.0, h->rhs))) return rc;
        if (p > 0) { int rcc = cip_copy(s, p, y, h->rhs + n); if (rcc) return rcc; }
        if (h->Npad > h->N) { int rcc = cip_zero(s, (h->Npad - h->N), h->rhs + h->N); if (rcc) return rcc; }
        if ((rc = graph_run(h, &h->gx_solve, [&]() { return cip_ldlt_solve(s, h->K, h->Npad, h->ldk, h->ws, h->rhs); }))) return rc;
        { int rcc = cip_copy(s, n, h->rhs, a); if (rcc) return rcc; }
        if (p > 0) { int rcc = cip_copy(s, p, h->rhs + n, b); if (rcc) return rcc; }
        if (m > 0) {
            if ((rc = mul_A(h, 1.0, h->rhs, 0.0, u))) return rc;
            if ((rc = apply_FtF_inv(h, u, tmp, u))) return rc;
            if ((rc = cip_axpby(s, m, 1.0, t, 0.0, c))) return rc;
            if ((rc = cip_axpby(s, m, -1.0, u, 1.0, c))) return rc;
        }
    } else {
        // [-F'F -A 0; -A' Q G'; 0 G 0] [c; a; b] = [-z; x; y]
        if (m > 0 && (rc = cip_axpby(s, m, -1.0, z, 0.0, h->rhs))) return rc;
        { int rcc = cip_copy(s, n, x, h->rhs + m); if (rcc) return rcc; }
        if (p > 0) { int rcc = cip_copy(s, p, y, h->rhs + m + n); if (rcc) return rcc; }
        if (h->Npad > h->N) { int rcc = cip_zero(s, (h->Npad - h->N), h->rhs + h->N); if (rcc) return rcc; }
        if ((rc = graph_run(h, &h->gx_solve, [&]() { return cip_ldlt_solve(s, h->K, h->Npad, h->ldk, h->ws, h->rhs); }))) return rc;
        if (m > 0) { int rcc = cip_copy(s, m, h->rhs, c); if (rcc) return rcc; }
        { int rcc = cip_copy(s, n, h->rhs + m, a); if (rcc) return rcc; }
        if (p > 0) { int rcc = cip_copy(s, p, h->rhs + m + n, b); if (rcc) return rcc; }
    }
    return 0;
}

// r = (x, y, z) - K3 (a, b, c) with the TRUE (unregularised) operator:
//   rx = x - (Q a + G' b - A' c) ;  ry = y - G a ;  rz = z - (A a + F'F c)
static int kkt3_residual(cip_handle *h, const double *x, const double *y, const double *z, const double *a, const double *b,
                         const double *c, double *rx, double *ry, double *rz) {
    hipStream_t s = h->stream;
    const int n = h->n, m = h->m, p = h->p;
    int rc;
    if ((rc = cip_axpby(s, n, 1.0, x, 0.0, rx))) return rc;
    if ((rc = cip_gemv_t(s, n, n, -1.0, h->Q, n, a, 1.0, rx))) return rc;
    if (p > 0) {
        if ((rc = cip_gemv_t(s, p, n, -1.0, h->G, p, b, 1.0, rx))) return rc;
        if ((rc = cip_axpby(s, p, 1.0, y, 0.0, ry))) return rc;
        if ((rc = cip_gemv_t(s, n, p, -1.0, h->Gt, n, a, 1.0, ry))) return rc;
    }
    if (m > 0) {
        if ((rc = mul_At(h, 1.0, c, 1.0, rx))) return rc;
        if ((rc = cip_axpby(s, m, 1.0, z, 0.0, rz))) return rc;
        if ((rc = mul_A(h, -1.0, a, 1.0, rz))) return rc;
        if ((rc = cip_cones_apply(s, h->cs, CIP_OP_F, c, h->mt2))) return rc;
        if ((rc = cip_cones_apply(s, h->cs, CIP_OP_FT, h->mt2, h->mt2))) return rc;
        if ((rc = cip_axpby(s, m, -1.0, h->mt2, 1.0, rz))) return rc;
    }
    return 0;
}

extern "C" int cip_solve3x3_dev(cip_handle *h, const double *x, const double *y, const double *z, double *a, double *b,
                                double *c) {
    if (!h) return CIP_E_INVALID;
    if (!h->factored) { cip_set_error("cip_solve3x3: no factorisation (call cip_factor first)"); return CIP_E_NOTFACTORED; }
    const int n = h->n, m = h->m, p = h->p;
    int rc;
    CipRange rg("cip:solve3x3");
    // speculative (no host wait, the flag is resolved only if its read-back has landed) once a factorisation of this handle
    // has been verified; before that the solve waits for the flag
    if ((rc = factor_resolve(h, !h->pivots_verified))) return rc;
    if (h->info_pending) h->spec_solves += 1;
    h->n_solve += 1;
    if (h->reg_rel <= 0.0) return solve3x3_once(h, x, y, z, a, b, c);
    // Regularised factor: iterative refinement against the true operator (the factor is of K + E, |E_ii| = reg_rel
    // times the row's largest entry): a few steps bring the residual to rounding level while ||K^-1 E|| < 1.
    hipStream_t s = h->stream;
    const size_t tot = (size_t)n + p + m;
    if (!h->ref) DMALLOC(h->ref, sizeof(double) * 3 * (tot + 8));
    double *xs = h->ref, *ys = xs + n, *zs = ys + p;              // private copy of the right-hand side (z may alias c)
    double *rx = h->ref + tot + 8, *ry = rx + n, *rz = ry + p;
    double *da = h->ref + 2 * (tot + 8), *db = da + n, *dc = db + p;
    { int rcc = cip_copy(s, n, x, xs); if (rcc) return rcc; }
    if (p > 0) { int rcc = cip_copy(s, p, y, ys); if (rcc) return rcc; }
    if (m > 0) { int rcc = cip_copy(s, m, z, zs); if (rcc) return rcc; }
    if ((rc = solve3x3_once(h, xs, ys, zs, a, b, c))) return rc;
    double prev = __builtin_inf();
    for (int it = 0; it < 6; ++it) {
        if ((rc = kkt3_residual(h, xs, ys, zs, a, b, c, rx, ry, rz))) return rc;
        const double *px[6] = {rx, ry, rz, xs, ys, zs};
        const int ln[6] = {n, p, m, n, p, m};
        double d6[6];
        if ((rc = cip_dots(s, 6, px, px, ln, h->dot_scratch, h->dot_ptrs, d6))) return rc;
        const double rn = sqrt(d6[0] + d6[1] + d6[2]), bn = sqrt(d6[3] + d6[4] + d6[5]);
        if (!(rn > 1e-14 * bn) || !(rn < prev)) break;           // converged, stagnating, or not finite
        prev = rn;
        if ((rc = solve3x3_once(h, rx, ry, rz, da, db, dc))) return rc;
        if ((rc = cip_axpby(s, n, 1.0, da, 1.0, a))) return rc;
        if (p > 0 && (rc = cip_axpby(s, p, 1.0, db, 1.0, b))) return rc;
        if (m > 0 && (rc = cip_axpby(s, m, 1.0, dc, 1.0, c))) return rc;
    }
    return 0;
}

extern "C" int cip_solve3x3(cip_handle *h, const double *x, const double *y, const double *z, double *a, double *b,
                            double *c) {
    if (!h) return CIP_E_INVALID;
    hipStream_t s = h->stream;
    const int n = h->n, m = h->m, p = h->p;
    double *in = h->stage, *out = h->stage + (n + p + m);
    CIP_HIP_CHECK(hipMemcpyAsync(in, x, sizeof(double) * n, hipMemcpyHostToDevice, s));
    if (p > 0) CIP_HIP_CHECK(hipMemcpyAsync(in + n, y, sizeof(double) * p, hipMemcpyHostToDevice, s));
    if (m > 0) CIP_HIP_CHECK(hipMemcpyAsync(in + n + p, z, sizeof(double) * m, hipMemcpyHostToDevice, s));
    int rc;
    if (h->factored && (rc = factor_resolve(h, true))) return rc;      // this entry point is synchronous anyway
    if ((rc = cip_solve3x3_dev(h, in, in + n, in + n + p, out, out + n, out + n + p))) return rc;
    CIP_HIP_CHECK(hipMemcpyAsync(a, out, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    if (p > 0) CIP_HIP_CHECK(hipMemcpyAsync(b, out + n, sizeof(double) * p, hipMemcpyDeviceToHost, s));
    if (m > 0) CIP_HIP_CHECK(hipMemcpyAsync(c, out + n + p, sizeof(double) * m, hipMemcpyDeviceToHost, s));
    CIP_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
}

// The 2x2 form of the plugin (src/ConicIP.jl:450-466; what `pivot` wraps, src/kktsolvers.jl:297-302, :316-349):
//   [Q + A'(F'F)^-1 A   G'] [dy]   [y]
//   [G                  0 ] [dw] = [w]      on the factor of the Schur route.
extern "C" int cip_solve2x2_dev(cip_handle *h, const double *y, const double *w, double *dy, double *dw) {
    if (!h) return CIP_E_INVALID;
    if (h->route != CIP_ROUTE_SCHUR) { cip_set_error("cip_solve2x2: needs the Schur route"); return CIP_E_UNSUPPORTED; }
    if (!h->factored) { cip_set_error("cip_solve2x2: no factorisation (call cip_factor first)"); return CIP_E_NOTFACTORED; }
    int rc;
    if ((rc = factor_resolve(h, !h->pivots_verified))) return rc;
    if (h->reg_rel > 0.0 && h->m > 0) {
        // regularised factor: go through the refined 3x3 solve with z = 0 (its first two components are the 2x2 solution)
        { int rcc = cip_zero(h->stream, h->m, h->mt1); if (rcc) return rcc; }
        if (!h->c2x2) DMALLOC(h->c2x2, sizeof(double) * h->m);
        return cip_solve3x3_dev(h, y, w, h->mt1, dy, dw, h->c2x2);
    }
    if (h->info_pending) h->spec_solves += 1;
    h->n_solve += 1;
    return solve3x3_once(h, y, w, nullptr, dy, dw, nullptr);
}
extern "C" int cip_solve2x2(cip_handle *h, const double *y, const double *w, double *dy, double *dw) {
    if (!h) return CIP_E_INVALID;
    hipStream_t s = h->stream;
    const int n = h->n, p = h->p;
    double *in = h->stage, *out = h->stage + (n + p + h->m);
    CIP_HIP_CHECK(hipMemcpyAsync(in, y, sizeof(double) * n, hipMemcpyHostToDevice, s));
    if (p > 0) CIP_HIP_CHECK(hipMemcpyAsync(in + n, w, sizeof(double) * p, hipMemcpyHostToDevice, s));
    int rc;
    if (h->factored && (rc = factor_resolve(h, true))) return rc;
    if ((rc = cip_solve2x2_dev(h, in, in + n, out, out + n))) return rc;
    CIP_HIP_CHECK(hipMemcpyAsync(dy, out, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    if (p > 0) CIP_HIP_CHECK(hipMemcpyAsync(dw, out + n, sizeof(double) * p, hipMemcpyDeviceToHost, s));
    CIP_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
}

// diag F (the packed scaling) when the cone set is R cones only -- the native loops then fuse the cone operations of their
// element-wise chains into the vector kernels (vecops.hip: k_loop_*); NULL otherwise (or with CIP_LOOP_FUSED_R=0)
const double *cip_loop_all_r(cip_handle *h) {
    static const int on = [] { const char *e = getenv("CIP_LOOP_FUSED_R"); return e ? atoi(e) : 1; }();
    if (!on || h->m <= 0) return nullptr;
    for (int q = 0; q < h->cs.ncones; ++q) if (h->h_cones[q].type != CIP_CONE_R) return nullptr;
    return h->cs.d_scal;
}
// solve4x4 (src/ConicIP.jl:684-692):  q = r.s (./) lambda ; t1 = F'q ; (dy,dw,dv) = solve3x3(r.y, r.w, r.v + t1) ;
// ds = t1 - F'(F dv).   r, dz are contiguous (y[n], w[p], v[m], s[m]).
extern "C" int cip_solve4x4_dev(cip_handle *h, const double *lambda, const double *r, double *dz) {
    if (!h) return CIP_E_INVALID;
    hipStream_t s = h->stream;
    const int n = h->n, m = h->m, p = h->p;
    const double *ry = r, *rw = r + n, *rv = r + n + p, *rs = r + n + p + m;
    double *dy = dz, *dw = dz + n, *dv = dz + n + p, *ds = dz + n + p + m;
    int rc;
    if (h->all_r < 0) {
        h->all_r = m > 0 ? 1 : 0;
        for (int q = 0; q < h->cs.ncones; ++q) if (h->cs.h_cones[q].type != CIP_CONE_R) h->all_r = 0;
        const char *e = getenv("CIP_S4_FUSED");
        if (e && atoi(e) == 0) h->all_r = 0;
    }
    // the fused kernels read r across threads (CSR gathers) while writing dz: an in-place call (dz overlapping r) takes the
    // element-wise generic path, which tolerates it
    const size_t len4 = (size_t)n + p + 2 * (size_t)m;
    const bool aliased = dz < r + len4 && r < dz + len4;
    if (h->all_r && !aliased && h->route == CIP_ROUTE_SCHUR && h->reg_rel <= 0.0) {
        // all cones R (F = diag(f), f = the packed scaling): the element-wise launches around the sweeps fused into one kernel
        // in front and one behind them (vecops.hip: k_s4_pre_r / k_s4_post_r), the same operations on every element
        if (!h->factored) { cip_set_error("cip_solve4x4: no factorisation (call cip_factor first)"); return CIP_E_NOTFACTORED; }
        CipRange rg("cip:solve3x3");
        if ((rc = factor_resolve(h, !h->pivots_verified))) return rc;
        if (h->reg_rel <= 0.0) {                       // (the resolve may have switched the handle to the regularised factorisation)
            if (h->info_pending) h->spec_solves += 1;
            h->n_solve += 1;
            const double *f = h->cs.d_scal;
            double *t = h->mt1, *u = h->mt3;
            if ((rc = cip_s4_pre_r(s, m, n, p, h->Npad, f, rs, lambda, rv, ry, rw, ds, t, h->rhs, h->A_sparse ? h->T_rp : nullptr, h->T_ci, h->T_v))) return rc;
            if (!h->A_sparse && (rc = mul_At(h, 1.0, t, 1.0, h->rhs))) return rc;
            if ((rc = graph_run(h, &h->gx_solve, [&]() { return cip_ldlt_solve(s, h->K, h->Npad, h->ldk, h->ws, h->rhs); }))) return rc;
            if (!h->A_sparse && (rc = mul_A(h, 1.0, h->rhs, 0.0, u))) return rc;
            return cip_s4_post_r(s, m, n, p, f, t, h->rhs, h->A_sparse ? nullptr : u, h->A_sparse ? h->A_rp : nullptr, h->A_ci, h->A_v, dy, dw, dv, ds);
        }
    }
    // ds is used as t1; dv temporarily holds r.v + t1 (input z of the 3x3 solve; solve3x3 copies it before writing c)
    if (m > 0) {
        if ((rc = cip_cones_div(s, h->cs, rs, lambda, ds))) return rc;          // q
        if ((rc = cip_cones_apply(s, h->cs, CIP_OP_FT, ds, ds))) return rc;     // t1 = F'q (in place)
        if ((rc = cip_axpby(s, m, 1.0, rv, 0.0, dv))) return rc;
        if ((rc = cip_axpby(s, m, 1.0, ds, 1.0, dv))) return rc;                // dv <- r.v + t1
    }
    if ((rc = cip_solve3x3_dev(h, ry, rw, dv, dy, dw, dv))) return rc;
    if (m > 0) {
        double *u = h->mt3;
        if ((rc = cip_cones_apply(s, h->cs, CIP_OP_F, dv, u))) return rc;
        if ((rc = cip_cones_apply(s, h->cs, CIP_OP_FT, u, u))) return rc;
        if ((rc = cip_axpby(s, m, -1.0, u, 1.0, ds))) return rc;                // ds = t1 - F'(F dv)
    }
    return 0;
}

// ------------------------------------------------------------------ cone algebra / vector helpers
extern "C" int cip_apply_F_dev(cip_handle *h, int mode, const double *x, double *out) {
    if (!h || mode < 0 || mode > 3) { cip_set_error("bad argument"); return CIP_E_INVALID; }
    return cip_cones_apply(h->stream, h->cs, mode, x, out);
}
extern "C" int cip_cone_prod_dev(cip_handle *h, const double *x, const double *y, double *out) {
    if (!h) return CIP_E_INVALID;
    return cip_cones_prod(h->stream, h->cs, x, y, out);
}
extern "C" int cip_cone_div_dev(cip_handle *h, const double *x, const double *y, double *out) {
    if (!h) return CIP_E_INVALID;
    return cip_cones_div(h->stream, h->cs, x, y, out);
}
extern "C" int cip_maxstep_dev(cip_handle *h, const double *x, const double *d, double scale, double *alpha_host) {
    if (!h || !alpha_host) return CIP_E_INVALID;
    return cip_cones_maxstep(h->stream, h->cs, x, d, scale, alpha_host);
}
extern "C" int cip_maxstep_pair_dev(cip_handle *h, const double *x1, const double *d1, const double *x2, const double *d2, double scale,
                                    double *alpha_host2) {
    if (!h || !alpha_host2) return CIP_E_INVALID;
    return cip_cones_maxstep2(h->stream, h->cs, x1, d1, x2, d2, scale, alpha_host2);
}
extern "C" int cip_cone_identity_dev(cip_handle *h, double *e) {
    if (!h) return CIP_E_INVALID;
    return cip_cones_identity(h->stream, h->cs, e);
}

extern "C" int cip_gemv_dev(cip_handle *h, int which, int trans, double alpha, const double *x, double beta, double *y) {
    if (!h) return CIP_E_INVALID;
    hipStream_t s = h->stream;
    switch (which) {
        case CIP_MAT_Q:                                                                     // Q symmetric
            if (h->symv_ws && !(((uintptr_t)x) & 15)) return cip_symv_lower(s, h->n, alpha, h->Q, h->n, x, beta, y, h->symv_ws);
            return cip_gemv_t(s, h->n, h->n, alpha, h->Q, h->n, x, beta, y);
        case CIP_MAT_A: return trans ? mul_At(h, alpha, x, beta, y) : mul_A(h, alpha, x, beta, y);
        case CIP_MAT_G:
            if (h->p == 0) {
                if (trans && beta == 0.0) { int rcc = cip_zero(s, h->n, y); if (rcc) return rcc; }
                else if (trans && beta != 1.0) return cip_axpby(s, h->n, 0.0, y, beta, y);
                return 0;
            }
            return trans ? cip_gemv_t(s, h->p, h->n, alpha, h->G, h->p, x, beta, y)
                         : cip_gemv_t(s, h->n, h->p, alpha, h->Gt, h->n, x, beta, y);
        default: cip_set_error("bad matrix id"); return CIP_E_INVALID;
    }
}
extern "C" int cip_dots_dev(cip_handle *h, int count, const double *const *x, const double *const *y, const int *len,
                            double *out_host) {
    if (!h) return CIP_E_INVALID;
    return cip_dots(h->stream, count, x, y, len, h->dot_scratch, h->dot_ptrs, out_host);
}
extern "C" int cip_axpby_dev(cip_handle *h, int len, double alpha, const double *x, double beta, double *y) {
    if (!h) return CIP_E_INVALID;
    return cip_axpby(h->stream, len, alpha, x, beta, y);
}

// ------------------------------------------------------------------ stand-alone LDL' / GEMM
extern "C" int cip_ldlt_workspace_bytes(int N, size_t *bytes) {
    if (N <= 0 || N % CIP_NB || !bytes) { cip_set_error("N must be a positive multiple of 128"); return CIP_E_INVALID; }
    *bytes = cip_ldlt_ws_bytes(N, -1);                  // room for the one-launch block steps whatever the mode is now or later
    return 0;
}
extern "C" int cip_ldlt_factor_dev(void *stream, double *K, int N, int ld, void *workspace, int *info_host) {
    if (!K || !workspace || N <= 0 || N % CIP_NB || ld < N || ld % 2) { cip_set_error("bad argument"); return CIP_E_INVALID; }
    LdltWorkspace ws{};
    cip_ldlt_ws_carve(workspace, N, &ws, cip_ldlt_fused_for(N));      // (the workspace has room for either mode: cip_ldlt_workspace_bytes)
    int rc = cip_ldlt_factor((hipStream_t)stream, K, N, ld, ws);
    if (rc) return rc;
    if (info_host) {
        CIP_HIP_CHECK(hipMemcpyAsync(info_host, ws.info, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
        CIP_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    }
    return 0;
}
extern "C" int cip_ldlt_solve_dev(void *stream, const double *K, int N, int ld, const void *workspace, double *rhs) {
    if (!K || !workspace || !rhs || N <= 0 || N % CIP_NB) { cip_set_error("bad argument"); return CIP_E_INVALID; }
    LdltWorkspace ws{};
    cip_ldlt_ws_carve((void *)workspace, N, &ws, cip_ldlt_fused_for(N));   // same mode as at the factorisation: cip_set_solve_fused must not change between a factor and its solves
    return cip_ldlt_solve((hipStream_t)stream, K, N, ld, ws, rhs);
}
extern "C" int cip_gemm_nt_dev(void *stream, int M, int N, int K, double alpha, const double *A, int lda, const double *B,
                               int ldb, double *C, int ldc, int lower_only) {
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = alpha;
    g.lower = lower_only ? 1 : 0;
    return cip_launch_gemm((hipStream_t)stream, EPI_ACCUM, g);
}

// ------------------------------------------------------------------ introspection
extern "C" int cip_kkt_order(const cip_handle *h, int *N, int *N_padded) {
    if (!h) return CIP_E_INVALID;
    if (N) *N = h->N;
    if (N_padded) *N_padded = h->Npad;
    return 0;
}
extern "C" int cip_get_kkt_matrix(cip_handle *h, double *K_host) {
    if (!h || !K_host) return CIP_E_INVALID;
    { const int rj = cip_ldlt_side_join(h->stream, h->ws, -1); if (rj) return rj; }
    CIP_HIP_CHECK(hipStreamSynchronize(h->stream));
    CIP_HIP_CHECK(hipMemcpy(K_host, h->K, sizeof(double) * (size_t)h->ldk * h->Npad, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int cip_stats(cip_handle *h, double *out8) {
    if (!h || !out8) return CIP_E_INVALID;
    out8[0] = h->n_factor; out8[1] = h->n_solve; out8[2] = h->ms_assemble; out8[3] = h->ms_ldlt; out8[4] = h->flops_ldlt;
    out8[5] = cip_ldlt_outer_block_for(h->Npad); out8[6] = h->N; out8[7] = h->Npad;
    return 0;
}
extern "C" int cip_set_timing(cip_handle *h, int enabled) {
    if (!h) return CIP_E_INVALID;
    h->timing = enabled != 0;
    return 0;
}
// per-launch timing of the LDL' trailing-update kernel (HIP events on the handle's stream)
extern "C" int cip_profile_trailing(cip_handle *h, int enabled) {
    if (!h) return CIP_E_INVALID;
    if (enabled && !h->ws.prof) h->ws.prof = cip_ldlt_profile_create();
    if (enabled) cip_ldlt_profile_stride(h->ws.prof, enabled);      // enabled = k > 1: every k-th factorisation (the first one included)
    if (!enabled && h->ws.prof) { cip_ldlt_profile_destroy(h->ws.prof); h->ws.prof = nullptr; }
    return 0;
}
// out3 = [launches, total ms, total algorithmic flops] accumulated since profiling was enabled
extern "C" int cip_profile_get(cip_handle *h, double *out3) {
    if (!h || !out3 || !h->ws.prof) { cip_set_error("profiling not enabled"); return CIP_E_INVALID; }
    return cip_ldlt_profile_collect(h->ws.prof, &out3[0], &out3[1], &out3[2]);
}
// HIP-event timing of further dominant kernels on the calling thread (cip_conicip runs on the caller's thread):
// slot 0 = LDL' trailing update (== cip_profile_trailing_thread), 1 = Schur formation with a dense A, 2 = one-sided Jacobi of
// a large S cone's NT scaling.  out3 = [launches, total ms, total algorithmic flops (0 for the latency-bound Jacobi)]
extern "C" int cip_profile_kernel_thread(int slot, int enabled) {
    if (cip_prof_slot_enable(slot, enabled)) { cip_set_error("cip_profile_kernel_thread: no slot %d", slot); return CIP_E_INVALID; }
    return 0;
}
extern "C" int cip_profile_kernel_thread_get(int slot, double *out3) {
    if (!out3) return CIP_E_INVALID;
    if (cip_prof_slot_collect(slot, &out3[0], &out3[1], &out3[2])) { cip_set_error("slot %d: profiling not enabled", slot); return CIP_E_INVALID; }
    return 0;
}
// the calling thread's own trailing-update profile: covers factorisations of handles it does not hold (lock-step batches)
extern "C" int cip_profile_trailing_thread(int enabled) { return cip_ldlt_profile_thread(enabled); }
extern "C" int cip_profile_thread_get(double *out3) {
    if (!out3) return CIP_E_INVALID;
    if (cip_ldlt_profile_thread_collect(&out3[0], &out3[1], &out3[2])) { cip_set_error("thread profiling not enabled"); return CIP_E_INVALID; }
    return 0;
}
extern "C" int cip_set_lazy_copy(int on) { return cip_lazy_copy_set(on); }
extern "C" int cip_set_sdp_lanczos(int on) { return cip_sdp_large_lanczos(on); }
extern "C" int cip_debug_chain_giveup(int n) { return cip_debug_chain_giveup_set(n); }
extern "C" int cip_get_chain_fallbacks(cip_handle *h) { return h ? h->n_chain_fallbacks : -1; }
extern "C" int cip_sdp_lanczos_fallbacks(cip_handle *h, int *count) {
    if (!h || !count) { cip_set_error("bad argument"); return CIP_E_INVALID; }
    int out2[2] = {0, 0};
    const int rc = cip_sdp_large_cert_stats(h->stream, h->cs.lg, out2);
    *count = out2[0];
    return rc;
}
extern "C" int cip_set_ldlt_fused_chain(int on) { return cip_ldlt_set_fused_chain(on); }
extern "C" int cip_set_ldlt_side_prep(int on) { return cip_ldlt_set_side_prep(on); }
extern "C" int cip_set_solve_block_max(int b) { return cip_solve_block_max_set(b); }
extern "C" int cip_set_solve_fused(int mode) { return cip_solve_fused_set(mode); }
extern "C" int cip_set_ldlt_outer_block(int nbo) { cip_ldlt_set_outer_block(nbo); return cip_ldlt_outer_block(); }
