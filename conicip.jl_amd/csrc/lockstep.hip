// Lock-step batches: B independent problems of IDENTICAL shape (n, m, p, cone list, route, dense / CSR A with the same
// number of non-zeros) advance through the interior-point loop of src/ConicIP.jl:468-939 together, every step of the
// loop ONE launch whose grid carries the problem index in blockIdx.z (BASELINE config 5: 64 dense QPs of n = 2048;
// SURVEY 7.4(5) "batch dimension").
//
// Why: a small system is a chain of tiny dependent launches (n = 2048: ~1500 launches and 16.8 ms of serial kernel time
// per problem, 30 % of it the single-workgroup diagonal kernel).  Host threads with one stream each (batch.hip) keep only
// ~3.4 kernels executing at once -- the streams are launch-latency-bound and share four hardware queues.  In lock-step
// the diagonal kernel of 64 problems is one launch of 64 workgroups, the TRSM one launch of 64 x 30, and the host pays
// one launch and one read-back per step for all of them.
//
// How: every device buffer of problem z is carved, in creation order, out of slab z of one arena, so that problem z's
// pointers are problem 0's + z * stride (cip_handle_alloc).  The library's host code then runs ONCE, on problem 0's
// handle, under a thread-local batch context (cip_internal.h: cip_launch_b appends {stride, mask} to every launch and
// multiplies grid.z by B; kernels shift their pointer arguments and drop out when their problem's mask bit is clear).
// The kernels, their grids in x / y and their arithmetic are those of the one-problem path: results are bit-identical
// to cip_conicip on each problem (tests/test_gpu_lockstep.py).  Per-problem control flow is the mask: problems that have
// reached a final status stop taking part; the refinement loop runs on the subset that still needs it.  A problem whose
// factorisation meets a bad pivot (it would switch to the regularised factorisation: LPs, singular Q with free
// variables) is taken out of the lock-step group and solved afterwards by cip_conicip on its own handle.
//
// Not supported in lock-step (the caller falls back to the thread pool of batch.hip): S cones of order >= 133 (their
// chip-wide kernels own one workspace), problems of differing shape.
#include "cip_driver.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace cipdrv;

thread_local CipBatchCtx cip_tl_bz = {1, 0, 1ull, nullptr, nullptr};

namespace {

struct BatchScope {           // activates the batch context for the calling thread; restores on exit
    CipBatchCtx saved;
    explicit BatchScope(const CipBatchCtx &c) : saved(cip_tl_bz) { cip_tl_bz = c; }
    ~BatchScope() { cip_tl_bz = saved; }
};

// the four pivot-flag words of a factorisation into slots 32..35 of the problem's row of the gather buffer (slots 0..31
// carry the dot products: the flags come back with the same device-to-host copy)
#define INFO_SLOT 32
#ifndef CIP_LOCKSTEP_SPLIT_MIN_DEFAULT
#define CIP_LOCKSTEP_SPLIT_MIN_DEFAULT 8    // smallest group a split may produce (2 x 4 problems side by side LOSE: 15.3 -> 16 ms per pass)
#endif
#ifndef CIP_LOCKSTEP_SPLIT_DEFAULT
#define CIP_LOCKSTEP_SPLIT_DEFAULT 2        // (round 6, measured: see cip_conicip_lockstep)
#endif
#define STEP_SLOT 40            // deferred max-step minima (cones.hip: cip_cones_maxstep with a defer slot)
__global__ void k_gather_info(const int *info, double *gather, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO1(cb, info);
    if (threadIdx.x < 4) gather[blockIdx.z * CIP_GATHER + INFO_SLOT + threadIdx.x] = (double)info[threadIdx.x];
}

static thread_local bool g_stats_accumulate = false;     // cip_conicip_mixed: the groups' statistics add up
bool same_shape(const cip_problem &a, const cip_problem &b) {
    if (a.n != b.n || a.m != b.m || a.p != b.p || a.ncones != b.ncones || a.route != b.route) return false;
    if ((a.A == nullptr) != (b.A == nullptr) || (a.flags & CIP_FLAG_DEVICE_PTRS) != (b.flags & CIP_FLAG_DEVICE_PTRS)) return false;
    for (int c = 0; c < a.ncones; ++c)
        if (a.cone_type[c] != b.cone_type[c] || a.cone_dim[c] != b.cone_dim[c]) return false;
    // CSR A: the slab layout depends on the number of non-zeros -- part of the shape when the row pointers can be read here
    // (host memory); device-resident row pointers are caught by the slab-layout check of lockstep_group instead
    auto host_rowptr = [](const cip_problem &q) {
        return q.A == nullptr && q.m > 0 && q.A_rowptr && ((q.flags & CIP_FLAG_CSR_HOST) || !(q.flags & CIP_FLAG_DEVICE_PTRS));
    };
    if (host_rowptr(a) && host_rowptr(b) && a.A_rowptr[a.m] != b.A_rowptr[b.m]) return false;
    return true;
}

// Up to four arenas are kept between calls (a bench or a service solves batch after batch of the same shape, possibly from several
// host threads at once; hipMalloc / hipFree of several GB cost up to 0.6 s and synchronise the device).
// cip_release_cached_memory() frees them.
struct ArenaCache {
    std::mutex mu;
    struct Slot { char *ptr; size_t bytes; int device; };
    std::vector<Slot> slots;
    // last probe: shape signature -> slab bytes
    std::vector<long> sig; size_t slab = 0;
} g_cache;
// *cap: the allocation's true size (a recycled arena may be larger than asked for; it goes back into the cache with that size)
int arena_acquire(size_t bytes, char **out, size_t *cap) {
    int dev = 0;
    CIP_HIP_CHECK(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        int best = -1;                                       // the smallest cached arena that fits
        for (int i = 0; i < (int)g_cache.slots.size(); ++i) {
            const auto &sl = g_cache.slots[i];
            if (sl.device == dev && sl.bytes >= bytes && (best < 0 || sl.bytes < g_cache.slots[best].bytes)) best = i;
        }
        if (best >= 0) {
            *out = g_cache.slots[best].ptr; *cap = g_cache.slots[best].bytes;
            g_cache.slots.erase(g_cache.slots.begin() + best);
            return 0;
        }
        // nothing fits: the cached arenas of this device make room (a larger batch follows a smaller one)
        for (int i = (int)g_cache.slots.size() - 1; i >= 0; --i)
            if (g_cache.slots[i].device == dev) { (void)hipFree(g_cache.slots[i].ptr); g_cache.slots.erase(g_cache.slots.begin() + i); }
    }
    CIP_HIP_CHECK(hipMalloc((void **)out, bytes));
    *cap = bytes;
    return 0;
}
void arena_release(char *ptr, size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_cache.mu);
    if (g_cache.slots.size() >= 4) {                         // full: the smallest one goes
        int small = 0;
        for (int i = 1; i < (int)g_cache.slots.size(); ++i) if (g_cache.slots[i].bytes < g_cache.slots[small].bytes) small = i;
        (void)hipFree(g_cache.slots[small].ptr);
        g_cache.slots.erase(g_cache.slots.begin() + small);
    }
    g_cache.slots.push_back({ptr, bytes, dev});
}
std::vector<long> shape_signature(const cip_problem &pr, int solve_block) {
    std::vector<long> sg = {pr.n, pr.m, pr.p, pr.ncones, pr.route, pr.A == nullptr, solve_block, cip_ldlt_outer_block()};
    for (int c = 0; c < pr.ncones; ++c) { sg.push_back(pr.cone_type[c]); sg.push_back(pr.cone_dim[c]); }
    return sg;
}

struct Group {
    int B = 0;
    char *arena = nullptr; size_t arena_bytes = 0;
    size_t stride = 0;
    double *gather_dev = nullptr, *gather_host = nullptr;
    hipStream_t stream = nullptr;
    std::vector<cip_handle *> h;
    ~Group() {
        for (cip_handle *x : h) if (x) cip_destroy(x);
        if (stream) (void)hipStreamDestroy(stream);
        if (arena) arena_release(arena, arena_bytes);      // the handles' stream was drained by cip_destroy
        if (gather_host) (void)hipHostFree(gather_host);
    }
};

thread_local int g_last_stats[3] = {0, 0, 0};      // groups, problems, problems that left their group (last call of this thread)
unsigned long long full_mask(int B) { return B >= 64 ? ~0ull : ((1ull << B) - 1ull); }

}   // namespace

// Solve-block limit of a lock-step call's handles, chosen from the size of the WHOLE call (count), not per group of 64: every
// problem of one cip_conicip_lockstep call is solved with the same block, so a problem's bits do not depend on which group
// it lands in (72 problems = 64 + 8: both groups take 256).  Never above the process-wide limit.
extern "C" int cip_lockstep_solve_block_for(int B) {
    // groups of up to 8 problems: 512 (the sweeps are launch chains on a mostly idle chip: half the block steps; 8 problems of
    // order 2048, 256 / 512 / 1024: 16.75 / 16.55 / 17.8 ms per pass); larger groups: 256 (the doubling GEMMs of a wider
    // block are real work for 64 problems).  CIP_LOCKSTEP_SOLVE_BLOCK overrides both.
    const char *e = getenv("CIP_LOCKSTEP_SOLVE_BLOCK");
    const int want = e ? atoi(e) : (B <= 8 ? 512 : 256), glob = cip_solve_block_max_set(0);
    return want < glob ? want : glob;
}
// One lock-step group (B <= CIP_BATCH_MAX problems of the same shape).  Returns 0 and fills res / y / w / v of every
// problem, or an error code (nothing meaningful written).
static int lockstep_group(int B, int call_count, const cip_problem *probs, const double *const *c, const double *const *b,
                          const double *const *d, const cip_options *opt_in, double *const *y, double *const *w,
                          double *const *v, cip_result *res) {
    const auto t_start = std::chrono::steady_clock::now();
    static const bool timing = getenv("CIP_LOCKSTEP_TIMING") != nullptr;
    auto since = [&]() { return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    double t_probe = 0, t_arena = 0, t_create = 0, t_loop = 0;
    const cip_options o = resolve_options(opt_in);
    const int n = probs[0].n, m = probs[0].m, p = probs[0].p;
    struct ExitTimer {            // reports what the destructors behind it (handles, arena) cost
        bool on; std::chrono::steady_clock::time_point t0;
        ~ExitTimer() { if (on) fprintf(stderr, "lockstep group: tear-down %.2f ms\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
    };
    ExitTimer exit_timer{false, std::chrono::steady_clock::now()};
    Group G;
    G.B = B;
    int rc;
    CIP_HIP_CHECK(hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking));
    // solve-block limit of the group's handles (see ldlt.hip: cip_solve_block)
    struct SolveBlockScope {
        int saved;
        explicit SolveBlockScope(int B_) : saved(cip_tl_solve_block_max) { cip_tl_solve_block_max = cip_lockstep_solve_block_for(B_); }
        ~SolveBlockScope() { cip_tl_solve_block_max = saved; }
    } solve_block_scope(call_count);
    // ---- slab size: create problem 0 once with ordinary allocations and count what it asked for
    size_t slab = 0;
    // CSR: the slab depends on the number of non-zeros -- part of the signature when the row pointers are host memory
    // (the probe is a full handle creation: 0.9 ms per call, 4 % of an 8-problem pass), else always probe
    const bool csr = probs[0].A == nullptr && m > 0;
    const bool csr_host = csr && probs[0].A_rowptr && ((probs[0].flags & CIP_FLAG_CSR_HOST) || !(probs[0].flags & CIP_FLAG_DEVICE_PTRS));
    std::vector<long> sig = shape_signature(probs[0], cip_tl_solve_block_max);
    if (csr_host) sig.push_back((long)probs[0].A_rowptr[m]);
    if (!csr || csr_host) {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        if (g_cache.sig == sig) slab = g_cache.slab;
    }
    if (slab == 0) {
        cip_handle *probe = nullptr;
        if ((rc = cip_create_ex(&probs[0], &probe))) return rc;
        slab = probe->alloc_bytes + ((cip_driver_bytes(probe) + 255) & ~(size_t)255);
        const bool large_S = probe->cs.nlarge > 0;
        cip_destroy(probe);
        if (large_S) { cip_set_error("lock-step batch: S cones of order >= %d are not supported", CIP_LARGE_S_MIN); return CIP_E_UNSUPPORTED; }
        std::lock_guard<std::mutex> lk(g_cache.mu);
        g_cache.sig = sig; g_cache.slab = slab;
    }
    t_probe = since();
    // odd number of 256-byte granules: the same buffer of consecutive problems does not land on the same HBM channel
    size_t gran = (slab + 255) / 256;
    if ((gran & 1) == 0) ++gran;
    G.stride = gran * 256;
    const size_t gather_bytes = sizeof(double) * (size_t)B * CIP_GATHER;
    G.arena_bytes = G.stride * (size_t)B + gather_bytes;
    if ((rc = arena_acquire(G.arena_bytes, &G.arena, &G.arena_bytes))) return rc;
    G.gather_dev = (double *)(G.arena + G.stride * (size_t)B);
    CIP_HIP_CHECK(hipHostMalloc((void **)&G.gather_host, gather_bytes, hipHostMallocDefault));
    G.h.assign(B, nullptr);
    t_arena = since();
    for (int z = 0; z < B; ++z) {
        if ((rc = cip_create_in_arena(&probs[z], G.arena + G.stride * (size_t)z, G.stride, G.stream, &G.h[z]))) return rc;
        cip_handle *hz = G.h[z];
        void *drv = nullptr;
        if ((rc = cip_handle_alloc(hz, &drv, cip_driver_bytes(hz)))) return rc;
        hz->drv = (double *)drv;
        if (hz->arena_overflow || (z > 0 && hz->arena_used != G.h[0]->arena_used)) {
            cip_set_error("lock-step batch: problem %d does not fit problem 0's slab layout", z);
            return CIP_E_UNSUPPORTED;
        }
    }
    t_create = since();
    cip_handle *h = G.h[0];
    hipStream_t s = G.stream;
    const int NT = n + p + 2 * m;
    Vectors V;
    V.carve(h->drv, n, m, p);
    double *c_d = V.c_d, *b_d = V.b_d, *d_d = V.d_d;
    Vec4 &zv = V.z, &r0 = V.r0, &rleft = V.rleft, &r = V.r, &daff = V.daff, &dz = V.dz, &dzr = V.dzr, &rIr = V.rIr, &rkkt = V.rkkt;
    double *e = V.e, *lam = V.lam, *mb1 = V.mb1, *mb2 = V.mb2, *mb3 = V.mb3;
    double *Qy = V.Qy, *pinf = V.pinf, *Ays = V.Ays, *Gy = V.Gy;
    const double conedim = cone_degree(h);
    const double *f = cip_loop_all_r(h);                   // diag F of problem 0 (the kernels shift it per problem) when every cone is an R cone

    // ---- per-problem host state
    std::vector<Norms> nm(B);
    std::vector<double> optBest(B, INFINITY);
    std::vector<int> n_factor(B, 0), n_solve(B, 0);
    std::vector<IterOutcome> outcome(B);
    std::vector<char> ejected(B, 0);
    for (int z = 0; z < B; ++z) {
        res[z] = cip_result{};
        res[z].prFeas = res[z].duFeas = res[z].muFeas = INFINITY; res[z].pobj = INFINITY; res[z].dobj = -INFINITY;
        nm[z] = host_norms(n, m, p, c[z], m > 0 ? b[z] : nullptr, p > 0 ? d[z] : nullptr);
    }

    CipBatchCtx ctx = {B, (long)G.stride, full_mask(B), G.gather_dev, G.gather_host};
    BatchScope scope(ctx);
    unsigned long long active = full_mask(B);
    auto set_mask = [&](unsigned long long mk) { cip_tl_bz.mask = mk; };
#define CK(x) do { if ((rc = (x)) != 0) return rc; } while (0)
    auto axpby = [&](int len, double alpha, const double *x, double beta, double *yy) { return len > 0 ? cip_axpby(s, len, alpha, x, beta, yy) : 0; };
    auto copy = [&](int len, const double *x, double *yy) { return axpby(len, 1.0, x, 0.0, yy); };
    // assembly + LDL' of every problem of the current mask; the pivot flags come back with the next read-back
    auto factor = [&]() -> int {
        int e2;
        h->reg_rel = 0.0;
        if ((e2 = cip_assemble(h, true))) return e2;
        if ((e2 = cip_ldlt_factor(s, h->K, h->Npad, h->ldk, h->ws))) return e2;
        h->factored = true; h->info_pending = false;
        for (int z = 0; z < B; ++z) if ((cip_tl_bz.mask >> z) & 1ull) ++n_factor[z];
        return 0;
    };
    // pivot flags of the last factorisation: gather_info enqueues their copy into the gather buffer (launched under the mask
    // the factorisation ran under; the words then ride on the next read-back of that buffer, normally the dots'), and
    // eject_bad_pivots takes the problems that met a bad pivot out of the group (read_back: fetch the buffer itself)
    auto gather_info = [&]() -> int {
        cip_launch_b(k_gather_info, dim3(1), dim3(64), 0, s, (const int *)h->ws.info, G.gather_dev);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    };
    auto eject_bad_pivots = [&](bool read_back, unsigned long long fmask) -> int {
        if (read_back) {
            CIP_HIP_CHECK(hipMemcpyAsync(G.gather_host, G.gather_dev, gather_bytes, hipMemcpyDeviceToHost, s));
            { const int rcw = cip_wait(s); if (rcw) return rcw; }
        }
        for (int z = 0; z < B; ++z) {
            if (!((fmask >> z) & 1ull)) continue;
            const double *gi = G.gather_host + (size_t)z * CIP_GATHER + INFO_SLOT;
            if (gi[1] != 0.0) { cip_set_error("LDL': a triangular sweep bailed out (problem %d)", z); return CIP_E_HIP; }
            // gi[3]: an in-launch wait of the fused panel launch gave up -- a GPU shared with other processes can keep a launch's
            // workgroups off the chip for longer than the bound.  The problem leaves the group like one with a bad pivot and is
            // solved alone afterwards, on the three-launch chain (no in-launch wait, same bits)
            if (gi[3] != 0.0) { G.h[z]->ws.unfused = 1; G.h[z]->n_chain_fallbacks += 1; h->ws.unfused = 1; }      // (h: the group's launches follow problem 0's workspace -- three launches per panel from here on)
            if (gi[0] != 0.0 || gi[3] != 0.0) { ejected[z] = 1; active &= ~(1ull << z); }
        }
        return 0;
    };
    std::vector<double> av(B), as(B), tmpB(B);
    // A max-step whose minima ride on the next read-back of the gather buffer (slot `slot` of every problem's row).  A group of
    // ONE problem is not a batch for the kernels (cip_tl_bz.B == 1: no gather buffer behind k_min_reduce), so it takes the
    // one-problem read-back and `direct[0]` holds the result on return; fetch_step() is then a no-op for that slot.
    auto maxstep_deferred = [&](const double *x, const double *dd, double scale, int slot, double *direct) -> int {
        if (B == 1) return cip_cones_maxstep(s, h->cs, x, dd, scale, direct);
        return cip_cones_maxstep(s, h->cs, x, dd, scale, nullptr, slot);
    };
    auto fetch_step = [&](int slot, std::vector<double> &dst) {
        if (B == 1) return;
        for (int z = 0; z < B; ++z) dst[z] = G.gather_host[(size_t)z * CIP_GATHER + slot];
    };

    // ---------------------------------------------------------------- initial point (:704-713)
    CK(cip_zero(s, (long)driver_doubles(n, m, p), h->drv));
    // the right-hand sides: problem z's vectors are problem 0's addresses + z * stride
    for (int z = 0; z < B; ++z) {
        const size_t off = G.stride * (size_t)z;
        CIP_HIP_CHECK(hipMemcpyAsync((char *)c_d + off, c[z], sizeof(double) * n, hipMemcpyHostToDevice, s));
        if (m > 0) CIP_HIP_CHECK(hipMemcpyAsync((char *)b_d + off, b[z], sizeof(double) * m, hipMemcpyHostToDevice, s));
        if (p > 0) CIP_HIP_CHECK(hipMemcpyAsync((char *)d_d + off, d[z], sizeof(double) * p, hipMemcpyHostToDevice, s));
    }
    if (m > 0) CK(cip_cones_identity(s, h->cs, e));
    set_mask(active);
    CK(cip_set_scaling_identity(h));
    CK(factor());
    CK(copy(n, c_d, r0.y)); CK(copy(p, d_d, r0.w)); CK(copy(m, b_d, r0.v));
    if (m > 0) CK(cip_zero(s, m, r0.s));
    CK(gather_info());
    CK(eject_bad_pivots(true, active));
    set_mask(active);
    if (active) {
        CK(cip_solve4x4_dev(h, e, r0.base, zv.base));
        for (int z = 0; z < B; ++z) if ((active >> z) & 1ull) ++n_solve[z];
        if (m > 0) {
            CK(maxstep_deferred(zv.v, nullptr, 1.0, STEP_SLOT, av.data()));
            CK(cip_cones_maxstep(s, h->cs, zv.s, nullptr, 1.0, as.data()));
            fetch_step(STEP_SLOT, av);
            for (int z = 0; z < B; ++z) { av[z] = -av[z]; as[z] = -as[z]; }
            CK(cip_axpby_ps(s, m, av.data(), e, 1.0, zv.v));
            CK(cip_axpby_ps(s, m, as.data(), e, 1.0, zv.s));
        }
    }

    std::vector<double> dt((size_t)B * 16), q4((size_t)B * 4), n2((size_t)B * 4), sigma(B), mu(B), mubar(B), alpha(B);
    int Iter = 1;
    for (; Iter <= o.maxIters && active; ++Iter) {                                         // :730
        set_mask(active);
        if (m > 0) CK(cip_cones_nt_scaling(s, h->cs, zv.v, zv.s, lam));                    // :732-735
        h->assembled = h->factored = false;
        CK(factor());                                                                      // :737 -> :682
        const unsigned long long factored_mask = active;
        // (round 5: the element-wise chains are the one-problem loop's fused kernels, driver.hip / vecops.hip: k_loop_*)
        if (m > 0 && !f) CK(cip_cones_prod(s, h->cs, lam, lam, rleft.s));                  // :746
        CK(cip_gemv_dev(h, CIP_MAT_Q, 0, 1.0, zv.y, 0.0, Qy));
        CK(copy(n, Qy, rleft.y));                                                          // :747-750
        if (p > 0) {
            CK(cip_gemv_dev(h, CIP_MAT_G, 1, 1.0, zv.w, 1.0, rleft.y));
            CK(cip_gemv_dev(h, CIP_MAT_G, 0, 1.0, zv.y, 0.0, rleft.w));
            CK(cip_gemv_dev(h, CIP_MAT_G, 1, 1.0, zv.w, 0.0, pinf));
        }
        if (m > 0) {
            CK(cip_gemv_dev(h, CIP_MAT_A, 1, -1.0, zv.v, 1.0, rleft.y));
            CK(cip_gemv_dev(h, CIP_MAT_A, 0, 1.0, zv.y, 0.0, rleft.v));
            CK(cip_gemv_dev(h, CIP_MAT_A, 1, -1.0, zv.v, p > 0 ? 1.0 : 0.0, pinf));
        } else if (p == 0) CK(cip_zero(s, n, pinf));
        CK(cip_loop_resid(s, n, m, p, rleft.base, zv.s, c_d, d_d, b_d, lam, f, r0.base, Gy, Ays));       // :753
        const double *px[16] = {zv.v, c_d, r0.y, r0.v, r0.s, zv.y, zv.w, zv.v, d_d, b_d, pinf, zv.y, zv.v, Ays, Gy, Qy};
        const double *py[16] = {zv.s, zv.y, r0.y, r0.v, r0.s, Qy, r0.w, r0.v, zv.w, zv.v, pinf, zv.y, zv.v, Ays, Gy, Qy};
        const int ln[16] = {m, n, n, m, m, n, p, m, p, m, n, n, m, m, p, n};
        CK(gather_info());                    // same mask as the factorisation; rides on the dots' read-back
        CK(cip_dots(s, 16, px, py, ln, h->dot_scratch, h->dot_ptrs, dt.data()));
        if (B == 1) CK(eject_bad_pivots(true, factored_mask));      // a group of one takes the one-problem read-back path
        else CK(eject_bad_pivots(false, factored_mask));
        for (int z = 0; z < B; ++z) {
            if (!((active >> z) & 1ull)) continue;
            IterDots dd;
            for (int i = 0; i < 16; ++i) dd.v[i] = dt[(size_t)z * 16 + i];
            outcome[z] = evaluate_iteration(dd, nm[z], conedim, m, p, o, Iter, &res[z], optBest[z], nullptr);
            mu[z] = outcome[z].mu; mubar[z] = outcome[z].mubar;
            if (outcome[z].status != CIP_STATUS_NONE) { res[z].status = outcome[z].status; active &= ~(1ull << z); }
        }
        if (!active) break;
        set_mask(active);

        // ------------------------------------------------------------ predictor (:879-887)
        CK(cip_solve4x4_dev(h, lam, r0.base, daff.base));
        for (int z = 0; z < B; ++z) if ((active >> z) & 1ull) ++n_solve[z];
        for (int z = 0; z < B; ++z) sigma[z] = 0.0;
        if (m > 0) {
            // one host round trip for the three results (round 4; it was three): the two max-steps leave their minima in slots
            // STEP_SLOT, STEP_SLOT + 1 of the gather buffer, which comes back with the dot products
            CK(maxstep_deferred(zv.v, daff.v, 1.0, STEP_SLOT, av.data()));
            CK(maxstep_deferred(zv.s, daff.s, 1.0, STEP_SLOT + 1, as.data()));
            const double *qx[4] = {zv.v, zv.v, daff.v, daff.v};
            const double *qy[4] = {zv.s, daff.s, zv.s, daff.s};
            const int ql[4] = {m, m, m, m};
            CK(cip_dots(s, 4, qx, qy, ql, h->dot_scratch, h->dot_ptrs, q4.data()));
            fetch_step(STEP_SLOT, av);
            fetch_step(STEP_SLOT + 1, as);
            for (int z = 0; z < B; ++z) {
                if (!((active >> z) & 1ull)) continue;
                const double a_aff = std::fmin(std::fmin(av[z], 1.0), as[z]);
                const double *q = &q4[(size_t)z * 4];
                const double rho = (q[0] - a_aff * q[1] - a_aff * q[2] + a_aff * a_aff * q[3]) / mubar[z];   // fts :162-163, :886
                const double cl = std::fmax(0.0, std::fmin(1.0, rho));
                sigma[z] = std::pow(cl, 3.0);
            }
        }

        // ------------------------------------------------------------ corrector (:893-901)
        if (m > 0 && !f) {
            CK(cip_cones_apply(s, h->cs, CIP_OP_FINVT, daff.s, mb1));
            CK(cip_cones_apply(s, h->cs, CIP_OP_F, daff.v, mb2));
            CK(cip_cones_prod(s, h->cs, mb1, mb2, mb3));
        }
        if (m > 0) {
            for (int z = 0; z < B; ++z) tmpB[z] = sigma[z] * mu[z];
            CK(cip_loop_corr(s, n, m, p, r0.base, daff.base, mb3, e, f, tmpB.data(), r.base));
        } else CK(copy(NT, r0.base, r.base));

        // ------------------------------------------------------------ Newton step + refinement (:907-921)
        CK(cip_solve4x4_dev(h, lam, r.base, dz.base));
        for (int z = 0; z < B; ++z) if ((active >> z) & 1ull) ++n_solve[z];
        unsigned long long refine = active;
        bool step_known = false;
        for (int it = 0; it < o.maxRefinementSteps && refine; ++it) {
            set_mask(refine);
            CK(cip_gemv_dev(h, CIP_MAT_Q, 0, 1.0, dz.y, 0.0, rkkt.y));
            if (p > 0) {
                CK(cip_gemv_dev(h, CIP_MAT_G, 1, 1.0, dz.w, 1.0, rkkt.y));
                CK(cip_gemv_dev(h, CIP_MAT_G, 0, 1.0, dz.y, 0.0, rkkt.w));
            }
            if (m > 0) {
                CK(cip_gemv_dev(h, CIP_MAT_A, 1, -1.0, dz.v, 1.0, rkkt.y));
                CK(cip_gemv_dev(h, CIP_MAT_A, 0, 1.0, dz.y, 0.0, rkkt.v));
                if (!f) {
                    CK(cip_cones_apply(s, h->cs, CIP_OP_F, dz.v, mb1));
                    CK(cip_cones_prod(s, h->cs, lam, mb1, mb2));
                    CK(cip_cones_apply(s, h->cs, CIP_OP_FINVT, dz.s, mb1));
                    CK(cip_cones_prod(s, h->cs, lam, mb1, mb3));
                }
            }
            CK(cip_loop_refine(s, n, m, p, rkkt.base, dz.base, r.base, lam, mb2, mb3, f, rIr.base));
            const double *nx[4] = {rIr.y, rIr.w, rIr.v, rIr.s};
            const int nl[4] = {n, p, m, m};
            // the step's two max-steps ride on this read-back (first pass only): when no problem asks for refinement -- the usual
            // case -- dz is final and the iteration has saved a host round trip; otherwise they are taken again behind the loop
            static const int spec_on = [] { const char *e = getenv("CIP_LOCKSTEP_SPEC_STEP"); return e ? atoi(e) : 1; }();
            const bool spec = spec_on && it == 0 && m > 0 && B > 1;
            if (spec) {
                CK(cip_cones_maxstep(s, h->cs, zv.v, dz.v, 1.0 / (1.0 - o.DTB), nullptr, STEP_SLOT));
                CK(cip_cones_maxstep(s, h->cs, zv.s, dz.s, 1.0 / (1.0 - o.DTB), nullptr, STEP_SLOT + 1));
            }
            CK(cip_dots(s, 4, nx, nx, nl, h->dot_scratch, h->dot_ptrs, n2.data()));
            for (int z = 0; z < B; ++z) {
                if (!((refine >> z) & 1ull)) continue;
                const double *q = &n2[(size_t)z * 4];
                const double rnorm = (nrm(q[0]) + (p > 0 ? nrm(q[1]) : 0.0) + (m > 0 ? nrm(q[2]) + nrm(q[3]) : 0.0)) / (n + 2 * m);   // :917
                if (rnorm < o.refinementThreshold) refine &= ~(1ull << z);
            }
            if (spec && !refine) {
                for (int z = 0; z < B; ++z) {
                    av[z] = G.gather_host[(size_t)z * CIP_GATHER + STEP_SLOT];
                    as[z] = G.gather_host[(size_t)z * CIP_GATHER + STEP_SLOT + 1];
                }
                step_known = true;
            }
            if (!refine) break;
            set_mask(refine);
            CK(cip_solve4x4_dev(h, lam, rIr.base, dzr.base));
            for (int z = 0; z < B; ++z) if ((refine >> z) & 1ull) ++n_solve[z];
            CK(axpby(NT, 1.0, dzr.base, 1.0, dz.base));                                    // :920
        }
        set_mask(active);

        // ------------------------------------------------------------ step (:927-932)
        for (int z = 0; z < B; ++z) alpha[z] = 1.0;
        if (m > 0) {
            if (!step_known) {
                // (one round trip for the pair: the v side rides on the s side's read-back)
                CK(maxstep_deferred(zv.v, dz.v, 1.0 / (1.0 - o.DTB), STEP_SLOT, av.data()));
                CK(cip_cones_maxstep(s, h->cs, zv.s, dz.s, 1.0 / (1.0 - o.DTB), as.data()));
                fetch_step(STEP_SLOT, av);
            }
            for (int z = 0; z < B; ++z) alpha[z] = std::fmin(std::fmin(av[z], 1.0), std::fmin(as[z], 1.0));
        }
        for (int z = 0; z < B; ++z) tmpB[z] = -alpha[z];
        CK(cip_axpby_ps(s, NT, tmpB.data(), dz.base, 1.0, zv.base));
    }
    // problems still active ran out of iterations (:936)
    for (int z = 0; z < B; ++z)
        if ((active >> z) & 1ull) { res[z].status = CIP_STATUS_ABANDONED; outcome[z] = IterOutcome{}; outcome[z].status = CIP_STATUS_ABANDONED; }

    if (timing) { (void)hipStreamSynchronize(s); t_loop = since(); }
    // ---- results of the lock-step problems
    for (int z = 0; z < B; ++z) {
        if (ejected[z]) continue;
        const size_t off = G.stride * (size_t)z;
        CIP_HIP_CHECK(hipMemcpyAsync(y[z], (char *)zv.y + off, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        if (p > 0) CIP_HIP_CHECK(hipMemcpyAsync(w[z], (char *)zv.w + off, sizeof(double) * p, hipMemcpyDeviceToHost, s));
        if (m > 0) CIP_HIP_CHECK(hipMemcpyAsync(v[z], (char *)zv.v + off, sizeof(double) * m, hipMemcpyDeviceToHost, s));
    }
    CIP_HIP_CHECK(hipStreamSynchronize(s));
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (timing)
        fprintf(stderr, "lockstep group B=%d n=%d: probe %.2f ms, arena %.2f, create %.2f, loop %.2f (%d iterations), download %.2f; slab %.1f MB\n",
                B, n, t_probe, t_arena - t_probe, t_create - t_arena, t_loop - t_create, Iter, 1e3 * wall - t_loop, G.stride / 1048576.0);
    for (int z = 0; z < B; ++z) {
        if (ejected[z]) continue;
        apply_certificate(outcome[z], n, m, p, y[z], p > 0 ? w[z] : nullptr, m > 0 ? v[z] : nullptr);
        res[z].n_factor = n_factor[z]; res[z].n_solve = n_solve[z];
        res[z].wall_s = wall;            // the group's wall time: the problems finished together
    }
#undef CK
    g_last_stats[0] += 1; g_last_stats[1] += B;
    for (int z = 0; z < B; ++z) g_last_stats[2] += ejected[z] ? 1 : 0;
    exit_timer.on = timing; exit_timer.t0 = std::chrono::steady_clock::now();
    // ---- problems that left the group: the one-problem loop on their own handle (regularised factorisation and all)
    {
        BatchScope single(CipBatchCtx{1, 0, 1ull, nullptr, nullptr});
        for (int z = 0; z < B; ++z) {
            if (!ejected[z]) continue;
            cip_handle *hz = G.h[z];
            hz->assembled = hz->factored = false; hz->info_pending = false; hz->reg_rel = 0.0; hz->n_regularized = 0;
            if ((rc = cip_conicip(hz, c[z], m > 0 ? b[z] : nullptr, p > 0 ? d[z] : nullptr, opt_in, y[z], p > 0 ? w[z] : nullptr,
                                  m > 0 ? v[z] : nullptr, &res[z], nullptr, 0)))
                return rc;
        }
    }
    return 0;
}

// groups of a large lock-step call that run side by side (see cip_conicip_lockstep); k < 1 only reads; returns the previous value
static std::atomic<int> g_lockstep_split{-1};
int cip_lockstep_split_set(int k) {
    if (g_lockstep_split.load() < 0) {
        const char *e = getenv("CIP_LOCKSTEP_SPLIT");
        int v = -1, want = e ? atoi(e) : CIP_LOCKSTEP_SPLIT_DEFAULT;
        want = want < 1 ? 1 : (want > 8 ? 8 : want);
        g_lockstep_split.compare_exchange_strong(v, want);
    }
    const int prev = g_lockstep_split.load();
    if (k >= 1 && k <= 8) g_lockstep_split.store(k);
    return prev;
}
extern "C" int cip_set_lockstep_split(int k) { return cip_lockstep_split_set(k); }

// Problems in, solutions out, in lock-step groups of up to 64.  CIP_E_UNSUPPORTED (nothing written): the problems differ
// in shape or hold S cones -- use cip_conicip_problems.
extern "C" int cip_conicip_lockstep(int count, const cip_problem *probs, const double *const *c, const double *const *b,
                                    const double *const *d, const cip_options *opt, double *const *y, double *const *w,
                                    double *const *v, cip_result *res) {
    if (count < 0 || (count > 0 && (!probs || !c || !y || !res))) { cip_set_error("cip_conicip_lockstep: null argument"); return CIP_E_INVALID; }
    if (count == 0) return 0;
    if ((probs[0].m > 0 && (!b || !v)) || (probs[0].p > 0 && (!d || !w))) { cip_set_error("cip_conicip_lockstep: null argument"); return CIP_E_INVALID; }
    for (int i = 1; i < count; ++i)
        if (!same_shape(probs[0], probs[i])) { cip_set_error("lock-step batch: problem %d differs in shape from problem 0", i); return CIP_E_UNSUPPORTED; }
    for (int c0 = 0; c0 < probs[0].ncones; ++c0)       // S cones of order >= 133 take chip-wide kernels with a single workspace (sdp_large.hip)
        if (probs[0].cone_type[c0] == CIP_CONE_S && probs[0].cone_dim[c0] >= CIP_LARGE_S_MIN * (CIP_LARGE_S_MIN + 1) / 2) {
            cip_set_error("lock-step batch: S cones of order >= %d are not supported", CIP_LARGE_S_MIN);
            return CIP_E_UNSUPPORTED;
        }
    if (cip_tl_builder) { cip_set_error("lock-step batch inside a graph recording"); return CIP_E_INVALID; }
    if (!g_stats_accumulate) g_last_stats[0] = g_last_stats[1] = g_last_stats[2] = 0;
    auto range = [&](int i0, int i1) -> int {                // groups of up to 64 over [i0, i1), one after the other, on the calling thread
        for (int g0 = i0; g0 < i1; g0 += CIP_BATCH_MAX) {
            const int B = (i1 - g0 < CIP_BATCH_MAX) ? (i1 - g0) : CIP_BATCH_MAX;
            const int rc = lockstep_group(B, count, probs + g0, c + g0, b ? b + g0 : nullptr, d ? d + g0 : nullptr, opt, y + g0,
                                          w ? w + g0 : nullptr, v ? v + g0 : nullptr, res + g0);
            if (rc) return rc;
        }
        return 0;
    };
    // Round 6: a large call as TWO (CIP_LOCKSTEP_SPLIT = k: k) lock-step groups side by side, each driven by its own host thread on its
    // own stream.  A group's loop alternates latency-bound launch chains (the panel chain, the triangular sweeps: most of the chip idle)
    // with throughput-bound ones (trailing updates, symv) and three host round trips per iteration; two groups fill each other's
    // gaps.  Measured first with two PROCESSES sharing one GPU (round 5: 64 problems of order 2048, 8819 against 7971 KKT solves/s for
    // one rank); this is the same overlap inside one process.  Per problem nothing changes: the solve block follows the size of the
    // whole call, every kernel is the group-size-independent code the bit-identity tests pin (tests/test_gpu_lockstep.py).
    // Measured (profiles/r6/lockstep_split.txt, problems of order 2048, ms per pass, one group -> two side by side): 64 problems 77.3-79.4 ->
    // 72.4-74.4 (8000 -> 8500-8650 KKT solves/s), 32: 44.9 -> 40.4-41.5, 24: 32.5 -> 30.8-31.7, 16: 25.2 -> 23.7-24.4; 8 problems as
    // 2 x 4 LOSE (15.3 -> 15.5-16.9: each panel launch owns whole CUs -- 160 KB of LDS per workgroup -- and two latency-bound chains only
    // get in each other's way), as do three or four groups (64 as 4 x 16: 77.2-78.8).  Hence: two groups, none smaller than 8.
    int nsplit = cip_lockstep_split_set(0);
    static const int split_min = [] { const char *e = getenv("CIP_LOCKSTEP_SPLIT_MIN"); const int k = e ? atoi(e) : CIP_LOCKSTEP_SPLIT_MIN_DEFAULT; return k < 1 ? 1 : k; }();
    while (nsplit > 1 && count / nsplit < split_min) --nsplit;
    if (nsplit <= 1) return range(0, count);
    int device = 0;
    CIP_HIP_CHECK(hipGetDevice(&device));
    std::vector<int> rcs(nsplit, 0);
    std::vector<std::string> errs(nsplit);
    std::vector<int> st(3 * (size_t)nsplit, 0);
    std::vector<std::thread> th;
    const int per = (count + nsplit - 1) / nsplit;
    for (int t = 0; t < nsplit; ++t)
        th.emplace_back([&, t] {
            (void)hipSetDevice(device);
            g_last_stats[0] = g_last_stats[1] = g_last_stats[2] = 0;
            const int i0 = t * per, i1 = (t + 1) * per < count ? (t + 1) * per : count;
            rcs[t] = i0 < i1 ? range(i0, i1) : 0;
            if (rcs[t]) errs[t] = cip_last_error();
            for (int q = 0; q < 3; ++q) st[3 * t + q] = g_last_stats[q];
        });
    for (auto &x : th) x.join();
    for (int t = 0; t < nsplit; ++t)
        for (int q = 0; q < 3; ++q) g_last_stats[q] += st[3 * t + q];
    for (int t = 0; t < nsplit; ++t)
        if (rcs[t]) { cip_set_error("%s", errs[t].c_str()); return rcs[t]; }
    return 0;
}

// Any mix of problems.  The ones that share a shape with at least one other problem of the batch (and qualify for
// lock-step: no chip-wide S cone) advance together, shape group by shape group in order of first appearance; the rest go
// through the thread pool (`in_flight` threads).  Results per problem are those of cip_conicip_lockstep / cip_conicip_problems.
extern "C" int cip_conicip_mixed(int count, const cip_problem *probs, const double *const *c, const double *const *b,
                                 const double *const *d, const cip_options *opt, double *const *y, double *const *w,
                                 double *const *v, cip_result *res, int in_flight) {
    if (count < 0 || (count > 0 && (!probs || !c || !y || !res))) { cip_set_error("cip_conicip_mixed: null argument"); return CIP_E_INVALID; }
    if (count == 0) return 0;
    auto lockstep_ok = [](const cip_problem &q) {
        for (int c0 = 0; c0 < q.ncones; ++c0)
            if (q.cone_type[c0] == CIP_CONE_S && q.cone_dim[c0] >= CIP_LARGE_S_MIN * (CIP_LARGE_S_MIN + 1) / 2) return false;
        return true;
    };
    std::vector<std::vector<int>> bins;                         // bins[k][0] is the representative
    for (int i = 0; i < count; ++i) {
        if ((probs[i].m > 0 && (!b || !v || !b[i] || !v[i])) || (probs[i].p > 0 && (!d || !w || !d[i] || !w[i])) || !c[i] || !y[i]) {
            cip_set_error("cip_conicip_mixed: null vector for problem %d", i);
            return CIP_E_INVALID;
        }
        size_t k = 0;
        for (; k < bins.size(); ++k)
            if (same_shape(probs[bins[k][0]], probs[i])) break;
        if (k == bins.size()) bins.emplace_back();
        bins[k].push_back(i);
    }
    g_last_stats[0] = g_last_stats[1] = g_last_stats[2] = 0;
    struct Acc { Acc() { g_stats_accumulate = true; } ~Acc() { g_stats_accumulate = false; } } acc;
    std::vector<int> rest;
    auto run = [&](const std::vector<int> &idx, bool lock) -> int {
        const size_t k = idx.size();
        std::vector<cip_problem> gp(k);
        std::vector<const double *> gc(k), gb(k), gd(k);
        std::vector<double *> gy(k), gw(k), gv(k);
        std::vector<cip_result> gr(k);
        for (size_t j = 0; j < k; ++j) {
            const int i = idx[j];
            gp[j] = probs[i]; gc[j] = c[i]; gy[j] = y[i];
            gb[j] = b ? b[i] : nullptr; gv[j] = v ? v[i] : nullptr;
            gd[j] = d ? d[i] : nullptr; gw[j] = w ? w[i] : nullptr;
        }
        const int rc = lock ? cip_conicip_lockstep((int)k, gp.data(), gc.data(), gb.data(), gd.data(), opt, gy.data(), gw.data(), gv.data(), gr.data())
                            : cip_conicip_problems((int)k, gp.data(), gc.data(), gb.data(), gd.data(), opt, gy.data(), gw.data(), gv.data(), gr.data(), in_flight);
        if (rc == 0 || !lock)                                   // the thread pool writes a status for every problem it reached
            for (size_t j = 0; j < k; ++j) res[idx[j]] = gr[j];
        return rc;
    };
    for (const auto &bin : bins) {
        if (bin.size() < 2 || !lockstep_ok(probs[bin[0]])) { rest.insert(rest.end(), bin.begin(), bin.end()); continue; }
        const int rc = run(bin, true);
        // a bin that turns out not to qualify (same shape, yet a different slab layout: CSR arrays in device memory with differing
        // numbers of non-zeros) has written nothing: its problems join the thread pool's share instead of failing the batch
        if (rc == CIP_E_UNSUPPORTED) { rest.insert(rest.end(), bin.begin(), bin.end()); continue; }
        if (rc) return rc;
    }
    if (!rest.empty()) {
        std::sort(rest.begin(), rest.end());
        const int rc = run(rest, false);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int cip_release_cached_memory(void) {
    std::lock_guard<std::mutex> lk(g_cache.mu);
    for (auto &sl : g_cache.slots) (void)hipFree(sl.ptr);
    g_cache.slots.clear();
    return 0;
}

// diagnostics of the calling thread's last cip_conicip_lockstep: {groups, problems, problems that left their group}
extern "C" int cip_lockstep_stats(int *out3) {
    if (!out3) return CIP_E_INVALID;
    for (int i = 0; i < 3; ++i) out3[i] = g_last_stats[i];
    return 0;
}
