// Dense symmetric LDL' (no pivoting) factor + solve for gfx950.
//
// Fills the role the reference gives to "dense factor + solve of the Newton system
// behind the kktsolver callback" (src/kktsolvers.jl:35 `qr(...)`, :257/:295 `lu(Z)`;
// solves at :39-48, :259, :299).  The matrix is the quasi-definite KKT matrix
//   [S G'; G 0]  (Schur route)   or   [-F'F -A 0; -A' Q G'; 0 G 0]  (full 3x3 route),
// whose LDL' exists for the static pivot order (S > 0, G full row rank).
//
// Blocked right-looking algorithm, two levels:
//   outer block NBO (512, 768 from order 4096 on) = NBO/128 inner panels of 128 columns
//   per inner panel:  diag 128x128 LDL' (1 workgroup, LDS) -> substitution TRSM (W = A21 inv(L11)', L = W D^-1) -> update of
//                     the block's remaining panel columns; on the serial schedule these are ONE launch per panel
//                     (diag.hip: k_ldlt_panel), in lock-step batches three
//   per outer block:  trailing update  C -= W L'  (lower tiles, K = NBO, MFMA GEMM)
// 3/4 of the N^3/3 flops at n = 8192 are in the trailing-update GEMM (gemm_f64.hip), the rest in the K = 128 in-block tiles.
//
// Solves (HBM-bound, L read once per sweep): blocked substitution with the stored inverses of the diagonal blocks
// (doubled up to 1024 wide), two gemv launches per block step and sweep.
#include "cip_internal.h"
#include <stdlib.h>
#include <string.h>

// Outer block.  0 = automatic: 896 (round 4; 768 before) from order 4096 on when the panel chain is fused (the in-block updates, K = 128 tiles
// that a wider block has more of, then run beside the diagonal kernels and a trailing update with K = 768 passes over C
// less often: same-session A/B at n = 8192, 512 / 768 / 1024 -> 130.6-132.0 / 134.5 / 133.9 KKT solves/s), else 512 (unfused
// chain, round 1: 256 / 384 / 512 -> 107.7 / 107.0 / 109.1; 1024 -> 125 against 128).
#include <atomic>
#include <mutex>
#include <vector>
static std::atomic<int> g_nbo{0};         // read by the library's worker threads (batch.hip) while a caller may set it
static std::atomic<int> g_fuse_diag{-1};
static std::once_flag g_fuse_once;
// 0: diag -> TRSM -> in-block update, three launches per panel; 1: the update inside the next diagonal kernel's launch
// (k_ldlt_diag_upd); 3: one launch per panel, the TRSM pipelined behind the diagonal kernel (k_ldlt_panel)
#define CIP_FUSE_DEFAULT 3
static void fuse_env(void) {
    std::call_once(g_fuse_once, [] {
        const char *e = getenv("CIP_FUSE_DIAG");
        const int v = e ? atoi(e) : CIP_FUSE_DEFAULT;
        g_fuse_diag = (v == 0) ? 0 : (v == 3) ? 3 : 1;
    });
}
int cip_ldlt_outer_block_for(int Npad) {
    { const int v = g_nbo.load(std::memory_order_relaxed); if (v > 0) return v; }
    fuse_env();
    // Round 4 re-tuned the width on the current chain (same-session A/B, wide last block in force; CIP_LDLT_NBO_AUTO overrides):
    // 640 / 768 / 896 / 1024 -> 185.4 / 189.1 / 191.3 / 188.7 KKT solves/s at n = 8192 (6 x 896 + 2816: one trailing update
    // fewer, 60.1 TFLOP/s), config 3 (order 4608) 31.3 -> 30.0 ms, the literal 3x3 route at N = 16384 35.7 -> 35.8.
    static const int nbo_auto = [] { const char *e = getenv("CIP_LDLT_NBO_AUTO"); const int v = e ? atoi(e) : 896; return (v >= 256 && v <= 1024 && v % CIP_NB == 0) ? v : 896; }();
    return (Npad >= 4096 && g_fuse_diag) ? nbo_auto : 512;    // not a function of the batch: lock-step groups reproduce the one-problem loop bit for bit
}
#define CIP_NBO_MAX 1024
#define CIP_TAIL_MAX 3072               // widest last block (CIP_LDLT_TAIL is clamped to it): Wbuf has room for it from order 4096 on
static size_t wbuf_cols(int Npad) { return Npad >= 4096 ? (Npad < CIP_TAIL_MAX ? Npad : CIP_TAIL_MAX) : CIP_NBO_MAX; }
// The LAST outer block takes everything that is left once that is no more than CIP_LDLT_TAIL columns (round 3; automatic
// widths only, orders from 4096 on).  At the bottom of the matrix a trailing update is a handful of tiles per CU behind a
// K = 768 loop (r = 1280 and r = 512 at n = 8192: 36 us each, 6 TFLOP/s) while the panel launches leave most of the chip idle:
// in a wide last block the same flops are in-block update tiles (K = 128, lower triangle only) that run BESIDE the diagonal
// kernels.  Same-session A/B at n = 8192, tail 0 / 1280 / 2048 / 2816 / 3584 / 4352: 180.5 / 181.8 / 183.1 / 185.4 / 184.7 /
// 182.5 KKT solves/s (beyond 2816 the block's first panel launches are bound by their ~900 tiles: 57 / 52 / 51 us).
// A function of the order and the column only, like the width itself.
static int ldlt_tail_cols(void) {
    static const int v = [] { const char *e = getenv("CIP_LDLT_TAIL"); const int t = e ? atoi(e) : 2816; return t > CIP_TAIL_MAX ? CIP_TAIL_MAX : t; }();
    return v;
}
static int outer_block_width(int Npad, int C0) {
    const int NBO = cip_ldlt_outer_block_for(Npad), left = Npad - C0;
    if (g_nbo == 0 && NBO >= 640 && left <= ldlt_tail_cols()) return left;
    return left < NBO ? left : NBO;
}
int cip_ldlt_outer_block(void) { return g_nbo; }
void cip_ldlt_set_outer_block(int nbo) {
    if (nbo == 0 || (nbo >= CIP_NB && nbo % CIP_NB == 0 && nbo <= 1024)) g_nbo = nbo;
}

// ---- optional instrumentation: HIP events around every trailing-update launch (bench.py roofline)
struct LdltProfile {
    std::vector<hipEvent_t> pool;     // event pairs
    size_t used = 0;
    std::vector<double> flops;        // algorithmic flops of each recorded launch
    double tot_launches = 0, tot_ms = 0, tot_flops = 0;
    int stride = 1;                   // every stride-th factorisation is timed (an event pair costs its launch ~8 us of chain)
    long nfact = 0;                   // factorisations seen
};
void cip_ldlt_profile_stride(LdltProfile *p, int stride) { if (p) { p->stride = stride > 1 ? stride : 1; p->nfact = 0; } }
LdltProfile *cip_ldlt_profile_create(void) { return new LdltProfile(); }
void cip_ldlt_profile_destroy(LdltProfile *p) {
    if (!p) return;
    for (hipEvent_t e : p->pool) (void)hipEventDestroy(e);
    delete p;
}
static int prof_event(LdltProfile *p, hipStream_t s) {
    if (p->used == p->pool.size()) {
        hipEvent_t e;
        CIP_HIP_CHECK(hipEventCreate(&e));
        p->pool.push_back(e);
    }
    CIP_HIP_CHECK(hipEventRecord(p->pool[p->used++], s));
    return 0;
}
int cip_ldlt_profile_collect(LdltProfile *p, double *launches, double *ms, double *flops) {
    if (!p) return -1;
    if (p->used) CIP_HIP_CHECK(hipEventSynchronize(p->pool[p->used - 1]));
    for (size_t i = 0; i + 1 < p->used; i += 2) {
        float t = 0;
        CIP_HIP_CHECK(hipEventElapsedTime(&t, p->pool[i], p->pool[i + 1]));
        p->tot_ms += t;
        p->tot_flops += p->flops[i / 2];
        p->tot_launches += 1;
    }
    p->used = 0;
    p->flops.clear();
    if (launches) *launches = p->tot_launches;
    if (ms) *ms = p->tot_ms;
    if (flops) *flops = p->tot_flops;
    return 0;
}

// A profile of the calling host thread, for factorisations whose workspace has none of its own: the handles of a
// lock-step batch are created inside cip_conicip_lockstep, whose host code runs on the caller's thread (bench.py measures
// config 5's trailing update this way).  One launch then covers every live problem of the batch: flops x popcount(mask).
static thread_local LdltProfile *g_tl_prof = nullptr;
int cip_ldlt_profile_thread(int enabled) {
    if (enabled && !g_tl_prof) g_tl_prof = cip_ldlt_profile_create();
    if (!enabled && g_tl_prof) { cip_ldlt_profile_destroy(g_tl_prof); g_tl_prof = nullptr; }
    return 0;
}
int cip_ldlt_profile_thread_collect(double *launches, double *ms, double *flops) {
    return cip_ldlt_profile_collect(g_tl_prof, launches, ms, flops);
}

// The same event-pair timing for other dominant kernels of the secondary configurations, per calling thread (bench.py's
// `secondary` object: cip_conicip runs on the caller's thread): slot 1 = Schur formation (assemble.hip, dense A), slot 2 = the
// one-sided Jacobi of a large S cone's NT scaling (sdp_large.hip).  Slot 0 is the trailing update's thread profile above.
static thread_local LdltProfile *g_tl_aux[CIP_PROF_SLOTS] = {nullptr, nullptr, nullptr};
int cip_prof_slot_enable(int slot, int enabled) {
    if (slot == 0) return cip_ldlt_profile_thread(enabled);
    if (slot < 0 || slot >= CIP_PROF_SLOTS) return -1;
    if (enabled && !g_tl_aux[slot]) g_tl_aux[slot] = cip_ldlt_profile_create();
    if (!enabled && g_tl_aux[slot]) { cip_ldlt_profile_destroy(g_tl_aux[slot]); g_tl_aux[slot] = nullptr; }
    return 0;
}
int cip_prof_slot_collect(int slot, double *launches, double *ms, double *flops) {
    if (slot == 0) return cip_ldlt_profile_thread_collect(launches, ms, flops);
    if (slot < 0 || slot >= CIP_PROF_SLOTS) return -1;
    return cip_ldlt_profile_collect(g_tl_aux[slot], launches, ms, flops);
}
// begin / end around a launch (or a launch set); no-ops unless the calling thread switched the slot on
int cip_prof_slot_begin(int slot, hipStream_t s, double work) {
    LdltProfile *p = (slot > 0 && slot < CIP_PROF_SLOTS && !cip_tl_builder) ? g_tl_aux[slot] : nullptr;
    if (!p) return 0;
    int rc = prof_event(p, s);
    if (rc) return rc;
    p->flops.push_back(work);
    return 0;
}
int cip_prof_slot_end(int slot, hipStream_t s) {
    LdltProfile *p = (slot > 0 && slot < CIP_PROF_SLOTS && !cip_tl_builder) ? g_tl_aux[slot] : nullptr;
    return p ? prof_event(p, s) : 0;
}

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// solve block: largest of {1024, 512, 256, 128} that divides the (128-padded) order
// Upper limit: process-wide knob (cip_set_solve_block_max / CIP_SOLVE_BLOCK, default 1024) or, when set, the calling
// thread's override (lock-step batches create their handles with 256: in a batch a block step is one launch for all
// problems, so launches are cheap and the doubled inverses -- 1.5 GFLOP per n = 2048 factorisation on top of its 2.9 --
// and the half-empty 1024-wide triangular blocks the solves stream are what cost).
static std::atomic<int> g_solve_block_max{-1};
thread_local int cip_tl_solve_block_max = 0;
int cip_solve_block_max_set(int b) {
    if (g_solve_block_max.load() < 0) { const char *e = getenv("CIP_SOLVE_BLOCK"); int v = -1; g_solve_block_max.compare_exchange_strong(v, e ? atoi(e) : 1024); }
    const int prev = g_solve_block_max.load();
    if (b == 128 || b == 256 || b == 512 || b == 1024) g_solve_block_max.store(b);
    return prev;
}
int cip_solve_block(int Npad) {
    int mx = cip_tl_solve_block_max > 0 ? cip_tl_solve_block_max : cip_solve_block_max_set(0);
    if (mx < CIP_NB) mx = CIP_NB;
    // (2048: only through the calling thread's override -- the order-2048 workspaces of large S cones want inv(L) of the WHOLE matrix,
    //  sdp_large.hip; the public knob stops at 1024)
    for (int b = 2048; b > CIP_NB; b >>= 1)
        if (b <= mx && Npad % b == 0) return b;
    return CIP_NB;
}

// One launch per block step of the triangular sweeps (k_solve_step) instead of two: needs the pre-multiplied neighbour blocks
// MT / PT (k_solve_premul, 2 (Npad/Bs - 1) triangular Bs^3 products per factorisation).  Mode 0 (default): two launches per step;
// 1: one launch for solve blocks of at most 512 columns; 2: always.  CIP_SOLVE_FUSED.
// Measured (round 5, same session, tools/ab_solve_fused.sh): the sweeps get 20 % faster -- solve4x4 at n = 8192 0.208 -> 0.165 ms
// with 17 launches instead of 32 -- and every configuration gets SLOWER, because the products cost more than the launches they
// save at the two to three solves an interior-point iteration makes per factorisation: n = 8192 15.7 GFLOP on the side stream,
// +0.24 ms per step exposed (196.2 -> 190.3 KKT solves/s); lock-step shards of 8 / 64 problems of order 2048 15.2 -> 16.5 /
// 78.0 -> 80.5 ms per pass (+28 % flops per factorisation); config 3 3.93 -> 4.0 ms per iteration.  It pays from about six
// solves per factorisation on: off by default, kept for callers that solve many right-hand sides with one factor.
static std::atomic<int> g_solve_fused{-1};
int cip_solve_fused_set(int mode) {
    if (g_solve_fused.load() < 0) { const char *e = getenv("CIP_SOLVE_FUSED"); int v = -1; g_solve_fused.compare_exchange_strong(v, e ? atoi(e) : 0); }
    const int prev = g_solve_fused.load();
    if (mode == 0 || mode == 1 || mode == 2) g_solve_fused.store(mode);
    return prev;
}
static int solve_fused_for(int Npad, int Bs) {
    const int mode = cip_solve_fused_set(-1);
    if (Npad / Bs < 2) return 0;
    return mode == 2 || (mode == 1 && Bs <= 512);
}

// Layout: everything whose size depends on (Npad, solve block) only, then -- LAST -- the pre-multiplied neighbour blocks MT / PT of the
// one-launch block steps.  `fused` is decided ONCE by the owner of the workspace and passed to both functions (round-5 advisor: each
// used to read the process-wide mode for itself, so a cip_set_solve_fused between the two -- or between handle creation's sizing and
// its carve -- placed MT / PT beyond the allocation).  fused < 0: by the current mode (cip_ldlt_ws_bytes only: the public sizing
// call reserves MT / PT whenever the order has two solve blocks, whatever the mode, so that a caller-owned workspace fits every mode).
int cip_ldlt_fused_for(int Npad) { return solve_fused_for(Npad, cip_solve_block(Npad)); }
size_t cip_ldlt_ws_bytes(int Npad, int fused) {
    const size_t nblk = Npad / CIP_NB;
    size_t b = 0;
    b += al256((size_t)Npad * wbuf_cols(Npad) * 8);       // Wbuf
    b += al256(nblk * CIP_NB * CIP_NB * 8) * 2;          // Linv, LinvT
    b += al256(nblk * 2048 * 8);                         // Xm
    const int Bs = cip_solve_block(Npad);
    const size_t nbk = Npad / Bs;
    b += al256(nbk * (size_t)Bs * Bs * 8) * 2;           // X, XT
    b += al256(nbk * (size_t)(Bs / 2) * (Bs / 2) * 8 + 256);   // Tt
    b += al256((size_t)Npad * 8);                        // zbuf
    b += al256((size_t)Npad * 8) * 3;                    // dinv, dvec, tmp
    b += al256(64 + 12 * nblk);                          // info (16 ints) + a `ready`, a `stage` and a tile-queue counter per 128-block (fused panel launches)
    if (fused < 0 ? nbk >= 2 : fused) b += al256(nbk * (size_t)Bs * Bs * 8) * 2;   // MT, PT
    return b;
}

void cip_ldlt_ws_carve(void *base, int Npad, LdltWorkspace *ws, int fused) {
    const size_t nblk = Npad / CIP_NB;
    char *p = (char *)base;
    ws->Wbuf = (double *)p;  p += al256((size_t)Npad * wbuf_cols(Npad) * 8);
    ws->Linv = (double *)p;  p += al256(nblk * CIP_NB * CIP_NB * 8);
    ws->LinvT = (double *)p; p += al256(nblk * CIP_NB * CIP_NB * 8);
    ws->Xm = (double *)p;    p += al256(nblk * 2048 * 8);
    ws->Bs = cip_solve_block(Npad);
    const size_t nbk = Npad / ws->Bs;
    ws->X = (double *)p;     p += al256(nbk * (size_t)ws->Bs * ws->Bs * 8);
    ws->XT = (double *)p;    p += al256(nbk * (size_t)ws->Bs * ws->Bs * 8);
    ws->Tt = (double *)p;    p += al256(nbk * (size_t)(ws->Bs / 2) * (ws->Bs / 2) * 8 + 256);
    ws->zbuf = (double *)p;  p += al256((size_t)Npad * 8);
    ws->dinv = (double *)p;  p += al256((size_t)Npad * 8);
    ws->dvec = (double *)p;  p += al256((size_t)Npad * 8);
    ws->tmp = (double *)p;   p += al256((size_t)Npad * 8);
    ws->info = (int *)p;     p += al256(64 + 12 * nblk);
    ws->fused = (fused != 0 && nbk >= 2) ? 1 : 0;
    ws->MT = ws->PT = nullptr;
    if (ws->fused) {
        ws->MT = (double *)p;  p += al256(nbk * (size_t)ws->Bs * ws->Bs * 8);
        ws->PT = (double *)p;  p += al256(nbk * (size_t)ws->Bs * ws->Bs * 8);
    }
    ws->prof = nullptr;
    ws->lazyC = nullptr; ws->lazy_ld = 0; ws->lazy_diag = nullptr;
    ws->signs = PivotSigns{-1, 0, 0};
    ws->x_zeroed = nullptr;
    ws->side = nullptr;
    ws->no_prep = 0;
    ws->unfused = 0;
}

// diag.hip
int cip_launch_diag_v2(hipStream_t s, double *Kb, long ld, double *xm_out, double *dvec, double *dinv, int *info, int col0,
                       PivotSigns sg);
int cip_launch_diag_upd(hipStream_t s, double *Kb, long ld, double *xm_out, double *dvec, double *dinv, int *info, int col0,
                        PivotSigns sg, unsigned *ready, const GemmArgs &g);
int cip_launch_diag_inverse(hipStream_t s, const double *K, long ld, int nblk, const double *xm_all, double *Linv,
                            double *LinvT, int Bs);
int cip_launch_panel(hipStream_t s, double *Kb, long ld, double *xm_out, double *dvec, double *dinv, int *info, int col0,
                     PivotSigns sg, unsigned *ready, unsigned *stage, unsigned *tileq, const GemmArgs *g, int rows, double *W, long ldw);
int cip_launch_trsm_subst(hipStream_t s, double *Ap, long ld, int rows, const double *L11, const double *xm,
                          const double *dinv, double *W, long ldw);

// right-looking update inside the outer block: after inner panel t, the remaining panel columns of the block
//   K[c0+128:, c0+128 : C0+wblk] -= W_t[c0+128:, :] * L_t[c0+128 : C0+wblk, :]'        (K = 128, wide and short:
// many quarter tiles, one short k-loop -- the latency-critical shape; the left-looking form had K up to 384)
static bool rest_of_block_args(double *K, int Npad, long ld, double *Wb, int C0, int wblk, int t, GemmArgs &g) {
    const int c1 = C0 + (t + 1) * CIP_NB;            // first row / column still to be factored in this block
    const int ncols = C0 + wblk - c1;
    if (ncols <= 0 || c1 >= Npad) return false;
    g = GemmArgs{};
    g.A = Wb + c1 + (long)(t * CIP_NB) * Npad; g.lda = Npad;
    g.B = K + c1 + (long)(c1 - CIP_NB) * ld; g.ldb = ld;
    g.C = K + c1 + (long)c1 * ld; g.ldc = ld;
    g.M = Npad - c1; g.N = ncols; g.K = CIP_NB; g.alpha = -1.0; g.lower = 0;
    return true;
}
static int update_rest_of_block(hipStream_t s, double *K, int Npad, long ld, double *Wb, int C0, int wblk, int t) {
    GemmArgs g;
    if (!rest_of_block_args(K, Npad, ld, Wb, C0, wblk, t, g)) return 0;
    return cip_launch_gemm(s, EPI_ACCUM, g);
}
// CIP_FUSE_DIAG=0 / cip_set_ldlt_fused_chain(0): the unfused chain (diag -> TRSM -> update per panel), for A/B runs and tests
int cip_ldlt_set_fused_chain(int on) { fuse_env(); const int prev = g_fuse_diag; if (on == 0 || on == 1 || on == 3) g_fuse_diag = on; return prev; }

// one inner-panel sweep of an outer block: [strip update] -> diagonal kernel -> TRSM, for each 128 columns
static int factor_outer_panels(hipStream_t s, double *K, int Npad, long ld, const LdltWorkspace &ws, double *Wb, int C0,
                               int wblk, int t0 = 0, int t1 = -1) {
    int rc;
    const int T = t1 < 0 ? wblk / CIP_NB : t1;               // panels [t0, T) of the block (all of them by default)
    for (int t = t0; t < T; ++t) {
        const int c0 = C0 + t * CIP_NB;
        const int jb = c0 / CIP_NB;
        const int r = Npad - c0 - CIP_NB;
        // MFMA micro-blocked diagonal kernel + substitution TRSM (the block inverses the solves
        // need are produced by one batched launch after the factorisation).  From the second panel of the block on, the
        // diagonal kernel's launch also carries the previous panel's in-block update (diag.hip: k_ldlt_diag_upd).
        GemmArgs gu;
        fuse_env();
        // fused when the chain has the chip to itself -- and in lock-step groups (grid.z = problems) while the group's
        // long-lived workgroups (diagonal kernel, producers, strips: 10 + Npad / 64 per problem at the first panel) are no more
        // than about four rounds of the chip: order 2048, 8 / 16 / 24 problems 19.9 -> 17.8, 28.0 -> 26.5, 37.6 -> 37.1 ms per
        // pass against the three batched launches per panel, 32 problems equal, 64 problems 83 -> 86 (a big batch is
        // throughput-bound and wants the tiles' occupancy).  Residency is a matter of speed, not of progress: a strip waits
        // for its own problem's workgroup 0 only, which was dispatched before it; a workgroup 0 waits for its producers, which
        // follow it in dispatch order and find a CU as soon as any earlier problem's workgroups retire -- and the first
        // problem's always can.
        static const int lsmax = [] { const char *e = getenv("CIP_LOCKSTEP_PANEL_MAX"); return e ? atoi(e) : 32; }();
        static const int lscus = [] {
            if (const char *e = getenv("CIP_LOCKSTEP_PANEL_CUS")) return atoi(e);
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return 4 * cus;
        }();
        const bool small_group = cip_in_batch() && g_fuse_diag == 3 && cip_tl_bz.B <= lsmax && (long)cip_tl_bz.B * (10 + Npad / 64) <= lscus;
        const bool fuse = g_fuse_diag && !ws.unfused && (!cip_in_batch() || small_group);
        if (fuse && g_fuse_diag == 3) {
            const bool upd = t > 0 && rest_of_block_args(K, Npad, ld, Wb, C0, wblk, t - 1, gu);
            unsigned *ctr = (unsigned *)(ws.info + 16);
            if ((rc = cip_launch_panel(s, K + c0 + (long)c0 * ld, ld, ws.Xm + (size_t)jb * 2048, ws.dvec + c0, ws.dinv + c0, ws.info, c0,
                                       ws.signs, ctr + jb, ctr + Npad / CIP_NB + jb, ctr + 2 * (Npad / CIP_NB) + jb, upd ? &gu : nullptr, r,
                                       Wb + (c0 + CIP_NB) + (long)(t * CIP_NB) * Npad, Npad)))
                return rc;
            continue;
        }
        if (fuse && t > 0 && rest_of_block_args(K, Npad, ld, Wb, C0, wblk, t - 1, gu)) {
            if ((rc = cip_launch_diag_upd(s, K + c0 + (long)c0 * ld, ld, ws.Xm + (size_t)jb * 2048, ws.dvec + c0, ws.dinv + c0,
                                          ws.info, c0, ws.signs, (unsigned *)(ws.info + 16) + jb, gu)))
                return rc;
        } else if ((rc = cip_launch_diag_v2(s, K + c0 + (long)c0 * ld, ld, ws.Xm + (size_t)jb * 2048, ws.dvec + c0,
                                            ws.dinv + c0, ws.info, c0, ws.signs)))
            return rc;
        if ((rc = cip_launch_trsm_subst(s, K + (c0 + CIP_NB) + (long)c0 * ld, ld, r, K + c0 + (long)c0 * ld,
                                        ws.Xm + (size_t)jb * 2048, ws.dinv + c0,
                                        Wb + (c0 + CIP_NB) + (long)(t * CIP_NB) * Npad, Npad)))
            return rc;
        if (!fuse && (rc = update_rest_of_block(s, K, Npad, ld, Wb, C0, wblk, t))) return rc;
    }
    return 0;
}

// zero fill as a kernel: these fills sit inside the launch sequences that small systems replay as hipGraphs, and a
// captured hipMemsetAsync node left the flag words of ws.info unset on ROCm 7.0 (the factorisation then reported a
// scheduler error out of uninitialised memory)
__global__ __launch_bounds__(256) void k_zero_words(unsigned *p, size_t nwords, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO1(cb, p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
static int zero_fill(hipStream_t s, void *p, size_t bytes) {
    const size_t nw = bytes / 4;
    size_t nb = (nw + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb == 0) return 0;
    cip_launch_b(k_zero_words, dim3((unsigned)nb), dim3(256), 0, s, (unsigned *)p, nw);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// upper triangle <- (strictly lower triangle)': gives the forward sweep the same coalesced
// "column-dot" access as the backward sweep (U[k, i] = L[i, k])
__global__ __launch_bounds__(256) void k_mirror_lower(double *K, long ld, int bj0, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO1(cb, K);
    __shared__ double t[32][33];
    const int bi = blockIdx.x + bj0, bj = blockIdx.y + bj0;      // 32x32 tile (row tile bi, column tile bj), bi >= bj; bj0: first column tile of the range
    if (bi < bj) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int q = 0; q < 32; q += 8) t[ty + q][tx] = K[(long)(bi * 32 + tx) + (long)(bj * 32 + ty + q) * ld];
    __syncthreads();
    for (int q = 0; q < 32; q += 8) {
        const int r = bj * 32 + tx, c = bi * 32 + ty + q;          // destination (r, c) = source (c, r)
        if (c > r) K[(long)r + (long)c * ld] = t[tx][ty + q];
    }
}
// After the factorisation: inverses of the Bs x Bs unit-lower diagonal blocks by doubling,
//   inv([L11 0; L21 L22]) = [X11 0; -X22 L21 X11  X22],
// two batched MFMA GEMMs per level (Tt = X11' L21', then X21 = -X22 Tt' together with its transpose), so that a
// triangular solve is Npad/Bs block steps (8 at n = 8192) instead of Npad/128 (64).
// the strictly upper blocks of X (lower of XT) are never written by the preparation: zeroed once per workspace, on the
// factorisation's own stream before anything else (the ranges below may run on two streams)
static int ensure_x_zeroed(hipStream_t s, int Npad, const LdltWorkspace &ws) {
    const int Bs = ws.Bs;
    if (Bs == CIP_NB || (ws.x_zeroed && *ws.x_zeroed)) return 0;
    int rc;
    const int nall = Npad / Bs;
    if ((rc = zero_fill(s, ws.X, sizeof(double) * (size_t)nall * Bs * Bs))) return rc;
    if ((rc = zero_fill(s, ws.XT, sizeof(double) * (size_t)nall * Bs * Bs))) return rc;
    if (ws.x_zeroed && (!cip_in_batch() || cip_tl_bz.mask == (cip_tl_bz.B >= 64 ? ~0ull : ((1ull << cip_tl_bz.B) - 1)))) *ws.x_zeroed = 1;
    return 0;
}
// The pre-multiplied neighbours of the one-launch block steps (k_solve_step), both kinds in ONE launch, 64 x 64 tiles:
//   blockIdx.y <  nm:  MT_J = U_{J-1,J} XT_J,  J = jm0 + blockIdx.y       (U = L' from the upper triangle; XT_J upper triangular:
//                      column tile j0 needs k < j0 + 64 only)
//   blockIdx.y >= nm:  PT_J = L_{J+1,J} X_J,   J = jp0 + blockIdx.y - nm  (X_J lower triangular: k >= j0 only)
#include "cip_gemm_tile.h"
__global__ __launch_bounds__(256, 4) void k_solve_premul(const double *K, long ld, const double *X, const double *XT, double *MT,
                                                         double *PT, int Bs, int jm0, int nm, int jp0, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    CIP_BATCH_GUARD(cb);
    CIP_BO5(cb, K, X, XT, MT, PT);
    const int tm = Bs / SB;
    const long i0 = (long)(blockIdx.x % tm) * SB, j0 = (long)(blockIdx.x / tm) * SB;
    const long bs2 = (long)Bs * Bs;
    GemmArgs g = {};
    g.alpha = 1.0; g.ldb = Bs; g.ldc = Bs; g.lda = ld;
    if ((int)blockIdx.y < nm) {
        const long J = jm0 + blockIdx.y;
        g.A = K + (J - 1) * Bs + J * Bs * ld;               // A[i, k] = U[C_{J-1} + i, C_J + k] = L[C_J + k, C_{J-1} + i]
        g.B = X + J * bs2;                                   // B[j, k] = X_J[j, k] = XT_J[k, j]: zero for k > j
        g.C = MT + J * bs2;
        g.K = (int)(j0 + SB);
    } else {
        const long J = jp0 + ((int)blockIdx.y - nm);
        g.A = K + (J + 1) * Bs + J * Bs * ld + j0 * ld;     // A[i, k] = L[C_{J+1} + i, C_J + k], k from j0 on
        g.B = XT + J * bs2 + j0 * Bs;                        // B[j, k] = XT_J[j, k] = X_J[k, j]: zero for k < j
        g.C = PT + J * bs2;
        g.K = (int)(Bs - j0);
    }
    gemm_tile_64<EPI_STORE>(g, lds, i0, j0);
}
// The doubling products of the LAST solve block of a 1024-wide blocking on 16x16 tiles (k_gemm_nt_16_batched; CIP_DOUBLING_TINY=0:
// never).  That block's preparation is the one piece of a factorisation nothing hides: with the side stream it runs under the first
// solve's forward sweep, which reaches the block after ~90 us, and its seven dependent launches took 135 us -- a 64x64 tile walks its
// whole K = h on one CU (512 dependent MFMAs per wave at h = 512: 19 us per launch on an idle chip), the 16x16 form spreads the same
// product over (h / 16)^2 workgroups.  Same-session, n = 8192: the first solve after a factorisation 0.282 -> 0.244 ms (the second:
// 0.205).  For every block, or in lock-step batches, the small tiles LOSE (more workgroups beside the panel chain: factor + 0.05 ms;
// 8 problems of order 2048: 15.2 -> 15.8 ms per pass).  A function of the block's position and the blocking only -- a side-stream
// group, a whole-matrix preparation (which splits the last block off) and a lock-step batch all produce the same bits.
static int doubling_tiny(int Bs, int J0, int nbk_all) {
    static const int on = [] { const char *e = getenv("CIP_DOUBLING_TINY"); return e ? atoi(e) : 1; }();
    return on && Bs == 1024 && J0 == nbk_all - 1;
}
static int build_solve_premul(hipStream_t s, double *K, long ld, const LdltWorkspace &ws, int nbk_all, int J0, int J1) {
    if (!ws.fused) return 0;
    const int Bs = ws.Bs;
    const double *X = (Bs == CIP_NB) ? ws.Linv : ws.X, *XT = (Bs == CIP_NB) ? ws.LinvT : ws.XT;
    const int jm0 = J0 > 1 ? J0 : 1, nm = J1 - jm0 > 0 ? J1 - jm0 : 0;                       // MT_J, J in [max(J0, 1), J1)
    const int jp1 = J1 < nbk_all - 1 ? J1 : nbk_all - 1, np = jp1 - J0 > 0 ? jp1 - J0 : 0;  // PT_J, J in [J0, min(J1, nbk - 1))
    if (nm + np == 0) return 0;
    const int tm = Bs / SB;
    cip_launch_b(k_solve_premul, dim3((unsigned)(tm * tm), (unsigned)(nm + np)), dim3(256), 0, s, (const double *)K, ld, X, XT, ws.MT, ws.PT, Bs, jm0, nm, J0);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

// [J0, J1): the range of Bs-wide diagonal blocks to prepare (their columns are final), all of them by default.  The mirror
// covers the same columns (every row below them).  Ranges are independent of each other (own slices of X / XT / Tt).
static int build_solve_blocks(hipStream_t s, double *K, int Npad, long ld, const LdltWorkspace &ws, int J0 = 0, int J1 = -1) {
    const int Bs = ws.Bs;
    if (J1 < 0) J1 = Npad / Bs;
    if (J1 - J0 > 1 && J1 == Npad / Bs && doubling_tiny(Bs, J1 - 1, Npad / Bs)) {      // the last block goes its own way (see doubling_tiny)
        const int rc0 = build_solve_blocks(s, K, Npad, ld, ws, J0, J1 - 1);
        return rc0 ? rc0 : build_solve_blocks(s, K, Npad, ld, ws, J1 - 1, J1);
    }
    const int nbk = J1 - J0;
    if (nbk <= 0) return 0;
    int rc;
    // (no mirror pass: L' is written with the factor, diag.hip: diag_store_panel_T / wave_store_T; CIP_LDLT_MIRROR=1 runs the
    //  old pass on top of it -- it rewrites the same values -- for A/B timing)
    static const int mirror_pass = [] { const char *e = getenv("CIP_LDLT_MIRROR"); return e ? atoi(e) : 0; }();
    if (mirror_pass) {
        const int bj0 = J0 * (Bs / 32), nct = nbk * (Bs / 32);
        cip_launch_b(k_mirror_lower, dim3(Npad / 32 - bj0, nct), dim3(256), 0, s, K, ld, bj0);
    }
    const int per = Bs / CIP_NB;
    const double *xm0 = ws.Xm + (size_t)J0 * per * 2048;
    const double *Kd = K + (long)J0 * Bs * (ld + 1);
    if (Bs == CIP_NB) {                                                 // X == Linv, XT == LinvT
        if ((rc = cip_launch_diag_inverse(s, Kd, ld, nbk, xm0, ws.Linv + (size_t)J0 * CIP_NB * CIP_NB, ws.LinvT + (size_t)J0 * CIP_NB * CIP_NB, CIP_NB))) return rc;
        return build_solve_premul(s, K, ld, ws, Npad / Bs, J0, J1);
    }
    const long bs2 = (long)Bs * Bs, tt2 = (long)(Bs / 2) * (Bs / 2);
    double *X0 = ws.X + (size_t)J0 * bs2, *XT0 = ws.XT + (size_t)J0 * bs2, *Tt0 = ws.Tt + (size_t)J0 * tt2;
    // the inverses of the 128-blocks, written straight into the diagonal of X / XT
    if ((rc = cip_launch_diag_inverse(s, Kd, ld, nbk * per, xm0, X0, XT0, Bs))) return rc;
    for (int h = CIP_NB; h < Bs; h *= 2) {
        const int P = Bs / (2 * h);                          // pairs per block
        const long pX = 2L * h * (Bs + 1);                   // pair stride inside a block of X / XT
        const long pK = 2L * h * (ld + 1);                   // pair stride along the diagonal of K
        GemmArgs g = {};
        g.M = g.N = g.K = h; g.lower = 0; g.overwrite = 1; g.by = nbk; g.bz = P;
        g.tiny16 = doubling_tiny(Bs, J0, Npad / Bs);
        // Tt = XT11 * L21'
        g.alpha = 1.0;
        g.A = XT0; g.lda = Bs; g.sAy = bs2; g.sAz = pX;
        g.B = Kd + h; g.ldb = ld; g.sBy = (long)Bs * (ld + 1); g.sBz = pK;
        g.C = Tt0; g.ldc = h; g.sCy = tt2; g.sCz = (long)h * h;
        if ((rc = cip_launch_gemm(s, EPI_ACCUM, g))) return rc;
        // X21 = -X22 * Tt'  and, from the same accumulators, XT12 = X21' (stored transposed by the epilogue)
        g.alpha = -1.0;
        g.A = X0 + h + (long)h * Bs; g.lda = Bs; g.sAy = bs2; g.sAz = pX;
        g.B = Tt0; g.ldb = h; g.sBy = tt2; g.sBz = (long)h * h;
        g.C = X0 + h; g.ldc = Bs; g.sCy = bs2; g.sCz = pX;
        g.Ct = XT0 + (long)h * Bs; g.ldct = Bs; g.sCty = bs2; g.sCtz = pX;
        if ((rc = cip_launch_gemm(s, EPI_ACCUM, g))) return rc;
        g.Ct = nullptr;
    }
    return build_solve_premul(s, K, ld, ws, Npad / Bs, J0, J1);
}

// ---- solve preparation BESIDE the panel chain (round 4).  The block inverses and the mirror image of columns that are
// final do not depend on the rest of the factorisation, and at the bottom of the matrix the panel launches are a latency-
// bound chain on a mostly idle chip.  When the last outer block is wide (>= 2 solve blocks begin or end in it) the
// preparation of every solve block whose columns are final is enqueued on a SIDE stream: everything to the left of the
// wide block when its panel chain starts, then one solve block each time the chain passes a multiple of Bs; only the last
// solve block is prepared behind the factorisation.  One event per fork (the main stream records, the side stream waits),
// one join at the end.  Off: CIP_SIDE_PREP=0, lock-step batches, graph capture.
struct LdltSide {
    hipStream_t s2 = nullptr;
    hipEvent_t fork[16] = {};         // recorded on the factorisation's stream: the columns of group g are final
    hipEvent_t done[16] = {};         // recorded on the side stream: group g's solve blocks are prepared
    int j0[16] = {};                  // first solve block of group g
    int nfork = 0;                    // groups of the last factorisation
    int waited = 0;                   // groups the factorisation's stream has waited for (cip_ldlt_side_join)
};
void cip_ldlt_side_destroy(LdltSide *sd) {
    if (!sd) return;
    if (sd->s2) (void)hipStreamSynchronize(sd->s2);          // its kernels read the handle's memory, which is freed next
    for (hipEvent_t e : sd->fork) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : sd->done) if (e) (void)hipEventDestroy(e);
    if (sd->s2) (void)hipStreamDestroy(sd->s2);
    delete sd;
}
static std::atomic<int> g_side_prep{-1};
static int side_prep_mode(void) {
    if (g_side_prep.load() < 0) { const char *e = getenv("CIP_SIDE_PREP"); int v = -1; g_side_prep.compare_exchange_strong(v, e ? (atoi(e) != 0) : 1); }
    return g_side_prep.load();
}
int cip_ldlt_set_side_prep(int on) { const int prev = side_prep_mode(); if (on == 0 || on == 1) g_side_prep.store(on); return prev; }
static LdltSide *side_get(const LdltWorkspace &ws) {
    if (!ws.side) return nullptr;
    if (*ws.side) return *ws.side;
    LdltSide *sd = new LdltSide();
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);             // lo = numerically greatest = LOWEST priority: the chain goes first
    bool ok = hipStreamCreateWithPriority(&sd->s2, hipStreamNonBlocking, lo) == hipSuccess;
    // (round 5: the LAST group -- what the first solve waits for -- on a third stream of the HIGHEST priority was measured and lost:
    //  the first solve unchanged, 0.246 -> 0.251 ms, and the factorisation 0.08-0.12 ms slower with such a queue in the process; on a
    //  third stream of DEFAULT priority: first solve 0.249 -> 0.249, factorisation + 0.02 ms)
    for (hipEvent_t &e : sd->fork) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for (hipEvent_t &e : sd->done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); cip_ldlt_side_destroy(sd); return nullptr; }
    *ws.side = sd;
    return sd;
}
// TEST SWITCH (CIP_DEBUG_SIDE_DELAY_US=k): every group of the side stream starts k microseconds late -- whoever reads what the group
// prepares without having waited for it then reads the previous factorisation's blocks, and the bits say so
__global__ void k_debug_spin(long ticks) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
static int side_fork(LdltSide *sd, hipStream_t s, double *K, int Npad, long ld, const LdltWorkspace &ws, int J0, int J1) {
    if (J1 <= J0) return 0;
    if (sd->nfork >= 16) return build_solve_blocks(s, K, Npad, ld, ws, J0, J1);
    const int g = sd->nfork++;
    sd->j0[g] = J0;
    hipStream_t t = sd->s2;
    CIP_HIP_CHECK(hipEventRecord(sd->fork[g], s));
    CIP_HIP_CHECK(hipStreamWaitEvent(t, sd->fork[g], 0));
    static const long delay_us = [] { const char *e = getenv("CIP_DEBUG_SIDE_DELAY_US"); return e ? atol(e) : 0L; }();
    if (delay_us > 0) { hipLaunchKernelGGL(k_debug_spin, dim3(1), dim3(64), 0, t, delay_us * 100); CIP_HIP_CHECK(hipGetLastError()); }
    const int rc = build_solve_blocks(t, K, Npad, ld, ws, J0, J1);
    CIP_HIP_CHECK(hipEventRecord(sd->done[g], t));
    return rc;
}
// The factorisation's stream waits for the side stream's groups that hold solve blocks <= J (J < 0: all of them).  The
// factorisation itself does NOT join: the last group -- the solve blocks whose columns the last panels produce -- is
// prepared while the first solve's element-wise kernel and the first block steps of its forward sweep run (they need the
// first blocks only); cip_ldlt_solve joins group by group as its sweep reaches their blocks, and whoever overwrites K or
// the block inverses next (the next assembly, cip_factor's timing read-out) joins everything.
int cip_ldlt_side_join(hipStream_t s, const LdltWorkspace &ws, int J) {
    LdltSide *sd = ws.side ? *ws.side : nullptr;
    if (!sd) return 0;
    while (sd->waited < sd->nfork && (J < 0 || sd->j0[sd->waited] <= J)) {
        CIP_HIP_CHECK(hipStreamWaitEvent(s, sd->done[sd->waited], 0));
        sd->waited += 1;
    }
    return 0;
}

// test hook (cip_debug_chain_giveup): the next `n` factorisations on a fused panel chain report that an in-launch wait gave up (info[3]),
// as a GPU shared with other processes can make them do -- so that the fall-back to the three-launch chain can be tested on one process
// (n = count + 65536 * skip: the first `skip` fused factorisations from now on are left alone, the `count` after them give up)
// (+ 2^28: they report a wrong-sign pivot at column 1 instead -- what the automatic regularisation answers)
static std::atomic<int> g_debug_giveup{0}, g_debug_giveup_skip{0}, g_debug_giveup_kind{0};
int cip_debug_chain_giveup_set(int n) {
    const int prev = g_debug_giveup.load();
    if (n >= 0) { g_debug_giveup_kind.store((n >> 28) & 1); g_debug_giveup_skip.store((n >> 16) & 0xfff); g_debug_giveup.store(n & 0xffff); }
    return prev;
}
__global__ void k_debug_set_word(int *p, int v, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO1(cb, p);
    if (threadIdx.x == 0) *p = v;
}
static int ldlt_factor_body(hipStream_t s, double *K, int Npad, long ld, const LdltWorkspace &ws);
int cip_ldlt_factor(hipStream_t s, double *K, int Npad, long ld, const LdltWorkspace &ws) {
    const int rc = ldlt_factor_body(s, K, Npad, ld, ws);
    if (rc == 0 && !ws.unfused && !cip_tl_builder && g_debug_giveup.load() > 0 && g_debug_giveup_skip.fetch_sub(1) <= 0 && g_debug_giveup.fetch_sub(1) > 0) {
        if (g_debug_giveup_kind.load()) cip_launch_b(k_debug_set_word, dim3(1), dim3(64), 0, s, ws.info + 0, 1);
        else cip_launch_b(k_debug_set_word, dim3(1), dim3(64), 0, s, ws.info + 3, -9);
        CIP_HIP_CHECK(hipGetLastError());
    }
    return rc;
}
static int ldlt_factor_body(hipStream_t s, double *K, int Npad, long ld, const LdltWorkspace &ws) {
    if (Npad % CIP_NB) { cip_set_error("ldlt: N must be a multiple of 128"); return -1; }
    int rc;
    // [0] bad pivot, [1] sweep bail-out, [2] dead pivot, [3] fused-launch wait timed out; from word 16: `ready` counters
    if ((rc = zero_fill(s, ws.info, 64 + 12 * (size_t)(Npad / CIP_NB)))) return rc;
    if ((rc = ensure_x_zeroed(s, Npad, ws))) return rc;
    const int Bs = ws.Bs;
    LdltProfile *prof_this = ws.prof ? ws.prof : (cip_tl_builder ? nullptr : g_tl_prof);
    if (prof_this && (prof_this->nfact++ % prof_this->stride) != 0) prof_this = nullptr;      // a sampled profile skips this factorisation
    LdltSide *sd = nullptr;
    int Jdone = 0;                                           // solve blocks [0, Jdone) are prepared or being prepared on the side stream
    // serial right-looking schedule: panels of the outer block, then ONE trailing update
    for (int C0 = 0, wblk = 0; C0 < Npad; C0 += wblk) {
        wblk = outer_block_width(Npad, C0);
        const bool last = C0 + wblk >= Npad;
        // side preparation only in a wide last block: at least one solve-block boundary strictly inside it or at its start,
        // and more than one solve block in the matrix
        if (last && side_prep_mode() && !cip_in_batch() && !cip_tl_builder && Bs > CIP_NB && Npad / Bs >= 2 &&
            wblk >= 2 * cip_ldlt_outer_block_for(Npad) && wblk > Bs && (sd = side_get(ws))) {
            if ((rc = cip_ldlt_side_join(s, ws, -1))) return rc;      // (a factorisation nobody solved with)
            sd->nfork = 0; sd->waited = 0;
            fuse_env();
            // panel by panel; from `side_from` columns before the end on, fork whenever the columns of another solve block
            // are final (the first fork takes everything to its left).  The wide block's FIRST panel launches carry ~900
            // update tiles each and fill the chip: side work beside them costs the chain what it saves (same-session A/B,
            // forking from the block's start: 187.5 against 188.7 KKT solves/s); the last ones are a chain on an idle chip.
            static const int side_from = [] { const char *e = getenv("CIP_SIDE_PREP_FROM"); return e ? atoi(e) : 2048; }();
            for (int c = C0; c < Npad; ) {
                // (forks no finer than 1024 columns: with a 512-wide solve block -- order 4608, config 3 -- a fork per block was five
                //  groups of tiny launches per factorisation, 0.2 ms of side work more than the serial preparation)
                if (c % (Bs > 1024 ? Bs : 1024) == 0 && Npad - c <= side_from && c / Bs > Jdone) {
                    if ((rc = side_fork(sd, s, K, Npad, ld, ws, Jdone, c / Bs))) return rc;
                    Jdone = c / Bs;
                }
                int cend = (c / Bs + 1) * Bs;               // next solve-block boundary
                if (cend > Npad) cend = Npad;
                // the panels [c, cend) of the wide block: factor_outer_panels on a sub-range keeps the block's W buffer layout
                if ((rc = factor_outer_panels(s, K, Npad, ld, ws, ws.Wbuf, C0, wblk, (c - C0) / CIP_NB, (cend - C0) / CIP_NB))) return rc;
                c = cend;
            }
            continue;                                        // (the last block has no trailing update)
        }
        if ((rc = factor_outer_panels(s, K, Npad, ld, ws, ws.Wbuf, C0, wblk))) return rc;
        const int r0 = C0 + wblk;
        if (r0 < Npad) {
            GemmArgs g = {};
            g.A = ws.Wbuf + r0; g.lda = Npad;
            g.B = K + r0 + (long)C0 * ld; g.ldb = ld;
            g.C = K + r0 + (long)r0 * ld; g.ldc = ld;
            g.M = Npad - r0; g.N = Npad - r0; g.K = wblk; g.alpha = -1.0; g.lower = 1;
            int epi = EPI_ACCUM;
            if (C0 == 0 && ws.lazyC) {                    // the trailing matrix is still in Q: read it from there, write K
                epi = EPI_LAZYC;
                g.Qin = ws.lazyC + r0 + (long)r0 * ws.lazy_ld; g.ldq = ws.lazy_ld; g.Cdiag = ws.lazy_diag + r0;
            }
            LdltProfile *prof = prof_this;
            if (prof) {
                if ((rc = prof_event(prof, s))) return rc;
                const double r = (double)(Npad - r0);
                const double live = cip_in_batch() ? (double)__builtin_popcountll(cip_tl_bz.mask) : 1.0;
                prof->flops.push_back(live * r * (r + 1.0) * (double)wblk);   // 2 flop/MAC on the lower triangle
            }
            if ((rc = cip_launch_gemm(s, epi, g))) return rc;
            if (prof && (rc = prof_event(prof, s))) return rc;
        }
    }
    if (sd && sd->nfork > 0) {
        // The LAST group -- the solve blocks whose columns the last panels produced -- runs under the first solve's forward sweep,
        // which waits for it at the last block (cip_ldlt_side_join).  Round 5 measured it on the factorisation's own stream instead
        // (CIP_SIDE_LAST_MAIN=1: no fork, no join, the group's seven launches alone on the chip: 63 us with the 16x16-tile doubling
        // products): the first solve 0.245 -> 0.223 ms, the step 0.03 ms SLOWER.  On the side stream by default.
        static const int last_main = [] { const char *e = getenv("CIP_SIDE_LAST_MAIN"); return e ? atoi(e) : 0; }();
        if (last_main && Npad / Bs - Jdone == 1) return build_solve_blocks(s, K, Npad, ld, ws, Jdone, Npad / Bs);
        return side_fork(sd, s, K, Npad, ld, ws, Jdone, Npad / Bs);      // joined by the solves (cip_ldlt_side_join)
    }
    if (ws.no_prep) return 0;
    return build_solve_blocks(s, K, Npad, ld, ws, Jdone, Npad / Bs);
}

// ---------------------------------------------------------------------------
// Solves
__global__ __launch_bounds__(256) void k_scale_vec(int n, const double *x, const double *d, double *y, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO3(cb, x, d, y);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i] * d[i];
}

// Two launches per block step (the diagonal block's product, then the update of everything below / above it): 4 N / Bs - 1
// dependent launches per solve, ~4 us of dependency latency each -- at N = 8192 the sweeps move their 0.67 GB in 31 launches
// and 0.27 ms, 30 % of the HBM peak, and it is the launch chain, not the traffic, that is the bound.  Both attempts to put
// the chain INSIDE a launch lost against it: round 2's one persistent kernel per sweep (per-workgroup flags and tickets:
// 0.51 ms per solve4x4), and round 3's one launch per block step (the Bs / 4 workgroups that own the next diagonal
// block's inputs wait for each other on a counter and compute its product while the rest of the grid streams the update:
// bit-identical, 17 launches instead of 31 -- and 0.36 ms: the fan-in of 256 workgroups plus the coherent reload under a
// streaming load costs ~11 us per step, more than the launch it replaces, as MI355X_MICROARCH.md's price list says).
//
// Round 5: ONE launch per block step.  With the neighbour blocks pre-multiplied by the block inverse (k_solve_premul),
//   forward   y_J = X_J r_J - (X_J L_{J,J-1}) y_{J-1},   r_J = b_J - sum_{I < J-1} L_{JI} y_I,
// nothing in step J depends on anything younger than y_{J-1}: the launch that forms y_J (part a) also applies y_{J-1} to the rows
// below block J (part b), and block J's own share of that update is the pre-multiplied term.  Mirror image for L' x = D^-1 y.
// Npad / Bs launches per sweep instead of 2 Npad / Bs - 1, the D^-1 scaling rides on the forward sweep's stores; the bytes are
// the same (MT_J / PT_J are read instead of L_{J,J-1} / L_{J+1,J}), the triangular halves of X no longer read as zeros.
struct SolveStepArgs {
    const double *T, *M, *v, *u;      // (a) out[j] = T[:, j] . v  -  M[:, j] . u   for j < Bs (T, M: Bs x Bs, pitch Bs; M == NULL: first step)
    double *out, *out2; const double *dsc;   // out2 != NULL: out2[j] = out[j] * dsc[j]
    int Bs, tri;                      // tri 1: T[:, j] is zero below row j (XT_J); 2: above row j (X_J)
    const double *Bm; long ldb; double *tgt; int ncols;   // (b) tgt[c] -= Bm[0:Bs, c] . u   for c < ncols
};
// this lane's share of a[r0:r1] . x[r0:r1]; r0, r1 multiples of 128, 16-byte aligned operands; four 16-byte loads in flight
__device__ __forceinline__ double lane_dot(const double *a, const double *x, int r0, int r1, int lane) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = r0 + lane * 2;
    for (; i + 384 < r1; i += 512) {
        const v2d a0 = *(const v2d *)(a + i), a1 = *(const v2d *)(a + i + 128), a2 = *(const v2d *)(a + i + 256), a3 = *(const v2d *)(a + i + 384);
        const v2d x0 = *(const v2d *)(x + i), x1 = *(const v2d *)(x + i + 128), x2 = *(const v2d *)(x + i + 256), x3 = *(const v2d *)(x + i + 384);
        s0 = fma(a0.x, x0.x, s0); s1 = fma(a0.y, x0.y, s1);
        s2 = fma(a1.x, x1.x, s2); s3 = fma(a1.y, x1.y, s3);
        s0 = fma(a2.x, x2.x, s0); s1 = fma(a2.y, x2.y, s1);
        s2 = fma(a3.x, x3.x, s2); s3 = fma(a3.y, x3.y, s3);
    }
    for (; i < r1; i += 128) {
        const v2d av = *(const v2d *)(a + i), xv = *(const v2d *)(x + i);
        s0 = fma(av.x, xv.x, s0); s1 = fma(av.y, xv.y, s1);
    }
    return (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void k_solve_step(SolveStepArgs a, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, a.T, a.M, a.v, a.u);
    CIP_BO5(cb, a.out, a.out2, a.dsc, a.Bm, a.tgt);
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j < a.Bs) {
        const int r0 = a.tri == 1 ? 0 : (j & ~127), r1 = a.tri == 1 ? ((j + 128) & ~127) : a.Bs;
        double s = lane_dot(a.T + (long)j * a.Bs, a.v, r0, r1, lane);
        if (a.M) s -= lane_dot(a.M + (long)j * a.Bs, a.u, 0, a.Bs, lane);
        s = cip_wave_sum(s);
        if (lane == 0) {
            a.out[j] = s;
            if (a.out2) a.out2[j] = s * a.dsc[j];
        }
        return;
    }
    const int c = j - a.Bs;
    if (c >= a.ncols) return;
    const double s = cip_wave_sum(lane_dot(a.Bm + (long)c * a.ldb, a.u, 0, a.Bs, lane));
    if (lane == 0) a.tgt[c] -= s;
}
static int solve_fused(hipStream_t s, const double *K, int Npad, long ld, const LdltWorkspace &ws, double *rhs) {
    const int Bs = ws.Bs, nbk = Npad / Bs;
    const double *X = (Bs == CIP_NB) ? ws.Linv : ws.X, *XT = (Bs == CIP_NB) ? ws.LinvT : ws.XT;
    const size_t bs2 = (size_t)Bs * Bs;
    double *y = ws.tmp, *z = ws.zbuf;
    int rc;
    for (int J = 0; J < nbk; ++J) {
        const long C0 = (long)J * Bs;
        if ((rc = cip_ldlt_side_join(s, ws, J))) return rc;
        SolveStepArgs a = {};
        a.T = XT + J * bs2; a.v = rhs + C0; a.out = y + C0; a.out2 = z + C0; a.dsc = ws.dinv + C0; a.Bs = Bs; a.tri = 1;
        if (J > 0) {
            a.M = ws.MT + J * bs2; a.u = y + C0 - Bs;
            a.ncols = Npad - (int)C0 - Bs;
            a.Bm = K + (C0 - Bs) + (C0 + Bs) * ld; a.ldb = ld; a.tgt = rhs + C0 + Bs;
        }
        cip_launch_b(k_solve_step, dim3((unsigned)((Bs + a.ncols + 3) / 4)), dim3(256), 0, s, a);
    }
    for (int J = nbk - 1; J >= 0; --J) {
        const long C0 = (long)J * Bs;
        SolveStepArgs a = {};
        a.T = X + J * bs2; a.v = z + C0; a.out = rhs + C0; a.Bs = Bs; a.tri = 2;
        if (J < nbk - 1) {
            a.M = ws.PT + J * bs2; a.u = rhs + C0 + Bs;
            a.ncols = (int)C0;
            a.Bm = K + (C0 + Bs); a.ldb = ld; a.tgt = z;
        }
        cip_launch_b(k_solve_step, dim3((unsigned)((Bs + a.ncols + 3) / 4)), dim3(256), 0, s, a);
    }
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_ldlt_solve(hipStream_t s, const double *K, int Npad, long ld, const LdltWorkspace &ws, double *rhs) {
    const int Bs = ws.Bs;
    const int nbk = Npad / Bs;
    const double *X = (Bs == CIP_NB) ? ws.Linv : ws.X;
    const double *XT = (Bs == CIP_NB) ? ws.LinvT : ws.XT;
    const size_t bs2 = (size_t)Bs * Bs;
    int rc;
    double *y = ws.tmp, *z = ws.zbuf;
    if (ws.fused && !(ld & 1) && !(((uintptr_t)K | (uintptr_t)rhs) & 15)) return solve_fused(s, K, Npad, ld, ws, rhs);
    for (int J = 0; J < nbk; ++J) {
        const long C0 = (long)J * Bs;
        if ((rc = cip_ldlt_side_join(s, ws, J))) return rc;          // (a no-op once every group has been waited for)
        if ((rc = cip_gemv_t(s, Bs, Bs, 1.0, XT + J * bs2, Bs, rhs + C0, 0.0, y + C0))) return rc;
        const int below = Npad - (int)C0 - Bs;
        if (below > 0 &&
            (rc = cip_gemv_t(s, Bs, below, -1.0, K + C0 + (C0 + Bs) * ld, ld, y + C0, 1.0, rhs + C0 + Bs)))
            return rc;
    }
    cip_launch_b(k_scale_vec, dim3((Npad + 255) / 256), dim3(256), 0, s, Npad, y, ws.dinv, z);
    for (int J = nbk - 1; J >= 0; --J) {
        const long C0 = (long)J * Bs;
        if ((rc = cip_gemv_t(s, Bs, Bs, 1.0, X + J * bs2, Bs, z + C0, 0.0, rhs + C0))) return rc;
        if (C0 > 0 && (rc = cip_gemv_t(s, Bs, (int)C0, -1.0, K + C0, ld, rhs + C0, 1.0, z))) return rc;
    }
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
