// The opaque handle behind the C ABI (one KKT system, resident on one GPU).
#pragma once
#include "cip_internal.h"
#include <vector>

struct cip_handle {
    int n = 0, m = 0, p = 0, ncones = 0, route = 0;
    int N = 0, Npad = 0;            // KKT order for the route, and padded to a multiple of 128
    int npad = 0, mpad = 0;         // n rounded up to 128, m rounded up to 16 (>= 16)
    int nq = 0, nqpad = 0;          // number of Q cones, rounded up to 16 (>= 16)
    hipStream_t stream = nullptr;
    int device = 0;
    // Device memory of the handle.  Normally one hipMalloc per buffer; a handle of a lock-step batch carves its buffers,
    // in creation order, out of its slab of the batch's arena (api.hip: cip_handle_alloc) so that problem z's buffers
    // sit at problem 0's addresses + z * stride.  alloc_bytes counts what creation asked for (256-byte granules).
    char *arena = nullptr;
    size_t arena_cap = 0, arena_used = 0, alloc_bytes = 0;
    bool arena_overflow = false;

    // ---- problem data, device resident for the lifetime of the handle (level 1)
    double *Q = nullptr;            // n x n, ld n
    double *symv_ws = nullptr;      // partial-sum tables of the symmetric mat-vec (n a multiple of 128, n >= 2048), else null
    bool A_sparse = false;
    bool A_one_per_row = false;     // CSR A with at most one entry per row (A = I, bound constraints): A'(F'F)^-1 A is diagonal for R cones
    double *kdiag = nullptr;        // n doubles: the Schur route's diagonal of K beyond the first outer block when the copy of Q is lazy
    int all_r = -1;                 // every cone is an R cone (F diagonal): solve4x4 takes the fused element-wise path; -1: not looked at yet
    double *A = nullptr;            // m x n, ld m            (dense A only)
    double *At = nullptr;           // npad x mpad, ld npad   (dense A only; zero padded)  At[i + r*npad] = A[r,i]
    int A_nnz = 0;
    int *A_rp = nullptr, *A_ci = nullptr; double *A_v = nullptr;   // CSR of A  (m rows)
    int *T_rp = nullptr, *T_ci = nullptr; double *T_v = nullptr;   // CSR of A' (n rows)
    int *row_cone = nullptr;        // m ints: cone index of every row of A
    // CSR A with S cones (round 4): the rows of the S cones as a dense transposed block, and its scaled image (Schur route)
    double *AtS = nullptr, *WtS = nullptr; int mS = 0, mSpad = 0;     // npad x mSpad each, ld npad; cone c's rows at columns [aoff, aoff + dim)
    double *G = nullptr;            // p x n, ld p
    double *Gt = nullptr;           // n x p, ld n

    // host staging of the small tables and of the CSR arrays: asynchronous uploads read them after the call has returned
    std::vector<int> st_rp, st_ci, st_trp, st_tci, st_rowcone, st_sidx, st_small, st_bigq, st_ritems, st_packq;
    std::vector<double> st_av, st_tv;
    bool staging_live = false;

    // ---- cones / scaling (level 2 input)
    std::vector<ConeDesc> h_cones;
    std::vector<WorkItem> h_items;
    ConeSet cs = {};

    // ---- KKT matrix + factor (level 2 output)
    double *K = nullptr; long ldk = 0;
    double *Wt = nullptr;           // npad x mpad   At * F^-1           (dense-A Schur route)
    double *syrk_ws = nullptr; int syrk_n = 1, syrk_len = 0;   // split-K images of the Schur formation (few output tiles, long K: config 4)
    double *Gm = nullptr;           // npad x nqpad  rank-1 columns of the Q cones (sparse-A Schur route)
    void *ws_base = nullptr; LdltWorkspace ws = {};
    struct LdltSide *ldlt_side = nullptr;   // side stream + events of the overlapped solve preparation (ldlt.hip), created on first use
    bool assembled = false, factored = false;
    // static regularisation K + delta diag(+1 .. -1 ..), delta = reg_rel * max|K_ii|: 0 until a factorisation meets a
    // bad pivot (auto_reg), then kept for the lifetime of the handle; the loops' iterative refinement absorbs it
    double reg_rel = 0.0;
    bool auto_reg = true;
    int n_regularized = 0;
    int n_chain_fallbacks = 0;     // factorisations redone with the three-launch chain after an in-launch wait of the fused one gave up
    int x_zeroed = 0;               // block-inverse storage zero-initialised
    // pivot flag of the last factorisation: read back asynchronously into pinned host memory, resolved lazily
    // (api.hip: factor_resolve) so that cip_factor never waits for the GPU
    int *info_host = nullptr;
    int info_seq = 0;                 // sequence number of the last factorisation's flag read-back (info_host[4] == info_seq: landed)
    bool info_pending = false;
    int spec_solves = 0;            // solves enqueued while the flag was still in flight
    // true once a factorisation of this handle has been resolved clean (or the regularised mode is on): until then the
    // *_dev solves WAIT for the pivot flag instead of going ahead speculatively -- the factorisation that meets a bad
    // pivot is almost always the first one (LPs, singular Q with free variables), and a caller doing factor -> solve*_dev
    // must not get rc 0 and a solution of a broken factor
    bool pivots_verified = false;

    // ---- scratch
    double *rhs = nullptr;          // Npad
    double *mt1 = nullptr, *mt2 = nullptr, *mt3 = nullptr;   // m-vectors
    double *nt1 = nullptr;          // n-vector
    double *pt1 = nullptr;          // p-vector
    double *dot_scratch = nullptr; void *dot_ptrs = nullptr;
    double *stage = nullptr;        // device staging for the host-pointer entry points: 2*(n+p+m) doubles
    double *ref = nullptr;          // right-hand side / residual / correction of the refinement inside solve3x3 (regularised factor only)
    double *c2x2 = nullptr;         // m-vector: discarded third component of a regularised 2x2 solve
    double *drv = nullptr;          // vectors of the native interior-point loop (cip_conicip), allocated on first use

    // ---- hipGraphs of the two launch-bound inner loops of small systems (opt-in, CIP_GRAPH=1; api.hip: graph_run)
    hipGraphExec_t gx_factor = nullptr, gx_solve = nullptr;
    // what the recorded factorisation baked in besides the handle's fixed pointers: where the first trailing update reads
    // its C operand (ws.lazyC: Q while the copy is lazy, else null).  A replay under another state would read stale data
    // (ADVICE r3): factor_enqueue drops the graph and records a new one when it differs.
    const double *gx_factor_lazyC = nullptr;
    int graph_state = 0;            // 0 undecided, 1 in use, -1 off (null stream, large system, capture failed, CIP_GRAPH=0)

    // ---- stats
    double n_factor = 0, n_solve = 0, ms_assemble = 0, ms_ldlt = 0, flops_ldlt = 0;
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
};

int cip_lazy_copy_set(int on);       // assemble.hip
int cip_scatter_AtS(cip_handle *h);   // assemble.hip: CSR A with S cones: their rows of A' as a dense block (h->AtS)
int cip_assemble(cip_handle *h, bool lazy_ok = false);     // assemble.hip; lazy_ok: the caller factors right away (see assemble_schur)
int cip_handle_alloc(cip_handle *h, void **out, size_t bytes);      // api.hip
int cip_create_in_arena(const struct cip_problem *pr, char *slab, size_t cap, hipStream_t stream, cip_handle **out);   // api.hip
size_t cip_driver_bytes(const cip_handle *h);                        // driver.hip: vectors of the interior-point loop
// api.hip: resolve the pivot flag of the last factorisation (wait = 0: only if its read-back has already landed)
int cip_factor_resolve(cip_handle *h, int wait);

// api.hip: roctx ranges (no-ops when the marker library is not present)
void cip_range_push(const char *name);
void cip_range_pop(void);
