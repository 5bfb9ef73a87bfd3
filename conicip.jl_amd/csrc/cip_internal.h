// Internal declarations shared by the libcipkkt translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define CIP_NB 128            // inner panel width == GEMM tile edge
#define CIP_KT 16             // GEMM k-tile

// ---------------------------------------------------------------- error plumbing
void cip_set_error(const char *fmt, ...);
#define CIP_HIP_CHECK(expr)                                                       \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) {                                                  \
            cip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                          __FILE__, __LINE__);                                    \
            return -3;                                                            \
        }                                                                         \
    } while (0)

// ---------------------------------------------------------------- launches that can be recorded into a hipGraph
// The LDL' factorisation and the triangular solves of a handle are fixed launch sequences (same pointers, same sizes
// every time).  For small systems they are recorded ONCE as an explicit graph -- kernel nodes added one after the other
// through cip_launch while a thread-local builder is active, no stream capture involved (capture in thread-local mode
// was invalidated at random by the other host threads of a batch on ROCm 7.0) -- and replayed with one hipGraphLaunch.
#include <tuple>
struct CipGraphBuilder { hipGraph_t graph; hipGraphNode_t last; bool have_last; bool ok; };
extern thread_local CipGraphBuilder *cip_tl_builder;
template <typename Tuple, size_t... I>
inline void cip_tuple_ptrs(Tuple &t, void **out, std::index_sequence<I...>) { ((out[I] = (void *)&std::get<I>(t)), ...); }
template <typename... KArgs, typename... Args>
inline void cip_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, Args... args) {
    if (!cip_tl_builder) { hipLaunchKernelGGL(kernel, grid, block, shmem, s, args...); return; }
    CipGraphBuilder *b = cip_tl_builder;
    std::tuple<KArgs...> targs{static_cast<KArgs>(args)...};
    void *ptrs[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1];
    cip_tuple_ptrs(targs, ptrs, std::index_sequence_for<KArgs...>{});
    hipKernelNodeParams kp = {};
    kp.func = (void *)kernel; kp.gridDim = grid; kp.blockDim = block; kp.sharedMemBytes = (unsigned)shmem;
    kp.kernelParams = ptrs; kp.extra = nullptr;
    hipGraphNode_t node;
    if (hipGraphAddKernelNode(&node, b->graph, b->have_last ? &b->last : nullptr, b->have_last ? 1 : 0, &kp) != hipSuccess) { b->ok = false; return; }
    b->last = node; b->have_last = true;
}

// ---------------------------------------------------------------- lock-step batches (batch dimension = blockIdx.z)
// A lock-step batch (csrc/lockstep.hip) holds B problems of identical shape whose device buffers were carved, in the
// same order, out of equally sized slabs of ONE arena: whatever pointer problem 0 uses, problem z uses the same pointer
// + z * stride.  The host code of the library therefore runs ONCE, on problem 0's handle, while a thread-local batch
// context is active: cip_launch_b multiplies grid.z by B and appends a CipBatch argument; the kernel drops out when its
// problem is masked off and shifts its pointer arguments (CIP_BOFF).  Outside a batch the context is {B = 1}: stride 0,
// mask 1 -- the same kernels, bit for bit.
struct CipBatch { long stride; unsigned long long mask; };            // stride in BYTES; bit z of mask: problem z takes part
struct CipBatchCtx {
    int B; long stride; unsigned long long mask;
    double *gather_dev; double *gather_host;       // B x CIP_GATHER doubles (device / pinned host): per-problem scalar results
};
#define CIP_GATHER 64
#define CIP_BATCH_MAX 64
struct CipScal64 { double v[CIP_BATCH_MAX]; };     // one scalar per problem, passed by value
extern thread_local CipBatchCtx cip_tl_bz;
inline bool cip_in_batch() { return cip_tl_bz.B > 1; }
template <typename T>
__device__ __forceinline__ T *cip_bo(T *p, const CipBatch &cb) { return p ? (T *)((char *)p + (long)blockIdx.z * cb.stride) : p; }
#define CIP_BATCH_GUARD(cb) do { if (!(((cb).mask >> blockIdx.z) & 1ull)) return; } while (0)
#define CIP_BO1(cb, a) a = cip_bo(a, cb)
#define CIP_BO2(cb, a, b) CIP_BO1(cb, a); CIP_BO1(cb, b)
#define CIP_BO3(cb, a, b, c) CIP_BO2(cb, a, b); CIP_BO1(cb, c)
#define CIP_BO4(cb, a, b, c, d) CIP_BO2(cb, a, b); CIP_BO2(cb, c, d)
#define CIP_BO5(cb, a, b, c, d, e) CIP_BO4(cb, a, b, c, d); CIP_BO1(cb, e)
#define CIP_BO6(cb, a, b, c, d, e, f) CIP_BO4(cb, a, b, c, d); CIP_BO2(cb, e, f)
#define CIP_BO7(cb, a, b, c, d, e, f, g) CIP_BO4(cb, a, b, c, d); CIP_BO3(cb, e, f, g)
#define CIP_BO8(cb, a, b, c, d, e, f, g, h) CIP_BO4(cb, a, b, c, d); CIP_BO4(cb, e, f, g, h)
template <typename... KArgs, typename... Args>
inline void cip_launch_b(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, Args... args) {
    const CipBatchCtx &c = cip_tl_bz;
    CipBatch cb{c.B > 1 ? c.stride : 0, c.B > 1 ? c.mask : 1ull};
    if (c.B > 1) grid.z *= (unsigned)c.B;
    cip_launch(kernel, grid, block, shmem, s, args..., cb);
}

// ---------------------------------------------------------------- GEMM (gemm_f64.hip)
enum { EPI_ACCUM = 0, EPI_SYRKQ = 2, EPI_STORE = 3, EPI_LAZYC = 4 };     // EPI_LAZYC: C = Cin + alpha acc with Cin = Qin (ldq), Cdiag[i] on its diagonal

struct GemmArgs {
    const double *A; long lda;   // M x K, element (i,k) at A[i + k*lda]
    const double *B; long ldb;   // N x K, element (j,k) at B[j + k*ldb]
    double *C; long ldc;         // M x N
    int M, N, K;                 // M, N multiples of 128; K multiple of 16
    double alpha;
    int lower;                   // 1: only tiles with bi >= bj (M == N); 2 / 3 (batched overwrite form): only 64-tiles with bi <= bj / bi >= bj
    // EPI_SYRKQ: C = Qin + acc for i,j < nvalid (lower tiles)
    const double *Qin; long ldq; int nvalid;
    // batching (EPI_ACCUM only): grid.y x grid.z independent problems, pointer strides in doubles
    int by, bz;
    long sAy, sAz, sBy, sBz, sCy, sCz;
    int overwrite;               // EPI_ACCUM: C = alpha*acc instead of C += alpha*acc
    double *Ct; long ldct, sCty, sCtz;   // 128-tile EPI_ACCUM only, optional: the result is also stored transposed, Ct[j + i*ldct]
    const double *Cdiag;         // EPI_LAZYC: the diagonal of Cin
    int force64;                 // plain accumulate form: quarter tiles (k_gemm_nt_64) whatever the tile count
    int tiny16;                  // batched overwrite form: one 16x16 tile per workgroup, k split over its waves (k_gemm_nt_16_batched; K % 128 == 0)
    // EPI_SYRKQ with few output tiles and a long K (round 4): the k range is cut into `ksplit_n` slices of `ksplit_len` columns, every
    // slice's product goes to its own M x M image in `ksplit_ws`, a second launch adds them up in slice order together with Qin
    double *ksplit_ws; int ksplit_n, ksplit_len;
};
// how cip_launch_gemm(EPI_SYRKQ) would split an (M x M, K) Schur formation: number of slices (1: no split) and their length
int cip_syrk_split(int M, int K, int *len);
int cip_launch_gemm(hipStream_t s, int epi, const GemmArgs &g);

// ---------------------------------------------------------------- small results back to the host (vecops.hip)
// The loop's scalars (max-step minima, dot products, the lock-step gather) are a few doubles the host must SEE before it can go
// on: a D2H copy into pageable memory is a staging kernel in front of the wait (eleven of them per iteration in the kernel trace).
// Instead: the reduction kernel stores straight into host-mapped pinned memory (one per host thread), the host records an event
// behind it and spins on it (0.0561 -> 0.0552 s to converge at n = 8192; spinning itself measures the same as a blocking
// hipStreamSynchronize on this stack: CIP_SPIN_WAIT=0).
struct CipHostScratch { double *host; double *dev; };       // 512 doubles; host == what the GPU wrote once cip_wait(s) has returned
int cip_host_scratch(CipHostScratch *out);
int cip_wait(hipStream_t s);

// ---------------------------------------------------------------- LDL' (ldlt.hip)
struct LdltProfile;           // optional per-launch event timing of the trailing-update kernel (ldlt.hip)
LdltProfile *cip_ldlt_profile_create(void);
void cip_ldlt_profile_destroy(LdltProfile *p);
void cip_ldlt_profile_stride(LdltProfile *p, int stride);     // time every stride-th factorisation only (default 1: all)
// synchronises; adds the elapsed time of every recorded trailing-update launch to the totals
int cip_ldlt_profile_collect(LdltProfile *p, double *launches, double *ms, double *flops);
int cip_ldlt_profile_thread(int enabled);      // a profile of the calling host thread (lock-step batches: bench.py, config 5)
int cip_ldlt_profile_thread_collect(double *launches, double *ms, double *flops);
// per-thread event timing of further dominant kernels (ldlt.hip): slot 1 Schur formation, slot 2 large-S Jacobi
#define CIP_PROF_SLOTS 3
#define CIP_PROF_SYRK 1
#define CIP_PROF_JACOBI 2
int cip_prof_slot_enable(int slot, int enabled);
int cip_prof_slot_collect(int slot, double *launches, double *ms, double *work);
int cip_prof_slot_begin(int slot, hipStream_t s, double work);
int cip_prof_slot_end(int slot, hipStream_t s);

// Expected pivot signs of a quasi-definite matrix in its static order: positive for columns in [p0, p1) and for the
// identity padding (>= N), negative elsewhere; p0 < 0: unknown, no sign check (stand-alone LDL' entry points)
struct PivotSigns { int p0, p1, N; };

struct LdltWorkspace {        // carved out of one device allocation
    double *Wbuf;             // Npad x NBO      (W = L*D panels of the current outer block)
    double *Linv;             // (Npad/128) x 128 x 128   inverse of each unit-lower diagonal block
    double *LinvT;            // same, transposed
    double *Xm;               // (Npad/128) x 8 x 256   inverses of the 16x16 unit-lower micro-blocks
    int Bs;                   // solve block: largest of {1024,512,256,128} dividing Npad
    double *X, *XT;           // (Npad/Bs) x Bs x Bs   inverse (and its transpose) of each Bs x Bs unit-lower diagonal block
    double *Tt;               // (Npad/Bs) x (Bs/2)^2  scratch of the block-inverse doubling
    // one launch per block step (round 5, ldlt.hip: k_solve_step): the pre-multiplied off-diagonal neighbours of every solve block,
    //   MT_J = (X_J L_{J,J-1})'  (J >= 1, forward sweep)   and   PT_J = L_{J+1,J} X_J  (J <= nbk - 2, backward sweep),
    // (Npad/Bs) x Bs x Bs each; `fused` == 0: two launches per block step, MT / PT unused (NULL)
    double *MT, *PT;
    int fused;
    double *zbuf;             // Npad scratch
    int *x_zeroed;            // host flag owned by the handle (NULL: zero X/XT on every factorisation)
    double *dinv;             // Npad   1/d
    double *dvec;             // Npad   d
    double *tmp;              // Npad   scratch vector for the solves
    int *info;                // device int: 0 ok, >0 = 1-based column of a bad pivot (zero, non-finite, or of the wrong sign)
    PivotSigns signs;
    LdltProfile *prof;        // host object or NULL
    // lazy copy (assemble.hip: assemble_schur): K beyond the first outer block has NOT been filled; the first trailing update
    // reads its C operand from lazyC (the problem's Q, column-major, leading dimension lazy_ld) with lazy_diag[i] on the
    // diagonal, and writes K.  Consumed (cleared) by the next cip_ldlt_factor.
    const double *lazyC; long lazy_ld; const double *lazy_diag;
    // solve preparation beside the panel chain of the last (wide) outer block (ldlt.hip: cip_ldlt_factor): where the owner
    // keeps the side stream / events, created on first use (NULL: always behind the factorisation, on its own stream)
    struct LdltSide **side;
    int no_prep;              // 1: the factor only (no block inverses for solves) -- inertia checks (sdp_large.hip)
    int unfused;              // 1: three launches per panel whatever the process-wide chain form (no in-launch wait: set after one gave up, api.hip)
};
struct LdltSide;
void cip_ldlt_side_destroy(struct LdltSide *sd);
int cip_ldlt_side_join(hipStream_t s, const struct LdltWorkspace &ws, int J);   // ldlt.hip: wait for the side stream's solve preparation (J < 0: all of it)
int cip_kernels_init(void);                // diag.hip: one-time kernel attributes (before any hipGraph capture)
int cip_ldlt_set_side_prep(int on);       // 1 (default): solve preparation beside the last outer block's panel chain; returns the previous setting
int cip_debug_chain_giveup_set(int n);    // test hook: the next n fused-chain factorisations report an in-launch wait that gave up; returns the previous count
int cip_ldlt_set_fused_chain(int on);     // 1 (default): diag + previous in-block update in one launch; returns the previous setting
int cip_solve_block(int Npad);
int cip_solve_block_max_set(int b);              // 128 | 256 | 512 | 1024 (0: query); returns the previous limit
int cip_solve_fused_set(int mode);               // 0 (default): two launches per block step; 1: one (pre-multiplied neighbours) for solve blocks <= 512; 2: always; < 0: query.  Returns the previous mode
extern thread_local int cip_tl_solve_block_max;  // > 0: this thread's limit for handles it creates
int cip_ldlt_fused_for(int Npad);                // does the CURRENT mode (cip_solve_fused_set) ask for the one-launch block steps at this order?
size_t cip_ldlt_ws_bytes(int Npad, int fused);     // fused: 1 / 0 = with / without MT, PT; < 0: reserve them whenever the order has two solve blocks
void cip_ldlt_ws_carve(void *base, int Npad, LdltWorkspace *ws, int fused);   // fused as passed to cip_ldlt_ws_bytes by the same owner (0 / 1)
int cip_ldlt_factor(hipStream_t s, double *K, int Npad, long ld, const LdltWorkspace &ws);
int cip_ldlt_solve(hipStream_t s, const double *K, int Npad, long ld, const LdltWorkspace &ws, double *rhs);
int cip_ldlt_outer_block(void);          // the knob: 0 = automatic
int cip_ldlt_outer_block_for(int Npad);  // NBO a factorisation of this order uses
void cip_ldlt_set_outer_block(int nbo);

// ---------------------------------------------------------------- cones (cones.hip)
struct ConeDesc {             // one per cone
    int type;                 // CIP_CONE_*
    int off;                  // offset into the m-vector
    int dim;                  // block length k
    int soff;                 // offset into the packed scaling storage
    int r;                    // S cone: matrix order
    int qidx;                 // Q cone: running index among Q cones (else -1)
    int item;                 // this cone's (first) slot in the partial-result array
    int aoff;                 // S cone: first column of its rows in the A' / A'F^-1 operands of the Schur scaling (= off for a dense A; CSR A:
                              // offset inside the dense block of the S cones' rows, api.hip)
};
struct WorkItem {             // unit of work (one 256-thread workgroup) of the per-cone kernels
    int cone;                 // index into ConeDesc[] (first cone of a pack)
    int start;                // first element (relative to the cone) -- R cones are chunked
    int len;                  // elements of the chunk; for a pack: number of cones
    int slot;                 // first slot of the per-cone partial results (max-step); a pack owns `len` consecutive slots
    int width;                // 0, or the lane-segment width of a pack of consecutive Q cones of dimension <= width <= 64
};
struct ConeSet {
    int ncones, nitems, nslots, m;
    ConeDesc *d_cones;        // device
    WorkItem *d_items;        // device
    double *d_scal;           // device packed scaling
    size_t scal_len;
    double *d_partial;        // nslots doubles (reductions)
    double *d_scalar;         // 8 doubles
    int has_S;
    int nbigq;                // Q cones of dimension > 64 (one work item each): the assembly gives them their own multi-workgroup kernels
    int nritems; int *d_ritems;   // the R cones' work items (chunks of <= 2048 rows): the Schur scaling splits each over 16 workgroups per row block
    int *d_bigq;              // device: their work-item indices (nbigq of them) -- those kernels' grids are per large cone, not per item
    int npackq; int *d_packq; // the packs of small Q cones (lane-segment items): k_scale_At's grid covers exactly these
    // S cones (sdp.hip)
    int ns, rmax, kmax;       // number of S cones, largest matrix order / vectorised length
    int *d_sidx;              // device: cone index of every S cone
    int sdp_slots;            // workgroup workspace slots
    double *d_sdpws;          // sdp_slots x 6 x rmax^2
    double *d_sdpvec;         // sdp_slots x 2 x kmax
    int *d_sdpflag;           // device int: Cholesky failure (iterate left the cone)
    // S cones of order >= 133 (sdp_large.hip): NT scaling, max-step and the Schur scaling take a multi-workgroup path
    int ns_small;             // number of S cones below that order, listed in d_sidx_small
    int *d_sidx_small;
    int nlarge;               // the others (at most CIP_MAX_LARGE_S)
    int large_cone[1024];     // their cone indices (CIP_MAX_LARGE_S entries)
    const ConeDesc *h_cones;  // host copy of the cone table (owned by the handle)
    struct LargeWs *lg;       // workspace of the large path (NULL when nlarge == 0)
};
#define CIP_MAX_LARGE_S 1024            // rounds 1-4: 8, round 5: 64, then 1024 -- every large cone costs 5 padded matrices (Rinv, Rinv', R, R', V) + its mat(a_i) images when they fit;
                                        // the cones' scalings are computed one after the other
#define CIP_LARGE_S_MIN 133
struct LargeWs;
int cip_sdp_large_create(int rmax_large, int nlarge, int ncols, LargeWs **out);
void cip_sdp_large_invalidate(LargeWs *w);      // the problem's A was replaced: drop the cached mat(a_i) images
void cip_sdp_large_destroy(LargeWs *w);
int cip_sdp_large_nt(hipStream_t s, LargeWs *w, const ConeDesc &cd, int li, const double *v, const double *sv, double *scal,
                     double *lambda, int *flag);
int cip_sdp_large_refresh(hipStream_t s, LargeWs *w, const ConeDesc &cd, int li, const double *scal);
int cip_sdp_large_apply(hipStream_t s, LargeWs *w, const ConeDesc &cd, int li, int mode, const double *x, double *out);
int cip_sdp_large_prod(hipStream_t s, LargeWs *w, const ConeDesc &cd, const double *x, const double *y, double *out);
int cip_sdp_large_div(hipStream_t s, LargeWs *w, const ConeDesc &cd, const double *x, const double *y, double *out, int *flag);
int cip_sdp_large_lanczos(int on);                  // sdp_large.hip: max-step eigenvalue by Lanczos (1), Lanczos + inertia certificate (2), tridiagonalisation (0); < 0 reads; returns the previous setting
int cip_sdp_large_cert_stats(hipStream_t s, struct LargeWs *w, int *out2);     // {certificates failed -> fallbacks taken, 0}
int cip_sdp_large_maxstep(hipStream_t s, LargeWs *w, const ConeDesc &cd, const double *x, const double *d, double scale,
                          double *partial, int side = 0);
bool cip_sdp_large_pairable(const LargeWs *w);            // the v- and s-side max-steps of a pair can run side by side
int cip_sdp_large_fork(hipStream_t s, LargeWs *w, hipStream_t *s2);
int cip_sdp_large_join(hipStream_t s, LargeWs *w);
int cip_sdp_large_scale_At(hipStream_t s, LargeWs *w, const ConeDesc &cd, int li, int n, const double *At, long ldat, double *Wt,
                           long ldwt);
int cip_cones_nt_scaling(hipStream_t s, const ConeSet &cs, const double *v, const double *sv, double *lambda);
int cip_cones_identity_scaling(hipStream_t s, const ConeSet &cs);
int cip_sdp_scaling_changed(hipStream_t s, const ConeSet &cs);     // sdp.hip: the packed scaling was replaced from outside
int cip_cones_apply(hipStream_t s, const ConeSet &cs, int mode, const double *x, double *out);
int cip_cones_prod(hipStream_t s, const ConeSet &cs, const double *x, const double *y, double *out);
int cip_cones_div(hipStream_t s, const ConeSet &cs, const double *x, const double *y, double *out);
int cip_cones_maxstep(hipStream_t s, const ConeSet &cs, const double *x, const double *d, double scale, double *alpha_host, int defer_slot = -1);
// the pair of the interior-point loop (src/ConicIP.jl:708-709, :881-882, :927-928): alpha_host2 = {maxstep(x1, d1), maxstep(x2, d2)},
// one wait; large S cones: the two sides on two streams
int cip_sdp_scale_At(hipStream_t s, const ConeSet &cs, int n, const double *At, long ldat, double *Wt, long ldwt);   // sdp.hip: the S cones' columns only
int cip_cones_maxstep2(hipStream_t s, const ConeSet &cs, const double *x1, const double *d1, const double *x2, const double *d2,
                       double scale, double *alpha_host2, int defer_slot = -1);
int cip_sdp_maxstep2(hipStream_t s, const ConeSet &cs, const double *x1, const double *d1, double *p1, const double *x2,
                     const double *d2, double *p2, double scale);
int cip_cones_identity(hipStream_t s, const ConeSet &cs, double *e);
// Wt[i, r] = (F^-T a_i)_r for every row i of At (n rows, ld ldat): Wt = At * F^-1
int cip_cones_scale_At(hipStream_t s, const ConeSet &cs, int n, const double *At, long ldat, double *Wt, long ldwt);

// ---------------------------------------------------------------- vector ops (vecops.hip)
int cip_symv_lower(hipStream_t s, int n, double alpha, const double *Q, long ldq, const double *x, double beta, double *y, double *ws);
int cip_gemv_t(hipStream_t s, int rows, int cols, double alpha, const double *A, long lda,
               const double *x, double beta, double *y);     // y[j] = alpha * sum_i A[i + j*lda] x[i] + beta y[j]
int cip_spmv_csr(hipStream_t s, int rows, const int *rowptr, const int *colind, const double *val,
                 double alpha, const double *x, double beta, double *y);
int cip_dots(hipStream_t s, int count, const double *const *x_host, const double *const *y_host,
             const int *len_host, double *scratch_dev, void *ptrs_dev, double *out_host);
int cip_axpby(hipStream_t s, int len, double alpha, const double *x, double beta, double *y);
// solve4x4 around the sweeps for all-R cone sets (vecops.hip): the element-wise launches fused, same arithmetic
int cip_s4_pre_r(hipStream_t s, int m, int n, int p, int Npad, const double *f, const double *rs, const double *lam, const double *rv,
                 const double *ry, const double *rw, double *t1_out, double *t_out, double *rhs, const int *T_rp, const int *T_ci, const double *T_v);
int cip_s4_post_r(hipStream_t s, int m, int n, int p, const double *f, const double *t, const double *rhs, const double *u_dense,
                  const int *A_rp, const int *A_ci, const double *A_v, double *dy, double *dw, double *dv, double *ds);
int cip_copy_neg(hipStream_t s, int len, const double *x, double *y, double scale);   // y = scale * x
int cip_axpby_ps(hipStream_t s, int len, const double *alpha_host, const double *x, double beta, double *y);   // batch: alpha per problem
int cip_zero(hipStream_t s, long len, double *y);                     // batch-aware memset(0) of doubles
int cip_copy(hipStream_t s, long len, const double *x, double *y);    // batch-aware device-to-device copy
struct cip_handle;
const double *cip_loop_all_r(cip_handle *h);     // api.hip: the packed scaling = diag F when every cone is an R cone (and CIP_LOOP_FUSED_R != 0), else NULL
// the element-wise chains of the interior-point loop, one kernel each (vecops.hip; f != NULL: all cones R, cone operations fused in)
int cip_loop_resid(hipStream_t s, int n, int m, int p, double *rl, const double *zs, const double *c, const double *d, const double *b,
                   const double *lam, const double *f, double *r0, double *Gy, double *Ays);
int cip_loop_corr(hipStream_t s, int n, int m, int p, const double *r0, const double *daff, const double *mb3, const double *e,
                  const double *f, const double *sigmu_host, double *r);
int cip_loop_refine(hipStream_t s, int n, int m, int p, double *rk, const double *dz, const double *r, const double *lam,
                    const double *mb2, const double *mb3, const double *f, double *rIr);

// ---- wave-level sums without LDS permutes (DPP row operations + v_permlane16/32_swap): every lane ends with the same bits.
// All lanes of the wave must be active.  (__shfl_xor is a ds_bpermute round trip per step: six dependent ones per 64-lane sum.)
#ifdef __HIPCC__
// sum over the 16 lanes of a DPP row, in every lane (xor 1, 2 as quad permutations, then the two mirrors)
template <int CTRL>
__device__ __forceinline__ double lz_dpp_add(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lz_sum16(double x) {
    x = lz_dpp_add<0xB1>(x); x = lz_dpp_add<0x4E>(x); x = lz_dpp_add<0x141>(x); return lz_dpp_add<0x140>(x);
}
// sum over the wave's four 16-lane rows, in every lane (v_permlane16_swap / v_permlane32_swap of two copies)
__device__ __forceinline__ double lz_sum_rows(double x) {
    {
        const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        x = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
// sum over the two 16-lane rows of a 32-lane half (rows 0, 1 or rows 2, 3), in every lane
__device__ __forceinline__ double lz_sum_row_pair(double x) {
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double cip_wave_sum(double x) { return lz_sum_rows(lz_sum16(x)); }
#endif
