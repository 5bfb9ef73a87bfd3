/*
 * libcipkkt -- MI355X (gfx950) native KKT-solve path for ConicIP-style
 * interior-point solvers.  C ABI: plain pointers and sizes only.
 *
 * This is the drop-in boundary for the reference's `kktsolver` plugin hook
 * (reference = MPF-Optimization-Laboratory/ConicIP.jl, paths relative to its
 * root):
 *
 *   level 1  kktsolver(Q, A, G, cone_dims)        src/ConicIP.jl:667
 *   level 2  solve3x3gen(F, F^-T)                 src/ConicIP.jl:682
 *   level 3  solve3x3(x, y, z) -> (a, b, c)       src/ConicIP.jl:688
 *
 * which the reference documents at src/ConicIP.jl:432-466 and
 * docs/src/guides/kkt_solvers.md:84-115, and implements three times in
 * src/kktsolvers.jl (kktsolver_qr :18-58, kktsolver_sparse :180-270,
 * pivot(kktsolver_2x2) :281-349).  The Julia-side `ccall` shim that binds these
 * entry points is shown in INTEGRATION.md.
 *
 * Conventions
 *   - all matrices column-major (Julia / LAPACK), fp64, 0-based C indices
 *   - the m-vector (v, s, lambda, z) is the concatenation of cone blocks in
 *     cone_dims order (src/ConicIP.jl:519-522); a Q block stores the bound t
 *     first; an S block stores `vecm` order (src/ConicIP.jl:128-151)
 *   - every entry point returns 0 on success, a negative CIP_E_* otherwise;
 *     cip_last_error() gives a message.  Nothing falls back to the CPU: when no
 *     HIP device is usable cip_create fails with CIP_E_NODEVICE.
 *   - `*_dev` entry points take DEVICE pointers and enqueue on the handle's
 *     stream without synchronising (except where a host scalar is returned);
 *     the un-suffixed ones take HOST pointers and are synchronous.
 */
#ifndef CIPKKT_H
#define CIPKKT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cip_handle cip_handle;

/* cone types: "R", "Q", "S" of cone_dims (src/ConicIP.jl:421-430) */
#define CIP_CONE_R 0
#define CIP_CONE_Q 1
#define CIP_CONE_S 2

/* elimination route */
#define CIP_ROUTE_SCHUR   0  /* pivot on the F'F block first (algebra of src/kktsolvers.jl:316-338), dense LDL' of [S G';G 0] */
#define CIP_ROUTE_FULL3X3 1  /* literal 3x3 assembly (src/kktsolvers.jl:254-256), symmetrised, dense LDL' of order n+p+m */

/* block-operator modes for cip_apply_F_dev (src/blockmatrices.jl:173-200) */
#define CIP_OP_F      0
#define CIP_OP_FT     1
#define CIP_OP_FINV   2
#define CIP_OP_FINVT  3

/* which problem matrix (cip_gemv_dev) */
#define CIP_MAT_Q 0
#define CIP_MAT_A 1
#define CIP_MAT_G 2

/* flags for cip_create_ex */
#define CIP_FLAG_DEVICE_PTRS 1   /* Q/A/G arrays are device pointers */
#define CIP_FLAG_CSR_HOST    2   /* ... except the CSR arrays of A, which are host pointers (the library needs them on the
                                    host anyway, to build the CSR of A') */

#define CIP_OK            0
#define CIP_E_INVALID    -1
#define CIP_E_NODEVICE   -2
#define CIP_E_HIP        -3
#define CIP_E_NOTFACTORED -4
#define CIP_E_SINGULAR   -5   /* zero / non-finite pivot met in the LDL' */
#define CIP_E_UNSUPPORTED -6

/* Problem description for cip_create_ex.  A may be given dense (A != NULL) or
 * in CSR (A == NULL, A_rowptr/A_colind/A_val != NULL, 0-based) -- with any mix of cone types (the rows of S cones are expanded
 * into a dense block on the device; R / Q rows stay CSR).  S cones: matrix order r <= 2048 and at most 1024 cones of order >= 133
 * (CIP_E_UNSUPPORTED beyond either; the reference has no limit, src/ConicIP.jl:196-210). */
typedef struct cip_problem {
    int n, m, p;
    int ncones;
    const int *cone_type;      /* ncones entries, CIP_CONE_* */
    const int *cone_dim;       /* ncones entries; for S: vectorised length k = r(r+1)/2 */
    const double *Q;  int ldq; /* n x n */
    const double *A;  int lda; /* m x n dense, or NULL */
    const int *A_rowptr; const int *A_colind; const double *A_val; /* CSR, m+1 / nnz / nnz */
    const double *G;  int ldg; /* p x n (may be NULL when p == 0) */
    int route;                 /* CIP_ROUTE_* */
    int flags;                 /* CIP_FLAG_* */
} cip_problem;

/* ---- level 1: kktsolver(Q, A, G, cone_dims)  (src/ConicIP.jl:667; src/kktsolvers.jl:18-28, :180-190, :281-285) */
int cip_create(int n, int m, int p, int ncones, const int *cone_type, const int *cone_dim,
               const double *Q, const double *A, const double *G, int route, cip_handle **out);
int cip_create_ex(const cip_problem *prob, cip_handle **out);
/* level 1 again on an existing handle: new Q / A / G of the same shape (n, m, p, cones, route, dense-or-CSR A with the
 * same nnz); keeps every device allocation (hipMalloc / hipFree synchronise the whole device) */
int cip_update_problem(cip_handle *h, const cip_problem *prob);
int cip_destroy(cip_handle *h);
const char *cip_last_error(void);
int cip_set_stream(cip_handle *h, void *hip_stream);

/* ---- level 2: solve3x3gen(F, F^-T)  (src/ConicIP.jl:682; src/kktsolvers.jl:30-35, :250-257, :287-295)
 * The scaling F is handed over in packed form, read off the reference's Block
 * elements (src/ConicIP.jl:189-192, :208, :598):
 *   R cone (k)   : k doubles        diag(F)
 *   Q cone (k)   : 1 + k doubles    beta, w      (F = diag(-beta, beta, ...) + w w')
 *   S cone (k)   : 2 r^2 doubles    R (r x r col-major), then inv(R) (r x r col-major)
 * cip_scaling_packed_len() returns the total length. */
size_t cip_scaling_packed_len(const cip_handle *h);
int cip_set_scaling_packed(cip_handle *h, const double *packedF);       /* host pointer */
int cip_set_scaling_identity(cip_handle *h);                            /* F = I (src/ConicIP.jl:704) */
/* nt_scaling on the device (src/ConicIP.jl:589-605, :165-210): F from (v, s); also
 * writes lambda = F v (src/ConicIP.jl:735) when lambda_out != NULL. */
int cip_set_scaling_from_iterate_dev(cip_handle *h, const double *v, const double *s, double *lambda_out);
int cip_get_scaling_packed(cip_handle *h, double *packedF);             /* host pointer (tests) */
/* assemble + factor the KKT system for the current scaling.  Asynchronous on the handle's stream: nothing waits for
 * the GPU.  The pivot flag is read back into pinned host memory behind the factorisation and resolved by
 * cip_check_factor, by the host-pointer solves (which are synchronous anyway) and by the *_dev solves: those WAIT for
 * the flag until one factorisation of the handle has been seen clean (the factorisation that meets a bad pivot is the
 * first one: LPs, singular Q), afterwards they resolve it only if the read-back has already landed and go ahead
 * speculatively otherwise (a later bad pivot then surfaces from the next resolving call as CIP_E_SINGULAR "repeat
 * them"; cip_check_factor after cip_factor rules that out).  A bad pivot triggers the regularised re-factorisation
 * described below; if that fails too the resolving call returns CIP_E_SINGULAR. */
int cip_factor(cip_handle *h);
/* wait for the factorisation and report its status: CIP_OK or CIP_E_SINGULAR */
int cip_check_factor(cip_handle *h);
/* Static regularisation of the quasi-definite LDL' (K + delta diag(+1.. -1..), delta = rel * max|K_ii|).  Default:
 * rel = 0 and automatic = 1 -- the first factorisation that meets a zero / non-finite / wrong-sign pivot (singular S:
 * LPs, free variables with singular Q; the reference's pivoting LU / QR, src/kktsolvers.jl:24,:231,:295, has no such
 * restriction) is redone with rel = 1e-13 per row (delta_i = rel * max_j |K_ij|), which then stays on; solve3x3 then
 * refines against the true operator (a few extra back-solves) so that callers see the unregularised solution. */
int cip_set_regularization(cip_handle *h, double rel, int automatic);
int cip_get_regularization(cip_handle *h, double *rel, int *times_switched_on);

/* ---- level 3: solve3x3(x, y, z) -> (a, b, c)  (src/ConicIP.jl:688; src/kktsolvers.jl:37-50, :324-330)
 *   Q a + G' b - A' c = x ;  G a = y ;  A a + F'F c = z */
int cip_solve3x3(cip_handle *h, const double *x, const double *y, const double *z,
                 double *a, double *b, double *c);                       /* host pointers */
int cip_solve3x3_dev(cip_handle *h, const double *x, const double *y, const double *z,
                     double *a, double *b, double *c);                   /* device pointers */

/* ---- the 2x2 plugin form  solve2x2(y, w) -> (dy, dw)  (src/ConicIP.jl:450-466; src/kktsolvers.jl:297-302;
 * what `pivot`, src/kktsolvers.jl:316-349, wraps):  [Q + A'(F'F)^-1 A, G'; G, 0][dy; dw] = [y; w].
 * Served by the factor of the Schur route (CIP_E_UNSUPPORTED on the full-3x3 route). */
int cip_solve2x2(cip_handle *h, const double *y, const double *w, double *dy, double *dw);      /* host pointers */
int cip_solve2x2_dev(cip_handle *h, const double *y, const double *w, double *dy, double *dw);  /* device pointers */

/* ---- the 4x4 -> 3x3 reduction of solve4x4 (src/ConicIP.jl:684-692), device pointers.
 * r and dz are 4-block vectors (y[n], w[p], v[m], s[m]) stored contiguously. */
int cip_solve4x4_dev(cip_handle *h, const double *lambda, const double *r, double *dz);

/* ---- block operator / cone algebra on the device (device pointers, length m) */
int cip_apply_F_dev(cip_handle *h, int mode, const double *x, double *out);              /* src/blockmatrices.jl:173-200 */
int cip_cone_prod_dev(cip_handle *h, const double *x, const double *y, double *out);     /* src/ConicIP.jl:637-665 */
int cip_cone_div_dev(cip_handle *h, const double *x, const double *y, double *out);      /* src/ConicIP.jl:607-635: solve y o out = x */
int cip_maxstep_dev(cip_handle *h, const double *x, const double *d, double scale, double *alpha_host); /* src/ConicIP.jl:571-587; d == NULL -> the `nothing` variant; steps along d*scale */
/* the pair the interior-point loop always asks for together (src/ConicIP.jl:708-709, :881-882, :927-928): alpha_host2[0] =
 * maxstep(x1, d1 * scale), alpha_host2[1] = maxstep(x2, d2 * scale), one wait; with S cones of order 133..256 the two sides run
 * side by side on two streams.  Same values as two cip_maxstep_dev calls. */
int cip_maxstep_pair_dev(cip_handle *h, const double *x1, const double *d1, const double *x2, const double *d2, double scale,
                         double *alpha_host2);
int cip_cone_identity_dev(cip_handle *h, double *e);                                     /* src/ConicIP.jl:559-565 */

/* ---- vector helpers for a device-resident driver loop (device pointers) */
int cip_gemv_dev(cip_handle *h, int which, int trans, double alpha, const double *x, double beta, double *y);
/* out_host[i] = dot(x_i[0:len_i], y_i[0:len_i]); pointer arrays live on the host */
int cip_dots_dev(cip_handle *h, int count, const double *const *x, const double *const *y,
                 const int *len, double *out_host);
int cip_axpby_dev(cip_handle *h, int len, double alpha, const double *x, double beta, double *y); /* y = alpha x + beta y */

/* ---- the whole interior-point loop of conicIP (src/ConicIP.jl:468-939) driven natively: every vector stays in
 * HBM, the host sees scalars only.  Same options and defaults as the reference's keyword
 * arguments (src/ConicIP.jl:498-509); a negative infeasTol / refinementThreshold selects the reference's derived
 * default (optTol, optTol/1e7).  c[n], b[m], d[p] and the outputs y[n], w[p], v[m] are host pointers.
 * trace (optional, trace_cap rows of CIP_TRACE_COLS doubles): Iter, mu, rDu, rPr, rCp, pobj, dobj, alpha, sigma. */
#define CIP_STATUS_NONE       0
#define CIP_STATUS_OPTIMAL    1   /* :Optimal    src/ConicIP.jl:786 */
#define CIP_STATUS_INFEASIBLE 2   /* :Infeasible src/ConicIP.jl:815-818 */
#define CIP_STATUS_UNBOUNDED  3   /* :Unbounded  src/ConicIP.jl:847-850 */
#define CIP_STATUS_ABANDONED  4   /* :Abandoned  src/ConicIP.jl:936 */
#define CIP_STATUS_ERROR      5   /* :Error      src/ConicIP.jl:870-873 */
#define CIP_TRACE_COLS 9
typedef struct cip_options {
    double optTol, DTB, infeasTol, refinementThreshold;
    int maxRefinementSteps, maxIters, verbose;
} cip_options;
typedef struct cip_result {          /* src/ConicIP.jl:384-398 (Solution) */
    int status, iter;
    double mu, prFeas, duFeas, muFeas, pobj, dobj;
    int n_factor, n_solve, trace_rows;
    double wall_s;
} cip_result;
int cip_conicip(cip_handle *h, const double *c, const double *b, const double *d, const cip_options *opt,
                double *y, double *w, double *v, cip_result *res, double *trace, int trace_cap);

/* ---- batches of independent problems on one GPU (BASELINE config 5).  A batch is an array of ordinary handles,
 * each on its own HIP stream; cip_batch_handle(b, i) is the "leading problem index": every entry point above works
 * on it.  cip_conicip_many / cip_batch_conicip keep `in_flight` interior-point loops running at once (one host
 * thread each); pointer arrays have one entry per problem (b / d / w / v arrays may be NULL when m or p is 0). */
typedef struct cip_batch cip_batch;
int cip_batch_create(int count, const cip_problem *probs, cip_batch **out);
int cip_batch_destroy(cip_batch *b);
int cip_batch_size(const cip_batch *b);
cip_handle *cip_batch_handle(cip_batch *b, int i);
int cip_batch_conicip(cip_batch *b, const double *const *c, const double *const *bvec, const double *const *d,
                      const cip_options *opt, double *const *y, double *const *w, double *const *v, cip_result *res,
                      int in_flight);
/* problems in, solutions out: `in_flight` host threads, each re-loading ONE handle (cip_update_problem) on its own
 * stream with the next problem of the queue -- level-1 upload of one problem overlaps the loops of the others */
int cip_conicip_problems(int count, const cip_problem *probs, const double *const *c, const double *const *bvec,
                         const double *const *d, const cip_options *opt, double *const *y, double *const *w,
                         double *const *v, cip_result *res, int in_flight);
/* the same, in LOCK-STEP: the problems must have identical shape (n, m, p, cone list, route, dense-or-CSR A) and no S
 * cone of matrix order >= 133; they advance through the loop together, every step ONE launch with the problem index in the grid (groups of up
 * to 64).  Results are bit-identical to cip_conicip on each problem run with the same solve block (lock-step handles use
 * min(cip_set_solve_block_max, cip_lockstep_solve_block_for(B)): a standalone handle of order >= 1024 sums its triangular solves in wider blocks unless
 * cip_set_solve_block_max(that value) is called first -- the difference is rounding).  Returns CIP_E_UNSUPPORTED (nothing written) when
 * the batch does not qualify -- fall back to cip_conicip_problems.  A problem whose factorisation meets a bad pivot leaves
 * the group and is solved by the one-problem loop afterwards. */
int cip_conicip_lockstep(int count, const cip_problem *probs, const double *const *c, const double *const *bvec,
                         const double *const *d, const cip_options *opt, double *const *y, double *const *w,
                         double *const *v, cip_result *res);
/* ANY mix of problems: those that share a shape with at least one other problem of the batch (and qualify for lock-step)
 * advance together, shape group by shape group; the others go through cip_conicip_problems' thread pool (`in_flight`
 * threads).  Per problem the result is that of the entry point it went through; cip_lockstep_stats adds up over the groups.
 * (The reference solves one problem per conicIP call, src/ConicIP.jl:472-480: a batch is N independent calls.) */
int cip_conicip_mixed(int count, const cip_problem *probs, const double *const *c, const double *const *bvec,
                      const double *const *d, const cip_options *opt, double *const *y, double *const *w,
                      double *const *v, cip_result *res, int in_flight);
/* the arena of the last lock-step group is kept for the next call (allocation of several GB is slow); this frees it */
int cip_release_cached_memory(void);
/* diagnostics of the calling thread's last cip_conicip_lockstep: {groups, problems, problems that left their group} */
int cip_lockstep_stats(int *out3);
/* a lock-step call of at least 2 x 8 problems of one shape runs as k groups (default 2) SIDE BY SIDE, each on its own host thread and stream (the
 * groups fill each other's launch-latency gaps; per problem nothing changes: same kernels, same solve block, same bits).  k = 1: one
 * group after the other (the form up to round 5).  Also CIP_LOCKSTEP_SPLIT.  Process-wide; returns the previous value (k < 1: query).
 * With k > 1 a call that cip_conicip_lockstep refuses with CIP_E_UNSUPPORTED for a slab-layout reason found inside a group (CSR arrays
 * in device memory with differing numbers of non-zeros) may already have written other groups' results. */
int cip_set_lockstep_split(int k);
int cip_conicip_many(cip_handle *const *handles, int count, const double *const *c, const double *const *bvec,
                     const double *const *d, const cip_options *opt, double *const *y, double *const *w,
                     double *const *v, cip_result *res, int in_flight);

/* ---- dense symmetric LDL' building blocks (device pointers), usable on their own.
 * K is N x N column-major with leading dimension ld; only the lower triangle is
 * referenced.  N and ld must be multiples of 128 (pad with an identity block).
 * After a factorisation K holds L strictly below the diagonal, D on it and L' above it OUTSIDE the 128 x 128 diagonal blocks;
 * the upper triangle of the diagonal blocks is undefined.  The workspace size covers either mode of cip_set_solve_fused; the
 * mode in force at cip_ldlt_factor_dev must still be in force at its cip_ldlt_solve_dev calls, and the solve-block limit
 * (cip_set_solve_block_max) must not change between the sizing call, the factorisation and its solves. */
int cip_ldlt_workspace_bytes(int N, size_t *bytes);
int cip_ldlt_factor_dev(void *hip_stream, double *K, int N, int ld, void *workspace, int *info_host);
int cip_ldlt_solve_dev(void *hip_stream, const double *K, int N, int ld, const void *workspace, double *rhs);
/* C(lower or full) += alpha * A * B'   (A: M x K, B: N x K, all column-major; M,N % 128 == 0, K % 16 == 0) */
int cip_gemm_nt_dev(void *hip_stream, int M, int N, int K, double alpha,
                    const double *A, int lda, const double *B, int ldb, double *C, int ldc, int lower_only);

/* ---- introspection (tests, bench) */
int cip_kkt_order(const cip_handle *h, int *N, int *N_padded);
int cip_get_kkt_matrix(cip_handle *h, double *K_host /* N_padded^2 */); /* assembled (before cip_factor) or factored */
int cip_assemble_only(cip_handle *h);                                   /* level-2 assembly without the factorisation */
/* stats: [0] factor calls, [1] solve calls, [2] last factor ms (assemble), [3] last factor ms (ldlt), [4] flops of last ldlt */
int cip_stats(cip_handle *h, double *out8);
int cip_set_timing(cip_handle *h, int enabled);
int cip_set_ldlt_outer_block(int nbo);    /* 0 = automatic (896 from order 4096 on, else 512); returns the knob's value */
/* tuning knob: widest block of the triangular solves' block-step form (128, 256, 512 or 1024; 0 = query).  Applies to
 * handles created afterwards; returns the previous value.  Lock-step batches use min(this, cip_lockstep_solve_block_for(B)) for their handles. */
int cip_set_solve_block_max(int b);
/* block steps of the triangular sweeps (also CIP_SOLVE_FUSED): 0 (default) = two launches per step (block product, then the
 * update of everything below / above); 1 = ONE launch per block step for solve blocks of at most 512 columns -- the neighbour
 * blocks X_J L_{J,J-1} / L_{J+1,J} X_J are formed with the block inverses at factorisation time, so a step depends on the
 * previous step's result only (sweeps 20 % faster, factorisation dearer: pays from about six solves per factorisation on);
 * 2 = for every block width.  Applies to handles created afterwards; the two forms differ in rounding.  Returns the
 * previous mode (other values: query). */
int cip_set_solve_fused(int mode);
/* the solve-block limit cip_conicip_lockstep gives the handles of a call of B problems in all (512 for B <= 8, else 256 -- chosen
 * from the size of the whole call, so the groups of 64 it is cut into all use the same one; never more than
 * cip_set_solve_block_max's value; CIP_LOCKSTEP_SOLVE_BLOCK overrides): a one-problem run with this limit
 * reproduces the call's iterates bit for bit */
int cip_lockstep_solve_block_for(int B);
/* panel chain of the serial schedule (also CIP_FUSE_DIAG): 3 (default) = one launch per 128-column panel -- diagonal kernel,
 * the previous panel's in-block update and this panel's TRSM, the TRSM following the diagonal kernel micro-panel by
 * micro-panel through a stage counter; 1 = diagonal kernel + previous panel's update in one launch, TRSM in its own;
 * 0 = three launches per panel.  Same factor bit for bit.  Process-wide; returns the previous setting (other values: query). */
int cip_set_ldlt_fused_chain(int on);
/* The fused panel chain waits INSIDE a launch for workgroups of the same launch (bounded: ~1 s, then the factorisation reports that the
 * wait gave up).  On a GPU shared with other processes the hardware scheduler can keep a launch's workgroups apart for longer than that
 * (seen with eight processes on one MI355X).  The library then redoes the factorisation with the three-launch chain -- no in-launch wait,
 * same bits -- and keeps that chain for the handle; a problem of a lock-step group leaves the group and is solved alone.  TEST HOOK: the
 * next n fused-chain factorisations of the process report such a give-up (n < 0: query; n = count + 65536 * skip: the `count` ones after
 * the next `skip`); returns the previous count. */
int cip_debug_chain_giveup(int n);
/* how many factorisations of this handle were redone on the three-launch chain after such a give-up (-1: NULL handle) */
int cip_get_chain_fallbacks(cip_handle *h);
/* solve preparation (block inverses for the triangular sweeps, mirror image of L) of every solve block whose columns are final,
 * on a side stream beside the last panels of the factorisation instead of behind it (also CIP_SIDE_PREP; CIP_SIDE_PREP_FROM =
 * columns before the end from which it forks, default 2048).  1 (default) on, 0 off.  Same bits.  Returns the previous setting. */
int cip_set_ldlt_side_prep(int on);
/* Schur route, CSR A with one entry per row, R cones, no equalities, order a multiple of 128 (the box-QP family): cip_factor
 * copies only the first outer block's columns of Q into K; the first trailing update of the LDL' reads the rest from Q
 * itself (also CIP_LAZY_COPY=0).  Same factor bit for bit.  1 (default) on, 0 off; returns the previous setting. */
int cip_set_lazy_copy(int on);
/* S cones of order 133..256: the max-step's extreme eigenvalue (the reference's eigmin / eigmax, src/ConicIP.jl:272-303) by
   Lanczos with full reorthogonalisation (1) or by a full tridiagonalisation + Sturm multisection (0); < 0 only
   reads.  Returns the previous setting.  Same eigenvalue to ~1e-11 relative either way; for A/B runs and tests.
   2 (DEFAULT): Lanczos + INERTIA CERTIFICATE -- a Krylov method started from a fixed vector can in principle settle on an interior
   eigenvalue (no component along the extreme eigenvector); mode 2 checks with an LDL' of theta' I - A (theta' = theta + 1e-9 |T|)
   that no eigenvalue lies beyond the returned one and recomputes the verdict by the tridiagonalisation when the check fails
   (0.09 ms per iteration of config 4, 1 %; also CIP_LG_LANCZOS=2).  cip_sdp_lanczos_fallbacks: how often that happened on this handle.
   3 (self-test): as 2 with the bound on the wrong side of theta, so that every certificate fails and every verdict is the fallback's. */
int cip_set_sdp_lanczos(int on);
/* (S cones of order >= 133, NT scaling, src/ConicIP.jl:196-210: the one-sided Jacobi of svd(Lz'Ls) runs as one launch per phase.  Rounds
   3-5 also offered one persistent launch with in-launch block hand-offs behind cip_set_sdp_jacobi_stepped; that form was measured to
   come out with other bits about once in 800 / 4000 / 40000 scalings at order 1024 / 512 / 256, the cause was never found, and round 6
   REMOVED it together with its switch: no public knob of this library documents nondeterminism.) */
int cip_sdp_lanczos_fallbacks(cip_handle *h, int *count);
/* HIP-event timing of the LDL' trailing-update launches (bench.py roofline): enable, then read
 * out3 = [launches, total ms, total algorithmic flops (r(r+1)K per launch)].  enabled = 1: every launch of every factorisation;
 * enabled = k > 1: of every k-th factorisation, starting with the next one (an event pair costs the chain ~8 us: a sampled
 * profile perturbs the timed region a k-th as much); 0: off (drops the totals) */
int cip_profile_trailing(cip_handle *h, int enabled);
int cip_profile_get(cip_handle *h, double *out3);
/* the same for every factorisation the CALLING THREAD enqueues on handles without a profile of their own (the handles of
 * cip_conicip_lockstep live inside the call): flops count every live problem of a lock-step launch */
int cip_profile_trailing_thread(int enabled);
int cip_profile_thread_get(double *out3);
/* the same event-pair timing for the dominant kernels of the other configurations, per calling thread: slot 0 = the trailing
 * update (as above), 1 = Schur formation S = Q + (A'F^-1)(A'F^-1)' with a dense A (m n^2 flop per call; src/kktsolvers.jl:33-34,
 * :290), 2 = the one-sided Jacobi of a large S cone's NT scaling (src/ConicIP.jl:204; latency-bound, flops reported as 0) */
#define CIP_PROFILE_TRAILING 0
#define CIP_PROFILE_SYRK     1
#define CIP_PROFILE_JACOBI   2
int cip_profile_kernel_thread(int slot, int enabled);
int cip_profile_kernel_thread_get(int slot, double *out3);

#ifdef __cplusplus
}
#endif
#endif /* CIPKKT_H */
