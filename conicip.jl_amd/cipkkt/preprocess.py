"""Pre-solve of the reference (src/preprocessor.jl:10-96) in front of the device solver.

`imcols` finds a maximal set of independent rows with a rank-revealing (column-pivoted) QR of A' and checks
that the system is consistent; `preprocess_conicIP` drops the dependent equality rows of G, adds 1 on the
diagonal of Q for the variables the dual equations do not determine, runs `conicIP` on the device and puts
zeros back for the removed multipliers.  This is the one-shot host-side step that guarantees the unpivoted
LDL' of [S G'; G 0] sees a full-row-rank G; it is done once per problem and
stays on the host, as in the reference (SuiteSparse QR there, LAPACK geqp3 here: the rows kept may differ,
the rank and the solution of the reduced problem do not).

Size (round 5).  The reference's QR is SPARSE (src/preprocessor.jl:17: `qr(sparse(A'))`); the dual test runs it on the
n x (n + m + p) matrix [Q A' G'].  A dense column-pivoted QR of that matrix is 8192 x 24576 at the headline size -- minutes
of level-2 BLAS.  scipy has no sparse rank-revealing QR, so the full-row-rank case -- every well-posed program, and the only
case in which nothing is changed -- is CERTIFIED first on the n x n Gram matrix (`_full_row_rank`: sigma_min from `eigvalsh`,
sparse blocks enter through sparse products, nothing wider than n x n is ever dense); only when the certificate does not hold
does the pivoted QR run, and above `DENSE_QR_LIMIT` entries it refuses with a message instead of allocating.
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from .driver import Solution, conicIP


def _dense(A):
    return np.asarray(A.todense()) if sp.issparse(A) else np.asarray(A, dtype=np.float64)


DENSE_QR_LIMIT = 1 << 26          # entries of the matrix handed to the dense pivoted QR (512 MB of fp64)


def _full_row_rank(blocks, eps):
    """True when M = [B_1 B_2 ...] (blocks with n rows each, dense or scipy-sparse) CERTAINLY has all n rows independent by the
    reference's criterion (every |R_ii| of the pivoted QR of (M / ||M||_F)' above eps, src/preprocessor.jl:19-23): for a
    triangular R, min |R_ii| >= sigma_min(R) = sigma_min(M) / ||M||_F, and sigma_min(M)^2 = lambda_min(M M') with
    M M' = sum B_i B_i' an n x n matrix.  `eigvalsh` resolves lambda_min to ~n * 1e-16 * lambda_max, hence the margin: the
    certificate needs sigma_min / ||M||_F >= max(100 eps, 1e-6).  False = "not certified" (the caller runs the QR)."""
    n = blocks[0].shape[0]
    gram = np.zeros((n, n))
    fro2 = 0.0
    for B in blocks:
        if B.shape[1] == 0:
            continue
        if sp.issparse(B):
            B = B.tocsr()
            gram += (B @ B.T).toarray()
            fro2 += float(B.multiply(B).sum())
        else:
            B = np.asarray(B, dtype=np.float64)
            gram += B @ B.T
            fro2 += float(np.vdot(B, B))
    if fro2 == 0.0:
        return False
    lam_min = float(sla.eigvalsh(gram, subset_by_index=[0, 0], check_finite=False)[0])
    return lam_min > 0.0 and np.sqrt(lam_min / fro2) >= max(100.0 * eps, 1e-6)


def imcols(A, b, eps=1e-8):
    """(rows, consistent): sorted 0-based indices of independent rows of A; consistency of A x = b
    (src/preprocessor.jl:10-30)."""
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    if A.shape[0] * A.shape[1] == 0:
        return [], True
    if A.shape[0] * A.shape[1] > DENSE_QR_LIMIT:
        raise ValueError("imcols: the rank-revealing QR of a %d x %d matrix would be dense here (the reference's is sparse, "
                         "src/preprocessor.jl:17) and the full-row-rank certificate did not hold: reduce the program, or call "
                         "conicIP directly if its rank conditions are known to hold" % A.shape)
    A = _dense(A)
    scale = np.linalg.norm(A)
    if scale == 0.0:
        # an all-zero matrix: the reference divides only the STORED entries of its sparse A by zero (none), the QR has
        # no pivot above eps and the empty-R branch answers ([], true) (src/preprocessor.jl:14, :25)
        return [], True
    A, b = A / scale, b / scale
    _, R, piv = sla.qr(A.T, mode="economic", pivoting=True)
    k = min(R.shape)
    rows = np.sort(piv[:k][np.abs(np.diag(R)[:k]) > eps])
    if rows.size == 0:
        return [], True
    x = np.linalg.lstsq(A[rows, :], b[rows], rcond=None)[0]
    # one refinement step: the consistency test is absolute (1e-8) on data scaled by 1/||A||, and LAPACK's residual
    # at ||b||_inf ~ 1e7 (the reference's Miles-3 scaling test) is 2e-8
    x = x + np.linalg.lstsq(A[rows, :], (b - A @ x)[rows], rcond=None)[0]
    if np.linalg.norm(A @ x - b, np.inf) < eps:
        return [int(i) for i in rows], True
    return [], False


def preprocess_conicIP(Q, c, A, b, cone_dims, G=None, d=None, **options):
    """`conicIP` behind the reference's rank pre-solve (src/preprocessor.jl:43-96); same keywords as `conicIP`."""
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    n, m = c.size, A.shape[0]
    G = np.zeros((0, n)) if G is None else G
    d = np.zeros(0) if d is None else np.asarray(d, dtype=np.float64).reshape(-1)
    p = G.shape[0]
    Gd = _dense(G)
    if p > 0 and _full_row_rank([G if sp.issparse(G) else Gd], 1e-8):
        keep_p, primal_ok = list(range(p)), True                    # G has full row rank: G y = d is consistent, nothing to drop
    else:
        keep_p, primal_ok = imcols(Gd, d)
    At = (A.T.tocsr() if sp.issparse(A) else np.asarray(A, dtype=np.float64).T) if m > 0 else np.zeros((n, 0))
    dual_blocks = [Q, At, Gd[keep_p, :].T]
    if _full_row_rank(dual_blocks, 1e-8):
        keep_d, dual_ok = list(range(n)), True                      # [Q A' G'] has full row rank: every dual equation is determined
    else:
        wide = n * (n + m + len(keep_p))
        if wide > DENSE_QR_LIMIT:
            imcols(sp.csr_matrix((n, n + m + len(keep_p))), c)     # raises the size error with the matrix's shape
        keep_d, dual_ok = imcols(np.hstack([_dense(Q), _dense(At), Gd[keep_p, :].T]), c)
    if not (primal_ok and dual_ok):
        return Solution(np.full(n, np.nan), np.full(p, np.nan), np.full(m, np.nan), status="Infeasible")
    if len(keep_d) == n:
        Q_aug = Q                                                   # nothing to augment: Q goes through untouched (dense, sparse or device tensor)
    else:
        free = np.ones(n)
        free[keep_d] = 0.0
        Q_aug = _dense(Q) + np.diag(free)
    sol = conicIP(Q_aug, c, A, b, cone_dims, Gd[keep_p, :] if keep_p else None,
                  d[keep_p] if keep_p else None, **options)
    w = np.zeros(p)
    if keep_p:
        w[keep_p] = sol.w
    sol.w = w
    return sol
