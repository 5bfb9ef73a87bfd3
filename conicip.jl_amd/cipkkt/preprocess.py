"""Pre-solve of the reference (src/preprocessor.jl:10-96) in front of the device solver.

`imcols` finds a maximal set of independent rows with a rank-revealing (column-pivoted) QR of A' and checks
that the system is consistent; `preprocess_conicIP` drops the dependent equality rows of G, adds 1 on the
diagonal of Q for the variables the dual equations do not determine, runs `conicIP` on the device and puts
zeros back for the removed multipliers.  This is the one-shot host-side step that guarantees the unpivoted
LDL' of [S G'; G 0] sees a full-row-rank G; it is O((n+m+p) n^2) once per problem and
stays on the host, as in the reference (SuiteSparse QR there, LAPACK geqp3 here: the rows kept may differ,
the rank and the solution of the reduced problem do not).
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from .driver import Solution, conicIP


def _dense(A):
    return np.asarray(A.todense()) if sp.issparse(A) else np.asarray(A, dtype=np.float64)


def imcols(A, b, eps=1e-8):
    """(rows, consistent): sorted 0-based indices of independent rows of A; consistency of A x = b
    (src/preprocessor.jl:10-30)."""
    A = _dense(A)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    if A.size == 0:
        return [], True
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
    Qd, Gd = _dense(Q), _dense(G)
    keep_p, primal_ok = imcols(Gd, d)
    keep_d, dual_ok = imcols(np.hstack([Qd, _dense(A).T, Gd[keep_p, :].T]), c)
    if not (primal_ok and dual_ok):
        return Solution(np.full(n, np.nan), np.full(p, np.nan), np.full(m, np.nan), status="Infeasible")
    free = np.ones(n)
    free[keep_d] = 0.0
    sol = conicIP(Qd + np.diag(free), c, A, b, cone_dims, Gd[keep_p, :] if keep_p else None,
                  d[keep_p] if keep_p else None, **options)
    w = np.zeros(p)
    if keep_p:
        w[keep_p] = sol.w
    sol.w = w
    return sol
