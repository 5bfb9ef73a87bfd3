"""ORACLE -- test infrastructure only.  CPU restatement of the reference's pre-solve
(src/preprocessor.jl:10-96): `imcols` (rank-revealing QR of A', consistency check) and
`preprocess_conicIP` (drop dependent equality rows, regularise the variables the dual
equations do not determine, solve, re-insert zeros).  The reference uses SuiteSparse's
sparse QR (`qr(sparse(A'))`, `F.pcol`); any column-pivoted QR reveals the same rank and
an equally valid maximal set of independent rows, so LAPACK's `geqp3` stands in for it
here (which rows are kept may differ; the solution of the reduced problem does not)."""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from .conicip import Solution, conicIP


def _dense(A):
    return np.asarray(A.todense()) if sp.issparse(A) else np.asarray(A, dtype=np.float64)


def imcols(A, b, eps=1e-8):
    """src/preprocessor.jl:10-30: indices (0-based, sorted) of a maximal set of independent rows of A and
    whether A x = b is consistent."""
    A = _dense(A)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    if A.size == 0:                                          # :16
        return [], True
    nA = np.linalg.norm(A)                                   # :14 (Frobenius norm of a sparse matrix)
    if nA == 0.0:                                            # sparse A/0 touches no stored entry -> zero R -> :25
        return [], True
    A = A / nA
    b = b / nA
    _, R, piv = sla.qr(A.T, mode="economic", pivoting=True)  # :18-22
    nr = min(R.shape)
    diag = np.abs(np.diag(R)[:nr])
    rows = np.sort(piv[:nr][diag > eps])                     # :23
    if rows.size == 0:                                       # :25
        return [], True
    x = np.linalg.lstsq(A[rows, :], b[rows], rcond=None)[0]  # :27  A[R,:] \ b[R]
    # one refinement step: the test below is ABSOLUTE (1e-8) on data scaled by 1/||A||; with ||b||_inf ~ 1e7 (Miles 3,
    # A and b scaled by 1e-4) LAPACK's least-squares residual is 2e-8 -- 2.5e-15 relative -- where SuiteSparseQR's
    # stays below the threshold (test/runtests.jl:630-637 expects :Optimal)
    x = x + np.linalg.lstsq(A[rows, :], (b - A @ x)[rows], rcond=None)[0]
    ok = np.linalg.norm(A @ x - b, np.inf) < eps
    return (list(rows), True) if ok else ([], False)


def preprocess_conicIP(Q, c, A, b, cone_dims, G=None, d=None, **options):
    """src/preprocessor.jl:43-96."""
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    n = c.size
    m = A.shape[0]
    G = np.zeros((0, n)) if G is None else G
    d = np.zeros(0) if d is None else np.asarray(d, dtype=np.float64).reshape(-1)
    p = G.shape[0]
    Qd, Ad, Gd = _dense(Q), _dense(A), _dense(G)
    IP, pcons = imcols(Gd, d)                                # :62
    ID, dcons = imcols(np.hstack([Qd, Ad.T, Gd[IP, :].T]), c)   # :63
    if not (pcons and dcons):                                # :65-68
        return Solution(np.full(n, np.nan), np.full(p, np.nan), np.full(m, np.nan), status="Infeasible")
    z = np.ones(n)
    z[ID] = 0.0                                              # :82
    sol = conicIP(Qd + np.diag(z), c, A, b, cone_dims, Gd[IP, :], d[IP], **options)   # :86-88
    w = np.zeros(p)
    w[IP] = sol.w                                            # :93
    sol.w = w
    return sol
