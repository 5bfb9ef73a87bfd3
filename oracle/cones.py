"""ORACLE (test infrastructure, NOT product code).

CPU/numpy restatement of the reference's per-cone algebra.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product path (``conicip.jl_amd/``) never does.

Parity status: the reference is Julia and cannot be run in the build container
(no ``julia`` binary, no network).  This restatement is pinned against every
deterministic known-answer test the reference's own suite holds for the path
(``test/runtests.jl`` — analytic projections, statuses, the pinned residual
Dicts at the reference's own 1e-3 tolerance, operator identities); see
``tests/test_oracle_*.py``.  At the granularity of WoodburyMatrices.jl internals
and for the large configs (n >= 2048) the reference holds no vectors:
**parity unpinned** there.

Every function cites the reference lines it follows (paths relative to the
reference repository root).
"""
import numpy as np

SQRT2 = np.sqrt(2.0)


# ---------------------------------------------------------------- mat / vecm
def ord_(x):
    """src/ConicIP.jl:85 -- matrix order r from vectorised length k=r(r+1)/2."""
    n = len(x)
    return int(round((np.sqrt(1 + 8 * n) - 1) / 2))


def mat(x):
    """src/ConicIP.jl:93-119 -- inverse of vecm (row-major upper triangle,
    off-diagonals scaled by 1/sqrt2)."""
    x = np.asarray(x, dtype=np.float64)
    n = ord_(x)
    Z = np.zeros((n, n))
    iu = np.triu_indices(n)
    Z[iu] = x
    off = iu[0] != iu[1]
    Z[iu[0][off], iu[1][off]] /= SQRT2
    Z = Z + np.triu(Z, 1).T
    return Z


def vecm(Z):
    """src/ConicIP.jl:128-151 -- for i=1..n, j=i..n; off-diagonals * sqrt2."""
    Z = np.asarray(Z, dtype=np.float64)
    n = Z.shape[0]
    iu = np.triu_indices(n)
    x = Z[iu].copy()
    x[iu[0] != iu[1]] *= SQRT2
    return x


# ------------------------------------------------------------ small helpers
def QF(r):
    """src/ConicIP.jl:160 -- r1^2 - ||r_2:||^2 written as 2 r1^2 - r.r."""
    return 2 * r[0] * r[0] - np.dot(r, r)


def Qxy(x, y):
    """src/ConicIP.jl:161 -- x' J y."""
    return 2 * x[0] * y[0] - np.dot(x, y)


def fts(x1, a1, y1, x2, a2, y2):
    """src/ConicIP.jl:162-163 -- (x1 - a1 y1)'(x2 - a2 y2)."""
    return (np.dot(x1, x2) - a2 * np.dot(x1, y2)
            - a1 * np.dot(y1, x2) + a1 * a2 * np.dot(y1, y2))


# ------------------------------------------------------------- NT scalings
def nestod_soc(z, s):
    """src/ConicIP.jl:165-194.  Returns (beta, w) such that the scaling block is
    SymWoodbury(Diagonal([-beta, beta, ..., beta]), w, 1.0) = diag + w w'."""
    z = np.array(z, dtype=np.float64)
    s = np.array(s, dtype=np.float64)
    beta = (QF(s) / QF(z)) ** 0.25
    z = z / np.sqrt(QF(z))
    s = s / np.sqrt(QF(s))
    gamma = np.sqrt((1 + np.dot(z, s)) / 2)
    z = -z
    z[0] = -z[0]                       # J z
    w = (1.0 / (2.0 * gamma)) * (s + z)
    w[0] = w[0] + 1
    w = w * (np.sqrt(2 * beta) / np.sqrt(2 * w[0]))
    return beta, w


def nestod_sdc(z, s):
    """src/ConicIP.jl:196-210.  Returns R with R' Z R = R^-1 S R^-T = Lambda."""
    Ls = np.linalg.cholesky(mat(s))
    Lz = np.linalg.cholesky(mat(z))
    U, lam, _ = np.linalg.svd(Lz.T @ Ls)
    R = np.linalg.solve(Lz.T, U) * np.sqrt(lam)[None, :]
    return R


# ---------------------------------------------------------------- max step
def maxstep_rp(x, d):
    """src/ConicIP.jl:212-240."""
    if d is None:
        if np.all(x > 0):
            return 0.0
        return -1 + np.min(x)
    pos = d > 0
    if not np.any(pos):
        return np.inf
    return np.min(x[pos] / d[pos])


def maxstep_soc(x, d):
    """src/ConicIP.jl:242-270."""
    if d is None:
        a = np.linalg.norm(x[1:]) - x[0]
        return 0.0 if a < 0 else -1 - a
    d = -d
    gamma = Qxy(x, x)
    xbar = x / np.sqrt(gamma)
    beta = Qxy(xbar, d)
    rho1 = beta / np.sqrt(gamma)
    mu = (beta + d[0]) / (xbar[0] + 1)
    rho2 = d[1:] - mu * xbar[1:]
    alpha = np.linalg.norm(rho2) / np.sqrt(gamma) - rho1
    if alpha < 0:
        return np.inf
    return 1 / alpha


def maxstep_sdc(x, d):
    """src/ConicIP.jl:272-303."""
    X = mat(x)
    if d is None:
        lam = np.linalg.eigvalsh(X)
        mn = np.min(lam)
        return 0.0 if mn > 0 else -1 + mn
    lamX, V = np.linalg.eigh(X)
    if np.any(lamX <= 0):
        return np.inf
    Xih = (V / np.sqrt(lamX)[None, :]) @ V.T
    D = mat(d)
    XDX = Xih @ D @ Xih
    XDX = 0.5 * (XDX + XDX.T)
    lam = np.linalg.eigvalsh(XDX)
    neg = lam < 0
    if np.all(neg):
        return np.inf
    return 1 / np.max(lam[~neg])


# ------------------------------------------------- Jordan product / division
def xrp(x, y):
    """src/ConicIP.jl:311-315."""
    return x * y


def drp(x, y):
    """src/ConicIP.jl:305-309 -- o = x ./ y."""
    return x / y


def xsoc(x, y):
    """src/ConicIP.jl:340-345."""
    o = np.empty_like(x)
    o[0] = np.dot(x, y)
    o[1:] = x[0] * y[1:] + y[0] * x[1:]
    return o


def dsoc(num, den):
    """src/ConicIP.jl:317-338 -- called as dsoc!(xI, yI, oI) from cone_div!
    (:630): first argument is the numerator, second the arrow matrix; solves
    den o o = num."""
    y1 = den[0]
    yb = den[1:]
    alpha = y1 * y1 - np.dot(yb, yb)
    x1 = num[0]
    xb = num[1:]
    o = np.empty_like(num)
    o[0] = (y1 * x1 - np.dot(yb, xb)) / alpha
    b1 = (-x1 / alpha) + np.dot(yb, xb) / (y1 * alpha)
    b2 = 1 / y1
    o[1:] = yb * b1 + xb * b2
    return o


def xsdc(x, y):
    """src/ConicIP.jl:355-360 -- vecm(XY + YX) (no 1/2)."""
    X = mat(x)
    Y = mat(y)
    return vecm(X @ Y + Y @ X)


def dsdc(x, y):
    """src/ConicIP.jl:347-353 -- vecm(lyap(Y, -X)): solve Y O + O Y = X."""
    X = mat(x)
    Y = mat(y)
    lam, V = np.linalg.eigh(Y)
    Xt = V.T @ X @ V
    Ot = Xt / (lam[:, None] + lam[None, :])
    return vecm(V @ Ot @ V.T)
