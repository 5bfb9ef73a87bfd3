"""ORACLE (test infrastructure, NOT product code).

numpy/scipy restatement of the reference's three shipped KKT solvers
(src/kktsolvers.jl).  All solve, for the current scaling F,

    [ Q   G'  -A' ] [a]   [x]
    [ G           ] [b] = [y]
    [ A       F'F ] [c]   [z]

through the same three-level closure interface as the reference
(kktsolver(Q,A,G,cone_dims) -> solve3x3gen(F,Finv_T) -> solve3x3(x,y,z)).
Dense LAPACK (OpenBLAS via numpy/scipy) stands in for Julia's stdlib
LinearAlgebra (OpenBLAS 0.3.29) and SuiteSparse UMFPACK -- third-party code not
present under the reference tree.
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp


def _dense(M):
    return M.toarray() if sp.issparse(M) else np.asarray(M, dtype=np.float64)


def kktsolver_qr(Q, A, G, cone_dims):
    """src/kktsolvers.jl:18-58 -- CVXOPT 'double QR' (default solver)."""
    Qd = _dense(Q)
    Ad = _dense(A)
    Gd = _dense(G).reshape(-1, Qd.shape[0])
    n = Qd.shape[0]
    p = Gd.shape[0]
    if p > 0:
        Q0, R1full = np.linalg.qr(Gd.T, mode="complete")      # :24-25
        R1 = R1full[:p, :p]                                   # :26
    else:
        Q0 = np.eye(n)
        R1 = np.zeros((0, 0))
    Q1 = Q0[:, :p]                                            # :27
    Q2 = Q0[:, p:]                                            # :28

    def solve3x3gen(F, FinvT_unused):
        Finv = F.inv()
        FinvT = Finv.matrix().T                               # :32
        Atil = FinvT @ Ad                                     # :33
        QpAtA = Qd + Atil.T @ Atil                            # :34
        Lq, Lr = np.linalg.qr(Q2.T @ QpAtA @ Q2)              # :35

        def Lsolve(rhs):
            return sla.solve_triangular(Lr, Lq.T @ rhs)

        def solve3x3(bx, by, bz):
            bx = np.asarray(bx, dtype=np.float64)
            by = np.asarray(by, dtype=np.float64)
            bz = np.asarray(bz, dtype=np.float64)
            if p > 0:
                Q1tx = sla.solve_triangular(R1, by, trans="T")          # :39
            else:
                Q1tx = np.zeros(0)
            t = bx + Atil.T @ (FinvT @ bz)
            Q2tx = Lsolve(Q2.T @ t - Q2.T @ (QpAtA @ (Q1 @ Q1tx)))       # :40-41
            if p > 0:
                y = sla.solve_triangular(
                    R1, Q1.T @ t - Q1.T @ (QpAtA @ (Q1 @ Q1tx))
                    - Q1.T @ (QpAtA @ (Q2 @ Q2tx)))                     # :42-44
            else:
                y = np.zeros(0)
            x = Q0 @ np.concatenate([Q1tx, Q2tx])                        # :45
            Fz = FinvT @ bz - Atil @ (Q1 @ Q1tx) - Atil @ (Q2 @ Q2tx)    # :46-47
            z = Finv.mul(Fz)                                             # :48
            return x, y, z

        return solve3x3

    return solve3x3gen


def kktsolver_sparse(Q, A, G, cone_dims):
    """src/kktsolvers.jl:180-270 -- literal 3x3 assembly
    Z = [Q G' -A'; G 0 0; A 0 F'F] (:254-256) + LU.  The lifted variant
    (:60-105, :195-240) is an equivalent sparse re-expression of the same linear
    system; the oracle solves the un-lifted system with dense LU."""
    Qd = _dense(Q)
    Ad = _dense(A)
    n = Qd.shape[0]
    Gd = _dense(G).reshape(-1, n)
    m = Ad.shape[0]
    p = Gd.shape[0]

    def solve3x3gen(F, FinvT):
        FtF = F.square().matrix()                             # :252
        Z = np.zeros((n + p + m, n + p + m))
        Z[:n, :n] = Qd
        Z[:n, n:n + p] = Gd.T
        Z[:n, n + p:] = -Ad.T
        Z[n:n + p, :n] = Gd
        Z[n + p:, :n] = Ad
        Z[n + p:, n + p:] = FtF
        lu = sla.lu_factor(Z)                                 # :257

        def solve3x3(dy, dw, dv):
            z = sla.lu_solve(lu, np.concatenate([dy, dw, dv]))
            return z[:n], z[n:n + p], z[n + p:]

        return solve3x3

    return solve3x3gen


def assemble3x3(Q, A, G, F):
    """The literal 3x3 matrix of src/kktsolvers.jl:254-256 (for parity tests of
    the device assembly kernel)."""
    Qd = _dense(Q)
    Ad = _dense(A)
    n = Qd.shape[0]
    Gd = _dense(G).reshape(-1, n)
    m = Ad.shape[0]
    p = Gd.shape[0]
    Z = np.zeros((n + p + m, n + p + m))
    Z[:n, :n] = Qd
    Z[:n, n:n + p] = Gd.T
    Z[:n, n + p:] = -Ad.T
    Z[n:n + p, :n] = Gd
    Z[n + p:, :n] = Ad
    Z[n + p:, n + p:] = F.square().matrix()
    return Z


def kktsolver_2x2(Q, A, G, cone_dims):
    """src/kktsolvers.jl:281-310 -- Schur system [Q + A'F^-1F^-T A, G'; G, 0]."""
    Qd = _dense(Q)
    Ad = _dense(A)
    n = Qd.shape[0]
    Gd = _dense(G).reshape(-1, n)
    p = Gd.shape[0]

    def solve2x2gen(F, FinvT):
        FiT = FinvT.matrix()                                  # :289
        S = Qd + Ad.T @ (FiT.T @ (FiT @ Ad))                  # :290
        Z = np.zeros((n + p, n + p))
        Z[:n, :n] = S
        Z[:n, n:] = Gd.T
        Z[n:, :n] = Gd
        lu = sla.lu_factor(Z)                                 # :295

        def solve2x2(dy, dw):
            z = sla.lu_solve(lu, np.concatenate([dy, dw]))
            return z[:n], z[n:]

        return solve2x2

    return solve2x2gen


def schur2x2(Q, A, G, F):
    """The 2x2 Schur matrix with the mathematically exact (F'F)^-1 (what the
    device Schur-assembly kernel must reproduce)."""
    Qd = _dense(Q)
    Ad = _dense(A)
    n = Qd.shape[0]
    Gd = _dense(G).reshape(-1, n)
    p = Gd.shape[0]
    FtFi = np.linalg.inv(F.square().matrix())
    Z = np.zeros((n + p, n + p))
    Z[:n, :n] = Qd + Ad.T @ FtFi @ Ad
    Z[:n, n:] = Gd.T
    Z[n:, :n] = Gd
    return Z


def pivotgen(kkt2x2, Q, A, G, cone_dims):
    """src/kktsolvers.jl:316-338.  Note (:326,:328): (F'F)^-1 is applied as
    F^-T(F^-T .), exact only for symmetric blocks (R, Q cones)."""
    solve2x2gen = kkt2x2(Q, A, G, cone_dims)

    def solve3x3gen(F, FinvT):
        solve2x2 = solve2x2gen(F, FinvT)

        def solve3x3(y, w, v):
            t1 = FinvT.mul(FinvT.mul(v))                      # :326
            dy, dw = solve2x2(y + A.T @ t1, w)                # :327
            t1 = t1 - FinvT.mul(FinvT.mul(A @ dy))            # :328
            return dy, dw, t1

        return solve3x3

    return solve3x3gen


def pivot(kkt2x2):
    """src/kktsolvers.jl:349."""
    return lambda Q, A, G, cone_dims: pivotgen(kkt2x2, Q, A, G, cone_dims)


def kktsolver_schur_exact(Q, A, G, cone_dims):
    """NOT a reference solver: the block elimination of `pivot` (src/kktsolvers.jl:316-338) with the mathematically
    exact (F'F)^-1 = F^-1 F^-T and without ever forming F as a dense m x m matrix (columns of A are pushed through
    the Block operator instead).  Same 3x3 system, same solution as `kktsolver_qr`; exists so that the oracle can run
    problems whose single S cone makes `kktsolver_qr`'s dense F (k x k, k = r(r+1)/2) too large -- config 4 at
    matrix order >= 128.  (`pivot` itself is exact only for symmetric scalings: SURVEY App. C.4.)"""
    Qd = _dense(Q)
    Ad = _dense(A)
    n = Qd.shape[0]
    Gd = _dense(G).reshape(-1, n)
    p = Gd.shape[0]

    def solve3x3gen(F, FinvT):
        Fi = FinvT.adjoint()                                  # F^-1
        W = FinvT.mul(Ad)                                     # F^-T A   (m x n, block operator on every column)
        Z = np.zeros((n + p, n + p))
        Z[:n, :n] = Qd + W.T @ W
        Z[:n, n:] = Gd.T
        Z[n:, :n] = Gd
        lu = sla.lu_factor(Z)

        def solve3x3(x, y, z):
            t = Fi.mul(FinvT.mul(z))                          # (F'F)^-1 z
            ab = sla.lu_solve(lu, np.concatenate([x + Ad.T @ t, y]))
            a = ab[:n]
            return a, ab[n:], t - Fi.mul(FinvT.mul(Ad @ a))

        return solve3x3

    return solve3x3gen
