"""ORACLE (test infrastructure, NOT product code).

Line-by-line numpy restatement of the reference's interior-point driver
``conicIP`` (src/ConicIP.jl:468-939): Mehrotra predictor-corrector with
Nesterov-Todd scaling, the 4x4 -> 3x3 reduction (``solve4x4gen`` :669-694),
iterative refinement (:907-921) and the infeasibility certificates (:790-852).
Quirks of the reference are kept: ``sol.y/w/v`` alias the
iterate, a factorisation also happens in the terminating iteration, ``rPr``
ignores the equality residual, ``norm(v4x1)`` is a sum of block 2-norms.
"""
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from . import cones
from .block import Block, Diagonal, SymWoodbury, VecCongurance, identity_block
from .kktsolvers import kktsolver_qr


class V4:
    """v4x1 (src/ConicIP.jl:57-66)."""

    def __init__(self, y, w, v, s):
        self.y, self.w, self.v, self.s = y, w, v, s

    def __add__(a, b):
        return V4(a.y + b.y, a.w + b.w, a.v + b.v, a.s + b.s)

    def __sub__(a, b):
        return V4(a.y - b.y, a.w - b.w, a.v - b.v, a.s - b.s)

    def norm(a):
        return sum(np.linalg.norm(t) if len(t) else 0.0 for t in (a.y, a.w, a.v, a.s))


@dataclass
class Solution:
    """src/ConicIP.jl:384-398."""
    y: np.ndarray
    w: np.ndarray
    v: np.ndarray
    status: str = "None"
    Iter: int = 0
    Mu: float = 0.0
    prFeas: float = np.inf
    duFeas: float = np.inf
    muFeas: float = np.inf
    pobj: float = np.inf
    dobj: float = -np.inf
    # oracle-only bookkeeping (not in the reference struct)
    n_factor: int = 0
    n_solve: int = 0
    trace: list = field(default_factory=list)


def _normsafe(x):
    return 0.0 if len(x) == 0 else float(np.linalg.norm(x))


def _cum_ranges(sizes):
    """cum_range (src/ConicIP.jl:158-159), 0-based slices."""
    out, c = [], 0
    for k in sizes:
        out.append(slice(c, c + k))
        c += k
    return out


def cone_identity(cone_dims):
    """e (src/ConicIP.jl:559-565) and conedim (:547-552)."""
    m = sum(k for _, k in cone_dims)
    e = np.zeros(m)
    conedim = 0
    for (t, k), I in zip(cone_dims, _cum_ranges([k for _, k in cone_dims])):
        if t == "R":
            e[I] = 1.0
            conedim += k
        elif t == "Q":
            e[I.start] = 1.0
            conedim += 1
        elif t == "S":
            r = cones.ord_(np.zeros(k))
            e[I] = cones.vecm(np.eye(r))
            conedim += r
        else:
            raise ValueError("unknown cone type %r" % (t,))
    return e, conedim


def make_cone_ops(cone_dims):
    """The cone-dispatch closures of src/ConicIP.jl:571-665."""
    types = [t for t, _ in cone_dims]
    ranges = _cum_ranges([k for _, k in cone_dims])

    def maxstep(x, d):                                        # :571-587
        mn = np.inf
        for t, I in zip(types, ranges):
            xI = x[I]
            dI = None if d is None else d[I]
            if t == "R":
                a = cones.maxstep_rp(xI, dI)
            elif t == "Q":
                a = cones.maxstep_soc(xI, dI)
            else:
                a = cones.maxstep_sdc(xI, dI)
            mn = min(a, mn)
        return mn

    def nt_scaling(x, y):                                     # :589-605
        blocks = []
        for t, I in zip(types, ranges):
            xI, yI = x[I], y[I]
            if t == "R":
                blocks.append(Diagonal(np.sqrt(yI / xI)))     # :598
            elif t == "Q":
                beta, w = cones.nestod_soc(xI, yI)            # :599
                k = len(xI)
                J = np.full(k, beta)
                J[0] = -beta
                blocks.append(SymWoodbury(J, w, 1.0))
            else:
                blocks.append(VecCongurance(cones.nestod_sdc(xI, yI)))   # :600
        return Block(blocks)

    def cone_div(x, y):                                       # :607-635
        o = np.zeros(len(x))
        for t, I in zip(types, ranges):
            if t == "R":
                o[I] = cones.drp(x[I], y[I])
            elif t == "Q":
                o[I] = cones.dsoc(x[I], y[I])
            else:
                o[I] = cones.dsdc(x[I], y[I])
        return o

    def cone_prod(x, y):                                      # :637-665
        o = np.zeros(len(x))
        for t, I in zip(types, ranges):
            if t == "R":
                o[I] = cones.xrp(x[I], y[I])
            elif t == "Q":
                o[I] = cones.xsoc(x[I], y[I])
            else:
                o[I] = cones.xsdc(x[I], y[I])
        return o

    return maxstep, nt_scaling, cone_div, cone_prod


def conicIP(Q, c, A, b, cone_dims, G=None, d=None, *,
            kktsolver=kktsolver_qr,
            optTol=1e-6, DTB=0.01, verbose=False,
            maxRefinementSteps=3, maxIters=100, cache_nestodd=False,
            infeasTol=None, refinementThreshold=None):
    """src/ConicIP.jl:468-939."""
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    n = len(c)
    if G is None:
        G = sp.csr_matrix((0, n))
    if d is None:
        d = np.zeros(0)
    d = np.asarray(d, dtype=np.float64).reshape(-1)
    if infeasTol is None:
        infeasTol = optTol                                    # :506
    if refinementThreshold is None:
        refinementThreshold = optTol / 1e7                    # :509

    m = A.shape[0]
    p = G.shape[0]
    AT = A.T
    GT = G.T

    normc = float(np.linalg.norm(c))
    normd = -np.inf if len(d) == 0 else float(np.linalg.norm(d))
    normb = _normsafe(b)

    # sanity checks :537-542
    if Q.shape[0] != Q.shape[1]:
        raise ValueError("Q is not square")
    if len(b) != m:
        raise ValueError("Inconsistency in inequalities")
    if Q.shape[0] != n or A.shape[1] != n:
        raise ValueError("Inconsistency in inequalities/objective")
    if len(d) != p:
        raise ValueError("Inconsistency in equalities")
    if G.shape[1] != n:
        raise ValueError("Inconsistency in equalities/objective")
    if sum(k for _, k in cone_dims) != m:
        raise ValueError("cone_dims do not cover the rows of A")

    block_sizes = [k for _, k in cone_dims]
    e, conedim = cone_identity(cone_dims)
    maxstep, nt_scaling, cone_div, cone_prod = make_cone_ops(cone_dims)

    solve3x3gen = kktsolver(Q, A, G, cone_dims)               # :667
    counts = {"factor": 0, "solve": 0}

    def solve4x4gen(lam, F, FinvT):                           # :669-694
        solve3x3 = solve3x3gen(F, FinvT)                      # :682
        counts["factor"] += 1

        def solve4x4(r):
            counts["solve"] += 1
            q = cone_div(r.s, lam)                            # :686
            t1 = F.tmul(q)                                    # :687
            dy, dw, dv = solve3x3(r.y, r.w, r.v + t1)         # :688
            t1 = t1 - F.tmul(F.mul(dv))                       # :689
            return V4(np.array(dy, dtype=np.float64), np.array(dw, dtype=np.float64),
                      np.array(dv, dtype=np.float64), t1)

        return solve4x4

    # initial point :704-713
    I = identity_block(block_sizes)
    r0 = V4(c, d, b, np.zeros(m))
    z = solve4x4gen(e, I, I)(r0)
    a_v = maxstep(z.v, None)
    a_s = maxstep(z.s, None)
    z.v = z.v - a_v * e
    z.s = z.s - a_s * e

    sol = Solution(z.y, z.w, z.v)                             # :726 (aliases z)
    optBest = np.inf
    for Iter in range(1, maxIters + 1):                       # :730
        F = nt_scaling(z.v, z.s)                              # :732
        FinvT = F.inv_adjoint()                               # :733
        lam = F.mul(z.v)                                      # :735
        solve = solve4x4gen(lam, F, FinvT)                    # :737

        lamlam = cone_prod(lam, lam)                          # :746
        Qy = Q @ z.y
        rleft = V4(Qy + GT @ z.w - AT @ z.v, G @ z.y, A @ z.y - z.s, lamlam)   # :747-750
        r0 = V4(rleft.y - c, rleft.w - d, rleft.v - b, rleft.s)               # :753

        mubar = float(np.dot(z.v, z.s))                       # :756
        mu = mubar / conedim if conedim > 0 else np.nan      # :757 (Julia: 0.0/0 = NaN, no exception)

        cTy = float(np.dot(c, z.y))                           # :763
        rDu = float(np.linalg.norm(r0.y)) / (1 + normc)       # :764
        rPr = _normsafe(r0.v) / (1 + normb)                   # :765
        rCp = _normsafe(r0.s) / (1 + abs(cTy))                # :766

        if max(rDu, rPr, rCp) < optBest:                      # :768-773
            sol.Iter = Iter
            sol.Mu = mu
            sol.duFeas, sol.prFeas, sol.muFeas = rDu, rPr, rCp
            optBest = max(rDu, rPr, rCp)

        pobj = 0.5 * float(np.dot(z.y, Qy)) - cTy             # :775
        dobj = pobj + float(np.dot(z.w, r0.w)) + float(np.dot(z.v, r0.v)) - mubar   # :776
        sol.pobj, sol.dobj = pobj, dobj
        sol.trace.append(dict(Iter=Iter, mu=mu, rDu=rDu, rPr=rPr, rCp=rCp,
                              pobj=pobj, dobj=dobj))

        if max(rDu, rPr, rCp) < optTol:                       # :786
            sol.status = "Optimal"

        if not (p == 0 and m == 0):                           # :790
            dTy_bTv = float(np.dot(d, z.w)) - float(np.dot(b, z.v))           # :808
            p_unscaled = float(np.linalg.norm(GT @ z.w - AT @ z.v))           # :810
            if dTy_bTv < 0:
                with np.errstate(divide="ignore", invalid="ignore"):
                    p_cvx = p_unscaled / (_normsafe(z.y) + _normsafe(z.v))    # :811
                    p_ecos = p_unscaled / (max(1, normc) * abs(dTy_bTv))      # :812
                p_infeas = _jlmax(p_cvx, p_ecos)
            else:
                p_infeas = np.nan
            if p_infeas < infeasTol:                          # :815-818
                sol.y = np.full(n, np.nan)
                sol.w = z.w / -dTy_bTv
                sol.v = z.v / -dTy_bTv
                sol.status = "Infeasible"

            if sol.status == "Infeasible":
                # the reference NaN-fills sol.y, which aliases z.y (:726,:816): every
                # quantity of the dual-infeasibility test below is then NaN -> skipped
                sol.n_factor, sol.n_solve = counts["factor"], counts["solve"]
                return sol
            d1 = -np.inf if m == 0 or n == 0 else float(np.linalg.norm(A @ z.y - z.s))   # :839
            d2 = -np.inf if p == 0 or n == 0 else float(np.linalg.norm(G @ z.y))         # :840
            d3 = float(np.linalg.norm(Qy)) if np.all(np.isfinite(z.y)) else np.nan       # :841
            if cTy > 0:
                d_cvx = _jlmax(_jlmax(d1 / max(1, normb), d2 / max(1, normd)),
                               d3 / max(1, normc)) / abs(cTy)                                 # :843
                d_ecos = _jlmax(_jlmax(d1, d2), d3) / float(np.linalg.norm(z.y))              # :844
                d_infeas = abs(_jlmax(d_cvx, d_ecos))
            else:
                d_infeas = np.nan
            if d_infeas < infeasTol:                          # :847-850
                sol.y = z.y / abs(cTy)
                sol.v = np.full(m, np.nan)
                sol.w = np.full(p, np.nan)
                sol.status = "Unbounded"

        if sol.status != "None":                              # :867
            sol.n_factor, sol.n_solve = counts["factor"], counts["solve"]
            if sol.status == "Optimal":
                sol.y, sol.w, sol.v = z.y, z.w, z.v
            return sol

        if not np.all(np.isfinite([mu, rDu, rPr, rCp])):      # :870-873
            sol.status = "Error"
            sol.n_factor, sol.n_solve = counts["factor"], counts["solve"]
            sol.y, sol.w, sol.v = z.y, z.w, z.v
            return sol

        # predictor :879-887
        d_aff = solve(r0)
        a_aff_v = min(maxstep(z.v, d_aff.v), 1)
        a_aff_s = min(maxstep(z.s, d_aff.s), 1)
        a_aff = min(a_aff_v, a_aff_s)
        rho = cones.fts(z.v, a_aff, d_aff.v, z.s, a_aff, d_aff.s) / mubar
        sigma = max(0, min(1, rho)) ** 3

        # corrector :893-901
        FiTdfs = FinvT.mul(d_aff.s)
        Fdfs = F.mul(d_aff.v)
        lc = cone_prod(FiTdfs, Fdfs)
        lc = -(lc - sigma * mu * e)
        r = V4(r0.y, r0.w, r0.v, rleft.s - lc)

        # newton step + refinement :907-921
        dz = solve(r)
        for rStep in range(1, maxRefinementSteps + 1):
            pb1 = cone_prod(lam, F.mul(dz.v))
            pb2 = cone_prod(lam, FinvT.mul(dz.s))
            rkkt = V4(Q @ dz.y + GT @ dz.w - AT @ dz.v, G @ dz.y, A @ dz.y - dz.s, pb1 + pb2)
            rIr = r - rkkt
            rnorm = rIr.norm() / (n + 2 * m)
            if rnorm < refinementThreshold:
                break
            dzr = solve(rIr)
            dz = dz + dzr

        # step :927-932
        a_v = min(maxstep(z.v, dz.v / (1 - DTB)), 1)
        a_s = min(maxstep(z.s, dz.s / (1 - DTB)), 1)
        alpha = min(a_v, a_s)
        z.y = z.y - alpha * dz.y
        z.w = z.w - alpha * dz.w
        z.v = z.v - alpha * dz.v
        z.s = z.s - alpha * dz.s
        sol.trace[-1]["alpha"] = alpha
        sol.trace[-1]["sigma"] = sigma

    sol.status = "Abandoned"                                  # :936
    sol.n_factor, sol.n_solve = counts["factor"], counts["solve"]
    sol.y, sol.w, sol.v = z.y, z.w, z.v
    return sol


def _jlmax(a, b):
    """Julia's max propagates NaN."""
    if np.isnan(a) or np.isnan(b):
        return np.nan
    return max(a, b)
