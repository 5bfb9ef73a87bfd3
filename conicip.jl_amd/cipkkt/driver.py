"""Host driver: the Mehrotra predictor-corrector loop of the reference's `conicIP`
(src/ConicIP.jl:468-939) with every vector resident in HBM and every vector / cone /
KKT operation executed by the HIP library through the C ABI.  The host only sees
scalars (residual norms, mu, step lengths) -- which is what the north-star asks for:
the loop stays on the host, the Newton step runs on the GPU.

Same keyword names and defaults as the reference (src/ConicIP.jl:498-509).  Quirks of
the reference are kept: a factorisation also happens in the
terminating iteration (:737 precedes :786), `rPr` ignores the equality residual (:765),
`norm(v4x1)` is the sum of block 2-norms (:61), the returned (y,w,v) is the last iterate.
"""
import math
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib as L
from .kkt import KKTSystem


@dataclass
class Solution:
    """src/ConicIP.jl:384-398."""
    y: np.ndarray
    w: np.ndarray
    v: np.ndarray
    status: str = "None"
    Iter: int = 0
    Mu: float = 0.0
    prFeas: float = math.inf
    duFeas: float = math.inf
    muFeas: float = math.inf
    pobj: float = math.inf
    dobj: float = -math.inf
    # extras (not in the reference struct)
    n_factor: int = 0
    n_solve: int = 0
    wall_s: float = 0.0
    trace: list = field(default_factory=list)


def _jlmax(*xs):
    for x in xs:
        if x != x:
            return math.nan
    return max(xs)


def _nrm(x2):
    return math.sqrt(x2) if x2 >= 0 else math.nan


def solution_from_result(res, y, w, v, trace=None):
    """cip_result (+ output vectors, optional trace rows) -> Solution."""
    sol = Solution(np.array(y, copy=True), np.array(w, copy=True), np.array(v, copy=True))
    sol.status = L.STATUS_NAMES[res.status]
    sol.Iter, sol.Mu = res.iter, res.mu
    sol.prFeas, sol.duFeas, sol.muFeas = res.prFeas, res.duFeas, res.muFeas
    sol.pobj, sol.dobj = res.pobj, res.dobj
    sol.n_factor, sol.n_solve = res.n_factor, res.n_solve
    sol.wall_s = res.wall_s
    names = ("Iter", "mu", "rDu", "rPr", "rCp", "pobj", "dobj", "alpha", "sigma")
    if trace is not None:
        for row in trace[:res.trace_rows]:
            d_ = dict(zip(names, row.tolist()))
            d_["Iter"] = int(d_["Iter"])
            if d_["alpha"] != d_["alpha"]:          # the terminating iteration takes no step
                del d_["alpha"], d_["sigma"]
            sol.trace.append(d_)
    return sol


def _conicIP_native(ks, c_h, b_h, d_h, n, m, p, optTol, DTB, infeasTol, refinementThreshold, maxRefinementSteps,
                    maxIters, verbose, t_start):
    """The loop of src/ConicIP.jl:730-934 inside the library (csrc/driver.hip: cip_conicip)."""
    import ctypes as C
    opt = L.CipOptions(optTol, DTB, infeasTol, refinementThreshold, maxRefinementSteps, maxIters, int(bool(verbose)))
    res = L.CipResult()
    y, w, v = np.zeros(max(n, 1)), np.zeros(max(p, 1)), np.zeros(max(m, 1))
    trace = np.zeros((max(maxIters, 1), L.TRACE_COLS))
    c_h, b_h, d_h = (np.ascontiguousarray(x, dtype=np.float64) for x in (c_h, b_h, d_h))
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    with torch.cuda.device(ks.device):
        L.check(ks.lib.cip_conicip(ks.h, ptr(c_h), ptr(b_h) if m else None, ptr(d_h) if p else None,
                                    C.byref(opt), ptr(y), ptr(w), ptr(v), C.byref(res), ptr(trace), int(maxIters)))
    sol = solution_from_result(res, y[:n], w[:p], v[:m], trace)
    sol.wall_s = time.perf_counter() - t_start
    return sol


def conicIP(Q, c, A, b, cone_dims, G=None, d=None, *,
            kktsolver="schur",
            optTol=1e-6, DTB=0.01, verbose=False,
            maxRefinementSteps=3, maxIters=100, cache_nestodd=False,
            infeasTol=None, refinementThreshold=None,
            device=None, system=None, keep_iterates=None, driver="native"):
    """minimize 1/2 y'Qy - c'y  s.t.  Ay - b in K,  Gy = d   (src/ConicIP.jl:411-430).

    `kktsolver` selects the elimination route of the HIP KKT path: "schur"
    (block elimination, ≙ pivot(kktsolver_2x2)) or "full3x3" (literal 3x3 assembly,
    ≙ kktsolver_sparse).  `system` may carry an already-built KKTSystem (level 1).

    `kktsolver` may also be a CALLABLE with the reference's plugin signature
    (src/ConicIP.jl:432-466): `kktsolver(Q, A, G, cone_dims) -> solve3x3gen`,
    `solve3x3gen(F, F_invT) -> solve3x3`, `solve3x3(x, y, z) -> (a, b, c)` -- e.g.
    `cipkkt.pivot(user_2x2)` as in test/runtests.jl:90-131.  The loop then stays on the
    device (scaling, cone algebra, residuals), the plugin receives host `cipkkt.blocks.Block`
    objects with the reference's element fields rebuilt from the device's packed scaling, and
    every 3x3 right-hand side / solution crosses PCIe (that is the reference's own boundary).

    `driver`: "native" runs the loop in C++ inside libcipkkt (`cip_conicip`, csrc/driver.hip); "python" runs
    the identical loop below through the per-operation C-ABI entry points (needed for `keep_iterates`)."""
    t_start = time.perf_counter()
    if infeasTol is None:
        infeasTol = optTol
    if refinementThreshold is None:
        refinementThreshold = optTol / 1e7
    c_h = np.asarray(c, dtype=np.float64).reshape(-1)
    b_h = np.asarray(b, dtype=np.float64).reshape(-1)
    n = c_h.size
    m = A.shape[0]
    d_h = np.zeros(0) if d is None else np.asarray(d, dtype=np.float64).reshape(-1)
    p = 0 if G is None else G.shape[0]
    # sanity checks (src/ConicIP.jl:537-542) -- raised before anything touches the GPU
    if Q.shape[0] != Q.shape[1]:
        raise ValueError("Q is not square")
    if b_h.size != m:
        raise ValueError("Inconsistency in inequalities")
    if Q.shape[0] != n or (m > 0 and A.shape[1] != n):
        raise ValueError("Inconsistency in inequalities/objective")
    if d_h.size != p:
        raise ValueError("Inconsistency in equalities")
    if p > 0 and G.shape[1] != n:
        raise ValueError("Inconsistency in equalities/objective")

    from . import kkt as _kkt
    plugin = None
    if callable(kktsolver):
        if kktsolver is _kkt.kktsolver_hip:
            kktsolver = "schur"
        elif kktsolver is _kkt.kktsolver_hip_full3x3:
            kktsolver = "full3x3"
        else:
            plugin, kktsolver = kktsolver, "schur"
    ks = system if system is not None else KKTSystem(Q, A, G, cone_dims, route=kktsolver, device=device)
    if driver == "native" and keep_iterates is None and plugin is None:
        return _conicIP_native(ks, c_h, b_h, d_h, n, m, p, optTol, DTB, infeasTol, refinementThreshold,
                               maxRefinementSteps, maxIters, verbose, t_start)
    dev = ks.device
    f64 = dict(dtype=torch.float64, device=dev)
    NT = n + p + 2 * m

    def vec4():
        return torch.zeros(NT, **f64)

    def parts(t):
        return t[:n], t[n:n + p], t[n + p:n + p + m], t[n + p + m:]

    c_d = torch.as_tensor(c_h, **f64)
    b_d = torch.as_tensor(b_h, **f64)
    d_d = torch.as_tensor(d_h, **f64)
    normc = float(np.linalg.norm(c_h))
    normd = -math.inf if p == 0 else float(np.linalg.norm(d_h))
    normb = float(np.linalg.norm(b_h)) if m > 0 else 0.0

    # conedim (:547-552) and e (:559-565)
    conedim = 0
    for t, k in ks.cone_dims:
        conedim += k if t == "R" else (1 if t == "Q" else int(round((math.sqrt(1 + 8 * k) - 1) / 2)))
    e = torch.zeros(max(m, 1), **f64)[:m]
    ks.cone_identity(e)

    z, r0, rleft, r, daff, dz, dzr, rIr, rkkt = (vec4() for _ in range(9))
    zy, zw, zv, zs = parts(z)
    lam = torch.zeros(max(m, 1), **f64)[:m]
    mb1 = torch.zeros(max(m, 1), **f64)[:m]
    mb2 = torch.zeros(max(m, 1), **f64)[:m]
    mb3 = torch.zeros(max(m, 1), **f64)[:m]
    Qy = torch.zeros(n, **f64)
    pinf = torch.zeros(n, **f64)
    Ays = torch.zeros(max(m, 1), **f64)[:m]
    Gy = torch.zeros(max(p, 1), **f64)[:p]
    counts = dict(factor=0, solve=0)

    user = dict(gen=None, solve=None)
    if plugin is not None:
        from .blocks import Block as _Block, Diagonal as _Diagonal, blocks_from_packed
        Gh = G if G is not None else np.zeros((0, n))
        user["gen"] = plugin(Q, A, Gh, ks.cone_dims)                       # level 1 (:667)
        t1_d = torch.zeros(max(m, 1), **f64)[:m]
        zin_d = torch.zeros(max(m, 1), **f64)[:m]
        t2_d = torch.zeros(max(m, 1), **f64)[:m]

    def factor(identity=False):                   # level 2 (:682)
        counts["factor"] += 1
        if plugin is None:
            ks.factor(check=False)                # enqueue only: the first solve resolves the pivot flag
            return
        if identity:
            I = _Block([_Diagonal(np.ones(k)) for _, k in ks.cone_dims])   # :704
            user["solve"] = user["gen"](I, I)
        else:
            F, FiT = blocks_from_packed(ks.cone_dims, ks.get_scaling_packed())
            user["solve"] = user["gen"](F, FiT)

    def solve4x4(lam_, rhs, out):                 # src/ConicIP.jl:684-692
        counts["solve"] += 1
        if plugin is None:
            ks.solve4x4_dev(lam_, rhs, out)
            return
        ry_, rw_, rv_, rs_ = parts(rhs)
        oy, ow, ov, os_ = parts(out)
        if m > 0:
            ks.cone_div(rs_, lam_, t1_d)                                   # q = r.s / lambda          (:686)
            ks.apply_F(L.OP_FT, t1_d, t1_d)                                # t1 = F'q                  (:687)
            zin_d.copy_(rv_)
            ks.axpby(1.0, t1_d, 1.0, zin_d)
        torch.cuda.synchronize(dev)
        a_, b_, c_ = user["solve"](ry_.cpu().numpy(), rw_.cpu().numpy(), zin_d.cpu().numpy())   # level 3 (:688)
        oy.copy_(torch.as_tensor(np.asarray(a_, dtype=np.float64).reshape(n)))
        if p > 0:
            ow.copy_(torch.as_tensor(np.asarray(b_, dtype=np.float64).reshape(p)))
        if m > 0:
            ov.copy_(torch.as_tensor(np.asarray(c_, dtype=np.float64).reshape(m)))
            ks.apply_F(L.OP_F, ov, t2_d)
            ks.apply_F(L.OP_FT, t2_d, t2_d)
            os_.copy_(t1_d)
            ks.axpby(-1.0, t2_d, 1.0, os_)                                  # ds = t1 - F'(F dv)         (:689)

    def kkt_apply(x, out):
        """out.y = Q x.y + G' x.w - A' x.v ; out.w = G x.y ; out.v = A x.y - x.s   (:747-749, :912-914)"""
        xy, xw, xv, xs = parts(x)
        oy, ow, ov, _ = parts(out)
        ks.gemv(L.MAT_Q, 0, 1.0, xy, 0.0, oy)
        if p > 0:
            ks.gemv(L.MAT_G, 1, 1.0, xw, 1.0, oy)
            ks.gemv(L.MAT_G, 0, 1.0, xy, 0.0, ow)
        if m > 0:
            ks.gemv(L.MAT_A, 1, -1.0, xv, 1.0, oy)
            ks.gemv(L.MAT_A, 0, 1.0, xy, 0.0, ov)
            ks.axpby(-1.0, xs, 1.0, ov)

    # ---------------------------------------------------------------- initial point (:704-713)
    ks.set_scaling_identity()
    factor(identity=True)
    if plugin is None:
        ks.check_factor()            # the solve below would otherwise run on an unverified factor (LPs fail here)
    r0y, r0w, r0v, r0s = parts(r0)
    r0y.copy_(c_d)
    r0w.copy_(d_d)
    r0v.copy_(b_d)
    r0s.zero_()
    solve4x4(e, r0, z)
    if m > 0:
        a_v, a_s = ks.maxstep_pair(zv, None, zs, None)
        ks.axpby(-a_v, e, 1.0, zv)
        ks.axpby(-a_s, e, 1.0, zs)

    sol = Solution(None, None, None)
    optBest = math.inf

    def finish(status):
        torch.cuda.synchronize(dev)
        zh = z.cpu().numpy()
        sol.status = status
        sol.y, sol.w, sol.v = zh[:n].copy(), zh[n:n + p].copy(), zh[n + p:n + p + m].copy()
        sol.n_factor, sol.n_solve = counts["factor"], counts["solve"]
        sol.wall_s = time.perf_counter() - t_start
        return sol

    rly, rlw, rlv, rls = parts(rleft)
    ry, rw, rv, rs = parts(r)
    for Iter in range(1, maxIters + 1):                                    # :730
        if keep_iterates is not None:
            keep_iterates.append(z.clone())
        if m > 0:
            ks.set_scaling_from_iterate(zv, zs, lam)                       # :732-735 (F, lambda = F v)
        factor()                                                           # :737 -> :682

        if m > 0:
            ks.cone_prod(lam, lam, rls)                                    # :746
        kkt_apply(z, rleft)                                                # :747-750
        # pieces needed by the certificates
        ks.gemv(L.MAT_Q, 0, 1.0, zy, 0.0, Qy)
        pinf.zero_()
        if p > 0:
            ks.gemv(L.MAT_G, 1, 1.0, zw, 0.0, pinf)
            Gy.copy_(rlw)
        if m > 0:
            ks.gemv(L.MAT_A, 1, -1.0, zv, 1.0, pinf)
            Ays.copy_(rlv)
        # r0 = rleft - (c, d, b, 0)   (:753)
        r0.copy_(rleft)
        ks.axpby(-1.0, c_d, 1.0, r0y)
        if p > 0:
            ks.axpby(-1.0, d_d, 1.0, r0w)
        if m > 0:
            ks.axpby(-1.0, b_d, 1.0, r0v)

        pairs = [(zv, zs), (c_d, zy), (r0y, r0y), (r0v, r0v), (r0s, r0s), (zy, Qy), (zw, r0w), (zv, r0v),
                 (d_d, zw), (b_d, zv), (pinf, pinf), (zy, zy), (zv, zv), (Ays, Ays), (Gy, Gy), (Qy, Qy)]
        (mubar, cTy, r0y2, r0v2, r0s2, yQy, wr0w, vr0v, dTw, bTv, pinf2, yy, vv, ays2, gy2, qy2) = ks.dots(pairs)
        mu = mubar / conedim if conedim > 0 else math.nan                  # :756-757
        rDu = _nrm(r0y2) / (1 + normc)                                     # :764
        rPr = (_nrm(r0v2) if m > 0 else 0.0) / (1 + normb)                 # :765
        rCp = (_nrm(r0s2) if m > 0 else 0.0) / (1 + abs(cTy))              # :766
        worst = _jlmax(rDu, rPr, rCp)
        if worst < optBest:                                                # :768-773
            sol.Iter, sol.Mu = Iter, mu
            sol.duFeas, sol.prFeas, sol.muFeas = rDu, rPr, rCp
            optBest = worst
        pobj = 0.5 * yQy - cTy                                             # :775
        dobj = pobj + wr0w + vr0v - mubar                                  # :776
        sol.pobj, sol.dobj = pobj, dobj
        sol.trace.append(dict(Iter=Iter, mu=mu, rDu=rDu, rPr=rPr, rCp=rCp, pobj=pobj, dobj=dobj))
        if verbose:
            print(" %6d | %-8.1e %-8.1e %-8.1e | % -8.1e % -8.1e" % (Iter, rDu, rPr, rCp, pobj, dobj))

        status = "None"
        if worst < optTol:                                                 # :786
            status = "Optimal"
        if not (p == 0 and m == 0):                                        # :790
            dTy_bTv = dTw - bTv                                            # :808
            if dTy_bTv < 0:
                p_unscaled = _nrm(pinf2)                                   # :810
                den = _nrm(yy) + (_nrm(vv) if m > 0 else 0.0)
                p_cvx = p_unscaled / den if den != 0 else math.inf         # :811
                p_ecos = p_unscaled / (max(1, normc) * abs(dTy_bTv))       # :812
                p_infeas = _jlmax(p_cvx, p_ecos)
            else:
                p_infeas = math.nan
            if p_infeas < infeasTol:                                       # :815-818
                finish("Infeasible")
                sol.y = np.full(n, np.nan)
                sol.w = sol.w / -dTy_bTv
                sol.v = sol.v / -dTy_bTv
                return sol
            d1 = -math.inf if m == 0 else _nrm(ays2)                       # :839
            d2 = -math.inf if p == 0 else _nrm(gy2)                        # :840
            d3 = _nrm(qy2)                                                 # :841
            if cTy > 0:
                d_cvx = _jlmax(d1 / max(1, normb), d2 / max(1, normd), d3 / max(1, normc)) / abs(cTy)   # :843
                ny = _nrm(yy)
                d_ecos = _jlmax(d1, d2, d3) / ny if ny != 0 else math.inf                             # :844
                d_infeas = abs(_jlmax(d_cvx, d_ecos))
            else:
                d_infeas = math.nan
            if d_infeas < infeasTol:                                       # :847-850
                finish("Unbounded")
                sol.y = sol.y / abs(cTy)
                sol.v = np.full(m, np.nan)
                sol.w = np.full(p, np.nan)
                return sol
        if status != "None":                                               # :867
            return finish(status)
        if not all(math.isfinite(x) for x in (mu, rDu, rPr, rCp)):         # :870-873
            return finish("Error")

        # ------------------------------------------------------------ predictor (:879-887)
        solve4x4(lam, r0, daff)
        _, _, dav, das = parts(daff)
        if m > 0:
            a_aff_v, a_aff_s = ks.maxstep_pair(zv, dav, zs, das)
            a_aff = min(a_aff_v, 1.0, a_aff_s)
            x1x2, x1y2, y1x2, y1y2 = ks.dots([(zv, zs), (zv, das), (dav, zs), (dav, das)])
            rho = (x1x2 - a_aff * x1y2 - a_aff * y1x2 + a_aff * a_aff * y1y2) / mubar    # fts :162-163,:886
            sigma = max(0.0, min(1.0, rho)) ** 3
        else:
            a_aff, sigma = 1.0, 0.0

        # ------------------------------------------------------------ corrector (:893-901)
        r.copy_(r0)
        if m > 0:
            ks.apply_F(L.OP_FINVT, das, mb1)                               # F^-T d_aff.s
            ks.apply_F(L.OP_F, dav, mb2)                                   # F d_aff.v
            ks.cone_prod(mb1, mb2, mb3)
            # lc = -(mb3 - sigma mu e) ; r.s = rleft.s - lc = rleft.s + mb3 - sigma mu e
            ks.axpby(1.0, mb3, 1.0, rs)
            ks.axpby(-sigma * mu, e, 1.0, rs)

        # ------------------------------------------------------------ Newton step + refinement (:907-921)
        solve4x4(lam, r, dz)
        dzy, dzw, dzv, dzs = parts(dz)
        rky, rkw, rkv, rks = parts(rkkt)
        for _ in range(maxRefinementSteps):
            kkt_apply(dz, rkkt)
            if m > 0:
                ks.apply_F(L.OP_F, dzv, mb1)
                ks.cone_prod(lam, mb1, mb2)
                ks.apply_F(L.OP_FINVT, dzs, mb1)
                ks.cone_prod(lam, mb1, mb3)
                rks.copy_(mb2)
                ks.axpby(1.0, mb3, 1.0, rks)
            rIr.copy_(r)
            ks.axpby(-1.0, rkkt, 1.0, rIr)
            iy, iw, iv, is_ = parts(rIr)
            n2 = ks.dots([(iy, iy), (iw, iw), (iv, iv), (is_, is_)] if (p > 0 and m > 0) else
                         ([(iy, iy), (iv, iv), (is_, is_)] if m > 0 else
                          ([(iy, iy), (iw, iw)] if p > 0 else [(iy, iy)])))
            rnorm = sum(_nrm(x) for x in n2) / (n + 2 * m)                 # :917 (norm(v4x1) :61)
            if rnorm < refinementThreshold:
                break
            solve4x4(lam, rIr, dzr)
            ks.axpby(1.0, dzr, 1.0, dz)                                    # :920

        # ------------------------------------------------------------ step (:927-932)
        if m > 0:
            a_v, a_s = ks.maxstep_pair(zv, dzv, zs, dzs, 1.0 / (1.0 - DTB))
            a_v, a_s = min(a_v, 1.0), min(a_s, 1.0)
            alpha = min(a_v, a_s)
        else:
            alpha = 1.0
        ks.axpby(-alpha, dz, 1.0, z)
        sol.trace[-1]["alpha"] = alpha
        sol.trace[-1]["sigma"] = sigma

    return finish("Abandoned")                                             # :936
