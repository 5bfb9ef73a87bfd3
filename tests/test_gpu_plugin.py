"""The drop-in boundary driven the way the reference drives it: the reference-shaped interior-point loop (the
oracle's line-by-line restatement of src/ConicIP.jl:468-939) with the HIP library plugged in as `kktsolver`
(src/ConicIP.jl:667, :682, :688) -- including the initial-point call with F = F^-T = Block([Diagonal(ones(k)) ...])
for every cone type (:704-706) -- must walk the same trajectory as with the reference-faithful `kktsolver_qr`.
Also the 2x2 form (`pivot`, src/kktsolvers.jl:316-349; test/runtests.jl:90-131) in both directions: the device as the
2x2 solver under the oracle's pivot / loop, and a user-written Python 2x2 plugin under the device-resident loop."""
import numpy as np
import pytest
import scipy.sparse as sp

import problems as P
from oracle import cones as oc
from oracle.conicip import conicIP as oracle_conicIP, make_cone_ops
from oracle import kktsolvers as ok

pytestmark = pytest.mark.gpu


def mixed_rqs(seed=2):
    """R + Q + S cones with an equality block, strictly feasible at y = 1 (pattern of benchmark/profile.jl:116-160)."""
    rng = np.random.default_rng(seed)
    n, p, r = 30, 3, 4
    k = r * (r + 1) // 2
    M = rng.standard_normal((n, n))
    Q = M.T @ M / n + 0.1 * np.eye(n)
    Aq = rng.standard_normal((6, n)) * 0.2
    Aq[0] = 0
    As = rng.standard_normal((k, n)) * 0.2
    A = np.vstack([np.eye(n), Aq, As])
    one = np.ones(n)
    b = np.concatenate([np.zeros(n) - 0.5, np.concatenate([[-(np.linalg.norm(Aq[1:] @ one) + 1.0)], np.zeros(5)]),
                        As @ one - oc.vecm(np.eye(r) * 2.0)])
    G = rng.standard_normal((p, n))
    return Q, rng.standard_normal(n), A, b, [("R", n), ("Q", 6), ("S", k)], G, G @ one


PROBLEMS = {
    "sphere": lambda: P.sphere(2)[:7],
    "combined": lambda: P.combined(10)[:7],
    "soc_direct": lambda: P.soc_direct()[:7],
    "psd_projection": lambda: P.psd_projection()[:7],
    "simplex": lambda: P.simplex(10)[:7],
    "mixed_rqs": mixed_rqs,
    "random_mixed_csr": lambda: P.random_mixed(dense_A=False)[:7],
}


@pytest.mark.parametrize("name", sorted(PROBLEMS))
@pytest.mark.parametrize("solver", ["kktsolver_hip", "kktsolver_hip_full3x3"])
def test_reference_loop_through_the_hip_plugin(name, solver):
    import cipkkt
    prob = PROBLEMS[name]()
    ref = oracle_conicIP(*prob, optTol=1e-7)                                   # kktsolver_qr
    got = oracle_conicIP(*prob, optTol=1e-7, kktsolver=getattr(cipkkt, solver))
    assert got.status == ref.status == "Optimal"
    assert got.Iter == ref.Iter and got.n_factor == ref.n_factor
    # the number of refinement solves depends on whether a residual of ~1e-14 lands above or below the threshold
    # (src/ConicIP.jl:917): solver rounding decides, the trajectory does not change
    assert got.Iter + 1 <= got.n_solve <= ref.n_solve + 3
    for a, b in ((got.y, ref.y), (got.w, ref.w), (got.v, ref.v)):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-8)
    for tg, tr in zip(got.trace, ref.trace):
        # per-iteration duality measure: 1e-5 relative (the last iterations sit at mu ~ 1e-8 where the two solvers'
        # rounding shows in the sixth digit), iterates above are held to 1e-6
        assert abs(tg["mu"] - tr["mu"]) <= 1e-5 * abs(tr["mu"]) + 1e-12


def test_solve2x2_matches_the_schur_system():
    import cipkkt
    rng = np.random.default_rng(4)
    Q, c, A, b, K, G, d, _ = P.random_mixed(n=40, nq=3, kq=6, p=4, seed=1)
    m = A.shape[0]
    _, nt_scaling, _, _ = make_cone_ops(K)
    v = np.concatenate([rng.random(40) + 0.1] + [np.r_[np.linalg.norm(x[1:]) + 0.5, x[1:]] for x in rng.standard_normal((3, 6))])
    s = np.concatenate([rng.random(40) + 0.1] + [np.r_[np.linalg.norm(x[1:]) + 0.5, x[1:]] for x in rng.standard_normal((3, 6))])
    F = nt_scaling(v, s)
    Z = ok.schur2x2(Q, A, G, F)
    y, w = rng.standard_normal(40), rng.standard_normal(4)
    want = np.linalg.solve(Z, np.concatenate([y, w]))
    gen = cipkkt.kktsolver_2x2_hip(Q, A, G, K)
    dy, dw = gen(F, F.inv_adjoint())(y, w)
    np.testing.assert_allclose(np.concatenate([dy, dw]), want, rtol=1e-9, atol=1e-11)
    # the 3x3 route refuses the 2x2 form
    ks = cipkkt.KKTSystem(Q, A, G, K, route="full3x3")
    ks.set_scaling_identity()
    ks.factor()
    with pytest.raises(cipkkt.CipError):
        ks.solve2x2(y, w)
    ks.close()
    gen.system.close()


@pytest.mark.parametrize("name", ["combined", "simplex", "random_mixed_csr"])
def test_device_2x2_under_the_reference_pivot(name):
    """pivot(kktsolver_2x2) of the reference with the device as the 2x2 solver: oracle pivot and product pivot."""
    import cipkkt
    prob = PROBLEMS[name]()
    ref = oracle_conicIP(*prob, optTol=1e-7, kktsolver=ok.pivot(ok.kktsolver_2x2))
    for piv in (ok.pivot, cipkkt.pivot):
        got = oracle_conicIP(*prob, optTol=1e-7, kktsolver=piv(cipkkt.kktsolver_2x2_hip))
        assert got.status == ref.status == "Optimal" and got.Iter == ref.Iter
        np.testing.assert_allclose(got.y, ref.y, rtol=1e-6, atol=1e-8)


def test_user_plugin_under_the_device_loop_box_qp():
    """test/runtests.jl:90-131: box-constrained QP, H = I/2, solved through a user-written diagonal 2x2 plugin wrapped
    by `pivot`.  Here the loop is cipkkt.conicIP (device-resident), the plugin a Python callable reading the Block it
    is handed exactly as the reference's does (`inv(F[1]*F[1]).diag`)."""
    import cipkkt
    n = 1000
    H, c, A, b, K, _, _, ystar = P.box_qp(n)
    calls = dict(gen=0, solve=0)

    def kktsolver_2x2_box(Q, A_, G_, cone_dims):
        def solve2x2gen(F, FinvT):
            calls["gen"] += 1
            v = (F[0] * F[0]).inv().diag
            D = v[:n] + v[n:]
            invHD = 1.0 / (0.5 + D)

            def solve2x2(rhs, rhs2):
                calls["solve"] += 1
                return invHD * rhs, np.zeros(0)
            return solve2x2
        return solve2x2gen

    sol = cipkkt.conicIP(H, c, A, b, K, kktsolver=cipkkt.pivot(kktsolver_2x2_box), optTol=1e-6, DTB=0.01,
                         maxRefinementSteps=3)
    ref = oracle_conicIP(H, c, A, b, K, kktsolver=ok.pivot(ok.kktsolver_2x2), optTol=1e-6)
    builtin = cipkkt.conicIP(H, c, A, b, K, optTol=1e-6)
    assert sol.status == ref.status == builtin.status == "Optimal"
    assert sol.Iter == ref.Iter == builtin.Iter
    assert calls["gen"] == sol.n_factor and calls["solve"] == sol.n_solve
    np.testing.assert_allclose(sol.y, ref.y, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(sol.y, builtin.y, rtol=1e-6, atol=1e-8)
    # the reference asserts the optimality condition of the box projection (test/runtests.jl:116), not closeness to 1
    # (the first coordinate sits exactly on its bound with a zero multiplier)
    g = 0.5 * (sol.y - np.arange(1.0, n + 1))
    assert np.linalg.norm(sol.y - np.clip(sol.y - g, -1, 1)) / n < 1e-3          # optcond(...) < tol, test/runtests.jl:13, :120


def test_user_3x3_plugin_with_q_and_s_cones():
    """A user plugin that is simply the oracle's kktsolver_qr, driven by the device loop on a mixed R+Q+S problem: the
    Blocks rebuilt from the packed device scaling must be what the reference would have handed over."""
    import cipkkt
    from oracle.block import Block, Diagonal, SymWoodbury, VecCongurance

    def adapt(F):
        out = []
        for blk in F:
            if hasattr(blk, "R"):
                out.append(VecCongurance(blk.R))
            elif hasattr(blk, "B"):
                out.append(SymWoodbury(blk.A.diag, blk.B[:, 0], float(blk.D[0, 0])))
            else:
                out.append(Diagonal(blk.diag))
        return Block(out)

    def plugin(Q, A, G, K):
        gen = ok.kktsolver_qr(Q, A, G, K)
        return lambda F, FiT: gen(adapt(F), adapt(FiT))

    prob = mixed_rqs()
    ref = oracle_conicIP(*prob, optTol=1e-7)
    got = cipkkt.conicIP(*prob, optTol=1e-7, kktsolver=plugin)
    assert got.status == ref.status == "Optimal" and got.Iter == ref.Iter
    np.testing.assert_allclose(got.y, ref.y, rtol=1e-6, atol=1e-8)
