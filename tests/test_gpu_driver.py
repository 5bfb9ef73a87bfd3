"""GPU parity of the whole Newton-step path inside the interior-point loop: the
product driver (HIP KKT path) against the oracle's restatement of conicIP on the
reference's own known-answer problems (test/runtests.jl) -- same status, same
iteration count, iterates equal to 1e-6 relative (the tolerance the north-star asks
to be stated), and the analytic answers the reference asserts."""
import os
import numpy as np
import pytest

import problems as P
from oracle.conicip import conicIP as oracle_conicIP

pytestmark = pytest.mark.gpu
TOL = 1e-3
OPT = 1e-7
ROUTES = ["schur", "full3x3"]


def run_both(prob, route, **kw):
    import cipkkt
    Q, c, A, b, K, G, d, expect = prob
    ref = oracle_conicIP(Q, c, A, b, K, G, d, **kw)
    got = cipkkt.conicIP(Q, c, A, b, K, G, d, kktsolver=route, **kw)
    return got, ref, expect


def assert_same_trajectory(got, ref, rtol=1e-6):
    assert got.status == ref.status
    assert got.Iter == ref.Iter, "iterations differ: hip %d vs oracle %d" % (got.Iter, ref.Iter)
    assert got.n_factor == ref.n_factor
    for tg, tr in zip(got.trace, ref.trace):
        assert tg["mu"] == pytest.approx(tr["mu"], rel=1e-3, abs=1e-14)   # late iterates amplify rounding
    if ref.status == "Optimal":
        scale = 1 + np.linalg.norm(ref.y)
        assert np.linalg.norm(got.y - ref.y) / scale < rtol
        assert np.linalg.norm(got.v - ref.v) / (1 + np.linalg.norm(ref.v)) < rtol
        if len(ref.w):
            assert np.linalg.norm(got.w - ref.w) / (1 + np.linalg.norm(ref.w)) < rtol
        assert got.pobj == pytest.approx(ref.pobj, rel=1e-7, abs=1e-9)


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", ["sphere", "combined", "simplex", "soc_direct", "lp_doc"])
def test_reference_kats(name, route):
    got, ref, expect = run_both(getattr(P, name)(), route, optTol=OPT, DTB=0.01, maxRefinementSteps=3)
    assert got.status == "Optimal"
    assert np.linalg.norm(got.y - expect) < TOL           # the reference's own assertion
    assert_same_trajectory(got, ref)


@pytest.mark.parametrize("route", ROUTES)
def test_statuses(route):
    import cipkkt
    Q, c, A, b, K, G, d, _ = P.simplex()
    assert cipkkt.conicIP(Q, c, A, b, K, G, d, optTol=OPT, maxIters=2, kktsolver=route).status == "Abandoned"
    for prob in (P.infeasible_box(), P.infeasible_eq()):
        Q, c, A, b, K, G, d, _ = prob
        assert cipkkt.conicIP(Q, c, A, b, K, G, d, optTol=OPT, kktsolver=route).status == "Infeasible"


def test_unbounded_status():
    """test/runtests.jl:487-505 (Q = 0: Schur matrix is A'F^-2A = F^-2 > 0)."""
    import cipkkt
    Q, c, A, b, K, G, d, _ = P.unbounded()
    sol = cipkkt.conicIP(Q, c, A, b, K, G, d, optTol=OPT)
    assert sol.status == "Unbounded"


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("dense_A", [True, False])
def test_random_mixed(route, dense_A):
    got, ref, _ = run_both(P.random_mixed(n=60, nq=4, kq=7, p=5, dense_A=dense_A), route, optTol=1e-8)
    assert got.status == "Optimal"
    assert_same_trajectory(got, ref)


def test_box_qp_n1000():
    """README / runtests.jl:90-131 box QP at n=1000 (m=2000, sparse A): optimality condition."""
    import cipkkt
    Q, c, A, b, K, G, d, _ = P.box_qp(1000)
    sol = cipkkt.conicIP(Q, c, A, b, K, optTol=OPT, DTB=0.01, maxRefinementSteps=3)
    assert sol.status == "Optimal"
    cvec = np.arange(1.0, 1001)
    grad = 0.5 * (sol.y - cvec)
    assert np.linalg.norm(sol.y - np.clip(sol.y - grad, -1, 1)) / 1000 < TOL
    ref = oracle_conicIP(Q, c, A, b, K, optTol=OPT, DTB=0.01, maxRefinementSteps=3,
                         kktsolver=__import__("oracle.kktsolvers", fromlist=["x"]).pivot(
                             __import__("oracle.kktsolvers", fromlist=["x"]).kktsolver_2x2))
    assert_same_trajectory(sol, ref)


def test_dense_qp_2048_properties():
    """BASELINE config family (dense QP, A = I, R cone) at n = 2048: size-independent
    properties -- KKT optimality conditions of the returned point and agreement of the two
    elimination routes."""
    import cipkkt
    rng = np.random.default_rng(7)
    n = 2048
    M = rng.standard_normal((n, n))
    Q = M.T @ M / n
    c = rng.standard_normal(n)
    import scipy.sparse as sp
    A = sp.identity(n, format="csr")
    b = np.zeros(n)
    K = [("R", n)]
    k1 = cipkkt.KKTSystem(Q, A, None, K, route="schur")
    s1 = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6, system=k1)
    h1 = k1.health(); k1.close()
    assert s1.status == "Optimal"
    y, v = s1.y, s1.v
    assert np.linalg.norm(Q @ y - c - v) / (1 + np.linalg.norm(c)) < 1e-6      # stationarity
    assert y.min() > -1e-6 and v.min() > -1e-9                                # primal / dual feasibility
    assert abs(y @ v) / n < 1e-5                                              # complementarity
    k2 = cipkkt.KKTSystem(Q, A, None, K, route="full3x3")
    s2 = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6, system=k2)
    h2 = k2.health(); k2.close()
    assert s2.status == "Optimal" and s2.Iter == s1.Iter
    dev = np.linalg.norm(s1.y - s2.y) / (1 + np.linalg.norm(s1.y))
    # the two routes factor different matrices (order 2048 and 4096) and still end within 1e-15 of each other on this problem
    # (every Newton step is refined against the same operator); the optimality tolerance alone would allow 1e-6 -- which is what
    # round 6 saw once in ~40 suite runs, a factorisation with a few stale strips (diag.hip: PANEL_STAGE_LAST).  1e-9 keeps such a
    # run from passing.  What is printed when it fails says which of the two routes left its usual path, and at which iteration.
    import hashlib
    diag = "schur: sha1 %s health %s\nfull3x3: sha1 %s health %s\n" % (
        hashlib.sha1(s1.y.tobytes() + s1.v.tobytes()).hexdigest()[:12], h1, hashlib.sha1(s2.y.tobytes() + s2.v.tobytes()).hexdigest()[:12], h2)
    diag += "\n".join("it %d  mu %.17g / %.17g  pobj %.17g / %.17g  alpha %s / %s" % (a["Iter"], a["mu"], b_["mu"], a["pobj"], b_["pobj"], a.get("alpha"), b_.get("alpha"))
                      for a, b_ in zip(s1.trace, s2.trace))
    if os.environ.get("CIP_TEST_DUMP") and dev > 1e-12:
        with open(os.environ["CIP_TEST_DUMP"], "a") as f: f.write("deviation %.3e\n%s\n\n" % (dev, diag))
    assert dev < 1e-9, diag


@pytest.mark.parametrize("name", ["sphere", "combined", "simplex", "soc_direct", "lp_doc", "psd_projection",
                                  "infeasible_box", "infeasible_eq", "unbounded"])
def test_native_driver_matches_per_operation_driver(name):
    """cip_conicip (csrc/driver.hip) and the Python loop over the per-operation entry points issue the same
    kernels in the same order: statuses, iteration counts and the whole trajectory agree (1e-12 relative)."""
    import cipkkt
    Q, c, A, b, K, G, d = getattr(P, name)()[:7]
    sols = [cipkkt.conicIP(Q, c, A, b, K, G, d, optTol=1e-7, driver=drv) for drv in ("native", "python")]
    nat, py = sols
    assert nat.status == py.status
    assert (nat.Iter, nat.n_factor, nat.n_solve, len(nat.trace)) == (py.Iter, py.n_factor, py.n_solve, len(py.trace))
    for tn, tp in zip(nat.trace, py.trace):
        assert set(tn) == set(tp)
        for key in tn:
            assert tn[key] == pytest.approx(tp[key], rel=1e-12, abs=1e-300), (key, tn["Iter"])
    for xn, xp in ((nat.y, py.y), (nat.w, py.w), (nat.v, py.v)):
        np.testing.assert_allclose(xn, xp, rtol=1e-12, atol=0, equal_nan=True)
    for f in ("Mu", "prFeas", "duFeas", "muFeas", "pobj", "dobj"):
        assert getattr(nat, f) == pytest.approx(getattr(py, f), rel=1e-12)


def test_native_driver_maxiters_abandoned():
    """test/runtests.jl:246-269: maxIters = 2 -> :Abandoned, through the native loop."""
    import cipkkt
    Q, c, A, b, K, G, d = P.simplex()[:7]
    sol = cipkkt.conicIP(Q, c, A, b, K, G, d, maxIters=2)
    assert sol.status == "Abandoned" and len(sol.trace) == 2


@pytest.mark.parametrize("drv", ["native", "python"])
def test_degenerate_shapes(drv):
    """No inequality rows / no cones (m = 0), with and without equalities, and a 1 x 1 problem: the reference's loop
    handles them (the first Newton solve is the answer); so do both drivers, in agreement with the oracle."""
    import cipkkt
    rng = np.random.default_rng(0)
    n = 7
    M = rng.standard_normal((n, n))
    Q, c = M.T @ M + np.eye(n), rng.standard_normal(n)
    A0, b0 = np.zeros((0, n)), np.zeros(0)
    s = cipkkt.conicIP(Q, c, A0, b0, [], driver=drv)
    r = oracle_conicIP(Q, c, A0, b0, [])
    assert s.status == r.status == "Optimal"
    np.testing.assert_allclose(s.y, np.linalg.solve(Q, c), rtol=1e-12)
    G, d = rng.standard_normal((2, n)), rng.standard_normal(2)
    s = cipkkt.conicIP(Q, c, A0, b0, [], G, d, driver=drv)
    r = oracle_conicIP(Q, c, A0, b0, [], G, d)
    assert s.status == r.status == "Optimal"
    np.testing.assert_allclose(s.y, r.y, rtol=1e-10, atol=1e-12)
    assert np.abs(G @ s.y - d).max() < 1e-12
    s = cipkkt.conicIP(np.array([[2.0]]), np.array([-1.0]), np.array([[1.0]]), np.array([0.0]), [("R", 1)], driver=drv)
    assert s.status == "Optimal" and abs(s.y[0]) < 1e-5


@pytest.mark.parametrize("route", ROUTES)
def test_graph_replay_matches_plain_launches(route, monkeypatch):
    """CIP_GRAPH=1 (opt-in): factor + solves of a handle are replayed from a captured hipGraph -- the one-launch-per-panel
    chain with its in-launch counters included.  Same kernels on the same data: the iterates must be bit-identical."""
    import cipkkt
    Q, c, A, b, K, G, d, _ = P.random_mixed(n=200, nq=3, kq=6, p=4, seed=77)
    monkeypatch.delenv("CIP_GRAPH", raising=False)
    plain = cipkkt.conicIP(Q, c, A, b, K, G, d, kktsolver=route)
    monkeypatch.setenv("CIP_GRAPH", "1")
    graph = cipkkt.conicIP(Q, c, A, b, K, G, d, kktsolver=route)
    assert plain.status == graph.status == "Optimal"
    assert (plain.Iter, plain.n_factor, plain.n_solve) == (graph.Iter, graph.n_factor, graph.n_solve)
    for f in ("y", "w", "v"):
        assert np.array_equal(getattr(plain, f), getattr(graph, f)), f
