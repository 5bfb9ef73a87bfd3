"""BASELINE.json configs 3-5 at (or near) full size on the GPU, checked through size-independent
properties: the optimality conditions of the returned point (src/ConicIP.jl:763-788), agreement of
the two elimination routes, and -- where the oracle finishes in seconds -- agreement with the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import cones as oc

pytestmark = pytest.mark.gpu


def socp_problem(n, ncones, kq, p, seed):
    """Config 3 family: ncones x ("Q", kq), dense A (benchmark/profile.jl:53-69 pattern: head row b = -1),
    equality block G with d = 0 -- strictly feasible at y = 0."""
    rng = np.random.default_rng(seed)
    m = ncones * kq
    A = rng.standard_normal((m, n)) / np.sqrt(n)
    b = np.zeros(m)
    b[::kq] = -1.0
    G = rng.standard_normal((p, n))
    d = np.zeros(p)
    Q = np.eye(n)
    c = rng.standard_normal(n)
    return Q, c, A, b, [("Q", kq)] * ncones, G, d


def check_optimality(Q, c, A, b, K, G, d, sol, tol):
    y, w, v = sol.y, sol.w, sol.v
    s = A @ y - b
    rdu = np.linalg.norm(Q @ y + (G.T @ w if len(w) else 0) - A.T @ v - c) / (1 + np.linalg.norm(c))
    assert rdu < tol, "dual residual %g" % rdu
    if len(w):
        assert np.linalg.norm(G @ y - d) / (1 + np.linalg.norm(d)) < tol
    off = 0
    for t, k in K:
        sb, vb = s[off:off + k], v[off:off + k]
        if t == "R":
            assert sb.min() > -tol and vb.min() > -tol
        elif t == "Q":
            assert sb[0] - np.linalg.norm(sb[1:]) > -tol and vb[0] - np.linalg.norm(vb[1:]) > -tol
        else:
            assert np.linalg.eigvalsh(oc.mat(sb)).min() > -tol and np.linalg.eigvalsh(oc.mat(vb)).min() > -tol
        off += k
    assert abs(s @ v) / max(1.0, abs(c @ y)) < 10 * tol


def test_config3_socp_small_vs_oracle():
    import cipkkt
    from oracle.conicip import conicIP as oracle_conicIP
    prob = socp_problem(n=128, ncones=16, kq=8, p=16, seed=3)
    ref = oracle_conicIP(*prob, optTol=1e-7)
    for route in ("schur", "full3x3"):
        sol = cipkkt.conicIP(*prob, optTol=1e-7, kktsolver=route)
        assert sol.status == ref.status == "Optimal" and sol.Iter == ref.Iter
        np.testing.assert_allclose(sol.y, ref.y, rtol=1e-6, atol=1e-8)


def test_config3_socp_full_size():
    """n = 4096, 512 x ("Q", 8), p = 512: Schur route N = 4608 (quasi-definite LDL')."""
    import cipkkt
    prob = socp_problem(n=4096, ncones=512, kq=8, p=512, seed=11)
    sol = cipkkt.conicIP(*prob, optTol=1e-6)
    assert sol.status == "Optimal"
    check_optimality(*prob, sol, 1e-5)


def test_config4_sdp_moderate():
    """Config 4 family at matrix order r = 24 (k = 300): single S cone + linear constraints,
    strictly feasible at y = 0 (b = -vecm(I)), both routes and the oracle."""
    import cipkkt
    from oracle.conicip import conicIP as oracle_conicIP
    rng = np.random.default_rng(5)
    r, n, p = 24, 40, 6
    k = r * (r + 1) // 2
    A = rng.standard_normal((k, n)) / np.sqrt(n)
    b = -oc.vecm(np.eye(r))
    G = rng.standard_normal((p, n))
    d = np.zeros(p)
    Q, c = np.eye(n), rng.standard_normal(n)
    prob = (Q, c, A, b, [("S", k)], G, d)
    ref = oracle_conicIP(*prob, optTol=1e-7)
    s1 = cipkkt.conicIP(*prob, optTol=1e-7)
    s2 = cipkkt.conicIP(*prob, optTol=1e-7, kktsolver="full3x3")
    assert s1.status == s2.status == ref.status == "Optimal"
    assert s1.Iter == ref.Iter
    np.testing.assert_allclose(s1.y, ref.y, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(s2.y, ref.y, rtol=1e-6, atol=1e-8)
    check_optimality(*prob, s1, 1e-6)


def test_config5_batch_of_qps():
    """Config 5 family: independent dense QPs (n = 2048, A = I), here 3 of them through solve_batch on
    one rank; every problem must converge and satisfy its optimality conditions."""
    import cipkkt
    from cipkkt.batch import solve_batch
    probs = []
    n = 2048
    for i in range(3):
        rng = np.random.default_rng(500 + i)
        M = rng.standard_normal((n, n))
        probs.append(dict(Q=M.T @ M / n, c=rng.standard_normal(n), A=sp.identity(n, format="csr"), b=np.zeros(n),
                          cone_dims=[("R", n)], kwargs=dict(optTol=1e-6)))
    sols, stats = solve_batch(probs)
    assert stats["n_problems"] == 3 and stats["n_optimal"] == 3
    for i, pr in enumerate(probs):
        check_optimality(pr["Q"], pr["c"], pr["A"], pr["b"], pr["cone_dims"], np.zeros((0, n)), np.zeros(0), sols[i], 1e-5)


@pytest.mark.parametrize("native", [True, "handles", False], ids=["cip_conicip_problems", "cip_conicip_many", "python-threads"])
def test_config5_batch_in_flight(native):
    """The same batch with several problems in flight on separate HIP streams: through the library's batch entry
    point (csrc/batch.hip) and through Python threads; both must reproduce the one-at-a-time solutions exactly
    (independent problems share nothing)."""
    from cipkkt.batch import solve_batch
    probs = []
    n = 512
    for i in range(5):
        rng = np.random.default_rng(700 + i)
        M = rng.standard_normal((n, n))
        G = rng.standard_normal((3, n)) if i % 2 else None
        probs.append(dict(Q=M.T @ M / n, c=rng.standard_normal(n), A=sp.identity(n, format="csr"), b=np.zeros(n),
                          cone_dims=[("R", n)], G=G, d=None if G is None else np.zeros(3), kwargs=dict(optTol=1e-7)))
    ref, st0 = solve_batch(probs)
    got, st1 = solve_batch(probs, concurrency=3, native=native)
    assert st1["n_optimal"] == 5 and st1["n_factor"] == st0["n_factor"] and st1["n_solve"] == st0["n_solve"]
    for i in range(5):
        assert got[i].status == ref[i].status == "Optimal" and got[i].Iter == ref[i].Iter
        np.testing.assert_allclose(got[i].y, ref[i].y, rtol=1e-12, atol=0)
        np.testing.assert_allclose(got[i].w, ref[i].w, rtol=1e-12, atol=1e-300)
