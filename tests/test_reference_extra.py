"""The remaining solver tests of the reference's suite (test/runtests.jl:271-356, :653-679) on the oracle (CPU) and on
the product (GPU): rank-one Hessians with equality constraints, equalities restated as inequality pairs, and the
three `imcols` cases.  The reference draws from Julia's RNG; what it asserts (status, feasibility, agreement of two
formulations, ranks) does not depend on the draw, so seeded numpy data of the same shapes is used."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle.conicip import conicIP as oracle_conicIP
from oracle.preprocess import imcols as oracle_imcols

TOL = 1e-3      # `tol`, test/runtests.jl:13
OPT = 1e-7      # `optTol`, :14


def rank_one_problem(seed=0, n=10):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n)
    H = np.outer(h, h)
    return H, H @ np.arange(1.0, n + 1), sp.identity(n, format="csr"), np.zeros(n), rng


def check_simplex_dense_h(solve):
    """:271-303  min 1/2 y'Hy - (Hc)'y  s.t.  y >= 0, sum(y) = 1, H = hh' (singular)."""
    H, Hc, A, b, _ = rank_one_problem()
    sol = solve(H, Hc, A, b, [("R", 10)], np.ones((1, 10)), np.array([1.0]), optTol=OPT)
    assert sol.status == "Optimal"
    assert abs(sol.y.sum() - 1.0) < 1e-6 and sol.y.min() > -1e-6
    assert max(sol.prFeas, sol.duFeas, sol.muFeas) < OPT


def check_random_projection_and_comparison(solve):
    """:305-356  six random equalities; the same feasible set written with the equalities doubled as inequality pairs
    must give the same minimiser."""
    H, Hc, A, b, rng = rank_one_problem()
    G, d = rng.random((6, 10)), np.zeros(6)
    s1 = solve(H, Hc, A, b, [("R", 10)], G, d, optTol=OPT)
    A2 = sp.vstack([A, sp.csr_matrix(G), sp.csr_matrix(-G)], format="csr")
    s2 = solve(H, Hc, A2, np.concatenate([b, d, -d]), [("R", 22)], G, d, optTol=OPT)
    assert s1.status == "Optimal" and s2.status == "Optimal"
    assert np.linalg.norm(s1.y - s2.y) < TOL
    assert np.abs(G @ s1.y).max() < 1e-6


def check_imcols(imcols):
    """:653-679"""
    rng = np.random.default_rng(42)
    A, b = rng.standard_normal((5, 10)), rng.standard_normal(5)
    R, ok = imcols(A, b)
    assert len(R) == np.linalg.matrix_rank(A) and ok
    A2, b2 = np.vstack([A, A[0:1] + A[1:2]]), np.concatenate([b, b[0:1] + b[1:2]])
    R2, ok2 = imcols(A2, b2)
    assert len(R2) == np.linalg.matrix_rank(A2) and ok2
    A3, b3 = np.vstack([A, A[0:1]]), np.concatenate([b, b[0:1] + 100.0])
    assert not imcols(A3, b3)[1]


def test_oracle_simplex_dense_h():
    check_simplex_dense_h(oracle_conicIP)


def test_oracle_linear_constraints_comparison():
    check_random_projection_and_comparison(oracle_conicIP)


def test_imcols_oracle_and_product():
    from cipkkt.preprocess import imcols
    check_imcols(oracle_imcols)
    check_imcols(imcols)


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["schur", "full3x3"])
def test_product_simplex_dense_h(route):
    import cipkkt
    check_simplex_dense_h(lambda *a, **k: cipkkt.conicIP(*a, kktsolver=route, **k))


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["schur", "full3x3"])
def test_product_linear_constraints_comparison(route):
    import cipkkt
    check_random_projection_and_comparison(lambda *a, **k: cipkkt.conicIP(*a, kktsolver=route, **k))


@pytest.mark.gpu
def test_product_matches_oracle_on_rank_one_hessian():
    import cipkkt
    H, Hc, A, b, rng = rank_one_problem(seed=4)
    G, d = rng.random((6, 10)), np.zeros(6)
    got = cipkkt.conicIP(H, Hc, A, b, [("R", 10)], G, d, optTol=OPT)
    ref = oracle_conicIP(H, Hc, A, b, [("R", 10)], G, d, optTol=OPT)
    assert got.status == ref.status == "Optimal" and got.Iter == ref.Iter
    np.testing.assert_allclose(got.y, ref.y, rtol=1e-6, atol=1e-8)


# ---- the documentation's worked examples with known answers (docs/src/tutorials/qp.jl:21-41, socp.jl:32-53)
def check_doc_examples(solve):
    n = 5
    p = np.arange(1.0, n + 1)                       # nearest point of the simplex to (1..5) is e_5
    sol = solve(sp.identity(n, format="csr"), p, sp.identity(n, format="csr"), np.zeros(n), [("R", n)], np.ones((1, n)),
                np.array([1.0]), optTol=1e-7)
    assert sol.status == "Optimal"
    # (y_4 sits exactly on its bound with a zero multiplier: no strict complementarity, so the interior-point iterate is
    #  only O(sqrt(optTol)) close there)
    np.testing.assert_allclose(sol.y, [0, 0, 0, 0, 1], atol=2e-3)
    n = 3
    a = np.ones(n)                                  # projection of (1,1,1) onto the unit ball: a / ||a||
    A = sp.vstack([sp.csr_matrix((1, n)), sp.identity(n)], format="csr")
    sol = solve(sp.identity(n, format="csr"), a, A, np.array([-1.0, 0, 0, 0]), [("Q", n + 1)], optTol=1e-7)
    assert sol.status == "Optimal"
    np.testing.assert_allclose(sol.y, a / np.linalg.norm(a), atol=1e-4)


def test_oracle_doc_examples():
    check_doc_examples(oracle_conicIP)


@pytest.mark.gpu
def test_product_doc_examples():
    import cipkkt
    check_doc_examples(cipkkt.conicIP)
