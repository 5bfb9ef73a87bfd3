"""Pre-solve (src/preprocessor.jl): the reference's three preprocessor tests (test/runtests.jl:362-440) on the
oracle restatement (CPU) and on the product's host-side pre-solve in front of the device solver (GPU), plus the
rank logic of `imcols` itself.  The reference draws its data from Julia's RNG; the properties it asserts do not
depend on the draw, so seeded numpy data of the same shape is used."""
import numpy as np
import pytest
import scipy.sparse as sp

import problems as P
from oracle.preprocess import imcols as o_imcols, preprocess_conicIP as o_pre

TOL = 1e-3       # `tol` of test/runtests.jl:13
OPT = 1e-7


def redundant_equalities(seed=0, n=10):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n)
    H = np.outer(h, h)
    c = np.arange(1.0, n + 1)
    G = rng.random((6, n))
    return H, H @ c, sp.identity(n, format="csr"), np.zeros(n), np.vstack([G, G]), np.zeros(12), G


def check_imcols(imcols):
    rng = np.random.default_rng(1)
    G = rng.standard_normal((4, 9))
    A = np.vstack([G, 2 * G[1:2], G[0:1] - G[3:4]])
    rows, ok = imcols(A, A @ rng.standard_normal(9))
    assert ok and len(rows) == 4 and np.linalg.matrix_rank(A[rows]) == 4
    rows, ok = imcols(np.array([[1.0, 0, 0], [1.0, 0, 0]]), np.array([1.0, -1.0]))   # inconsistent duplicate
    assert (rows, ok) == ([], False)
    assert imcols(np.zeros((0, 5)), np.zeros(0)) == ([], True)
    assert imcols(np.zeros((2, 5)), np.array([1.0, 0.0])) == ([], True)      # all-zero rows: the reference's empty-R branch


def check_redundant(pre):
    """test/runtests.jl:357-388: duplicated equality rows; the same problem with the equalities also stated as
    inequalities has the same solution."""
    H, Hc, A, b, G2, d2, G = redundant_equalities()
    n = 10
    s1 = pre(H, Hc, A, b, [("R", n)], G2, d2, optTol=OPT)
    A2 = sp.vstack([A, sp.csr_matrix(G2), sp.csr_matrix(-G2)], format="csr")
    s2 = pre(H, Hc, A2, np.concatenate([b, d2, -d2]), [("R", n + 24)], G2, d2, optTol=OPT)
    assert s1.status == "Optimal" and s2.status == "Optimal"
    assert np.linalg.norm(s1.y - s2.y) < TOL
    assert s1.w.shape == (12,) and np.count_nonzero(s1.w) <= 6       # zeros re-inserted for the dropped rows
    assert np.abs(G @ s1.y).max() < 1e-5


def check_bad_dual(pre):
    """test/runtests.jl:390-408: Q = 0 and A = [I I] leave n directions undetermined; the augmented Q picks y = 0."""
    n = 10
    Q = np.zeros((2 * n, 2 * n))
    A = sp.hstack([sp.identity(n), sp.identity(n)], format="csr")
    sol = pre(Q, -np.ones(2 * n), A, np.zeros(n), [("R", n)], optTol=OPT)
    assert np.linalg.norm(sol.y) < TOL


def check_infeasible(pre):
    """test/runtests.jl:410-438: y1 = 1 and y1 = -1."""
    H, Hc, A, b, _, _, _ = redundant_equalities()
    G = np.zeros((2, 10))
    G[:, 0] = 1.0
    sol = pre(H, Hc, A, b, [("R", 10)], G, np.array([1.0, -1.0]), optTol=OPT)
    assert sol.status == "Infeasible" and np.isnan(sol.y).all()


def test_oracle_imcols():
    check_imcols(o_imcols)


def test_oracle_preprocess_redundant():
    check_redundant(o_pre)


def test_oracle_preprocess_bad_dual():
    check_bad_dual(o_pre)


def test_oracle_preprocess_infeasible():
    check_infeasible(o_pre)


def test_product_imcols_properties():
    """The product's `imcols` against what `src/preprocessor.jl:10-30` promises, not against the oracle's copy of the
    same LAPACK call: the rows kept are independent, as many as the rank numpy computes on its own (SVD), every dropped
    row lies in their span, and the consistency verdict is that of a least-squares residual."""
    from cipkkt.preprocess import imcols
    check_imcols(imcols)
    rng = np.random.default_rng(5)
    for (mr, rk, nc) in [(7, 5, 12), (12, 3, 6), (9, 9, 9), (20, 4, 40), (6, 1, 3)]:
        A = rng.standard_normal((mr, rk)) @ rng.standard_normal((rk, nc))
        x0 = rng.standard_normal(nc)
        rows, ok = imcols(A, A @ x0)
        r = np.linalg.matrix_rank(A)
        assert ok and len(rows) == r == rk and rows == sorted(set(rows))
        sv = np.linalg.svd(A[rows], compute_uv=False)
        assert sv[-1] > 1e-8 * sv[0]                                   # kept rows independent
        others = [i for i in range(mr) if i not in rows]
        if others:                                                     # dropped rows are combinations of the kept ones
            coef = np.linalg.lstsq(A[rows].T, A[others].T, rcond=None)[0]
            assert np.abs(A[rows].T @ coef - A[others].T).max() < 1e-9 * np.abs(A).max()
        # inconsistent right-hand side: move b off the range of A (only possible when A has dependent rows)
        if r < mr:
            u = np.linalg.svd(A)[0][:, r]                              # a direction orthogonal to range(A)
            assert imcols(A, A @ x0 + u) == ([], False)
        # scaling of the data does not change the verdict (the reference normalises by ||A||, :14)
        assert imcols(1e6 * A, 1e6 * (A @ x0))[0] == rows
    # sparse input takes the same path
    As = sp.csr_matrix(np.vstack([np.eye(4), np.eye(4)[:2]]))
    rows, ok = imcols(As, np.array([1.0, 2, 3, 4, 1, 2]))
    assert ok and len(rows) == 4 and np.linalg.matrix_rank(As.toarray()[rows]) == 4


@pytest.mark.gpu
@pytest.mark.parametrize("check", [check_redundant, check_bad_dual, check_infeasible],
                         ids=["redundant", "bad_dual", "infeasible"])
def test_product_preprocess(check):
    import cipkkt
    check(cipkkt.preprocess_conicIP)


@pytest.mark.gpu
def test_product_preprocess_matches_oracle():
    import cipkkt
    H, Hc, A, b, G2, d2, _ = redundant_equalities(seed=3)
    got = cipkkt.preprocess_conicIP(H, Hc, A, b, [("R", 10)], G2, d2, optTol=OPT)
    ref = o_pre(H, Hc, A, b, [("R", 10)], G2, d2, optTol=OPT)
    assert got.status == ref.status == "Optimal" and got.Iter == ref.Iter
    np.testing.assert_allclose(got.y, ref.y, rtol=1e-6, atol=1e-8)


# ---- Miles's counterexamples (test/runtests.jl:592-651) through the pre-solve: reference data, reference verdicts
def _miles(k, kc=1.0, ka=1.0):
    c, A, b, con, var = P.miles_problem(k)
    return P.mpb_to_conicip(kc * c, ka * A, ka * b, con, var)


def check_miles(pre):
    assert pre(*_miles(1)).status == "Optimal"                       # :598-606
    assert pre(*_miles(2)).status == "Infeasible"                    # :608-616
    for kappa in (1e-8, 1e-6, 1e-4, 1.0, 1e4, 1e6, 1e8):             # :621-628  (c, A, b all scaled)
        assert pre(*_miles(3, kappa, kappa)).status == "Optimal", kappa
    for kappa in (1e-4, 1.0, 1e4, 1e6):                              # :630-637  (A, b scaled)
        assert pre(*_miles(3, 1.0, kappa)).status == "Optimal", kappa


def test_oracle_miles_counterexamples():
    check_miles(o_pre)


@pytest.mark.gpu
def test_product_miles_counterexamples():
    import cipkkt
    check_miles(cipkkt.preprocess_conicIP)


def test_full_row_rank_certificate_agrees_with_the_pivoted_qr():
    """round 5: the n x n Gram-matrix certificate that lets the pre-solve skip the (dense) pivoted QR of [Q A' G'] -- it may only
    say "full row rank" when the reference's criterion (every |R_ii| of the pivoted QR of the Frobenius-normalised matrix above
    1e-8, src/preprocessor.jl:19-23) keeps every row; "not certified" is always allowed."""
    from cipkkt import preprocess as pp
    rng = np.random.default_rng(3)
    n = 40
    Q = rng.standard_normal((n, n)); Q = Q @ Q.T / n
    A = sp.identity(n, format="csr")
    assert pp._full_row_rank([Q, A.T.tocsr(), np.zeros((n, 0))], 1e-8)                 # box QP: [Q I] has full row rank
    rows, ok = pp.imcols(np.hstack([Q, A.toarray().T]), rng.standard_normal(n))
    assert ok and rows == list(range(n))
    # rank deficient: Q = 0 and A' = [I; I] leave n of 2n rows independent -> never certified
    Z = np.zeros((2 * n, 2 * n))
    A2 = sp.hstack([sp.identity(n), sp.identity(n)], format="csr")
    assert not pp._full_row_rank([Z, A2.T.tocsr()], 1e-8)
    rows, ok = pp.imcols(np.hstack([Z, A2.T.toarray()]), np.ones(2 * n))
    assert len(rows) == n
    # nearly dependent rows (sigma_min / ||M||_F ~ 1e-7 < the certificate's margin 1e-6, above the reference's 1e-8): not certified
    G = rng.standard_normal((3, n))
    G[2] = G[0] + 1e-6 * rng.standard_normal(n)
    assert not pp._full_row_rank([G], 1e-8)
    assert pp._full_row_rank([rng.standard_normal((3, n))], 1e-8)
    assert not pp._full_row_rank([np.zeros((3, n))], 1e-8)


def test_presolve_refuses_a_dense_qr_beyond_the_size_limit(monkeypatch):
    """a rank-deficient program whose [Q A' G'] is larger than DENSE_QR_LIMIT raises a ValueError that names the shape instead
    of allocating an n x (n + m + p) dense matrix (limit lowered here so that the case is small)"""
    from cipkkt import preprocess as pp
    monkeypatch.setattr(pp, "DENSE_QR_LIMIT", 500)
    n = 10
    Q = np.zeros((2 * n, 2 * n))
    A = sp.hstack([sp.identity(n), sp.identity(n)], format="csr")
    with pytest.raises(ValueError, match=r"20 x 30"):
        pp.preprocess_conicIP(Q, -np.ones(2 * n), A, np.zeros(n), [("R", n)])
