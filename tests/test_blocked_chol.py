"""oracle/blocked_chol.py (bench.py's second strong-CPU contender): the in-place blocked Cholesky on the host BLAS agrees
with LAPACK's potrf and solves to rounding, for both forms of the trailing update and a ragged last block."""
import numpy as np
import pytest
import scipy.linalg as sla

from oracle.blocked_chol import blocked_cholesky, cholesky_solve


@pytest.mark.parametrize("n, nb", [(300, 128), (512, 128), (77, 512)])
@pytest.mark.parametrize("trailing", ["syrk", "gemm"])
def test_blocked_cholesky_matches_lapack(n, nb, trailing):
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    S0 = np.asfortranarray(M @ M.T / n + np.eye(n))
    S = S0.copy(order="F")
    assert blocked_cholesky(S, nb, trailing) == 0
    L = np.tril(S)
    Lref = sla.cholesky(S0, lower=True)
    assert np.allclose(L, Lref, rtol=1e-11, atol=1e-12)
    if trailing == "syrk":                                                # (the dgemm form also updates the diagonal blocks' upper halves)
        assert np.array_equal(np.triu(S, 1), np.triu(S0, 1))             # the strict upper triangle is not touched
    b = rng.standard_normal(n)
    x = cholesky_solve(S, b)
    assert np.linalg.norm(S0 @ x - b) <= 1e-12 * np.linalg.norm(b) * np.linalg.cond(S0)


def test_blocked_cholesky_reports_an_indefinite_block():
    S = np.asfortranarray(np.eye(200))
    S[150, 150] = -1.0
    assert blocked_cholesky(S, 64, "syrk") == 151                          # LAPACK's 1-based index of the failing pivot
