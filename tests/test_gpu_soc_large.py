"""large second-order cones at LOOP level (round-3 review: "one assembly-level case, no end-to-end run").

The reference treats a large SOC block specially -- `lift` (src/kktsolvers.jl:60-105) replaces the dense k x k block
beta^2 (2 wbar wbar' - J) of F'F by a diagonal-plus-two-columns sparse form inside kktsolver_sparse (:192-240), and
`pivot(kktsolver_2x2)` forms A'F^-1F^-T A (:289-293) -- and benchmarks exactly two SOC shapes (benchmark/profile.jl:43-69,
benchmark/report.md:57-62: "Single SOC (500)" 6 iterations, "Many SOCs (250 x 3)" 9 iterations with every solver).  On the
device a Q cone never becomes a dense block on the Schur route: F^-1 is applied to the rows of A' in O(mn) (dense A), or
A'F^-2A = A' diag A + one rank-1 column per cone (CSR A: `k_schur_rows` + `k_schur_qcols` + a rank-nq update).

Here: both benchmark shapes and one ("Q", 4097) cone behind a CSR and behind a dense A, both elimination routes, against the
ORACLE'S TRAJECTORY (live at n = 500; committed fixtures at n = 4096, tests/golden/make_fullsize_fixtures.py soc) plus the
analytic minimiser of the single-SOC problem (projection of c onto the unit ball)."""
import json
import os

import numpy as np
import pytest

from cipkkt import workloads as W
from oracle.conicip import conicIP as oracle_conicIP
from oracle import kktsolvers as ok
from test_gpu_configs import check_optimality
from test_gpu_configs_full import assert_same_trajectory, FULLSIZE

pytestmark = pytest.mark.gpu


def _dense(M):
    return M.toarray() if hasattr(M, "toarray") else np.asarray(M)


def _same_as_live_oracle(sol, ref, what):
    assert sol.status == ref.status == "Optimal", (what, sol.status, ref.status)
    assert (sol.Iter, sol.n_factor) == (ref.Iter, ref.n_factor), (what, sol.Iter, ref.Iter)
    np.testing.assert_allclose(sol.y, ref.y, rtol=1e-6, atol=1e-8, err_msg=what)
    np.testing.assert_allclose(sol.v, ref.v, rtol=1e-6, atol=1e-8, err_msg=what)
    for it, (tg, tr) in enumerate(zip(sol.trace, ref.trace)):
        assert abs(tg["mu"] - tr["mu"]) <= 1e-6 * abs(tr["mu"]), (what, it)
        if tr.get("alpha") is not None:
            assert abs(tg["alpha"] - tr["alpha"]) <= 1e-6, (what, it)


@pytest.mark.parametrize("route", ["schur", "full3x3"])
@pytest.mark.parametrize("dense", [False, True], ids=["csrA", "denseA"])
def test_reference_single_soc_benchmark(route, dense):
    """benchmark/profile.jl:43-52 at its own size (n = 500, one ("Q", 501)): the oracle's trajectory (6 iterations -- the
    count benchmark/report.md:57-59 publishes for all three reference solvers), the analytic answer c / |c|."""
    import cipkkt
    prob = W.soc_single(500, seed=42, dense=dense)
    ref = oracle_conicIP(*prob, optTol=1e-6, kktsolver=ok.pivot(ok.kktsolver_2x2))
    assert ref.Iter == 6
    for driver in ("native", "python"):
        sol = cipkkt.conicIP(*prob, optTol=1e-6, kktsolver=route, driver=driver)
        _same_as_live_oracle(sol, ref, "single SOC n=500 %s %s" % (route, driver))
    c = prob[1]
    assert np.linalg.norm(sol.y - c / np.linalg.norm(c)) < 1e-6


@pytest.mark.parametrize("route", ["schur", "full3x3"])
def test_reference_many_small_socs_benchmark(route):
    """benchmark/profile.jl:54-69: 250 x ("Q", 3), A = sprandn(750, 500, 0.1) (CSR on the device: rank-1 columns per cone),
    9 iterations (benchmark/report.md:60-62)."""
    import cipkkt
    prob = W.soc_many_small()
    ref = oracle_conicIP(*prob, optTol=1e-6, kktsolver=ok.pivot(ok.kktsolver_2x2))
    assert ref.Iter == 9
    sol = cipkkt.conicIP(*prob, optTol=1e-6, kktsolver=route)
    _same_as_live_oracle(sol, ref, "250 x Q(3) %s" % route)
    Q, c, A, b, K = prob
    check_optimality(_dense(Q), c, _dense(A), b, K, np.zeros((0, 500)), np.zeros(0), sol, 1e-5)


@pytest.mark.parametrize("route", ["schur", "full3x3"])
@pytest.mark.parametrize("which", ["soc_single_n4096_seed42", "soc_dense_n4096_seed21"])
def test_large_soc_q4097(which, route):
    """One ("Q", 4097) cone, n = 4096, behind A = [0; I] in CSR and behind a dense A, Schur route (N = 4096) and literal 3x3
    route (N = 8193): the oracle's trajectory (fixture: pivot(kktsolver_2x2) on the same SplitMix64 inputs) and the
    optimality conditions of the returned point."""
    import cipkkt
    prob = W.soc_single(4096, seed=42) if which.startswith("soc_single") else W.soc_large_dense(4096, seed=21)
    sol = cipkkt.conicIP(*prob, optTol=1e-6, kktsolver=route)
    assert_same_trajectory(sol, FULLSIZE[which], "%s %s" % (which, route))
    Q, c, A, b, K = prob
    check_optimality(_dense(Q), c, _dense(A), b, K, np.zeros((0, 4096)), np.zeros(0), sol, 1e-5)
    if which.startswith("soc_single"):
        assert np.linalg.norm(sol.y - c / np.linalg.norm(c)) < 1e-6
