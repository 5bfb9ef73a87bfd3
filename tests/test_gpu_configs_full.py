"""Every BASELINE.json configuration at its stated size on the GPU ("configs untested"), inputs from the
portable generator (cipkkt/workloads.py).  At full size the checks are (1) IDENTITY OF THE TRAJECTORY WITH THE ORACLE --
status, iteration count, factorisations, solves, per-iteration mu / alpha / residuals, sampled entries of the iterates:
the north-star's "identical iteration count to convergence" -- against tests/golden/fullsize_trajectories.json, which
tests/golden/make_fullsize_fixtures.py writes by running the oracle on the same SplitMix64 inputs in the build container
(config 2 at n = 8192, config 3 at full size, all 64 problems of config 5), and (2) size-independent
properties: the optimality conditions of the returned point (src/ConicIP.jl:763-788) and a KKT backward error."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from cipkkt import workloads as W
from oracle.conicip import conicIP as oracle_conicIP
from oracle import kktsolvers as ok
from test_gpu_configs import check_optimality

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_trajectories.json")) as _f:
    FULLSIZE = json.load(_f)


def assert_same_trajectory(sol, fx, what):
    """`sol` (product) walked the oracle's trajectory `fx` (fixture record): equal status / Iter / n_factor / n_solve; mu
    to 1e-6 relative and alpha to 1e-6 absolute in every iteration; the residual triples to 1e-6 of their scale (they are
    norms of differences of O(1) vectors: absolute agreement 1e-9); the final iterate by its norms and 64 sampled entries."""
    assert sol.status == fx["status"], (what, sol.status, fx["status"])
    assert (sol.Iter, sol.n_factor, sol.n_solve) == (fx["Iter"], fx["n_factor"], fx["n_solve"]), \
        (what, (sol.Iter, sol.n_factor, sol.n_solve), (fx["Iter"], fx["n_factor"], fx["n_solve"]))
    if sol.trace:
        assert len(sol.trace) == len(fx["trace"]), (what, len(sol.trace), len(fx["trace"]))
        for it, (tg, tr) in enumerate(zip(sol.trace, fx["trace"])):
            assert abs(tg["mu"] - tr["mu"]) <= 1e-6 * abs(tr["mu"]), (what, it, tg["mu"], tr["mu"])
            if tr.get("alpha") is not None:
                assert abs(tg["alpha"] - tr["alpha"]) <= 1e-6, (what, it, tg["alpha"], tr["alpha"])
                assert abs(tg["sigma"] - tr["sigma"]) <= 1e-6, (what, it, tg["sigma"], tr["sigma"])
            for k in ("rDu", "rPr", "rCp"):
                assert abs(tg[k] - tr[k]) <= 1e-6 * abs(tr[k]) + 1e-9, (what, it, k, tg[k], tr[k])
    ny, nv = np.linalg.norm(sol.y), np.linalg.norm(sol.v)
    assert abs(ny - fx["norm_y"]) <= 1e-6 * fx["norm_y"] and abs(nv - fx["norm_v"]) <= 1e-6 * max(fx["norm_v"], 1e-300), what
    np.testing.assert_allclose(sol.y[fx["idx_y"]], fx["y"], rtol=1e-6, atol=1e-8 * fx["norm_y"], err_msg=what)
    np.testing.assert_allclose(sol.v[fx["idx_v"]], fx["v"], rtol=1e-6, atol=1e-8 * max(fx["norm_v"], 1.0), err_msg=what)


def test_c1_readme_boxqp_n1000_vs_oracle_qr():
    """Config 1 as written in README.md:56-65 (Q = B'B, B 10 %-dense, c = 1, A = I): reference-faithful kktsolver_qr
    restatement vs the HIP path, both routes."""
    import cipkkt
    Q, c, A, b, K = W.c1_readme_boxqp(1000, seed=42)
    ref = oracle_conicIP(Q, c, A, b, K, optTol=1e-6)
    assert ref.status == "Optimal"
    for route in ("schur", "full3x3"):
        got = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6, kktsolver=route)
        assert got.status == "Optimal" and got.Iter == ref.Iter and got.n_factor == ref.n_factor
        np.testing.assert_allclose(got.y, ref.y, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got.v, ref.v, rtol=1e-6, atol=1e-7)


def test_c2_family_n2048_identical_trajectory_to_oracle():
    """Config 2's family at n = 2048 against the oracle's pivot(kktsolver_2x2) (src/kktsolvers.jl:281-349): same number
    of iterations, factorisations and solves, same iterates, same per-iteration mu."""
    import cipkkt
    Q, c, A, b, K = W.c2_problem(2048, seed=1234)
    ref = oracle_conicIP(Q, c, A, b, K, optTol=1e-6, kktsolver=ok.pivot(ok.kktsolver_2x2))
    got = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6)
    assert got.status == ref.status == "Optimal"
    assert (got.Iter, got.n_factor, got.n_solve) == (ref.Iter, ref.n_factor, ref.n_solve)
    np.testing.assert_allclose(got.y, ref.y, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(got.v, ref.v, rtol=1e-6, atol=1e-8)
    for tg, tr in zip(got.trace, ref.trace):
        assert abs(tg["mu"] - tr["mu"]) <= 1e-6 * abs(tr["mu"])
        assert abs(tg.get("alpha", 0) - tr.get("alpha", 0)) <= 1e-6


def test_c2_headline_n8192():
    """Config 2 at full size, inputs generated in HBM: the ORACLE'S TRAJECTORY (fixture: pivot(kktsolver_2x2) on the same
    inputs, 9 iterations) with both loops, the optimality conditions, and the backward error of one
    3x3 solve at a late-iteration scaling (||K x - rhs|| / (||K|| ||x|| + ||rhs||) < 1e-12, evaluated with the problem
    operators on the host)."""
    import torch
    import cipkkt
    n = 8192
    Q, c, A, b, K = W.c2_problem(n, seed=1234, device="cuda")
    ks = cipkkt.KKTSystem(Q, A, None, K)
    its = []
    sol = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6, system=ks, keep_iterates=its, driver="python")
    fx = FULLSIZE["c2_n8192_seed1234"]
    assert_same_trajectory(sol, fx, "c2 n=8192, per-operation loop")
    nat = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6, system=ks)
    assert_same_trajectory(nat, fx, "c2 n=8192, native loop")
    assert np.array_equal(nat.y, sol.y)
    Qh = Q.cpu().numpy()
    check_optimality(Qh, c, A, b, K, np.zeros((0, n)), np.zeros(0), sol, 1e-5)
    # a late-iteration NT scaling (two before the last), one solve3x3 through the host-pointer ABI
    z = its[-2]
    v, s = z[n:2 * n].clone(), z[2 * n:3 * n].clone()
    ks.set_scaling_from_iterate(v, s)
    ks.factor()
    rng = np.random.default_rng(0)
    x, zz = rng.standard_normal(n), rng.standard_normal(n)
    a, _, cc = ks.solve3x3(x, np.zeros(0), zz)
    d2 = (s / v).cpu().numpy()                                   # F'F for an R cone
    r1 = Qh @ a - cc - x                                         # Q a - A'c = x   (A = I)
    r2 = a + d2 * cc - zz                                        # A a + F'F c = z
    normK = max(np.abs(Qh).sum(axis=1).max() + 1, d2.max() + 1)
    berr = np.sqrt(r1 @ r1 + r2 @ r2) / (normK * np.sqrt(a @ a + cc @ cc) + np.sqrt(x @ x + zz @ zz))
    assert berr < 1e-12, berr
    ks.close()


def test_c2_headline_full3x3_n8192():
    """Config 2 at full size through the LITERAL 3x3 assembly (src/kktsolvers.jl:254-257: [Q G' -A'; G 0 0; A 0 F'F], here
    symmetrised and in the pivot order (3,1,2), order N = n + m = 16384): it is the same linear system as the Schur
    route's, so the oracle's trajectory fixture of `test_c2_headline_n8192` applies unchanged -- status / Iter / n_factor /
    n_solve equal, mu and alpha at 1e-6 per iteration, the iterate at 1e-6 -- and one solve3x3 at a late scaling has a backward
    error < 1e-12 against the problem's own operators (round-3 review, NS-1)."""
    import cipkkt
    n = 8192
    Q, c, A, b, K = W.c2_problem(n, seed=1234, device="cuda")
    ks = cipkkt.KKTSystem(Q, A, None, K, route="full3x3")
    assert ks.N == 2 * n
    its = []
    sol = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6, system=ks, keep_iterates=its, driver="python")
    fx = FULLSIZE["c2_n8192_seed1234"]
    assert_same_trajectory(sol, fx, "c2 n=8192, literal 3x3 route (N = 16384), per-operation loop")
    nat = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6, system=ks)
    assert_same_trajectory(nat, fx, "c2 n=8192, literal 3x3 route (N = 16384), native loop")
    assert np.array_equal(nat.y, sol.y)
    Qh = Q.cpu().numpy()
    check_optimality(Qh, c, A, b, K, np.zeros((0, n)), np.zeros(0), sol, 1e-5)
    z = its[-2]
    v, s = z[n:2 * n].clone(), z[2 * n:3 * n].clone()
    ks.set_scaling_from_iterate(v, s)
    ks.factor()
    rng = np.random.default_rng(0)
    x, zz = rng.standard_normal(n), rng.standard_normal(n)
    a, _, cc = ks.solve3x3(x, np.zeros(0), zz)
    d2 = (s / v).cpu().numpy()
    r1 = Qh @ a - cc - x
    r2 = a + d2 * cc - zz
    normK = max(np.abs(Qh).sum(axis=1).max() + 1, d2.max() + 1)
    berr = np.sqrt(r1 @ r1 + r2 @ r2) / (normK * np.sqrt(a @ a + cc @ cc) + np.sqrt(x @ x + zz @ zz))
    assert berr < 1e-12, berr
    ks.close()


def test_c3_socp_full_size_portable_inputs():
    import cipkkt
    prob = W.c3_socp()
    sol = cipkkt.conicIP(*prob, optTol=1e-6)
    assert sol.status == "Optimal"
    check_optimality(*prob, sol, 1e-5)
    # the oracle's run with the reference's DEFAULT solver (kktsolver_qr restatement) on the same inputs
    assert_same_trajectory(sol, FULLSIZE["c3_socp_seed11"], "c3 full size")


@pytest.mark.parametrize("r,n,p", [(64, 96, 8), (140, 64, 4)])
def test_c4_sdp_vs_oracle(r, n, p):
    """Config 4's family against the oracle at r = 64 (LDS-resident S-cone kernels) and r = 140 (the path for matrix
    orders 133-512).  The oracle runs the exact block elimination (oracle.kktsolvers.kktsolver_schur_exact): the dense
    k x k F of kktsolver_qr would be 780 MB at r = 140."""
    import cipkkt
    prob = W.c4_sdp(r=r, n=n, p=p, seed=5)
    ref = oracle_conicIP(*prob, optTol=1e-6, kktsolver=ok.kktsolver_schur_exact)
    got = cipkkt.conicIP(*prob, optTol=1e-6)
    assert got.status == ref.status == "Optimal" and got.Iter == ref.Iter
    np.testing.assert_allclose(got.y, ref.y, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(got.v, ref.v, rtol=1e-5, atol=1e-6)


def test_c4_sdp_r256_full_size():
    """The literal reading of config 4: matrix order 256, ("S", 32896), n = 1024, p = 16: the oracle's trajectory (fixture:
    the exact block elimination of the 3x3 system, oracle.kktsolvers.kktsolver_schur_exact -- kktsolver_qr's dense F would be
    32896 x 32896) with both loops, and the optimality conditions of the returned point."""
    import cipkkt
    prob = W.c4_sdp(r=256, n=1024, p=16, seed=5)
    sol = cipkkt.conicIP(*prob, optTol=1e-6)
    assert sol.status == "Optimal"
    check_optimality(*prob, sol, 2e-5)
    fx = FULLSIZE["c4_sdp_r256_seed5"]
    # n_solve: the oracle's plugin counts its solves the same way; the iterates agree to 1e-6 (the S-cone scaling is unique up
    # to an orthogonal factor only, its invariants are what the loop sees)
    assert_same_trajectory(sol, fx, "c4 r=256, native loop")
    py = cipkkt.conicIP(*prob, optTol=1e-6, driver="python")
    assert_same_trajectory(py, fx, "c4 r=256, per-operation loop")


def test_c5_all_64_problems():
    """Config 5 in full: 64 independent dense QPs, n = 2048, seeds 4000 + i, generated in HBM, through the batch entry
    point (lock-step): every problem walks the oracle's trajectory (fixture: Iter / n_factor / n_solve equal per problem,
    final iterates at 1e-6)."""
    from cipkkt.batch import solve_batch
    probs = W.c5_batch(64, 2048, seed=4000, device="cuda")
    sols, st = solve_batch(probs, concurrency=4, native=True)
    assert st["n_problems"] == 64 and st["n_optimal"] == 64
    fx = FULLSIZE["c5_n2048_seed4000"]["problems"]
    for i in range(64):
        assert_same_trajectory(sols[i], fx[str(i)], "c5 problem %d" % i)
    assert st["iters"] == sum(fx[str(i)]["Iter"] for i in range(64))
    for i in (0, 17, 63):                                       # spot-check optimality conditions
        pr = probs[i]
        check_optimality(pr["Q"].cpu().numpy(), pr["c"], pr["A"], pr["b"], pr["cone_dims"], np.zeros((0, 2048)),
                         np.zeros(0), sols[i], 1e-5)
