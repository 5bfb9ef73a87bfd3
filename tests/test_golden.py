"""Golden vectors (tests/golden/*.json, generator committed beside them):
  * reference_pins.json   -- literal values the reference's own test-suite pins (data)
  * primitives.json       -- per-primitive vectors, oracle output cross-checked at 50 digits
  * oracle_trajectories.json -- per-iteration traces of the deterministic KATs
CPU tests pin the oracle; the gpu-marked tests pin the HIP kernels / driver to the same files."""
import json
import os

import numpy as np
import pytest

import problems as P
from oracle import cones
from oracle.conicip import conicIP as oracle_conicIP

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PRIM = json.load(open(os.path.join(G, "primitives.json")))
PINS = json.load(open(os.path.join(G, "reference_pins.json")))
TRAJ = json.load(open(os.path.join(G, "oracle_trajectories.json")))
SDC = json.load(open(os.path.join(G, "primitives_sdc.json")))      # S-cone invariants, mpmath at 50 digits


def test_oracle_primitives_match_golden():
    for r in PRIM["nestod_soc"]:
        beta, w = cones.nestod_soc(np.array(r["z"]), np.array(r["s"]))
        assert beta == pytest.approx(r["beta"], rel=1e-13)
        np.testing.assert_allclose(w, r["w"], rtol=1e-12, atol=1e-14)
    for r in PRIM["maxstep_soc"]:
        a = cones.maxstep_soc(np.array(r["x"]), np.array(r["d"]))
        assert (r["alpha"] is None and np.isinf(a)) or a == pytest.approx(r["alpha"], rel=1e-11)
    for r in PRIM["dsoc"]:
        np.testing.assert_allclose(cones.dsoc(np.array(r["num"]), np.array(r["den"])), r["out"], rtol=1e-12)
    for r in PRIM["vecm_mat"]:
        np.testing.assert_allclose(cones.vecm(np.array(r["X"])), r["v"], rtol=1e-15)
        np.testing.assert_allclose(cones.mat(np.array(r["v"])), r["X"], rtol=1e-15)
    for r in PRIM["nestod_sdc"]:
        R = cones.nestod_sdc(np.array(r["z"]), np.array(r["s"]))
        lam = np.sort(np.diag(R.T @ cones.mat(np.array(r["z"])) @ R))
        np.testing.assert_allclose(lam, r["lam"], rtol=1e-10)


def test_oracle_sdc_primitives_match_mpmath():
    for r in SDC["nestod_sdc"]:
        Z = cones.mat(np.array(r["z"]))
        R = cones.nestod_sdc(np.array(r["z"]), np.array(r["s"]))
        np.testing.assert_allclose(np.sort(np.diag(R.T @ Z @ R)), r["lam"], rtol=1e-11)
        Ri = np.linalg.inv(R)
        np.testing.assert_allclose(np.sort(np.diag(Ri @ cones.mat(np.array(r["s"])) @ Ri.T)), r["lam"], rtol=1e-9)
    for r in SDC["maxstep_sdc"]:
        a = cones.maxstep_sdc(np.array(r["x"]), np.array(r["d"]))
        assert (r["alpha"] is None and np.isinf(a)) or a == pytest.approx(r["alpha"], rel=1e-10)
    for r in SDC["dsdc"]:
        np.testing.assert_allclose(cones.dsdc(np.array(r["x"]), np.array(r["y"])), r["out"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("name", ["sphere", "combined", "simplex", "psd_projection"])
def test_oracle_against_reference_pins(name):
    """the reference's `compare` (test/runtests.jl:15-21) on its own pinned Dicts"""
    pin = PINS[name]
    Q, c, A, b, K, Gm, d, _ = getattr(P, name)()
    sol = oracle_conicIP(Q, c, A, b, K, Gm, d, optTol=1e-7, DTB=0.01, maxRefinementSteps=3)
    tol = PINS["tol"]
    assert sol.status == pin["status"]
    for key in ("prFeas", "Mu", "muFeas", "duFeas"):
        assert abs(getattr(sol, key) - pin[key]) < tol


def test_oracle_trajectories_reproducible():
    for name, t in TRAJ.items():
        Q, c, A, b, K, Gm, d, _ = getattr(P, name)()
        sol = oracle_conicIP(Q, c, A, b, K, Gm, d, optTol=1e-7, DTB=0.01, maxRefinementSteps=3)
        assert sol.status == t["status"] and sol.Iter == t["Iter"]
        np.testing.assert_allclose(sol.y, t["y"], rtol=1e-7, atol=1e-9)


# ------------------------------------------------------------------ GPU side
@pytest.mark.gpu
def test_hip_cone_kernels_match_golden():
    import torch
    import cipkkt
    dv = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float64, device="cuda")
    for r in PRIM["nestod_soc"]:
        k = len(r["z"])
        ks = cipkkt.KKTSystem(np.eye(2), np.zeros((k, 2)), None, [("Q", k)])
        lam = torch.zeros(k, dtype=torch.float64, device="cuda")
        ks.set_scaling_from_iterate(dv(r["z"]), dv(r["s"]), lam)
        packed = ks.get_scaling_packed()
        assert packed[0] == pytest.approx(r["beta"], rel=1e-13)
        np.testing.assert_allclose(packed[1:], r["w"], rtol=1e-12, atol=1e-14)
        ks.close()
    for r in PRIM["maxstep_soc"]:
        k = len(r["x"])
        ks = cipkkt.KKTSystem(np.eye(2), np.zeros((k, 2)), None, [("Q", k)])
        a = ks.maxstep(dv(r["x"]), dv(r["d"]))
        assert (r["alpha"] is None and np.isinf(a)) or a == pytest.approx(r["alpha"], rel=1e-10)
        ks.close()
    for r, rx in zip(PRIM["dsoc"], PRIM["xsoc"]):
        k = len(r["num"])
        ks = cipkkt.KKTSystem(np.eye(2), np.zeros((k, 2)), None, [("Q", k)])
        out = torch.zeros(k, dtype=torch.float64, device="cuda")
        ks.cone_div(dv(r["num"]), dv(r["den"]), out)
        np.testing.assert_allclose(out.cpu().numpy(), r["out"], rtol=1e-11, atol=1e-13)
        ks.cone_prod(dv(rx["x"]), dv(rx["y"]), out)
        np.testing.assert_allclose(out.cpu().numpy(), rx["out"], rtol=1e-13, atol=1e-14)
        ks.close()


@pytest.mark.gpu
def test_hip_sdc_kernels_match_mpmath():
    """S-cone kernels against the 50-digit invariants (the reference holds no numeric pin at this granularity)."""
    import torch
    import cipkkt
    dv = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float64, device="cuda")
    for r, rm, rd in zip(SDC["nestod_sdc"], SDC["maxstep_sdc"], SDC["dsdc"]):
        k = len(r["z"])
        ks = cipkkt.KKTSystem(np.eye(2), np.zeros((k, 2)), None, [("S", k)])
        lam = torch.zeros(k, dtype=torch.float64, device="cuda")
        ks.set_scaling_from_iterate(dv(r["z"]), dv(r["s"]), lam)
        got = np.sort(np.linalg.eigvalsh(cones.mat(lam.cpu().numpy())))      # lambda = vecm(diag(Lambda)) up to ordering
        np.testing.assert_allclose(got, r["lam"], rtol=1e-11)
        # F^-T s = lambda as well (R^-1 S R^-T = Lambda)
        t = torch.zeros(k, dtype=torch.float64, device="cuda")
        ks.apply_F(cipkkt.OP_FINVT, dv(r["s"]), t)
        np.testing.assert_allclose(np.sort(np.linalg.eigvalsh(cones.mat(t.cpu().numpy()))), r["lam"], rtol=1e-9)
        a = ks.maxstep(dv(rm["x"]), dv(rm["d"]))
        assert (rm["alpha"] is None and np.isinf(a)) or a == pytest.approx(rm["alpha"], rel=1e-10)
        ks.cone_div(dv(rd["x"]), dv(rd["y"]), t)
        np.testing.assert_allclose(t.cpu().numpy(), rd["out"], rtol=1e-8, atol=1e-10)
        ks.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sphere", "combined", "simplex", "soc_direct", "lp_doc", "psd_projection"])
def test_hip_driver_matches_golden_trajectory(name):
    import cipkkt
    t = TRAJ[name]
    Q, c, A, b, K, Gm, d, _ = getattr(P, name)()
    sol = cipkkt.conicIP(Q, c, A, b, K, Gm, d, optTol=1e-7, DTB=0.01, maxRefinementSteps=3)
    assert sol.status == t["status"] and sol.Iter == t["Iter"] and sol.n_factor == t["n_factor"]
    np.testing.assert_allclose(sol.y, t["y"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(sol.v, t["v"], rtol=1e-5, atol=1e-8)
    if name in PINS:
        pin = PINS[name]
        for key in ("prFeas", "Mu", "muFeas", "duFeas"):
            assert abs(getattr(sol, key) - pin[key]) < PINS["tol"]
