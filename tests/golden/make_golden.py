#!/usr/bin/env python3
"""Generates tests/golden/*.json.

The reference (ConicIP.jl) is Julia and cannot be imported or run in the build image, so
these vectors do NOT come from the reference binary.  They are (a) the literal values the
reference's own test-suite pins (test/runtests.jl: the `Dict`s at :122-127, :157-162,
:197-202, :235-240, :542-547 and the analytic answers it asserts), copied as data, and
(b) per-primitive input/output vectors produced by the oracle restatement and
cross-checked here against 50-digit mpmath evaluation of the closed forms the reference
implements (src/ConicIP.jl:165-194 nestod_soc, :242-262 maxstep_soc, :317-338 dsoc!).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cones  # noqa: E402

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))


def mp_nestod_soc(z, s):
    z = [mp.mpf(float(x)) for x in z]
    s = [mp.mpf(float(x)) for x in s]
    qf = lambda r: r[0] * r[0] - sum(x * x for x in r[1:])
    beta = (qf(s) / qf(z)) ** mp.mpf("0.25")
    zn = [x / mp.sqrt(qf(z)) for x in z]
    sn = [x / mp.sqrt(qf(s)) for x in s]
    gamma = mp.sqrt((1 + sum(a * b for a, b in zip(zn, sn))) / 2)
    jz = [zn[0]] + [-x for x in zn[1:]]
    w = [(a + b) / (2 * gamma) for a, b in zip(sn, jz)]
    w[0] += 1
    sc = mp.sqrt(2 * beta) / mp.sqrt(2 * w[0])
    return float(beta), [float(x * sc) for x in w]


def mp_maxstep_soc(x, d):
    x = [mp.mpf(float(t)) for t in x]
    d = [-mp.mpf(float(t)) for t in d]
    Q = lambda a, b: a[0] * b[0] - sum(p * q for p, q in zip(a[1:], b[1:]))
    gam = Q(x, x)
    xb = [t / mp.sqrt(gam) for t in x]
    beta = Q(xb, d)
    rho1 = beta / mp.sqrt(gam)
    mu = (beta + d[0]) / (xb[0] + 1)
    rho2 = [a - mu * b for a, b in zip(d[1:], xb[1:])]
    alpha = mp.sqrt(sum(t * t for t in rho2)) / mp.sqrt(gam) - rho1
    return float("inf") if alpha < 0 else float(1 / alpha)


def main():
    rng = np.random.default_rng(20260220)
    prim = {"nestod_soc": [], "maxstep_soc": [], "dsoc": [], "xsoc": [], "vecm_mat": [], "nestod_sdc": []}
    for k in (3, 8, 8, 21):
        z = rng.standard_normal(k); z[0] = np.linalg.norm(z[1:]) + rng.random() + 0.05
        s = rng.standard_normal(k); s[0] = np.linalg.norm(s[1:]) + rng.random() + 0.05
        beta, w = cones.nestod_soc(z, s)
        mb, mw = mp_nestod_soc(z, s)
        assert abs(beta - mb) < 1e-13 * abs(mb) and np.allclose(w, mw, rtol=1e-12, atol=1e-14)
        prim["nestod_soc"].append(dict(z=z.tolist(), s=s.tolist(), beta=mb, w=mw))
        d = rng.standard_normal(k)
        a = cones.maxstep_soc(z, d)
        ma = mp_maxstep_soc(z, d)
        assert (np.isinf(a) and np.isinf(ma)) or abs(a - ma) < 1e-11 * abs(ma)
        prim["maxstep_soc"].append(dict(x=z.tolist(), d=d.tolist(), alpha=(None if np.isinf(ma) else ma)))
        x = rng.standard_normal(k)
        o = cones.dsoc(x, s)
        assert np.allclose(cones.xsoc(s, o), x, rtol=1e-11, atol=1e-12)
        prim["dsoc"].append(dict(num=x.tolist(), den=s.tolist(), out=o.tolist()))
        prim["xsoc"].append(dict(x=x.tolist(), y=s.tolist(), out=cones.xsoc(x, s).tolist()))
    # vecm / mat: the reference's own doc example (src/ConicIP.jl:96-99, :131-132; docs/src/tutorials/sdp.jl:53-70)
    X = np.array([[1.0, 2, 3], [2, 5, 6], [3, 6, 9]])
    prim["vecm_mat"].append(dict(X=X.tolist(), v=cones.vecm(X).tolist()))
    for r in (3, 6):
        M = rng.standard_normal((r, r)); Z = M @ M.T + np.eye(r)
        M = rng.standard_normal((r, r)); S = M @ M.T + np.eye(r)
        R = cones.nestod_sdc(cones.vecm(Z), cones.vecm(S))
        lam1 = np.diag(R.T @ Z @ R)
        prim["nestod_sdc"].append(dict(z=cones.vecm(Z).tolist(), s=cones.vecm(S).tolist(),
                                       lam=np.sort(lam1).tolist()))   # R is unique up to column signs: pin Lambda
    json.dump(prim, open(os.path.join(HERE, "primitives.json"), "w"), indent=1)

    # the reference's own pinned values (data copied from test/runtests.jl)
    ref = {
        "sphere": dict(lines="test/runtests.jl:137-166", status="Optimal", prFeas=0.0, Mu=2.866608128093695e-7,
                       muFeas=1.621702501927476e-7, duFeas=3.2367552452111847e-16, Iter=5,
                       y=[2 ** -0.5, 2 ** -0.5]),
        "combined": dict(lines="test/runtests.jl:168-206", status="Optimal", prFeas=7.764421906286858e-17,
                         Mu=4.663886012743681e-7, muFeas=1.7037397157416066e-7, duFeas=2.77947804665922e-17, Iter=10),
        "simplex": dict(lines="test/runtests.jl:208-244", status="Optimal", prFeas=1.4506364239112378e-16,
                        Mu=2.7686402945528533e-9, muFeas=2.897827518851058e-9, duFeas=2.70780035221441e-17, Iter=11),
        "psd_projection": dict(lines="test/runtests.jl:527-552", status="Optimal", prFeas=4.2341217602756234e-16,
                               Mu=3.4583513329836624e-10, muFeas=1.48267911727847e-9,
                               duFeas=4.2341217602756234e-16, Iter=6),
        "box_qp": dict(lines="test/runtests.jl:90-131", status="Optimal", prFeas=0, Mu=0, muFeas=0, duFeas=0, Iter=7),
        "tol": 1e-3,
    }
    json.dump(ref, open(os.path.join(HERE, "reference_pins.json"), "w"), indent=1)

    # oracle trajectories of the deterministic KATs (per-iteration mu, residuals, step) + final iterate
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import problems as P
    from oracle.conicip import conicIP
    traj = {}
    for name in ("sphere", "combined", "simplex", "soc_direct", "lp_doc", "psd_projection"):
        Q, c, A, b, K, G, d, _ = getattr(P, name)()
        sol = conicIP(Q, c, A, b, K, G, d, optTol=1e-7, DTB=0.01, maxRefinementSteps=3)
        traj[name] = dict(status=sol.status, Iter=sol.Iter, n_factor=sol.n_factor, n_solve=sol.n_solve,
                          y=sol.y.tolist(), w=sol.w.tolist(), v=sol.v.tolist(),
                          trace=[{k: float(v) for k, v in t.items()} for t in sol.trace])
    json.dump(traj, open(os.path.join(HERE, "oracle_trajectories.json"), "w"), indent=1)
    print("wrote primitives.json, reference_pins.json, oracle_trajectories.json")


if __name__ == "__main__":
    main()
