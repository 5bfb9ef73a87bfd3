#!/usr/bin/env python3
"""Generates tests/golden/primitives_sdc.json: S-cone vectors checked at 50 digits with mpmath.

The reference pins nothing numerically for the S cone beyond `Iter == 6` and 1e-3 Dict values of its PSD-projection test
(test/runtests.jl:542-547) and the VecCongurance identities (:68-83), so these are invariants of the closed forms it
implements, evaluated independently of numpy/LAPACK:
  nestod_sdc (src/ConicIP.jl:196-210):  R'ZR = R^-1 S R^-T = Lambda, Lambda_i = sqrt(eig_i(Lz' S Lz))
  maxstep_sdc (:272-303):               1 / lambda_max(L^-1 D L^-T), X = L L'  (Inf when lambda_max < 0)
  dsdc! (:347-353):                     Y O + O Y = X
Run from the repo root:  python tests/golden/make_golden_sdc.py
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


def M(a):
    return mp.matrix([[mp.mpf(float(x)) for x in row] for row in a])


def mp_lambda(Z, S):
    L = mp.cholesky(M(Z))
    E, _ = mp.eigsy(L.T * M(S) * L)
    return sorted(float(mp.sqrt(e)) for e in E)


def mp_maxstep(X, D):
    L = mp.cholesky(M(X))
    Li = L ** -1
    E, _ = mp.eigsy(Li * M(D) * Li.T)
    mx = max(E)
    return None if mx < 0 else float(1 / mx)


def main():
    rng = np.random.default_rng(20261002)
    out = {"nestod_sdc": [], "maxstep_sdc": [], "dsdc": []}
    for r in (2, 3, 6, 10, 17):
        A = rng.standard_normal((r, r)); Z = A @ A.T / r + 0.3 * np.eye(r)
        A = rng.standard_normal((r, r)); S = A @ A.T / r + 0.3 * np.eye(r)
        lam = mp_lambda(Z, S)
        R = cones.nestod_sdc(cones.vecm(Z), cones.vecm(S))
        assert np.allclose(np.sort(np.diag(R.T @ Z @ R)), lam, rtol=1e-11)
        out["nestod_sdc"].append(dict(z=cones.vecm(Z).tolist(), s=cones.vecm(S).tolist(), lam=lam))
        A = rng.standard_normal((r, r)); D = 0.5 * (A + A.T)
        if r == 6:
            D = -(A @ A.T)                                  # negative definite direction: no bound -> Inf
        a = mp_maxstep(Z, D)
        o = cones.maxstep_sdc(cones.vecm(Z), cones.vecm(D))
        assert (a is None and np.isinf(o)) or abs(o - a) < 1e-10 * abs(a)
        out["maxstep_sdc"].append(dict(x=cones.vecm(Z).tolist(), d=cones.vecm(D).tolist(), alpha=a))
        X = 0.5 * (A + A.T)
        O = cones.mat(cones.dsdc(cones.vecm(X), cones.vecm(S)))
        res = M(S) * M(O) + M(O) * M(S) - M(X)
        assert max(abs(res[i, j]) for i in range(r) for j in range(r)) < mp.mpf("1e-12")
        out["dsdc"].append(dict(x=cones.vecm(X).tolist(), y=cones.vecm(S).tolist(), out=cones.vecm(O).tolist()))
    json.dump(out, open(os.path.join(HERE, "primitives_sdc.json"), "w"), indent=1)
    print("wrote primitives_sdc.json")


if __name__ == "__main__":
    main()
