"""Repeat tests/test_gpu_driver.py::test_dense_qp_2048_properties' two solves and print a hash of every solution: are the two
routes reproducible run to run, and how far apart are their end points?   usage: python tools/qp2048_repeat.py [reps]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp
import cipkkt
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(7)
n = 2048
M = rng.standard_normal((n, n))
Q = M.T @ M / n
c = rng.standard_normal(n)
A = sp.identity(n, format="csr")
b = np.zeros(n)
seen = {"schur": {}, "full3x3": {}}
last = {}
for rep in range(reps):
    for route in ("schur", "full3x3"):
        kw = {} if route == "schur" else {"kktsolver": "full3x3"}
        s = cipkkt.conicIP(Q, c, A, b, [("R", n)], optTol=1e-6, **kw)
        hv = hashlib.sha1(s.y.tobytes() + s.v.tobytes()).hexdigest()[:12]
        if hv not in seen[route]:
            print("rep %d %s: NEW bits %s status %s Iter %d pobj %.12f" % (rep, route, hv, s.status, s.Iter, s.pobj if hasattr(s, "pobj") else float("nan")), flush=True)
            seen[route][hv] = s.y.copy()
        last[route] = s
    d = np.linalg.norm(last["schur"].y - last["full3x3"].y) / (1 + np.linalg.norm(last["schur"].y))
    if rep < 3 or d > 5e-7: print("rep %d: |y_schur - y_full| / (1 + |y|) = %.3e" % (rep, d), flush=True)
print("distinct results: schur %d, full3x3 %d over %d repetitions" % (len(seen["schur"]), len(seen["full3x3"]), reps))
