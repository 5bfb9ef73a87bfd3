"""Soak of the n = 2048 dense QP of tests/test_gpu_driver.py::test_dense_qp_2048_properties (both routes, a fresh handle per solve, as
the test does): every run's iteration trace against the first run's.  A run that leaves the usual path is printed with the first
iteration at which it does.   usage: python tools/qp2048_soak.py <seconds> [route ...]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp
import cipkkt
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
routes = sys.argv[2:] or ["schur", "full3x3"]
rng = np.random.default_rng(7)
n = 2048
M = rng.standard_normal((n, n))
Q = M.T @ M / n
c = rng.standard_normal(n)
A = sp.identity(n, format="csr")
b = np.zeros(n)
K = [("R", n)]
ref = {}
count = {r: 0 for r in routes}
odd = {r: 0 for r in routes}
t0 = time.time()
while time.time() - t0 < budget:
    for route in routes:
        ks = cipkkt.KKTSystem(Q, A, None, K, route=route)
        s = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6, system=ks)
        hl = ks.health(); ks.close()
        hv = hashlib.sha1(s.y.tobytes() + s.v.tobytes()).hexdigest()[:12]
        count[route] += 1
        if route not in ref:
            ref[route] = (hv, s)
            print("%s: reference %s, %d iterations" % (route, hv, s.Iter), flush=True)
            continue
        if hv != ref[route][0]:
            odd[route] += 1
            r = ref[route][1]
            first = next((i for i, (a, b_) in enumerate(zip(s.trace, r.trace)) if any(a.get(k) != b_.get(k) for k in ("mu", "pobj", "dobj", "rPr", "rDu", "alpha", "sigma"))), None)
            print("%s run %d: OTHER BITS %s  status %s Iter %d  health %s  dev %.3e  first differing iteration %s" % (
                route, count[route], hv, s.status, s.Iter, hl, np.linalg.norm(s.y - r.y) / (1 + np.linalg.norm(r.y)), first), flush=True)
            if first is not None:
                for i in range(max(0, first - 1), min(len(s.trace), first + 3)):
                    print("    it %d  this %s" % (i, {k: s.trace[i].get(k) for k in ("mu", "pobj", "rPr", "rDu", "alpha", "sigma")}), flush=True)
                    print("          ref  %s" % ({k: r.trace[i].get(k) for k in ("mu", "pobj", "rPr", "rDu", "alpha", "sigma")},), flush=True)
print("runs: %s   runs with other bits: %s   (%.0f s)" % (count, odd, time.time() - t0))
