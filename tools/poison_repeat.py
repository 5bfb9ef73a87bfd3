"""Are the two routes of the n = 2048 dense QP of tests/test_gpu_driver.py reproducible when the device memory the library's fresh
allocations land on holds NaNs / garbage (torch tensors filled, freed and handed back to the driver before every solve)?
usage: python tools/poison_repeat.py [reps] [poison: nan|rand|none]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp, torch
import cipkkt
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
poison = sys.argv[2] if len(sys.argv) > 2 else "nan"
rng = np.random.default_rng(7)
n = 2048
M = rng.standard_normal((n, n))
Q = M.T @ M / n
c = rng.standard_normal(n)
A = sp.identity(n, format="csr")
b = np.zeros(n)
K = [("R", n)]
seen = {"schur": {}, "full3x3": {}}
prng = np.random.default_rng(99)
for rep in range(reps):
    for route in ("schur", "full3x3"):
        if poison != "none":
            ts = []
            for _ in range(int(prng.integers(1, 5))):
                k = int(prng.integers(1 << 20, 1 << 27))
                t = torch.empty(k, dtype=torch.float64, device="cuda")
                if poison == "nan": t.fill_(float("nan"))
                else: t.uniform_(-1e30, 1e30)
                ts.append(t)
            torch.cuda.synchronize()
            del ts, t
            torch.cuda.empty_cache()
        ks = cipkkt.KKTSystem(Q, A, None, K, route=route)
        s = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6, system=ks)
        hl = ks.health()
        ks.close()
        hv = hashlib.sha1(s.y.tobytes() + s.v.tobytes()).hexdigest()[:12]
        if hv not in seen[route] or hl["n_regularized"] or hl["n_chain_fallbacks"]:
            print("rep %d %s: bits %s%s status %s Iter %d pobj %.12f health %s" % (rep, route, hv, "" if hv in seen[route] else " NEW", s.status, s.Iter, s.pobj, hl), flush=True)
            seen[route][hv] = 1
print("distinct results: schur %d, full3x3 %d over %d repetitions (poison: %s)" % (len(seen["schur"]), len(seen["full3x3"]), reps, poison))
