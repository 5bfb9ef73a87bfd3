"""C4 (SDP) probe: one S cone of order r, n variables, p equalities; prints wall time; use under rocprofv3 for
the kernel breakdown.  usage: c4_probe.py [r] [n] [maxIters]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd', ROOT + '/tests'): sys.path.insert(0, p)
import numpy as np, torch, cipkkt
from oracle import cones as oc
r = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rng = np.random.default_rng(5); p = 16; k = r * (r + 1) // 2
A = rng.standard_normal((k, n)) / np.sqrt(n)
prob = (np.eye(n), rng.standard_normal(n), A, -oc.vecm(np.eye(r)), [("S", k)], rng.standard_normal((p, n)), np.zeros(p))
ks = cipkkt.KKTSystem(prob[0], prob[2], prob[5], prob[4])
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sol = cipkkt.conicIP(*prob, system=ks, optTol=1e-6, maxIters=mi)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"r={r} n={n} k={k}: {sol.status} iters={sol.Iter} wall={t1 - t0:.3f}s")
