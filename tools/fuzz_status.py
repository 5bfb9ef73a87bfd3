"""Status fuzz: random small conic programs with degenerate structure (tests/problems.py::random_degenerate) through the
oracle's and the product's pre-solve + interior-point loop; prints every case whose outcomes differ.
usage: fuzz_status.py [cases]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd', ROOT + '/tests'): sys.path.insert(0, p)
import numpy as np, cipkkt, problems as P
from oracle.preprocess import preprocess_conicIP as o_pre
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad, hist = 0, {}
for seed in range(N):
    prob = P.random_degenerate(seed)
    r = o_pre(*prob, optTol=1e-7, maxIters=80)
    g = cipkkt.preprocess_conicIP(*prob, optTol=1e-7, maxIters=80)
    hist[r.status] = hist.get(r.status, 0) + 1
    ok = r.status == g.status and (r.status != "Optimal" or np.linalg.norm(r.y - g.y) <= 1e-4 * (1 + np.linalg.norm(r.y)))
    if not ok:
        bad += 1
        print("seed", seed, prob[4], "p", prob[5].shape[0], "oracle", r.status, r.Iter, "product", g.status, g.Iter)
print("cases", N, "mismatches", bad, "oracle statuses", hist)
