"""As fuzz_status.py with badly scaled data: c by 10^[-6,6], (A, b) and (G, d) by 10^[-3,3] (the feasible set and the
status do not change).  Development tool."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd', ROOT + '/tests'): sys.path.insert(0, p)
import numpy as np, cipkkt, problems as P
from oracle.preprocess import preprocess_conicIP as o_pre
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad, hist = 0, {}
for seed in range(N):
    Q, c, A, b, K, G, d = P.random_degenerate(seed)
    rng = np.random.default_rng(10_000 + seed)
    kc, ka, kg = 10.0 ** rng.integers(-6, 7), 10.0 ** rng.integers(-3, 4), 10.0 ** rng.integers(-3, 4)
    prob = (Q, kc * c, ka * A, ka * b, K, kg * G, kg * d)
    r = o_pre(*prob, optTol=1e-7, maxIters=100)
    g = cipkkt.preprocess_conicIP(*prob, optTol=1e-7, maxIters=100)
    hist[(r.status, g.status)] = hist.get((r.status, g.status), 0) + 1
    if r.status != g.status:
        bad += 1
        print("seed", seed, "kc ka kg", kc, ka, kg, K, "p", G.shape[0], "oracle", r.status, r.Iter, "product", g.status, g.Iter)
print("cases", N, "mismatches", bad, hist)
