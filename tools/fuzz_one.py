import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd', ROOT + '/tests'): sys.path.insert(0, p)
import numpy as np, cipkkt, problems as P, ctypes as C
from oracle.preprocess import preprocess_conicIP as o_pre
seed = int(sys.argv[1])
Q, c, A, b, K, G, d = P.random_degenerate(seed)
rng = np.random.default_rng(10_000 + seed)
kc, ka, kg = 10.0 ** rng.integers(-6, 7), 10.0 ** rng.integers(-3, 4), 10.0 ** rng.integers(-3, 4)
prob = (Q, kc * c, ka * A, ka * b, K, kg * G, kg * d)
r = o_pre(*prob, optTol=1e-7, maxIters=100)
for drv in ("native",):
    g = cipkkt.preprocess_conicIP(*prob, optTol=1e-7, maxIters=100, driver=drv)
    print(drv, g.status, len(g.trace), "n_solve", g.n_solve, "n_factor", g.n_factor)
print("oracle", r.status, len(r.trace), r.n_solve)
for i in range(min(16, len(r.trace), len(g.trace))):
    a, bb = r.trace[i], g.trace[i]
    print(i + 1, "mu %.3e %.3e  rDu %.2e %.2e  rPr %.2e %.2e  rCp %.2e %.2e alpha %s %s" % (a["mu"], bb["mu"], a["rDu"], bb["rDu"], a["rPr"], bb["rPr"], a["rCp"], bb["rCp"], a.get("alpha"), bb.get("alpha")))
