import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd', ROOT + '/tests'): sys.path.insert(0, p)
import numpy as np, cipkkt, problems as P
from oracle.preprocess import preprocess_conicIP as o_pre
c, A, b, con, var = P.miles_problem(3)
for kc, ka in [(k, k) for k in (1e-8, 1e-6, 1e-4, 1.0, 1e4, 1e6, 1e8)] + [(1.0, k) for k in (1e-4, 1e4, 1e6)]:
    prob = P.mpb_to_conicip(kc * c, ka * A, ka * b, con, var)
    for route in ("schur", "full3x3"):
        s = cipkkt.preprocess_conicIP(*prob, kktsolver=route)
        print(kc, ka, route, s.status, s.Iter, len(s.trace), "mu", s.trace[-1]["mu"] if s.trace else None)
    r = o_pre(*prob)
    print(kc, ka, "oracle", r.status, r.Iter, len(r.trace))
