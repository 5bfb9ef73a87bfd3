"""S cones of the large path (order 133..256: Lanczos max-step, paired max-steps) against the oracle on random instances of
config 4's family: status, iteration count, iterates.   usage: python tools/fuzz_sdp.py [count] [orders, comma separated]
(default orders 133..256; e.g. "300,400,640,900,1100,1300" draws from the padded-512 / 1024 / 2048 paths: n = 6..12 there)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import numpy as np, cipkkt
from cipkkt import workloads as W
from oracle.conicip import conicIP as oracle_conicIP
from oracle import kktsolvers as ok
count = int(sys.argv[1]) if len(sys.argv) > 1 else 6
orders = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [133, 140, 160, 192, 200, 256]
rng = np.random.default_rng(2026)
bad = 0
for t in range(count):
    r = int(rng.choice(orders)); n = int(rng.choice([24, 40, 64] if r <= 256 else [6, 8, 12])); p = int(rng.choice([0, 2, 4])); seed = int(rng.integers(1, 10**6))
    prob = W.c4_sdp(r=r, n=n, p=p, seed=seed)
    t0 = time.time(); ref = oracle_conicIP(*prob, optTol=1e-6, kktsolver=ok.kktsolver_schur_exact); t1 = time.time()
    got = cipkkt.conicIP(*prob, optTol=1e-6)
    dy = np.linalg.norm(got.y - ref.y) / (1 + np.linalg.norm(ref.y))
    okk = got.status == ref.status and got.Iter == ref.Iter and dy < 1e-5
    bad += not okk
    print("r %3d n %2d p %d seed %6d: oracle %s %d it (%.0f s) | gpu %s %d it | rel dy %.1e %s" % (r, n, p, seed, ref.status, ref.Iter, t1 - t0, got.status, got.Iter, dy, "" if okk else "  <-- MISMATCH"), flush=True)
print("mismatches:", bad)
