"""Config 4 at its literal size through the native loop: time per iteration (and, with CIP_LG_LANCZOS_STATS=1, the histogram
of Lanczos steps per max-step at exit)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import torch, cipkkt
from cipkkt import workloads as W
r = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob = W.c4_sdp(r=r, n=1024, p=16, seed=5)
Q, c, A, b, K, G, d = prob
ks = cipkkt.KKTSystem(Q, A, G, K)
cipkkt.conicIP(*prob, optTol=1e-6, system=ks, maxIters=2)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sol = cipkkt.conicIP(*prob, optTol=1e-6, system=ks)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("conicIP: %s, %d iterations, %.3f s -> %.2f ms per iteration" % (sol.status, sol.Iter, dt, dt / sol.Iter * 1e3))
ks.close()
