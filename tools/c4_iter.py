"""config 4 (matrix order r, ("S", r (r + 1) / 2), n = 1024, p = 16): the native loop to convergence, ms per iteration; CIP_LG_DEBUG=1
prints the Jacobi's sweep count of every NT scaling.  usage: python tools/c4_iter.py [r] [n] [p]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p_)
import cipkkt
from cipkkt import workloads as W
r = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
p = int(sys.argv[3]) if len(sys.argv) > 3 else 16
Q, c, A, b, K, G, d = W.c4_sdp(r, n, p, seed=5)
ks = cipkkt.KKTSystem(Q, A, G, K)
for rep in range(3):
    sol = cipkkt.conicIP(Q, c, A, b, K, G, d, optTol=1e-6, system=ks)
    print("rep %d: %s %d iterations, %.3f ms per iteration (%.1f ms), mu %.6e pobj %.12e" % (rep, sol.status, sol.Iter, 1e3 * sol.wall_s / sol.Iter, 1e3 * sol.wall_s, sol.Mu, sol.pobj), flush=True)
ks.close()
