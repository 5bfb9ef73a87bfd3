"""The order-640 S-cone program of tests/test_gpu_sdp.py, solved repeatedly: iteration count and objective of every run (a run-to-run
difference is a race).  usage: python tools/sdp640_repeat.py [runs] [r]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p_)
import cipkkt
from cipkkt import workloads as W
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = int(sys.argv[2]) if len(sys.argv) > 2 else 640
prob = W.c4_sdp(r=r, n=12, p=0, seed=31337)
for k in range(runs):
    sol = cipkkt.conicIP(*prob, optTol=1e-6)
    print("run %d: %s Iter %d pobj %.15e mu %.6e" % (k, sol.status, sol.Iter, sol.pobj, sol.Mu), flush=True)
