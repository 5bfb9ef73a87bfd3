import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/conicip.jl_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import cipkkt
from cipkkt import workloads as W
prob = W.c4_sdp(r=140, n=64, p=4, seed=5)
got = cipkkt.conicIP(*prob, optTol=1e-6)
print(os.environ.get("CIPKKT_LIB", "default"), got.status, got.Iter, got.n_factor, got.n_solve, got.Mu, got.prFeas, got.duFeas, got.muFeas)
for t in got.trace[-4:]: print(t)
