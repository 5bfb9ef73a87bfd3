import sys, os, time
ROOT='/root/repo'
for p in (ROOT, ROOT+'/conicip.jl_amd', ROOT+'/tests'): sys.path.insert(0,p)
import numpy as np, torch, cipkkt
from oracle import cones as oc
r=128; rng=np.random.default_rng(5); n,p=256,16; k=r*(r+1)//2
A=rng.standard_normal((k,n))/np.sqrt(n)
prob=(np.eye(n), rng.standard_normal(n), A, -oc.vecm(np.eye(r)), [("S",k)], rng.standard_normal((p,n)), np.zeros(p))
ks=cipkkt.KKTSystem(prob[0],prob[2],prob[5],prob[4])
sol=cipkkt.conicIP(*prob, system=ks, optTol=1e-6, maxIters=3)
print(sol.status, sol.Iter)
