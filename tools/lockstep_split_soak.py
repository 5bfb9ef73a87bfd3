"""Soak of the two-groups-side-by-side form of a lock-step call (round 6, lockstep.hip: cip_conicip_lockstep): `passes` passes over
`count` config-5 problems of order n with the split on; every pass's iterates must have the bits of the FIRST pass with the split off
(one group after the other).  A difference is a race between the two host threads / streams.
usage: python tools/lockstep_split_soak.py [count] [n] [passes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p_)
import numpy as np, torch
from cipkkt import _lib as L
from cipkkt.batch import _solve_problems_native
from cipkkt.workloads import c5_batch
count = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lib = L.load(); dev = torch.device("cuda:0")
prs = c5_batch(count=count, n=n, seed=4000, device=dev)
lib.cip_set_lockstep_split(1)
ref = _solve_problems_native(prs, dev, 1, "lockstep")
lib.cip_set_lockstep_split(2)
bad = 0; t0 = time.time()
for k in range(passes):
    got = _solve_problems_native(prs, dev, 1, "lockstep")
    for a, b in zip(got, ref):
        if not (a.status == b.status and a.Iter == b.Iter and np.array_equal(a.y, b.y) and np.array_equal(a.v, b.v)):
            bad += 1; break
print("%d problems of order %d, %d passes with two groups side by side: %d passes differ from one group after the other; %.1f ms per pass, %d Optimal"
      % (count, n, passes, bad, 1e3 * (time.time() - t0) / passes, sum(s.status == "Optimal" for s in ref)))
