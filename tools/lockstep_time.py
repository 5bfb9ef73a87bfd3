"""config 5 on one GPU: lock-step batch against the thread pool (csrc/lockstep.hip vs csrc/batch.hip).
usage: python tools/lockstep_time.py [count] [n] [passes] [lockstep|threads|both] [outer block]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conicip.jl_amd"))
from cipkkt.batch import _solve_problems_native  # noqa: E402
from cipkkt.workloads import c5_batch  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
prs = c5_batch(count=count, n=n, seed=4000, device=dev)
which = sys.argv[4] if len(sys.argv) > 4 else "both"
if len(sys.argv) > 5:
    from cipkkt import _lib as L
    L.load().cip_set_ldlt_outer_block(int(sys.argv[5]))
for mode, inflight in (("lockstep", 1), ("threads", 8)):
    if which not in ("both", mode):
        continue
    for k in range(passes + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sols = _solve_problems_native(prs, dev, inflight, mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nf = sum(s.n_factor for s in sols)
        if k > 0:
            print("%-9s pass %d: %.1f ms, %d factorisations -> %.0f KKT solves/s, %d Optimal" %
                  (mode, k, 1e3 * dt, nf, nf / dt, sum(s.status == "Optimal" for s in sols)))
