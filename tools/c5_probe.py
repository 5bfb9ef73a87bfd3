"""C5 probe: one dense QP of order n (default 2048) through the native driver; prints wall time and per-iteration cost.
Run under rocprofv3 --kernel-trace --stats to compare the kernel-time sum with the wall clock."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd', ROOT + '/tests'): sys.path.insert(0, p)
import numpy as np, torch, cipkkt
import scipy.sparse as sp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
drv = sys.argv[2] if len(sys.argv) > 2 else "native"
rng = np.random.default_rng(3)
M = rng.standard_normal((n, n)); Q = M.T @ M / n
c = rng.standard_normal(n)
ks = cipkkt.KKTSystem(Q, sp.identity(n, format="csr"), None, [("R", n)])
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sol = cipkkt.conicIP(Q, c, sp.identity(n, format="csr"), np.zeros(n), [("R", n)], system=ks, driver=drv)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"n={n} {drv}: {sol.status} iters={len(sol.trace)} factor={sol.n_factor} solve={sol.n_solve} wall={1e3*(t1-t0):.2f} ms  per factorisation {1e3*(t1-t0)/sol.n_factor:.2f} ms")
