"""BASELINE config 4 at its literal size (matrix order 256, ("S", 32896), n = 1024, p = 16): time per interior-point
iteration and per S-cone entry point."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import numpy as np, torch, cipkkt
from cipkkt import workloads as W
r = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob = W.c4_sdp(r=r, n=1024, p=16, seed=5)
Q, c, A, b, K, G, d = prob
ks = cipkkt.KKTSystem(Q, A, G, K)
k = K[0][1]
rng = np.random.default_rng(0)
def psd():
    M = rng.standard_normal((r, r)); S = M @ M.T / r + 0.5 * np.eye(r)
    iu = np.triu_indices(r); v = S[iu] * np.sqrt(2.0); v[np.cumsum(np.concatenate([[0], np.arange(r, 1, -1)]))] = np.diag(S); return v
dev = lambda x: torch.as_tensor(x, dtype=torch.float64, device="cuda")
v, s, dd = dev(psd()), dev(psd()), dev(rng.standard_normal(k))
lam = torch.zeros(k, dtype=torch.float64, device="cuda")
def timeit(name, fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); print("%-28s %8.3f ms" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)
timeit("nt_scaling", lambda: ks.set_scaling_from_iterate(v, s, lam))
timeit("maxstep(x, d)", lambda: ks.maxstep(v, dd))
timeit("maxstep(x, nothing)", lambda: ks.maxstep(dd, None))
out = torch.zeros_like(v)
timeit("apply F", lambda: ks.apply_F(0, dd, out))
timeit("cone_prod", lambda: ks.cone_prod(v, dd, out))
timeit("cone_div by lambda", lambda: ks.cone_div(dd, lam, out))
timeit("factor (scale A', SYRK, LDL')", lambda: ks.factor())
ks.check_factor()
cipkkt.conicIP(*prob, optTol=1e-6, system=ks, maxIters=2)
torch.cuda.synchronize(); t0 = time.perf_counter()
sol = cipkkt.conicIP(*prob, optTol=1e-6, system=ks)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("conicIP: %s, %d iterations, %.3f s -> %.1f ms per iteration" % (sol.status, sol.Iter, dt, dt / sol.Iter * 1e3))
