"""Two-stream look-ahead (cip_set_ldlt_lookahead(3)) against the serial schedule at n = 8192: bit-identity of the factor
and time per assemble + factor.  Usage: python tools/la2_time.py [n]   (CIP_LA2_RESERVE, CIP_LA2_MIN in the environment)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import numpy as np, torch, cipkkt
from cipkkt import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
Q, c, A, b, K = W.c2_problem(n, seed=1234, device="cuda")
ks = cipkkt.KKTSystem(Q, A, None, K)
g = torch.Generator(device="cuda"); g.manual_seed(3)
v = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
s = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
lam = torch.zeros(n, dtype=torch.float64, device="cuda")
ref = None
for mode in (0, 3, 0, 3):
    ks.lib.cip_set_ldlt_lookahead(mode)
    ts = []
    for rep in range(8):
        ks.set_scaling_from_iterate(v, s, lam)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ks.factor()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ks.check_factor()
    Kf = np.tril(ks.kkt_matrix())
    if ref is None:
        ref = Kf
    same = bool(np.array_equal(ref, Kf))
    print("mode %d: assemble+factor ms %s | factor identical to the serial one: %s" %
          (mode, " ".join("%.3f" % (t * 1e3) for t in ts), same), flush=True)
