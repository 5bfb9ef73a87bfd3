"""solve4x4 time at n = 8192 (50 calls, three repetitions; bitwise repeatability and a checksum)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import torch, cipkkt
from cipkkt import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
Q, c, A, b, K = W.c2_problem(n, seed=1234, device="cuda")
ks = cipkkt.KKTSystem(Q, A, None, K)
g = torch.Generator(device="cuda"); g.manual_seed(3)
v = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
s = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
lam = torch.zeros(n, dtype=torch.float64, device="cuda")
rhs = torch.randn(3 * n, generator=g, dtype=torch.float64, device="cuda")
dz = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
ks.set_scaling_from_iterate(v, s, lam); ks.factor(); ks.check_factor()
outs = []
for rep in range(3):
    ks.solve4x4_dev(lam, rhs, dz); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): ks.solve4x4_dev(lam, rhs, dz)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    outs.append(dz.clone())
print("solve4x4 ms", dt * 1e3, "bitwise repeatable", bool(torch.equal(outs[0], outs[2])),
      "checksum", float(dz.double().abs().sum()))
