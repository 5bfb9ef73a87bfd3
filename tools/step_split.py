"""Where a bench step's time goes on the GPU timeline: events after the scaling + assembly + factorisation's launches and after each of
the two solve4x4 (n = 8192).  solve 1 runs while the side stream still prepares the last solve block (ldlt.hip: side_fork): the
difference solve 1 - solve 2 is what that costs the step."""
import os, sys
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
def step(ev=None):
    if ev: ev[0].record()
    ks.set_scaling_from_iterate(v, s, lam); ks.factor(check=False)
    if ev: ev[1].record()
    ks.solve4x4_dev(lam, rhs, dz)
    if ev: ev[2].record()
    ks.solve4x4_dev(lam, rhs, dz)
    if ev: ev[3].record()
for _ in range(5): step()
torch.cuda.synchronize()
tot = [0.0, 0.0, 0.0]
R = 20
for _ in range(R):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    step(ev)
    torch.cuda.synchronize()
    for i in range(3): tot[i] += ev[i].elapsed_time(ev[i + 1])
print("per step (GPU timeline, ms): scaling+assembly+factor %.4f | solve 1 %.4f | solve 2 %.4f | sum %.4f" % (tot[0] / R, tot[1] / R, tot[2] / R, sum(tot) / R))
ks.check_factor(); ks.close()
