"""BASELINE config 5 at full size on one GPU: 64 independent dense QPs, n = m = 2048, R cone; solve_batch with
several problems in flight.  Prints throughput and checks every problem converged."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd', ROOT + '/tests'): sys.path.insert(0, p)
import numpy as np, torch, scipy.sparse as sp
from cipkkt.batch import solve_batch
n, count = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 64
probs = []
for i in range(count):
    g = torch.Generator(device="cuda"); g.manual_seed(4000 + i)
    M = torch.randn(n, n, generator=g, dtype=torch.float64, device="cuda")
    probs.append(dict(Q=(M.t() @ M / n), c=np.random.default_rng(i).standard_normal(n), A=sp.identity(n, format="csr"),
                      b=np.zeros(n), cone_dims=[("R", n)], kwargs=dict(optTol=1e-6)))
solve_batch(probs[:2])
for conc in (1, 2, 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sols, st = solve_batch(probs, concurrency=conc)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps(dict(config="C5 %d x dense QP n=2048, %d in flight" % (count, conc), n_optimal=st["n_optimal"], iters=st["iters"],
                          n_factor=st["n_factor"], wall_s=round(dt, 3), kkt_solves_per_s=round(st["n_factor"] / dt, 1),
                          problems_per_s=round(count / dt, 1))), flush=True)
