"""One BASELINE config through the native interior-point loop, three timed runs (for rocprofv3 kernel traces: tools/iter_timeline.py).
usage: python tools/loop_run.py c2|c2f|c3|c4 [size]     (c2f: the literal 3x3 route)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import torch, cipkkt
from cipkkt import workloads as W
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
kw = dict(optTol=1e-6)
if which in ("c2", "c2f"):
    prob = W.c2_problem(int(sys.argv[2]) if len(sys.argv) > 2 else 8192, 1234, device=torch.device("cuda:0"))
    if which == "c2f": kw["kktsolver"] = "full3x3"
elif which == "c3":
    prob = W.c3_socp()
else:
    prob = W.c4_sdp(r=int(sys.argv[2]) if len(sys.argv) > 2 else 256)
Q, c, A, b, K = prob[:5]
G, d = (prob[5], prob[6]) if len(prob) > 5 else (None, None)
ks = cipkkt.KKTSystem(Q, A, G, K, route=kw.get("kktsolver", "schur"))
cipkkt.conicIP(Q, c, A, b, K, G, d, system=ks, maxIters=2, **kw)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sol = cipkkt.conicIP(Q, c, A, b, K, G, d, system=ks, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    import hashlib
    hv = hashlib.sha1(b"".join(torch.as_tensor(t).cpu().numpy().tobytes() for t in (sol.y, sol.w, sol.v) if t is not None)).hexdigest()[:12]
    print("%s: %s, %d iterations, %d factorisations, %d solves, %.4f s -> %.3f ms per iteration   sha1(y,w,v) %s" % (which, sol.status, sol.Iter, sol.n_factor, sol.n_solve, dt, dt / sol.Iter * 1e3, hv), flush=True)
ks.close()
