"""Soak of a whole BASELINE config through the native interior-point loop: the same problem again and again on one handle (as an outer
loop -- MPC, parameter sweeps -- uses the library), every solution's bits against the first.
usage: python tools/loop_soak.py c2|c2f|c3|c4|c5 <seconds> [size]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import numpy as np, torch, cipkkt
from cipkkt import workloads as W
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
dev = torch.device("cuda:0")
def sha(sol):
    return hashlib.sha1(b"".join(np.ascontiguousarray(torch.as_tensor(t).cpu().numpy()).tobytes() for t in (sol.y, sol.w, sol.v) if t is not None)).hexdigest()[:12]
t0 = time.time(); runs = 0; odd = 0; nfac = 0
if which == "c5":
    from cipkkt.batch import _solve_problems_native
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    prs = W.c5_batch(count=count, n=2048, seed=4000, device=dev)
    ref = None
    while time.time() - t0 < budget:
        sols = _solve_problems_native(prs, dev, 1, "lockstep")
        hs = [sha(s) for s in sols]
        runs += 1; nfac += sum(s.n_factor for s in sols)
        if ref is None: ref = hs; print("c5 x%d reference %s ..." % (count, hs[:3]), flush=True)
        elif hs != ref:
            odd += 1
            print("pass %d: problems with other bits: %s" % (runs, [i for i, (a, b) in enumerate(zip(hs, ref)) if a != b]), flush=True)
else:
    kw = dict(optTol=1e-6)
    if which in ("c2", "c2f"):
        prob = W.c2_problem(int(sys.argv[3]) if len(sys.argv) > 3 else 8192, 1234, device=dev)
        if which == "c2f": kw["kktsolver"] = "full3x3"
    elif which == "c3": prob = W.c3_socp()
    else: prob = W.c4_sdp(r=int(sys.argv[3]) if len(sys.argv) > 3 else 256)
    Q, c, A, b, K = prob[:5]
    G, d = (prob[5], prob[6]) if len(prob) > 5 else (None, None)
    ks = cipkkt.KKTSystem(Q, A, G, K, route=kw.get("kktsolver", "schur"))
    ref = None
    while time.time() - t0 < budget:
        sol = cipkkt.conicIP(Q, c, A, b, K, G, d, system=ks, **kw)
        h = sha(sol); runs += 1; nfac += sol.n_factor
        if ref is None: ref = (h, sol); print("%s reference %s: %s, %d iterations" % (which, h, sol.status, sol.Iter), flush=True)
        elif h != ref[0]:
            odd += 1
            first = next((i for i, (a, b_) in enumerate(zip(sol.trace, ref[1].trace)) if any(a.get(k) != b_.get(k) for k in ("mu", "pobj", "rPr", "alpha", "sigma"))), None)
            print("run %d: OTHER BITS %s status %s Iter %d first differing iteration %s health %s" % (runs, h, sol.status, sol.Iter, first, ks.health()), flush=True)
    print("health at the end: %s" % (ks.health(),))
    ks.close()
print("%s: %d runs, %d factorisations in %.0f s, %d with other bits" % (which, runs, nfac, time.time() - t0, odd))
