"""The max-step of ONE large S cone (order r > 256: LDL' + two GEMMs + the cooperative tridiagonalisation + Sturm) from one fixed
(x, d), repeated: every repetition must return the first one's bits.  usage: python tools/maxstep1024_repeat.py [r] [reps] [spread]"""
import os, sys, time, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p_)
import numpy as np, torch, cipkkt
r = int(sys.argv[1]) if len(sys.argv) > 1 else 640
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
spread = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
k = r * (r + 1) // 2
n = 4
rng = np.random.default_rng(7)
A = rng.standard_normal((k, n)) / np.sqrt(k)
ks = cipkkt.KKTSystem(torch.from_numpy(np.eye(n)).cuda(), torch.from_numpy(A).cuda(), None, [("S", k)])
def vecm(M):
    iu = np.triu_indices(r)
    out = M[iu].copy(); out[iu[0] != iu[1]] *= np.sqrt(2.0)
    return out
g = np.random.default_rng(1); B = g.standard_normal((r, r)) / np.sqrt(r)
if spread > 0:
    Qm, _ = np.linalg.qr(B); X = (Qm * 10.0 ** (-spread * g.random(r))) @ Qm.T
else:
    X = B @ B.T + 0.5 * np.eye(r)
D = g.standard_normal((r, r)); D = 0.5 * (D + D.T)
x = torch.from_numpy(vecm(X)).cuda(); d = torch.from_numpy(vecm(D)).cuda()
first = None; bad = 0; t0 = time.time()
for it in range(reps):
    a = ks.maxstep(x, d, 1.0)
    bits = struct.pack("d", a)
    if first is None: first = (a, bits)
    elif bits != first[1]:
        bad += 1; print("rep %d: %.17g against %.17g" % (it, a, first[0]), flush=True)
print("order %d: %d repetitions, %d differ, %.2f ms each, alpha %.17g" % (r, reps, bad, 1e3 * (time.time() - t0) / reps, first[0]))
ks.close()
