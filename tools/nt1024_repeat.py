"""The NT scaling of ONE large S cone (order r, padded to 1024 above 512) from ONE fixed interior pair (v, s), repeated: the packed
scaling of every repetition must have the first one's bits.  Counts the repetitions that differ (a race in the cooperative kernels of
sdp_large.hip: the one-sided Jacobi's block exchange).  usage: python tools/nt1024_repeat.py [r] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p_)
import numpy as np, torch, cipkkt
r = int(sys.argv[1]) if len(sys.argv) > 1 else 640
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
k = r * (r + 1) // 2
n = 4
rng = np.random.default_rng(7)
A = rng.standard_normal((k, n)) / np.sqrt(k)
Q = np.eye(n)
ks = cipkkt.KKTSystem(torch.from_numpy(Q).cuda(), torch.from_numpy(A).cuda(), None, [("S", k)])
def vecm(M):
    iu = np.triu_indices(r)
    out = M[iu].copy(); out[iu[0] != iu[1]] *= np.sqrt(2.0)
    return out
spread = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0        # > 0: eigenvalues 10^-spread .. 1 (a late interior-point iterate)
def spd(seed):
    g = np.random.default_rng(seed); B = g.standard_normal((r, r)) / np.sqrt(r)
    if spread <= 0: return B @ B.T + 0.5 * np.eye(r)
    Qm, _ = np.linalg.qr(B)
    return (Qm * 10.0 ** (-spread * g.random(r))) @ Qm.T
v = torch.from_numpy(vecm(spd(1))).cuda(); s = torch.from_numpy(vecm(spd(2))).cuda()
lam = torch.zeros(k, dtype=torch.float64, device="cuda")
first = None; bad = 0; t0 = time.time()
for it in range(reps):
    ks.set_scaling_identity()                    # (the next NT scaling starts its Jacobi cold: same input, same work)
    ks.set_scaling_from_iterate(v, s, lam)
    out = ks.get_scaling_packed()
    if first is None: first = out
    elif not np.array_equal(first, out):
        bad += 1
        print("rep %d differs: max |d| %.3e (relative to max |F| %.3e)" % (it, np.abs(first - out).max(), np.abs(first).max()), flush=True)
import zlib
print("order %d: %d repetitions, %d differ, %.1f ms each, crc32 of the packed scaling %08x" % (r, reps, bad, 1e3 * (time.time() - t0) / reps, zlib.crc32(first.tobytes())))
ks.close()
