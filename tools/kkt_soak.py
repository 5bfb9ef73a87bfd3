"""Level-2/3 soak of one KKT system: alternate between two scalings (so that anything left over from the previous factorisation shows),
factor, solve twice with one right-hand side, compare every output with the first time that scaling was used -- bit for bit, on the device.
usage: python tools/kkt_soak.py <route> <n> <seconds>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp, torch
import cipkkt
route = sys.argv[1] if len(sys.argv) > 1 else "full3x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
rng = np.random.default_rng(7)
M = rng.standard_normal((n, n))
Q = M.T @ M / n
A = sp.identity(n, format="csr")
K = [("R", n)]
ks = cipkkt.KKTSystem(Q, A, None, K, route=route)
dev = torch.device("cuda:0")
f64 = dict(dtype=torch.float64, device=dev)
sc = [(torch.as_tensor(rng.random(n) + 0.1, **f64), torch.as_tensor(rng.random(n) * (10.0 ** rng.uniform(-3, 1, n)) + 1e-3, **f64)) for _ in range(2)]
x = torch.as_tensor(rng.standard_normal(n), **f64); y = torch.zeros(0, **f64); z = torch.as_tensor(rng.standard_normal(n), **f64)
lam = torch.zeros(n, **f64)
def once(k):
    v, s = sc[k]
    ks.set_scaling_from_iterate(v, s, lam)
    ks.factor(check=False)
    outs = []
    for _ in range(2):
        a = torch.empty(n, **f64); b = torch.empty(0, **f64); c = torch.empty(n, **f64)
        ks.solve3x3_dev(x, y, z, a, b, c)
        outs.append((a, c))
    return lam.clone(), outs
ref = [once(0), once(1)]
import hashlib
print("reference solutions: %s" % [hashlib.sha1(ref[k][1][0][0].cpu().numpy().tobytes()).hexdigest()[:12] for k in (0, 1)], flush=True)
Kref = []
for k in (0, 1):
    once(k); Kref.append(np.tril(ks.kkt_matrix()).copy())          # the factor (L below the diagonal, D on it) of either scaling
for k in (0, 1):
    assert torch.equal(ref[k][1][0][0], ref[k][1][1][0]), "the two solves of one factorisation differ in the reference itself"
reps = 0
odd = []
t0 = time.time()
while time.time() - t0 < budget:
    for k in (0, 1):
        l, outs = once(k)
        reps += 1
        e_l = not torch.equal(l, ref[k][0])
        e1 = not (torch.equal(outs[0][0], ref[k][1][0][0]) and torch.equal(outs[0][1], ref[k][1][0][1]))
        e2 = not (torch.equal(outs[1][0], ref[k][1][1][0]) and torch.equal(outs[1][1], ref[k][1][1][1]))
        if e_l or e1 or e2:
            r1 = ((outs[0][0] - ref[k][1][0][0]).norm() / ref[k][1][0][0].norm()).item()
            r2 = ((outs[1][0] - ref[k][1][1][0]).norm() / ref[k][1][1][0].norm()).item()
            # how close is the odd first solve to the OTHER scaling's answer (a left-over of the previous factorisation)?
            ro = ((outs[0][0] - ref[1 - k][1][0][0]).norm() / ref[1 - k][1][0][0].norm()).item()
            nbad = int((outs[0][0] != ref[k][1][0][0]).sum().item())
            odd.append((reps, k, e_l, e1, e2, r1, r2))
            Kodd = np.tril(ks.kkt_matrix())
            dif = Kodd != Kref[k]
            if dif.any():
                rr, cc = np.nonzero(dif)
                cols = np.unique(cc)
                print("    factor: %d entries differ, rows %d..%d, columns %d..%d (%d distinct columns; first columns %s), max |diff| %.3e, of which on the diagonal %d" % (
                    dif.sum(), rr.min(), rr.max(), cc.min(), cc.max(), len(cols), cols[:12].tolist(), np.abs(Kodd - Kref[k])[dif].max(), int((rr == cc).sum())), flush=True)
                c0 = cc.min()
                r_in = rr[cc == c0]
                print("    first differing column %d: rows %s%s" % (c0, r_in[:16].tolist(), " ..." if len(r_in) > 16 else ""), flush=True)
                # per 128-column panel: how many entries differ
                pan = np.bincount(cc // 128, minlength=Kodd.shape[0] // 128)
                print("    entries per 128-column panel: %s" % pan.tolist(), flush=True)
            else:
                print("    factor: identical to the reference -- the solve preparation (block inverses) or the sweeps differ", flush=True)
            a3 = torch.empty(n, **f64); b3 = torch.empty(0, **f64); c3 = torch.empty(n, **f64)
            ks.solve3x3_dev(x, y, z, a3, b3, c3)
            print("    a third solve on the same factor: %s" % ("ODD" if not torch.equal(a3, ref[k][1][0][0]) else "ok"), flush=True)
            print("rep %d scaling %d: lambda %s, first solve %s (rel %.3e, %d entries, rel to the other scaling's answer %.3e), second solve %s (rel %.3e)  health %s" % (
                reps, k, "ODD" if e_l else "ok", "ODD" if e1 else "ok", r1, nbad, ro, "ODD" if e2 else "ok", r2, ks.health()), flush=True)
print("%s n=%d: %d factorisations (+ 2 solves each) in %.0f s, %d with other bits" % (route, n, reps, time.time() - t0, len(odd)))
ks.close()
