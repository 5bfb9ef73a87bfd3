"""The n = 2048 dense QP of tests/test_gpu_driver.py on both routes, with the k-th factorisation of the run reporting an in-launch
give-up of the fused panel chain (cip_debug_chain_giveup), k = 0 .. 9: the bits must not depend on it."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp
import cipkkt
from cipkkt import _lib as L
lib = L.load()
rng = np.random.default_rng(7)
n = 2048
M = rng.standard_normal((n, n))
Q = M.T @ M / n
c = rng.standard_normal(n)
A = sp.identity(n, format="csr")
b = np.zeros(n)
K = [("R", n)]
KIND = (1 << 28) if (len(sys.argv) > 1 and sys.argv[1] == 'pivot') else 0
def run(route, arm=None):
    ks = cipkkt.KKTSystem(Q, A, None, K, route=route)
    if arm is not None: lib.cip_debug_chain_giveup(1 + 65536 * arm + KIND)
    try:
        s = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6, system=ks)
        left = lib.cip_debug_chain_giveup(-1)
    finally:
        lib.cip_debug_chain_giveup(0)
    hl = ks.health(); ks.close()
    return hashlib.sha1(s.y.tobytes() + s.v.tobytes()).hexdigest()[:12], s, hl, left
bad = 0
for route in ("schur", "full3x3"):
    h0, s0, hl0, _ = run(route)
    print("%s: reference bits %s, %d iterations, %d factorisations, health %s" % (route, h0, s0.Iter, s0.n_factor, hl0), flush=True)
    for k in range(s0.n_factor + 1):
        h, s, hl, left = run(route, k)
        same = h == h0
        bad += (not same)
        print("  give-up at factorisation %d: bits %s %s  status %s Iter %d  health %s  hook left %d  dev %.3e pobj %.13f dobj %.13f rPr %.15e" % (
            k, h, "same" if same else "DIFFERENT", s.status, s.Iter, hl, left, np.linalg.norm(s.y - s0.y) / (1 + np.linalg.norm(s0.y)), s.pobj, s.dobj, s.trace[-1]["rPr"]), flush=True)
print("runs with other bits: %d" % bad)
