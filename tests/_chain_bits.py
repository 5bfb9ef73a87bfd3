"""Helper of tests/test_gpu_panel_chain.py (run as a subprocess so that CIPKKT_LIB can select a test build of the library): factor the
KKT matrix of a dense QP under two scalings on both routes, solve, print one line of hashes."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conicip.jl_amd"))
import numpy as np
import scipy.sparse as sp
import torch
import cipkkt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(n)
M = rng.standard_normal((n, n))
Q = M.T @ M / n
A = sp.identity(n, format="csr")
f64 = dict(dtype=torch.float64, device="cuda:0")
out = []
for route in ("schur", "full3x3"):
    ks = cipkkt.KKTSystem(Q, A, None, [("R", n)], route=route)
    x = torch.as_tensor(rng.standard_normal(n), **f64); y = torch.zeros(0, **f64); z = torch.as_tensor(rng.standard_normal(n), **f64)
    h = hashlib.sha1()
    for rep in range(reps):
        v = torch.as_tensor(rng.random(n) + 0.1, **f64); s = torch.as_tensor(rng.random(n) + 1e-3, **f64)
        ks.set_scaling_from_iterate(v, s)
        ks.factor(check=True)
        a = torch.empty(n, **f64); b = torch.empty(0, **f64); c = torch.empty(n, **f64)
        ks.solve3x3_dev(x, y, z, a, b, c)
        h.update(a.cpu().numpy().tobytes()); h.update(c.cpu().numpy().tobytes())
    ks.close()
    out.append(route + ":" + h.hexdigest()[:16])
print("BITS " + " ".join(out))
