"""Stand-alone LDL' of one matrix, factored + solved repeatedly: every solution must have the first one's bits (a difference is a race in
the factorisation / the block-inverse doubling).  usage: python tools/doubling_repeat.py [N] [runs]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p_)
import numpy as np, torch
from cipkkt import _lib as L
lib = L.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.default_rng(N)
M = rng.standard_normal((N, N)); Kmat = M @ M.T / N + np.eye(N)
rhs = rng.standard_normal(N)
nbytes = C.c_size_t(); L.check(lib.cip_ldlt_workspace_bytes(N, C.byref(nbytes)))
ws = torch.zeros(nbytes.value // 8 + 8, dtype=torch.float64, device="cuda")
K0 = torch.from_numpy(np.asfortranarray(Kmat).ravel(order="F").copy()).cuda()
first = None; bad = 0
for k in range(runs):
    dK = K0.clone(); info = C.c_int(-1)
    L.check(lib.cip_ldlt_factor_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), C.byref(info)))
    x = torch.from_numpy(rhs.copy()).cuda()
    L.check(lib.cip_ldlt_solve_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), x.data_ptr()))
    torch.cuda.synchronize()
    if first is None: first = x.clone()
    elif not torch.equal(first, x): bad += 1; print("run", k, "differs: max |dx|", float((first - x).abs().max()))
print("N", N, "runs", runs, "differing", bad)
