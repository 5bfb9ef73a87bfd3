"""Stress of the in-launch hand-offs of the one-launch-per-panel chain: many factorisations of one matrix, every factor
compared bit for bit with the first, a second stream streaming memory beside them.
usage: python tools/chain_stress.py [N] [repetitions]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import torch, cipkkt
from cipkkt import _lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
lib = L.load()
nbytes = C.c_size_t()
L.check(lib.cip_ldlt_workspace_bytes(N, C.byref(nbytes)))
ws = torch.zeros(nbytes.value // 8 + 8, dtype=torch.float64, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(N)
M = torch.randn(N, N, generator=g, dtype=torch.float64, device="cuda")
K0 = (M @ M.t() / N + torch.eye(N, dtype=torch.float64, device="cuda")).contiguous()
def factor():
    dK = K0.clone(); info = C.c_int(-1)
    L.check(lib.cip_ldlt_factor_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), C.byref(info)))
    torch.cuda.synchronize()
    assert info.value == 0, info.value
    return dK
ref = torch.tril(factor().t())
side = torch.cuda.Stream()
a = torch.empty(1 << 26, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
bad = 0
for r in range(reps):
    if r % 3 == 0:
        with torch.cuda.stream(side):
            b.copy_(a, non_blocking=True); a.copy_(b, non_blocking=True)
    if not torch.equal(torch.tril(factor().t()), ref): bad += 1
side.synchronize()
print("N", N, "repetitions", reps, "factors that differ from the first:", bad)
