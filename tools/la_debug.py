"""Diagnostic for the look-ahead schedule: reconstruction error of L D L' under the three schedules."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import torch, cipkkt
from cipkkt import _lib as L
lib = L.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator(device="cuda"); g.manual_seed(1)
M = torch.randn(N, N, generator=g, dtype=torch.float64, device="cuda")
K0 = (M @ M.t() / N + torch.eye(N, dtype=torch.float64, device="cuda")).contiguous()
nb = C.c_size_t(); L.check(lib.cip_ldlt_workspace_bytes(N, C.byref(nb)))
ws = torch.zeros(nb.value // 8 + 8, dtype=torch.float64, device="cuda")
def run(mode):
    lib.cip_set_ldlt_lookahead(mode)
    dK = K0.clone(); info = C.c_int(-1)
    L.check(lib.cip_ldlt_factor_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), C.byref(info)))
    torch.cuda.synchronize()
    F = dK.t()
    Lf = torch.tril(F, -1) + torch.eye(N, dtype=torch.float64, device="cuda"); D = torch.diagonal(F)
    rec = (Lf * D[None, :]) @ Lf.t()
    err = torch.tril(rec - K0).abs()
    blk = err.reshape(N // 512, 512, N // 512, 512).amax(dim=(1, 3))
    return dK, info.value, float(err.max()), blk
ref = None
for mode in (0, 2, 1, 1):
    dK, info, e, blk = run(mode)
    print("mode", mode, "info", info, "max |LDL'-K| (lower)", e)
    if e > 1e-9:
        print((blk > 1e-9).int().cpu().numpy())
    if mode == 2: ref = dK
    if mode == 1:
        d = (torch.tril(dK.t()) != torch.tril(ref.t()))
        print("  differing entries vs mode 2:", int(d.sum()), " first differing 64-tiles:", torch.nonzero(d.reshape(N//64,64,N//64,64).any(dim=3).any(dim=1))[:10].tolist())
# timing of the factorisation under each schedule
import time
for mode in (0, 1):
    lib.cip_set_ldlt_lookahead(mode)
    dK = K0.clone(); info = C.c_int(-1)
    for rep in range(3):
        dK.copy_(K0); torch.cuda.synchronize(); t0 = time.perf_counter()
        L.check(lib.cip_ldlt_factor_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), None))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("mode", mode, "factor ms", dt * 1e3, "TFLOP/s", N ** 3 / 3 / dt / 1e12)
