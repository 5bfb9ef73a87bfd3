import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import torch, cipkkt
from cipkkt import _lib as L
lib = L.load(); N = 8192
nbytes = C.c_size_t(); L.check(lib.cip_ldlt_workspace_bytes(N, C.byref(nbytes)))
ws = torch.zeros(nbytes.value // 8 + 8, dtype=torch.float64, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(N)
M = torch.randn(N, N, generator=g, dtype=torch.float64, device="cuda")
K0 = (M @ M.t() / N + torch.eye(N, dtype=torch.float64, device="cuda")).contiguous()
for rep in range(4):
    dK = K0.clone(); info = C.c_int(-1)
    L.check(lib.cip_ldlt_factor_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), C.byref(info)))
    torch.cuda.synchronize()
