import ctypes as C, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, ROOT + "/conicip.jl_amd"): sys.path.insert(0, p)
import torch, cipkkt
from cipkkt import _lib as L
lib = L.load()
N, quasi = 4608, 512
nbo = int(sys.argv[1]); reps = int(sys.argv[2])
lib.cip_set_ldlt_outer_block(nbo)
nbytes = C.c_size_t(); L.check(lib.cip_ldlt_workspace_bytes(N, C.byref(nbytes)))
ws = torch.zeros(nbytes.value // 8 + 8, dtype=torch.float64, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(N + quasi + 11)
M = torch.randn(N, N, generator=g, dtype=torch.float64, device="cuda")
K0 = M @ M.t() / N + torch.eye(N, dtype=torch.float64, device="cuda")
K0[N - quasi:, N - quasi:] = 0.0
K0 = K0.contiguous()
def factor(mode):
    prev = lib.cip_set_ldlt_lookahead(mode)
    dK = K0.clone(); info = C.c_int(-1)
    L.check(lib.cip_ldlt_factor_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), C.byref(info)))
    torch.cuda.synchronize(); lib.cip_set_ldlt_lookahead(prev)
    return dK, info.value
ref, i0 = factor(2)
bad = 0; infos = set()
for r in range(reps):
    got, inf = factor(1); infos.add(inf)
    if not torch.equal(torch.tril(got.t()), torch.tril(ref.t())): bad += 1
print("nbo", nbo, "reps", reps, "mismatches", bad, "info values", infos, "ref info", i0)
