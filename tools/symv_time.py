import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, ROOT + "/conicip.jl_amd"): sys.path.insert(0, p)
import torch, cipkkt
from cipkkt import _lib as L, workloads as W
for n in (2048, 8192):
    Q, c, A, b, K = W.c2_problem(n, seed=5, device="cuda")
    ks = cipkkt.KKTSystem(Q, A, None, K)
    x = torch.randn(n + 2, dtype=torch.float64, device="cuda"); y = torch.zeros(n, dtype=torch.float64, device="cuda")
    for off, name in ((0, "symv (aligned x)"), (1, "gemv_t (x misaligned by 8 bytes)")):
        xp = x.data_ptr() + 8 * off
        for _ in range(3): L.check(ks.lib.cip_gemv_dev(ks.h, L.MAT_Q, 0, 1.0, xp, 0.0, y.data_ptr()))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): L.check(ks.lib.cip_gemv_dev(ks.h, L.MAT_Q, 0, 1.0, xp, 0.0, y.data_ptr()))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print("n=%d %-34s %.1f us  (full Q bytes / time = %.2f TB/s)" % (n, name, dt * 1e6, n * n * 8 / dt / 1e12))
    ks.close()
