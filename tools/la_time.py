"""Factorisation time at n = 8192 through the KKT handle under the serial and the look-ahead schedule + worker stats.
Usage: python tools/la_time.py [n] [modes e.g. 01]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/conicip.jl_amd'): sys.path.insert(0, p)
import torch, cipkkt
from cipkkt import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
modes = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "01")]
Q, c, A, b, K = W.c2_problem(n, seed=1234, device="cuda")
ks = cipkkt.KKTSystem(Q, A, None, K)
g = torch.Generator(device="cuda"); g.manual_seed(3)
v = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
s = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
lam = torch.zeros(n, dtype=torch.float64, device="cuda")
for mode in modes:
    ks.lib.cip_set_ldlt_lookahead(mode)
    ts = []
    for rep in range(6):
        ks.set_scaling_from_iterate(v, s, lam)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ks.factor()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ks.check_factor()
    msg = "mode %d: assemble+factor ms %s" % (mode, " ".join("%.3f" % (t * 1e3) for t in ts))
    if mode == 1:
        st = ks.profile_lookahead()
        per_tile_us = st["busy_ticks"] / max(st["tiles"], 1) / 2200.0      # s_memtime ticks are shader cycles (~2.2 GHz under load)
        flops = sum((n - 512 * (J + 1)) * (n - 512 * (J + 1) + 1) * 512.0 for J in range(n // 512 - 1))
        busy_s = st["busy_ticks"] / 2.2e9 / max(st["workers"], 1)
        msg += " | workers %d tiles %d avg tile %.1f us, busy/worker %.3f ms -> %.1f TFLOP/s while busy" % (
            st["workers"], st["tiles"], per_tile_us, busy_s * 1e3, flops / busy_s / 1e12)
    print(msg, flush=True)
