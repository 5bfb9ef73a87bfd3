"""Phase stamps of the LAST k_ldlt_panel<true> launch of a factorisation (library built with -DPANEL_TIMING:
tools/build_variant.sh ptim diag.hip -DPANEL_TIMING; CIPKKT_LIB=...).  Times in us relative to workgroup 0's entry.
usage: CIPKKT_LIB=conicip.jl_amd/build/variants/libcipkkt_ptim.so python tools/panel_stamps.py [n] [stop_after_panels]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "conicip.jl_amd"))
import cipkkt
from cipkkt import workloads as W, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Q, c, A, b, K = W.c2_problem(n, seed=1234, device="cuda")
ks = cipkkt.KKTSystem(Q, A, None, K)
v = torch.ones(n, dtype=torch.float64, device="cuda"); s = torch.ones(n, dtype=torch.float64, device="cuda")
ks.set_scaling_from_iterate(v, s)
lib = _lib.load()
if len(sys.argv) > 2:
    lib.cip_debug_panel_col(int(sys.argv[2]))          # stamp the launch of the panel at this column only
names = {0: "wg0 entry", 4: "ready seen", 2: "block in LDS", 6: "A(0) done", 7: "last pivot", 8: "wg0 end", 9: "producer wg1 entry",
         10: "producer (0,0): operands in", 11: "its MFMAs done", 28: "its stores landed", 12: "strip 0 update tiles done", 13: "strip 0 TRSM done",
         14: "last strip TRSM done", 15: "worker 0 starts", 16: "worker 0 dry", 17: "strip 0 helper starts", 18: "strip 0 helper dry", 19: "last worker dry"}
for rep in range(3):
    ks.factor(); torch.cuda.synchronize()
    t = (ctypes.c_long * 32)()
    assert lib.cip_debug_panel_stamps(t) == 0
    t0 = t[0]
    print("rep", rep, " | ".join("%s %.2f" % (names[i], (t[i] - t0) / 100.0) for i in (0, 9, 10, 11, 28, 4, 2, 6, 7, 8, 12, 13, 14, 15, 16, 17, 18, 19) if t[i]))
    if t[20]:
        print("      serial wave, A(1..7) done at", " ".join("%.2f" % ((t[20 + k] - t0) / 100.0) for k in range(7)))
