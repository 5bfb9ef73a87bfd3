#!/usr/bin/env python3
"""what a LARGE second-order cone costs on the device (round-3 review item 3).

One ("Q", n+1) cone, n = 4096 (or argv[1]), behind (a) a dense A and (b) A = [0; I] in CSR (the reference's own "single large
SOC" benchmark shape, benchmark/profile.jl:43-52).  Prints one JSON line per case with the Schur formation split out:

  assemble_q_ms   cip_assemble with the Q cone:  dense A: W = A'F^-1 (two O(mn) passes: the rank-1 form of F^-1 on every row of
                  A') + the m n^2 SYRK;  CSR A: A' diag A (O(nnz)) + one rank-1 column + a rank-16 update of K
  assemble_r_ms   the SAME A with an R cone of the same size (diagonal F): the SYRK / the O(nnz) pass alone
  => q_cone_part_ms = the difference: the Q-cone's own cost, to be read beside the SYRK and the LDL'
  ldlt_ms, solve3x3_ms, the interior-point run (iterations = the oracle's fixture).
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/profile_r4.sh does)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "conicip.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import cipkkt
from cipkkt import workloads as W


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def interior_q(cones, rng):
    xs = []
    for _, k in cones:
        x = rng.standard_normal(k)
        x[0] = np.linalg.norm(x[1:]) * 1.5 + 0.5
        xs.append(x)
    return np.concatenate(xs)


def case(name, prob, route="schur"):
    Q, c, A, b, K = prob
    n, m = Q.shape[0], A.shape[0]
    rng = np.random.default_rng(0)
    dev = lambda x: torch.as_tensor(x, dtype=torch.float64, device="cuda")
    out = dict(case=name, n=n, m=m, route=route)
    for tag, cones in (("q", K), ("r", [("R", m)])):
        ks = cipkkt.KKTSystem(Q, A, None, cones, route=route)
        if tag == "q":
            v, s = dev(interior_q(K, rng)), dev(interior_q(K, rng))
        else:
            v, s = dev(rng.random(m) + 0.5), dev(rng.random(m) + 0.5)
        out["nt_scaling_%s_ms" % tag] = timed(lambda: ks.set_scaling_from_iterate(v, s))
        out["assemble_%s_ms" % tag] = timed(ks.assemble_only)
        if tag == "q":
            ks.set_timing(True)
            for _ in range(3):
                ks.factor()
            torch.cuda.synchronize()
            st = ks.stats()
            out["ldlt_ms"] = st["ms_ldlt"]
            out["kkt_order"] = int(st["N"])
            ks.set_timing(False)
            x, z = rng.standard_normal(n), rng.standard_normal(m)
            xd, zd = dev(x), dev(z)
            a, cc, y0 = torch.zeros_like(xd), torch.zeros_like(zd), torch.zeros(0, dtype=torch.float64, device="cuda")
            ks.factor()
            out["solve3x3_ms"] = timed(lambda: ks.solve3x3_dev(xd, y0, zd, a, y0, cc))
            cipkkt.conicIP(*prob, optTol=1e-6, system=ks, kktsolver=route, maxIters=2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sol = cipkkt.conicIP(*prob, optTol=1e-6, system=ks, kktsolver=route)
            torch.cuda.synchronize()
            out.update(status=sol.status, iters=sol.Iter, n_factor=sol.n_factor, n_solve=sol.n_solve,
                       converge_ms=(time.perf_counter() - t0) * 1e3)
        ks.close()
    out["q_cone_part_ms"] = out["assemble_q_ms"] - out["assemble_r_ms"]
    out["syrk_flops"] = float(m) * n * n if not hasattr(A, "tocsr") else 0.0
    print(json.dumps(out), flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    case("one Q(%d) cone, dense A" % (n + 1), W.soc_large_dense(n, seed=21))
    case("one Q(%d) cone, A = [0; I] CSR (benchmark/profile.jl:43-52 scaled up)" % (n + 1), W.soc_single(n, seed=42))
    case("one Q(%d) cone, dense A, literal 3x3 route" % (n + 1), W.soc_large_dense(n, seed=21), route="full3x3")
    case("reference size: one Q(501) cone, A = [0; I] CSR (benchmark/report.md:57-59)", W.soc_single(500, seed=42))
    case("reference size: 250 x Q(3), sprandn A (benchmark/report.md:60-62)", W.soc_many_small())


if __name__ == "__main__":
    main()
