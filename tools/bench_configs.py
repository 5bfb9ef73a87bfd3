#!/usr/bin/env python3
"""Secondary measurements (not the bench.py headline): BASELINE.json configs 3, 4, 5 on one GPU.
Prints one JSON line per config: wall-clock to converge, iterations, factorisations/solves, KKT solves/s."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "conicip.jl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import scipy.sparse as sp
import torch

import cipkkt
from oracle import cones as oc
from test_gpu_configs import socp_problem


def run(name, prob, **kw):
    Q, c, A, b, K, G, d = prob
    t0 = time.perf_counter()
    ks = cipkkt.KKTSystem(Q, A, G, K)                # level 1: upload + workspaces (once per problem)
    torch.cuda.synchronize()
    t_l1 = time.perf_counter() - t0
    cipkkt.conicIP(*prob, maxIters=2, system=ks, **kw)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sol = cipkkt.conicIP(*prob, system=ks, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(config=name, status=sol.status, iters=sol.Iter, n_factor=sol.n_factor, n_solve=sol.n_solve,
                          level1_s=t_l1, wall_s=dt, kkt_solves_per_s=sol.n_factor / dt)), flush=True)
    ks.close()


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ''
    # C3: SOCP n=4096, 512 x Q(8), p=512 (p is not fixed by BASELINE.json: stated choice)
    run("C3 SOCP n=4096 512xQ(8) p=512 dense A", socp_problem(4096, 512, 8, 512, 11), optTol=1e-6)
    if only == "c3":
        return
    if only == "c5":
        pass
    # C4: single S cone.  ("S",256) is not a legal cone spec (256 is not triangular, src/ConicIP.jl:85);
    # measured here at matrix order r=64 (k=2080) and r=128 (k=8256), n=256, p=16; the literal size (r=256, n=1024):
    # tools/c4_time.py
    for r in (64, 128):
        rng = np.random.default_rng(5)
        n, p = 256, 16
        k = r * (r + 1) // 2
        A = rng.standard_normal((k, n)) / np.sqrt(n)
        prob = (np.eye(n), rng.standard_normal(n), A, -oc.vecm(np.eye(r)), [("S", k)], rng.standard_normal((p, n)), np.zeros(p))
        run("C4 SDP r=%d (k=%d) n=%d p=%d" % (r, k, n, p), prob, optTol=1e-6)
    # C5: batch of independent dense QPs n=2048 (8 of the 64 here; one GPU)
    from cipkkt.batch import solve_batch
    probs = []
    n = 2048
    for i in range(8):
        g = torch.Generator(device="cuda"); g.manual_seed(900 + i)
        M = torch.randn(n, n, generator=g, dtype=torch.float64, device="cuda")
        Q = (M.t() @ M / n).cpu().numpy()
        probs.append(dict(Q=Q, c=np.random.default_rng(i).standard_normal(n), A=sp.identity(n, format="csr"),
                          b=np.zeros(n), cone_dims=[("R", n)], kwargs=dict(optTol=1e-6)))
    solve_batch(probs[:1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sols, st = solve_batch(probs)
    dt = time.perf_counter() - t0
    st["wall_s"] = dt
    print(json.dumps(dict(config="C5 batch 8 x dense QP n=2048 (sequential on 1 GPU)", **st,
                          kkt_solves_per_s=st["n_factor"] / dt)), flush=True)
    for conc in (2, 4, 8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sols, st = solve_batch(probs, concurrency=conc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st["wall_s"] = dt
        print(json.dumps(dict(config="C5 batch 8 x dense QP n=2048, %d in flight (cip_conicip_many)" % conc, **st,
                              kkt_solves_per_s=st["n_factor"] / dt)), flush=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sols, st = solve_batch(probs, concurrency=conc, native=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st["wall_s"] = dt
        print(json.dumps(dict(config="C5 batch 8 x dense QP n=2048, %d in flight (Python threads)" % conc, **st,
                              kkt_solves_per_s=st["n_factor"] / dt)), flush=True)


if __name__ == "__main__":
    main()
