"""Independent-problem batches sharded one problem per GPU (BASELINE config 5).

A single KKT system never leaves its GPU (north-star), so the only multi-GPU structure
is data parallelism over independent problems: problem i -> rank (i mod world).  There
is no data-path collective; one tiny all-reduce (RCCL over xGMI on the GPU box, gloo in
the CPU tests) combines convergence / timing statistics.
"""
import time

import numpy as np
import torch


def shard_indices(n_problems, rank, world):
    """Static round-robin assignment: problem i -> rank i % world."""
    return list(range(rank, n_problems, world))


def _solve_on_stream(pr, device):
    """One problem on its own HIP stream: small systems (n ~ 2048) are latency-bound -- a factorisation is a
    chain of tiny dependent launches -- so several of them in flight on different streams fill the chip."""
    from .driver import conicIP
    from .kkt import KKTSystem
    st = torch.cuda.Stream(device=device)
    with torch.cuda.stream(st):
        ks = KKTSystem(pr["Q"], pr["A"], pr.get("G"), pr["cone_dims"], device=device)
        ks.set_stream(st.cuda_stream)
        try:
            sol = conicIP(pr["Q"], pr["c"], pr["A"], pr["b"], pr["cone_dims"], pr.get("G"), pr.get("d"), system=ks,
                          **pr.get("kwargs", {}))
        finally:
            st.synchronize()
            ks.close()
    return sol


def solve_batch(problems, solve_fn=None, rank=0, world=1, dist=None, device=None, concurrency=1):
    """problems: list of dicts(Q, c, A, b, cone_dims, G, d, kwargs).  Each rank solves its
    shard with `solve_fn` (default: the HIP-backed cipkkt.conicIP) and the statistics are
    reduced over ranks:  SUM(iters, n_factor, n_solve, n_optimal, n_problems), MAX(wall).
    `concurrency` > 1 (default solver only) keeps that many problems in flight on separate
    HIP streams, one host thread each (ctypes releases the GIL during library calls).
    Returns (local_solutions, stats_dict)."""
    default_solver = solve_fn is None
    if solve_fn is None:
        from .driver import conicIP as solve_fn
    mine = shard_indices(len(problems), rank, world)
    sols = {}
    t0 = time.perf_counter()
    if default_solver and concurrency > 1 and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor
        dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        with ThreadPoolExecutor(max_workers=concurrency) as ex:
            futs = {i: ex.submit(_solve_on_stream, problems[i], dev) for i in mine}
            for i, f in futs.items():
                sols[i] = f.result()
    else:
        for i in mine:
            pr = problems[i]
            sols[i] = solve_fn(pr["Q"], pr["c"], pr["A"], pr["b"], pr["cone_dims"], pr.get("G"), pr.get("d"),
                               **pr.get("kwargs", {}))
    wall = time.perf_counter() - t0
    sums = np.array([sum(s.Iter for s in sols.values()),
                     sum(getattr(s, "n_factor", 0) for s in sols.values()),
                     sum(getattr(s, "n_solve", 0) for s in sols.values()),
                     sum(1 for s in sols.values() if s.status == "Optimal"),
                     len(sols)], dtype=np.float64)
    mx = np.array([wall], dtype=np.float64)
    if dist is not None and world > 1:
        tdev = device if device is not None else "cpu"
        ts = torch.as_tensor(sums, device=tdev)
        tm = torch.as_tensor(mx, device=tdev)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        sums, mx = ts.cpu().numpy(), tm.cpu().numpy()
    stats = dict(iters=int(sums[0]), n_factor=int(sums[1]), n_solve=int(sums[2]), n_optimal=int(sums[3]),
                 n_problems=int(sums[4]), wall_s=float(mx[0]))
    return sols, stats
