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


def solve_batch(problems, solve_fn=None, rank=0, world=1, dist=None, device=None):
    """problems: list of dicts(Q, c, A, b, cone_dims, G, d, kwargs).  Each rank solves its
    shard with `solve_fn` (default: the HIP-backed cipkkt.conicIP) and the statistics are
    reduced over ranks:  SUM(iters, n_factor, n_solve, n_optimal, n_problems), MAX(wall).
    Returns (local_solutions, stats_dict)."""
    if solve_fn is None:
        from .driver import conicIP as solve_fn
    mine = shard_indices(len(problems), rank, world)
    sols = {}
    t0 = time.perf_counter()
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
