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


def _solve_many_native(prs, device, in_flight):
    """All problems of this rank through `cip_conicip_many` (csrc/batch.hip): one handle + HIP stream per problem,
    `in_flight` native interior-point loops at once on the library's own host threads (no Python in the loop)."""
    import ctypes as C
    from . import _lib as L
    from .driver import solution_from_result
    from .kkt import KKTSystem
    lib = L.load()
    kw = prs[0].get("kwargs", {})
    opt = L.CipOptions(kw.get("optTol", 1e-6), kw.get("DTB", 0.01), kw.get("infeasTol", -1.0) or -1.0,
                       kw.get("refinementThreshold", -1.0) or -1.0, kw.get("maxRefinementSteps", 3),
                       kw.get("maxIters", 100), 0)
    systems, streams, keep = [], [], []
    try:
        for pr in prs:
            st = torch.cuda.Stream(device=device)
            with torch.cuda.stream(st):
                ks = KKTSystem(pr["Q"], pr["A"], pr.get("G"), pr["cone_dims"], device=device)
            ks.set_stream(st.cuda_stream)
            systems.append(ks)
            streams.append(st)
        k = len(prs)
        vp = C.c_void_p * k

        def host(key, i, size):
            x = prs[i].get(key)
            a = np.zeros(max(size, 1)) if x is None else np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
            keep.append(a)
            return a

        cs = [host("c", i, ks.n) for i, ks in enumerate(systems)]
        bs = [host("b", i, ks.m) for i, ks in enumerate(systems)]
        ds = [host("d", i, ks.p) for i, ks in enumerate(systems)]
        ys = [np.zeros(max(ks.n, 1)) for ks in systems]
        ws = [np.zeros(max(ks.p, 1)) for ks in systems]
        vs = [np.zeros(max(ks.m, 1)) for ks in systems]
        arr = lambda xs: vp(*[x.ctypes.data for x in xs])
        res = (L.CipResult * k)()
        handles = vp(*[ks.h.value for ks in systems])
        with torch.cuda.device(device):
            L.check(lib.cip_conicip_many(handles, k, arr(cs), arr(bs), arr(ds), C.byref(opt), arr(ys), arr(ws), arr(vs),
                                         res, int(in_flight)))
        return [solution_from_result(res[i], ys[i][:ks.n], ws[i][:ks.p], vs[i][:ks.m]) for i, ks in enumerate(systems)]
    finally:
        for st in streams:
            st.synchronize()
        for ks in systems:
            ks.close()


def solve_batch(problems, solve_fn=None, rank=0, world=1, dist=None, device=None, concurrency=1, native=False):
    """problems: list of dicts(Q, c, A, b, cone_dims, G, d, kwargs).  Each rank solves its
    shard with `solve_fn` (default: the HIP-backed cipkkt.conicIP) and the statistics are
    reduced over ranks:  SUM(iters, n_factor, n_solve, n_optimal, n_problems), MAX(wall).
    `concurrency` > 1 (default solver only) keeps that many problems in flight on separate
    HIP streams, one Python thread each: a thread builds its problem's handle, runs the native loop
    (`cip_conicip`, GIL released) and frees it, so level-1 setup of one problem overlaps the solve of another
    (n = 2048, 8 problems: 583 KKT solves/s one at a time, 899 with 2 in flight).  `native=True` instead builds all
    handles first and hands them to the library's batch entry point (`cip_conicip_many`: host threads inside the
    library, what a C caller uses): 605-683 KKT solves/s on the same batch, the setup is not overlapped.
    Returns (local_solutions, stats_dict)."""
    default_solver = solve_fn is None
    if solve_fn is None:
        from .driver import conicIP as solve_fn
    mine = shard_indices(len(problems), rank, world)
    sols = {}
    t0 = time.perf_counter()
    same_opts = all(problems[i].get("kwargs", {}) == problems[mine[0]].get("kwargs", {}) for i in mine) if mine else True
    if default_solver and native and concurrency > 1 and len(mine) > 1 and same_opts:
        dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        for i, sol in zip(mine, _solve_many_native([problems[i] for i in mine], dev, concurrency)):
            sols[i] = sol
    elif default_solver and concurrency > 1 and len(mine) > 1:
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
