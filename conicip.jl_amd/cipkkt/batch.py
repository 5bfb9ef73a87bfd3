"""Independent-problem batches sharded one problem per GPU (BASELINE config 5).

A single KKT system never leaves its GPU (north-star), so the only multi-GPU structure
is data parallelism over independent problems: problem i -> rank (i mod world).  There
is no data-path collective; one tiny all-reduce (RCCL over xGMI on the GPU box, gloo in
the CPU tests) combines convergence / timing statistics.
"""
import os
import time

import numpy as np
import torch


def release_cached_memory():
    """Give back the device arena the library keeps between lock-step batches (cip_release_cached_memory)."""
    from . import _lib as L
    L.check(L.load().cip_release_cached_memory())


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


def _solve_problems_native(prs, device, in_flight, mode="auto"):
    """All problems of this rank through the library's batch entry points.
    mode "lockstep": `cip_conicip_lockstep` (csrc/lockstep.hip) -- problems of identical shape advance through the loop
    together, every step one launch with the problem index in the grid; "threads": `cip_conicip_problems`
    (csrc/batch.hip) -- `in_flight` host threads inside the library, each re-loading ONE handle on its own HIP stream
    with the next problem of the queue; "auto": `cip_conicip_mixed` -- every group of problems that share a shape (and
    hold no chip-wide S cone) in lock-step, the others through the threads.  CIP_BATCH=threads|lockstep overrides "auto"."""
    import ctypes as C
    from . import _lib as L
    from .driver import solution_from_result
    from .kkt import make_problem
    lib = L.load()
    kw = prs[0].get("kwargs", {})
    opt = L.CipOptions(kw.get("optTol", 1e-6), kw.get("DTB", 0.01), kw.get("infeasTol", -1.0) or -1.0,
                       kw.get("refinementThreshold", -1.0) or -1.0, kw.get("maxRefinementSteps", 3),
                       kw.get("maxIters", 100), 0)
    k = len(prs)
    keep = []
    structs = (L.CipProblem * k)()
    dims = []
    with torch.cuda.device(device):
        for i, pr in enumerate(prs):
            cd = [(str(t), int(kk)) for t, kk in pr["cone_dims"]]
            st, kp, _ = make_problem(pr["Q"], pr["A"], pr.get("G"), cd, kw.get("kktsolver", "schur"), device)
            structs[i] = st
            keep.append(kp)
            dims.append((st.n, st.m, st.p))
        torch.cuda.current_stream(device).synchronize()     # the staging transposes ran on torch's stream
        vp = C.c_void_p * k

        def host(key, i, size):
            x = prs[i].get(key)
            a = np.zeros(max(size, 1)) if x is None else np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
            keep.append(a)
            return a

        cs = [host("c", i, dims[i][0]) for i in range(k)]
        bs = [host("b", i, dims[i][1]) for i in range(k)]
        ds = [host("d", i, dims[i][2]) for i in range(k)]
        ys = [np.zeros(max(dims[i][0], 1)) for i in range(k)]
        ws = [np.zeros(max(dims[i][2], 1)) for i in range(k)]
        vs = [np.zeros(max(dims[i][1], 1)) for i in range(k)]
        arr = lambda xs: vp(*[x.ctypes.data for x in xs])
        res = (L.CipResult * k)()
        if mode == "auto":
            mode = os.environ.get("CIP_BATCH", "auto")
        if mode == "lockstep":
            L.check(lib.cip_conicip_lockstep(k, structs, arr(cs), arr(bs), arr(ds), C.byref(opt), arr(ys), arr(ws), arr(vs), res))
        elif mode == "threads":
            L.check(lib.cip_conicip_problems(k, structs, arr(cs), arr(bs), arr(ds), C.byref(opt), arr(ys), arr(ws), arr(vs),
                                             res, int(in_flight)))
        else:
            L.check(lib.cip_conicip_mixed(k, structs, arr(cs), arr(bs), arr(ds), C.byref(opt), arr(ys), arr(ws), arr(vs),
                                          res, int(in_flight)))
    return [solution_from_result(res[i], ys[i][:dims[i][0]], ws[i][:dims[i][2]], vs[i][:dims[i][1]]) for i in range(k)]


def solve_batch(problems, solve_fn=None, rank=0, world=1, dist=None, device=None, concurrency=1, native=False,
                reduce_device=None):
    """problems: list of dicts(Q, c, A, b, cone_dims, G, d, kwargs).  Each rank solves its
    shard with `solve_fn` (default: the HIP-backed cipkkt.conicIP) and the statistics are
    reduced over ranks:  SUM(iters, n_factor, n_solve, n_optimal, n_problems), MAX(wall).
    `concurrency` > 1 (default solver only) keeps that many problems in flight on separate
    HIP streams, one Python thread each: a thread builds its problem's handle, runs the native loop
    (`cip_conicip`, GIL released) and frees it, so level-1 setup of one problem overlaps the solve of another
    (n = 2048, 8 problems: 583 KKT solves/s one at a time, 899 with 2 in flight).  `native=True` hands the problems to
    the library's batch entry point `cip_conicip_problems` (host threads inside the library, one re-loaded handle per
    thread: what a C caller uses); `native="handles"` builds all handles first and calls `cip_conicip_many`
    (605-683 KKT solves/s on the same batch: the setup is not overlapped).
    Returns (local_solutions, stats_dict)."""
    default_solver = solve_fn is None
    if solve_fn is None:
        from .driver import conicIP as solve_fn
    mine = shard_indices(len(problems), rank, world)
    sols = {}
    t0 = time.perf_counter()
    same_opts = all(problems[i].get("kwargs", {}) == problems[mine[0]].get("kwargs", {}) for i in mine) if mine else True
    if default_solver and native and len(mine) > 0 and same_opts:
        dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        shard = [problems[i] for i in mine]
        if native == "handles":
            out = _solve_many_native(shard, dev, max(1, concurrency))
        else:
            out = _solve_problems_native(shard, dev, max(1, concurrency), native if native in ("lockstep", "threads") else "auto")
        for i, sol in zip(mine, out):
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
    if dist is not None:                 # also with one rank (bench.py's CIP_BENCH_FORCE_DIST exercises RCCL on one GPU)
        # the statistics travel as tensors on `reduce_device` (default: the compute device -- RCCL; "cpu" under gloo)
        tdev = reduce_device if reduce_device is not None else (device if device is not None else "cpu")
        ts = torch.as_tensor(sums, device=tdev)
        tm = torch.as_tensor(mx, device=tdev)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        sums, mx = ts.cpu().numpy(), tm.cpu().numpy()
    stats = dict(iters=int(sums[0]), n_factor=int(sums[1]), n_solve=int(sums[2]), n_optimal=int(sums[3]),
                 n_problems=int(sums[4]), wall_s=float(mx[0]))
    return sols, stats


def run_config5(rank, world, dist, device, steps, warmup, problems=None, count=64, n=2048, seed=4000, in_flight=4,
                solve_fn=None, barrier=None, reduce_device=None):
    """BASELINE config 5 as a timed job (bench.py --gpus N, N > 1; `--workload c5` on one GPU): `count` independent
    problems, problem i -> rank i mod world, every rank's shard resident in HBM before the timed region; a step is one
    pass over the whole batch (each rank its shard, `in_flight` problems at once through cip_conicip_problems).
    Returns (stats of one pass reduced over ranks, elapsed seconds for `steps` passes = MAX over ranks).
    `problems` / `solve_fn` are injectable (the gloo test runs the sharding and the reduction on CPU)."""
    if problems is None:
        from .workloads import c5_batch
        mine = shard_indices(count, rank, world)
        local = c5_batch(count, n, seed, device=device, indices=mine)
        problems = [None] * count                      # only this rank's shard is materialised
        for i, pr in zip(mine, local):
            problems[i] = pr
    sync = (lambda: torch.cuda.synchronize(device)) if device is not None and str(device) != "cpu" else (lambda: None)
    if barrier is None:
        barrier = (lambda: dist.barrier()) if dist is not None else (lambda: None)

    def one_pass(reduce):
        return solve_batch(problems, solve_fn=solve_fn, rank=rank, world=world, dist=dist if reduce else None,
                           device=device, concurrency=in_flight, native=solve_fn is None, reduce_device=reduce_device)

    for _ in range(warmup):
        one_pass(False)
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass(False)
    sync()
    busy = time.perf_counter() - t0                    # this rank's own time for its shard (before it waits for the others)
    barrier(); sync()
    elapsed = time.perf_counter() - t0
    _, stats = one_pass(True)                          # untimed: the reduced statistics of one pass
    if solve_fn is None:
        release_cached_memory()                        # the lock-step arena (GBs) is not kept beyond the job
    busy_min = busy_max = busy
    if dist is not None:
        dev = reduce_device if reduce_device is not None else (device if device is not None else "cpu")
        t = torch.tensor([elapsed, busy, -busy], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, busy_max, busy_min = float(t[0].item()), float(t[1].item()), -float(t[2].item())
    # load imbalance over the ranks (problems need different iteration counts): per-pass busy time of the fastest / slowest rank
    stats = dict(stats, rank_busy_ms_min=1e3 * busy_min / max(steps, 1), rank_busy_ms_max=1e3 * busy_max / max(steps, 1))
    return stats, elapsed
