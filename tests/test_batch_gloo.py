"""Multi-rank path (problem-per-GPU sharding + stats all-reduce) on CPU: world_size 2,
gloo backend.  The per-problem solver is injected (the oracle here, since the HIP path
needs a GPU); what is under test is the sharding and the reduction."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "conicip.jl_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import problems as P
    from cipkkt.batch import solve_batch
    from oracle.conicip import conicIP as oracle_conicIP
    probs = []
    for i in range(5):
        Q, c, A, b, K, G, d, _ = P.random_mixed(n=20, nq=2, kq=4, p=2, seed=100 + i)
        probs.append(dict(Q=Q, c=c, A=A, b=b, cone_dims=K, G=G, d=d, kwargs=dict(optTol=1e-7)))
    sols, stats = solve_batch(probs, solve_fn=oracle_conicIP, rank=rank, world=world, dist=dist)
    q.put((rank, sorted(sols.keys()), stats, {i: s.Iter for i, s in sols.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]          # problem i -> rank i % 2
    assert res[0][2] == res[1][2]                                  # every rank sees the reduced stats
    st = res[0][2]
    iters = {**res[0][3], **res[1][3]}
    assert st["n_problems"] == 5 and st["n_optimal"] == 5
    assert st["iters"] == sum(iters.values())
    assert st["wall_s"] > 0


def _worker_c5(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "conicip.jl_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cipkkt import workloads as W
    from cipkkt.batch import run_config5, shard_indices
    from oracle.conicip import conicIP as oracle_conicIP
    # config 5's shard map at its full count (64 problems, seeds 4000 + i) with tiny systems and the oracle injected
    # as the per-problem solver: what runs here is bench.py's multi-GPU code path minus the GPU.
    probs = W.c5_batch(64, 12, seed=4000)
    seen = []

    def solver(Q, c, A, b, K, G=None, d=None, **kw):
        seen.append(float(c[0]))
        return oracle_conicIP(Q, c, A, b, K, G, d, **kw)

    stats, elapsed = run_config5(rank, world, dist, None, steps=1, warmup=0, problems=probs, solve_fn=solver, in_flight=1)
    mine = shard_indices(64, rank, world)
    per_pass = seen[:len(mine)]
    q.put((rank, stats, elapsed, per_pass == [float(probs[i]["c"][0]) for i in mine], len(seen)))
    dist.barrier()
    dist.destroy_process_group()


def test_config5_shard_map_and_reduction_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c5, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, st0, el0, ok0, n0), (r1, st1, el1, ok1, n1) = res
    assert ok0 and ok1                          # rank r solved exactly problems r, r+2, ... in order
    assert n0 == n1 == 64                       # one timed pass + the untimed pass that collects the statistics, 32 each
    assert st0 == st1 and st0["n_problems"] == 64 and st0["n_optimal"] == 64
    assert st0["n_factor"] == st0["iters"] + 64           # one factorisation per iteration + the initial point
    assert el0 == el1 > 0                       # MAX over ranks, seen by both


def test_shard_indices():
    sys.path.insert(0, os.path.join(ROOT, "conicip.jl_amd"))
    from cipkkt.batch import shard_indices
    assert shard_indices(64, 3, 8) == list(range(3, 64, 8))
    allv = sorted(sum((shard_indices(10, r, 4) for r in range(4)), []))
    assert allv == list(range(10))
