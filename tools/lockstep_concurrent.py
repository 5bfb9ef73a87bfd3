"""Experiment: a rank's shard of B problems as G lock-step groups solved CONCURRENTLY from G host threads (each call of
cip_conicip_lockstep has its own stream and arena), against one group of B.   usage: python tools/lockstep_concurrent.py [B] [n] [groups]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conicip.jl_amd"))
import torch
from cipkkt.batch import _solve_problems_native
from cipkkt.workloads import c5_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
prs = c5_batch(count=B, n=n, seed=4000, device=dev)
def run_groups(g):
    parts = [prs[i::g] for i in range(g)]
    out = [None] * g
    def work(i): out[i] = _solve_problems_native(parts[i], dev, 1, "lockstep")
    th = [threading.Thread(target=work, args=(i,)) for i in range(g)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, sum(s.n_factor for o in out for s in o)
for g in (1, G, 1, G, 1, G):
    for rep in range(3):
        dt, nf = run_groups(g)
    print("%d problems as %d concurrent lock-step group(s): %.1f ms per pass, %d factorisations" % (B, g, dt * 1e3, nf), flush=True)
