#!/usr/bin/env python3
"""bench.py -- KKT solves/sec (+ wall-clock to converge) of the MI355X-native KKT path.

Workload (BASELINE.json configs[1]): dense random QP, n = m = 8192, p = 0,
K = [("R", 8192)], Q = M'M/n (M iid N(0,1)), c ~ N(0,1), A = I (sparse), b = 0, fp64,
optTol = 1e-6 -- the README box-QP form at the headline size.

A "step" = one KKT solve = one Newton system of the interior-point loop:
    NT scaling from the iterate (device)  +  KKT assembly  +  dense LDL' factorisation
    +  SOLVES_PER_FACTOR back-solves (cip_solve4x4: cone division, 3x3 solve, ds recovery)
with SOLVES_PER_FACTOR = the rounded mean the real solve used (predictor, corrector,
refinement).  The scaling comes from a mid-trajectory iterate of the same problem; all
inputs are resident in HBM before the timed region.

`--gpus N` with N > 1: run under `python -m torch.distributed.run --nproc-per-node N ...` (the driver's way: WORLD_SIZE
must then equal N) or plainly as `python bench.py --gpus N` -- bench.py then starts that launcher itself as a CHILD
process before anything touches the GPU and exits with its return code; fewer than N visible GPUs is an error.
N > 1 (one rank per GPU over RCCL) measures the north-star's
multi-GPU configuration, BASELINE.json configs[4]: 64 independent dense QPs with n = 2048
(seeds 4000 + i), problem i -> rank i mod N, each rank running its shard through the library's
batch entry point -- in LOCK-STEP (cip_conicip_lockstep: the rank's problems advance through the loop
together, every step one launch with the problem index in the grid; `--batch-mode threads` selects the
thread pool of cip_conicip_problems instead); no data-path collective, one all-reduce of the
counts (SUM) and of the wall time (MAX).  A step is then one pass over the whole batch and
`value` = KKT solves (factorisations) of all ranks per second; total work is fixed as N grows
(`"scaling": "strong"`).  `--workload c5` runs that workload on one GPU; the default N = 1 line
(the n = 8192 headline) carries the same figure as `c5_single_gpu`, and every N > 1 line carries its own
`c5_single_gpu` (rank 0 alone on all 64 problems, timed before the sharded passes), `ranks_seen`, a `roofline` of the
batch's trailing-update launches and a bounded `cpu_baseline`, so that a 1 -> N curve can be read off the N > 1 lines.

Inputs come from the portable SplitMix64 generator (cipkkt/workloads.py), generated in HBM.

Output: ONE JSON line on rank 0 with `roofline` (LDL' trailing-update kernel, fp64 MFMA,
HIP-event timed per launch on the launch stream) and `cpu_baseline` (the oracle's
reference-faithful kktsolver_qr restatement timed on the host cores).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "conicip.jl_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

TRAILING_KERNELS = ("k_ldlt_trailing_64",)
FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X vendor fp64 matrix peak (dense); see DESIGN.md §5
# what the chip can issue at the clock it holds under this kernel: 256 CUs x 4 SIMDs x 2048 flop / 64 cycles x 2.12 GHz
# (GRBM_GUI_ACTIVE / 8 / duration of profiles/r3/final_pmc_mfma.csv: 2.117 GHz, 2.17 in round 2; tools/mfma_peak.hip measures 77.6 at 2.37 GHz unloaded)
FP64_MFMA_CLOCK_LIMITED_TFLOPS = 256 * 4 * 32 * 2.12e9 / 1e12
HBM_PEAK_GBS = 8000.0
PMC_ROUNDS = ("r6", "r5", "r4", "r3", "r2", "r1")   # newest first: the replay fallback of roofline.traffic uses the latest committed passes


def pmc_dir():
    for r in PMC_ROUNDS:
        d = os.path.join(ROOT, "profiles", r)
        if os.path.exists(os.path.join(d, "final_pmc_fetch.csv")) and os.path.exists(os.path.join(d, "final_pmc_write.csv")):
            return d
    return None


def launch_ranks(n_gpus):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process (never an
    exec, and before this process has touched the GPU) and return its exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()            # counts devices without initialising the runtime
    if have < n_gpus and not (os.environ.get("CIP_BENCH_SHARE_GPU") and have >= 1):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this box" % (n_gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(Q_host, n, solves_per_factor, budget_s=100.0):
    """CPU leg, timed on the host cores in this run (reported baseline, not the target).

    (i) reference-faithful: the oracle's restatement of kktsolver_qr (src/kktsolvers.jl:18-58: dense F^-T,
        Atil = F^-T A, Q + Atil'Atil, QR, null-space solve) -- measured at the full n when the n/2 run
        predicts it fits `budget_s`, otherwise at n/2, scaled by (n/n_s)^3
        (`sample` says which);
    (ii) strong CPU: the same Schur + Cholesky route the GPU takes (src/kktsolvers.jl:281-338 with LAPACK potrf / potrs in
        place of UMFPACK), ALWAYS measured at the full n.  Only LAPACK is inside its timed regions: S is formed before
        (diagonal added in place), potrf and the potrs pair are timed separately (best of two), the thread count is the
        fastest of a potrf probe at n = 4096, and the rates are printed beside the host's measured dgemm rate so a reader
        can see whether LAPACK was starved (round-3 review)."""
    from oracle.block import Block, Diagonal
    from oracle.kktsolvers import kktsolver_qr
    import scipy.linalg as sla
    logical = os.cpu_count() or 1
    cores = cores_potrf = logical
    dgemm_gflops = potrf_probe_gflops = None
    probe = {}
    limiter = None
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    if threadpool_limits is not None:
        # OpenBLAS does not scale to every hardware thread of a 2-socket box: probe dgemm (the faithful leg is GEMM + QR) and
        # potrf (the strong leg) at a few thread counts and run each leg at its fastest one.
        rngp = np.random.default_rng(0)
        npb = 4096 if n >= 4096 else max(256, n)
        Mp = rngp.standard_normal((npb, npb))
        Sp = np.asfortranarray(Mp @ Mp.T / npb + np.eye(npb))      # column-major, as LAPACK wants it (a C-ordered input is transposed first)
        best_g, best_p = (1e30, cores), (1e30, cores)
        for nt in sorted({c for c in (8, 16, 32, 64, 128, logical) if c <= logical}):
            with threadpool_limits(limits=nt):
                Mp[:512] @ Mp                      # (thread pool warm-up)
                dg = dp = 1e30
                for _ in range(2):                 # best of two
                    t0 = time.perf_counter()
                    Mp @ Mp
                    dg = min(dg, time.perf_counter() - t0)
                    Sc = Sp.copy(order="F")
                    t0 = time.perf_counter()
                    sla.cho_factor(Sc, lower=True, overwrite_a=True, check_finite=False)
                    dp = min(dp, time.perf_counter() - t0)
            probe[nt] = dict(dgemm_gflops=2.0 * npb ** 3 / dg / 1e9, potrf_gflops=npb ** 3 / 3.0 / dp / 1e9)
            if dg < best_g[0]:
                best_g = (dg, nt)
            if dp < best_p[0]:
                best_p = (dp, nt)
        cores, cores_potrf = best_g[1], best_p[1]
        dgemm_gflops = 2.0 * npb ** 3 / best_g[0] / 1e9
        potrf_probe_gflops = npb ** 3 / 3.0 / best_p[0] / 1e9
        del Mp, Sp
        limiter = threadpool_limits(limits=cores)
    rng = np.random.default_rng(1)

    def scaling(nn):
        return Block([Diagonal(np.exp(np.random.default_rng(1).standard_normal(nn)))])

    def faithful(nn):
        Qs = np.ascontiguousarray(Q_host[:nn, :nn])
        F = scaling(nn)
        t0 = time.perf_counter()
        gen = kktsolver_qr(Qs, np.eye(nn), np.zeros((0, nn)), [("R", nn)])
        t1 = time.perf_counter()
        s3 = gen(F, None)
        t2 = time.perf_counter()
        for _ in range(solves_per_factor):
            s3(rng.standard_normal(nn), np.zeros(0), rng.standard_normal(nn))
        t3 = time.perf_counter()
        return dict(level1=t1 - t0, factor=t2 - t1, solves=t3 - t2)

    def strong(nn, variant="potrf"):
        """variant "potrf": LAPACK dpotrf + dpotrs; "blocked-syrk" / "blocked-gemm": the right-looking blocked Cholesky of
        oracle/blocked_chol.py (block 512: dpotrf on the diagonal block, dtrsm, the trailing update as one dsyrk / as dgemms per
        block column) + two dtrsm per solve -- the flops sit in level-3 BLAS, which threads where potrf's panels do not"""
        from oracle.blocked_chol import blocked_cholesky, cholesky_solve
        F = scaling(nn)
        dinv = 1.0 / F.Blocks[0].diag ** 2
        idx = np.arange(nn)
        best = None
        for _ in range(2):
            S = np.array(Q_host[:nn, :nn], order="F")          # formed OUTSIDE the timed regions
            S[idx, idx] += dinv
            rhs = [rng.standard_normal(nn) for _ in range(solves_per_factor)]
            t0 = time.perf_counter()
            if variant == "potrf":
                cf = sla.cho_factor(S, lower=True, overwrite_a=True, check_finite=False)
            else:
                if blocked_cholesky(S, 512, "syrk" if variant == "blocked-syrk" else "gemm"):
                    raise RuntimeError("blocked Cholesky: not positive definite")
            t1 = time.perf_counter()
            for r in rhs:
                if variant == "potrf":
                    sla.cho_solve(cf, r, overwrite_b=True, check_finite=False)
                else:
                    cholesky_solve(S, r)
            t2 = time.perf_counter()
            cur = (t1 - t0, t2 - t1)
            if best is None or sum(cur) < sum(best):
                best = cur
            del S
        return best

    # ladder: n/2 first (an eighth of the work); the full size only if eight times that fits the budget
    nn = n
    t = None
    if n > 2048:
        half = faithful(n // 2)
        if (half["factor"] + half["solves"]) * 8.0 > budget_s:
            nn, t = n // 2, half
    if t is None:
        t = faithful(nn)
    step_s = (t["factor"] + t["solves"]) * (n / nn) ** 3
    # flops of the faithful factorisation with m = n, p = 0 (src/kktsolvers.jl:32-35 as restated): F^-T A 2n^3, Atil'Atil 2n^3,
    # Q2'(.)Q2 with the dense Q2 = I 4n^3, QR 4/3 n^3 + forming its Q 4/3 n^3
    faithful_gflops = (32.0 / 3.0) * nn ** 3 / t["factor"] / 1e9
    if limiter is not None:
        limiter.restore_original_limits()
        limiter = threadpool_limits(limits=cores_potrf)
    t_potrf, t_potrs = strong(n)
    variants = {"potrf": dict(threads=cores_potrf, factor_s=t_potrf, solves_s=t_potrs, gflops=n ** 3 / 3.0 / t_potrf / 1e9)}
    # round-4 review: potrf reached 41 % of the host's dgemm rate.  The blocked factorisation runs at the dgemm probe's thread
    # count (its flops are dsyrk / dgemm) and the fastest of the three is the strong leg.
    if limiter is not None:
        limiter.restore_original_limits()
        limiter = threadpool_limits(limits=cores)
    for var in ("blocked-syrk", "blocked-gemm"):
        try:
            tf_, ts_ = strong(n, var)
            variants[var] = dict(threads=cores, factor_s=tf_, solves_s=ts_, gflops=n ** 3 / 3.0 / tf_ / 1e9)
        except Exception as e:                      # (scipy without the cython_blas capsules: keep potrf)
            variants[var] = dict(error=repr(e))
    strong_variant = min((k for k in variants if "factor_s" in variants[k]), key=lambda k: variants[k]["factor_s"] + variants[k]["solves_s"])
    t_potrf, t_potrs = variants[strong_variant]["factor_s"], variants[strong_variant]["solves_s"]
    cores_potrf = variants[strong_variant]["threads"]
    strong_s = t_potrf + t_potrs
    sample = ("1 factorisation + %d solves of the kktsolver_qr restatement at n=%d%s (level-1 setup %.2fs excluded); "
              "strong variant measured at n=%d"
              % (solves_per_factor, nn, " (measured at full size)" if nn == n else " scaled to n=%d by (n/n_s)^3" % n,
                 t["level1"], n))
    if limiter is not None:
        limiter.restore_original_limits()
    return dict(value=1.0 / step_s, unit="KKT solves/s", cores=cores, kind="port", sample=sample,
                host_logical_cpus=logical, host_physical_cores=physical_cores(),
                measured_at_full_size=bool(nn == n),
                faithful_cpu_gflops=faithful_gflops,
                host_dgemm_gflops=dgemm_gflops, host_potrf_probe_gflops=potrf_probe_gflops,
                blas_thread_probe=probe,
                strong_cpu_value=1.0 / strong_s, strong_cpu_seconds_per_step=strong_s,
                strong_cpu_potrf_s=t_potrf, strong_cpu_potrs_s=t_potrs, strong_cpu_threads=cores_potrf,
                strong_cpu_gflops=n ** 3 / 3.0 / t_potrf / 1e9,
                strong_cpu_variant=strong_variant, strong_cpu_variants=variants,
                strong_over_dgemm=(n ** 3 / 3.0 / t_potrf / 1e9 / dgemm_gflops) if dgemm_gflops else None,
                strong_cpu_note="Schur + Cholesky on the host (same elimination route as the GPU, src/kktsolvers.jl:281-338): the "
                                "FASTEST of LAPACK potrf (+ potrs) and a right-looking blocked Cholesky over dtrsm / dsyrk / dgemm "
                                "(block 512, oracle/blocked_chol.py; + 2 dtrsm per solve); factorisation and %d solves at n=%d timed "
                                "separately, best of two, S formed outside the timed region; potrf at the fastest thread count "
                                "of a potrf probe at n=4096, the blocked forms at the dgemm probe's; strong_over_dgemm = "
                                "strong_cpu_gflops / host_dgemm_gflops (a Cholesky cannot beat the GEMM it is made of)"
                                % (solves_per_factor, n))


def live_pmc_traffic(args):
    """HBM-side bytes per trailing-update launch MEASURED IN THIS RUN: two child processes under `rocprofv3 --pmc` (FETCH_SIZE,
    then WRITE_SIZE: separate passes, counters only -- no tracing beside them, MI355X_MICROARCH.md), each running this script on
    the same workload for three steps (one of them its warm-up: same kernels, same shapes -- the averages include it) without
    the CPU leg.  Started BEFORE this process initialises the GPU for itself (children, never an exec; the parent has only
    counted devices so far).  Returns (bytes per trailing-update launch, launches, bytes per solve4x4 or None) or
    (None, reason, None); the caller then falls back to the committed passes and says so (`traffic_measured_live`)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    tot = {}
    nl = {}
    gv = {}
    N = args.n if args.route == "schur" else 2 * args.n
    small, big = 256 * 256, 256 * ((N - 1024) // 4)          # k_gemv_t grids of the triangular sweeps (see pmc_solve_traffic)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="cip_pmc_", dir="/tmp")
            env = dict(os.environ, CIP_BENCH_PMC_CHILD="1", TMPDIR="/tmp")
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-converge", "--no-c5", "--no-live-pmc",
                   "--n", str(args.n), "--route", args.route]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                shutil.rmtree(d, ignore_errors=True)
                return None, "rocprofv3 --pmc %s child failed (rc %d)" % (ctr, r.returncode), None
            t, k = 0.0, 0
            g_tot, g_small = 0.0, 0
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] != ctr:
                        continue
                    name = row["Kernel_Name"].replace("void ", "")
                    if name.startswith(TRAILING_KERNELS):
                        t += float(row["Counter_Value"]); k += 1
                    elif name.startswith("k_gemv_t") and N % 1024 == 0 and int(row["Grid_Size"]) <= big:
                        g_tot += float(row["Counter_Value"]); g_small += int(row["Grid_Size"]) == small
            shutil.rmtree(d, ignore_errors=True)
            if k == 0:
                return None, "no trailing-update dispatch in the %s pass" % ctr, None
            tot[ctr], nl[ctr] = t, k
            gv[ctr] = (g_tot, g_small)
        if nl["FETCH_SIZE"] != nl["WRITE_SIZE"]:
            return None, "the two passes saw different launch counts", None
        # gfx950: FETCH_SIZE counts half the bytes of wide streaming reads -> doubled (MI355X_MICROARCH.md, section HBM); both in KiB
        solve_bytes = None
        per = 2 * (N // 1024 + 1)                            # k_gemv_t dispatches of the 1024-column grid per solve
        (gf, n1), (gw, n2) = gv["FETCH_SIZE"], gv["WRITE_SIZE"]
        if n1 > 0 and n1 == n2 and n1 % per == 0:
            solve_bytes = (2.0 * gf + gw) * 1024.0 / (n1 // per)
        return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / nl["FETCH_SIZE"], nl["FETCH_SIZE"], solve_bytes
    except Exception as e:                                   # never let the measurement of a side figure take the bench down
        return None, "live PMC pass failed: %r" % (e,), None


def pmc_traffic_per_launch():
    """Average HBM bytes per trailing-update launch from the committed PMC passes (gfx950: FETCH_SIZE counts half
    the bytes of wide streaming reads -> doubled, MI355X_MICROARCH.md section HBM).  None when absent."""
    import csv
    try:
        pdir = pmc_dir()

        def load(name, counter):
            out = {}
            with open(os.path.join(pdir, name)) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] == counter and r["Kernel"].startswith(TRAILING_KERNELS):
                        out[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
            return out
        fe = load("final_pmc_fetch.csv", "FETCH_SIZE")
        wr = load("final_pmc_write.csv", "WRITE_SIZE")
        if not fe or not wr or len(fe) != len(wr):
            return None
        return (2.0 * sum(fe.values()) + sum(wr.values())) * 1024.0 / len(fe)
    except Exception:
        return None


def pmc_solve_traffic(N):
    """HBM-side bytes of ONE solve4x4 from the committed PMC passes: the k_gemv_t dispatches of the triangular sweeps
    (1024-row column-dot gemvs; the Q gemvs of the loop have 8192 rows and a larger grid), 2*FETCH_SIZE + WRITE_SIZE as for
    the trailing update.  A solve has 2 * (N/1024 + 1) dispatches of the 1024-column grid.  None when absent / other N."""
    import csv
    try:
        if N % 1024:
            return None
        pdir = pmc_dir()
        small, big = 256 * 256, 256 * ((N - 1024) // 4)

        def load(name, counter):
            tot, nsmall = 0.0, 0
            with open(os.path.join(pdir, name)) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] == counter and r["Kernel"].startswith("k_gemv_t") and int(r["Grid_Size"]) <= big:
                        tot += float(r["Counter_Value"])
                        nsmall += int(r["Grid_Size"]) == small
            return tot, nsmall
        fe, n1 = load("final_pmc_fetch.csv", "FETCH_SIZE")
        wr, n2 = load("final_pmc_write.csv", "WRITE_SIZE")
        per = 2 * (N // 1024 + 1)
        if n1 == 0 or n1 != n2 or n1 % per:
            return None
        return (2.0 * fe + wr) * 1024.0 / (n1 // per)
    except Exception:
        return None


def plugin_boundary(ks, v_mid, s_mid, lam, spf, reps=5):
    """What a `ccall` caller of the three plugin levels gets at this size (src/ConicIP.jl:667, :682, :688): every call with HOST
    pointers, synchronous, fresh output arrays per level-3 call as the reference requires (:690) -- the path
    integration/ConicIPHIP's `kktsolver_hip` drives.  Untimed region of the bench (never `value`)."""
    n, m, p = ks.n, ks.m, ks.p
    ks.set_scaling_from_iterate(v_mid, s_mid, lam)
    packed = ks.get_scaling_packed()                       # the packed F a Julia shim reads off the Block (R: sqrt(s/v), :598)
    rng = np.random.default_rng(5)
    x, y, z = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    ks.set_scaling_packed(packed); ks.factor(check=True); ks.solve3x3(x, y, z)      # warm
    t2 = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ks.set_scaling_packed(packed)                      # cip_set_scaling_packed: host pointer, 8 m bytes up
        ks.factor(check=True)                              # cip_factor + cip_check_factor: assembly + LDL', waited for
        t2.append(time.perf_counter() - t0)
    t3 = []
    for _ in range(reps * 2):
        t0 = time.perf_counter()
        ks.solve3x3(x, y, z)                               # cip_solve3x3: 8 (n + p + m) bytes up, as many down, fresh outputs
        t3.append(time.perf_counter() - t0)
    l2, l3 = float(np.median(t2)), float(np.median(t3))
    return {"level2_ms": l2 * 1e3, "level3_ms": l3 * 1e3, "solves_per_factor": spf,
            "kkt_solves_per_s": 1.0 / (l2 + spf * l3),
            "bytes_per_level2_call": 8 * packed.size, "bytes_per_level3_call": 16 * (n + p + m),
            "note": "host-pointer C ABI as ccall drives it: level 2 = cip_set_scaling_packed + cip_factor + cip_check_factor, "
                    "level 3 = cip_solve3x3 (synchronous, fresh output vectors); medians of %d / %d calls; the level-3 call is "
                    "the 3x3 solve alone -- the 4x4 -> 3x3 reduction (src/ConicIP.jl:684-692) stays with the caller's loop"
                    % (reps, 2 * reps)}


def host_loop_through_plugin(Qh, c, A, b, cone_dims, route):
    """CPU-baseline leg (the only place bench.py touches oracle/): the REFERENCE-SHAPED HOST LOOP -- the oracle's restatement of
    src/ConicIP.jl:468-939, i.e. what a Julia user's `conicIP(...; kktsolver = kktsolver_hip)` executes on the host: cone algebra,
    five Q*y / A*y products per iteration and five per refinement pass on the CPU -- with the PRODUCT's plugin as its kktsolver.
    Wall-clock to converge, to be read beside the native loop's (`converge.wall_s`)."""
    import cipkkt
    from oracle.conicip import conicIP as oracle_conicIP
    ksolver = cipkkt.kktsolver_hip if route == "schur" else cipkkt.kktsolver_hip_full3x3
    holder = {}

    def kkt(Q, A_, G, cd):
        t0 = time.perf_counter()
        gen = ksolver(Q, A_, G, cd)
        holder["level1_s"] = time.perf_counter() - t0
        holder["gen"] = gen
        return gen
    t0 = time.perf_counter()
    sol = oracle_conicIP(Qh, c, A, b, cone_dims, optTol=1e-6, kktsolver=kkt)
    wall = time.perf_counter() - t0
    holder["gen"].system.close()
    return {"wall_s": wall, "level1_s": holder["level1_s"], "loop_s": wall - holder["level1_s"], "iters": sol.Iter,
            "status": sol.status, "n_factor": getattr(sol, "n_factor", None), "n_solve": getattr(sol, "n_solve", None),
            "blas_threads": os.cpu_count() or 1,
            "note": "oracle.conicIP (host restatement of the reference's loop) with kktsolver = cipkkt.kktsolver_hip: level 1 "
                    "(upload of Q, A) once, then one level-2 call per iteration + the initial point and 2-3 level-3 calls per "
                    "iteration through host pointers; the host's own work is the loop's numpy mat-vecs and cone algebra"}


def secondary_config(name, lib, device):
    """BASELINE configs[2] / configs[3] on this GPU, after the timed region: the native loop (cip_conicip) to convergence on the
    stated workload -- one warm-up solve, one timed solve, one more with HIP events around the dominant kernels (thread
    profile slots, include/cipkkt.h: cip_profile_kernel_thread)."""
    import cipkkt
    from cipkkt import workloads
    C = cipkkt._lib.C
    if name == "c3":
        Q, c, A, b, K, G, d = workloads.c3_socp()
        what = "SOCP n=4096, 512 x (Q,8) (m=4096), dense A, p=512: Schur order 4608"
    else:
        Q, c, A, b, K, G, d = workloads.c4_sdp(256, 1024, 16)
        what = 'SDP: one ("S", 32896) cone = matrix order 256, n=1024, dense A, p=16'
    n, m = Q.shape[0], A.shape[0]
    t0 = time.perf_counter()
    ks = cipkkt.KKTSystem(Q, A, G, K, device=device)
    torch.cuda.synchronize()
    level1_s = time.perf_counter() - t0
    try:
        cipkkt.conicIP(Q, c, A, b, K, G, d, optTol=1e-6, system=ks)                       # warm-up
        sol = cipkkt.conicIP(Q, c, A, b, K, G, d, optTol=1e-6, system=ks)                 # timed: the loop's own wall clock
        for slot in (0, 1, 2):
            lib.cip_profile_kernel_thread(slot, 1)
        cipkkt.conicIP(Q, c, A, b, K, G, d, optTol=1e-6, system=ks)
        prof = {}
        for slot, key in ((0, "trailing"), (1, "syrk"), (2, "jacobi")):
            o3 = (C.c_double * 3)()
            cipkkt._lib.check(lib.cip_profile_kernel_thread_get(slot, o3))
            prof[key] = (o3[0], o3[1], o3[2])
            lib.cip_profile_kernel_thread(slot, 0)
    finally:
        ks.close()
    out = {"workload": what, "status": sol.status, "iters": sol.Iter, "n_factor": sol.n_factor, "n_solve": sol.n_solve,
           "wall_s": sol.wall_s, "ms_per_iter": 1e3 * sol.wall_s / max(1, sol.Iter), "level1_s": level1_s}

    def mfma(key, kernel):
        l, ms, fl = prof[key]
        if l <= 0 or ms <= 0:
            return None
        ach = fl / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP64_MFMA_PEAK_TFLOPS, "launches": l, "avg_launch_ms": ms / l, "algorithmic_flops_per_launch": fl / l,
                "traffic": None}
    out["roofline"] = mfma("syrk", "Schur formation S = Q + (A'F^-1)(A'F^-1)' (m n^2 flop; %s)"
                           % ("k_syrkq_64" if name == "c3" else "k_syrk_splitk_128 images + k_syrk_reduce"))
    out["roofline_trailing"] = mfma("trailing", "LDL' trailing update k_ldlt_trailing_64")
    if name == "c4" and prof["jacobi"][0] > 0:
        l, ms, _ = prof["jacobi"]
        out["nt_scaling_svd"] = {"bound": "latency", "kernel": "one-sided Jacobi of svd(Lz'Ls) (src/ConicIP.jl:204)",
                                 "launches": l, "avg_launch_ms": ms / l, "share_of_iteration": (ms / l) / max(1e-9, out["ms_per_iter"])}
    fx = fixture_iters("c3_socp_seed11" if name == "c3" else "c4_sdp_r256_seed5")
    if fx is not None and isinstance(fx, dict) and "Iter" in fx:
        out["iters_cpu"] = fx["Iter"]
    return out


def full3x3_headline(Q, c_host, A, b, cone_dims, device, n):
    """The literal 3x3 route (src/kktsolvers.jl:254-257: the whole KKT matrix [-F'F -A 0; -A' Q G'; 0 G 0], N = m + n + p = 2n) on the
    headline workload, after the timed region: the native loop to convergence (iterations = the Schur route's = the oracle's), one
    timed factorisation (assembly + LDL' of order 2n, solve preparation joined) and HIP events around its trailing-update launches."""
    import cipkkt
    ks = cipkkt.KKTSystem(Q, A, None, cone_dims, route="full3x3", device=device)
    try:
        cipkkt.conicIP(Q, c_host, A, b, cone_dims, optTol=1e-6, system=ks, kktsolver="full3x3")                    # warm-up
        sol = cipkkt.conicIP(Q, c_host, A, b, cone_dims, optTol=1e-6, system=ks, kktsolver="full3x3")
        N = ks.N
        ks.profile_trailing(1)
        ks.set_timing(True)
        ks.factor()
        st = ks.stats()
        ks.set_timing(False)
        prof = ks.profile_get()
        ks.profile_trailing(False)
        fl = N ** 3 / 3.0
        out = {"workload": "headline QP through CIP_ROUTE_FULL3X3: LDL' of the literal 3x3 KKT matrix, order N = %d" % N,
               "kkt_order": N, "status": sol.status, "iters": sol.Iter, "n_factor": sol.n_factor, "n_solve": sol.n_solve,
               "wall_s": sol.wall_s, "ms_per_iter": 1e3 * sol.wall_s / max(1, sol.Iter),
               "ms_assemble": st["ms_assemble"], "ms_ldlt_factor": st["ms_ldlt"],
               "ldlt_tflops_whole_factor": fl / (st["ms_ldlt"] * 1e-3) / 1e12 if st["ms_ldlt"] > 0 else None,
               "algorithmic_flops_per_factor": fl}
        if prof["ms"] > 0:
            ach = prof["flops"] / (prof["ms"] * 1e-3) / 1e12
            out["roofline_trailing"] = {"bound": "mfma", "kernel": "LDL' trailing update k_ldlt_trailing_64 (K = outer block) at order %d" % N,
                                        "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                                        "launches": prof["launches"], "avg_launch_ms": prof["ms"] / max(1.0, prof["launches"]),
                                        "algorithmic_flops_per_launch": prof["flops"] / max(1.0, prof["launches"]), "traffic": None}
        fx = fixture_iters("c2_n%d_seed1234" % n)
        if fx is not None and "Iter" in fx:
            out["iters_cpu"] = fx["Iter"]
        return out
    finally:
        ks.close()


def c5_shard_passes(device, in_flight, full=None):
    """Config 5 as the 8 / 4 / 2 / 1-GPU job would hand it to RANK 0: lock-step passes over rank 0's shard of 64 / G problems
    (problem i -> rank i mod G), on this one GPU, after the timed region.  ms per pass of the shard is what the G-GPU job's slowest-
    rank time is made of (plus the imbalance between the ranks' shards, which one GPU cannot show): a projection of the 1 -> 8 curve
    from driver-observed numbers when no 8-GPU node is available, NOT a scaling measurement."""
    from cipkkt.batch import run_config5
    out = {}
    for G in (8, 4, 2, 1):
        if G == 1 and full is not None:
            out["64"] = {"gpus_of_the_job": 1, "problems": 64, "ms_per_pass": full["ms_per_pass"], "n_factor": full["n_factor"],
                         "iters": full["iters"], "n_optimal": full["n_optimal"]}
            continue
        stats, el = run_config5(0, G, None, device, 3, 1, count=64, n=2048, seed=4000, in_flight=in_flight)
        out[str(64 // G)] = {"gpus_of_the_job": G, "problems": stats["n_problems"], "ms_per_pass": el / 3 * 1e3,
                             "n_factor": stats["n_factor"], "iters": stats["iters"], "n_optimal": stats["n_optimal"]}
    if "64" in out and "8" in out:
        out["projected_speedup_8_gpus"] = out["64"]["ms_per_pass"] / out["8"]["ms_per_pass"]
        out["note"] = ("rank 0's shard of the G-GPU job (problems 0, G, 2G, ...) in lock-step on ONE GPU, 3 passes after 1 warm-up; "
                       "projected_speedup_8_gpus = ms(64 problems) / ms(8 problems): the other ranks' shards differ in iteration "
                       "counts, so the real curve is the driver's SCALE record")
    return out


def c5_cpu_baseline(seed=4000, n=2048):
    """Bounded CPU leg of the batch workload: problem 0 of config 5 (n = 2048) through the oracle's conicIP with
    pivot(kktsolver_2x2) (src/kktsolvers.jl:281-349) on the host cores -- a few seconds."""
    from cipkkt import workloads
    from oracle.conicip import conicIP as oracle_conicIP
    from oracle import kktsolvers as ok
    Q, c, A, b, K = workloads.c2_problem(n, seed)
    t0 = time.perf_counter()
    sol = oracle_conicIP(Q, c, A, b, K, optTol=1e-6, kktsolver=ok.pivot(ok.kktsolver_2x2))
    dt = time.perf_counter() - t0
    return dict(value=sol.n_factor / dt, unit="KKT solves/s", cores=os.cpu_count() or 1, kind="port",
                sample="problem 0 of the batch (n=2048, seed %d) to convergence through the oracle's conicIP + "
                       "pivot(kktsolver_2x2): %d factorisations, %d iterations, %s, %.2f s; BLAS threads = all logical CPUs"
                       % (seed, sol.n_factor, sol.Iter, sol.status, dt),
                iters=sol.Iter, status=sol.status)


def fixture_iters(key):
    """Iteration count of the ORACLE on the same inputs, from the committed full-size fixture
    (tests/golden/fullsize_trajectories.json, generated by tests/golden/make_fullsize_fixtures.py)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "fullsize_trajectories.json")) as f:
            return json.load(f)[key]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--route", default="schur", choices=["schur", "full3x3"])
    ap.add_argument("--nbo", type=int, default=0, help="LDL' outer block (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-converge", action="store_true")
    ap.add_argument("--workload", default=None, choices=["c2", "c5"],
                    help="c2: dense QP n=8192 (default at --gpus 1); c5: 64 x n=2048 batch (default at --gpus > 1)")
    ap.add_argument("--in-flight", type=int, default=8, help="c5: problems in flight per GPU (measured on one GPU: 4 -> 1703, 8 -> 2024 KKT solves/s)")
    ap.add_argument("--batch-mode", default="lockstep", choices=["lockstep", "threads"],
                    help="c5: lock-step batch (one launch per step for all problems of the rank) or host threads + streams")
    ap.add_argument("--no-c5", action="store_true", help="c2: skip the single-GPU config-5 figure")
    ap.add_argument("--no-secondary", action="store_true", help="c2: skip the configs[2] / configs[3] runs after the timed region")
    ap.add_argument("--no-plugin-boundary", action="store_true", help="c2: skip the host-pointer plugin-level timings")
    ap.add_argument("--profile-stride", type=int, default=4,
                    help="c2: HIP events around the trailing-update launches of every k-th timed step (1 = every step)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="c2: do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes before the GPU is "
                         "touched); replay the committed passes instead")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: start one rank per GPU as a child process -- before anything here touches the GPU
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    live_traffic = (None, "not requested", None)
    if (world == 1 and args.gpus == 1 and (args.workload in (None, "c2")) and not args.no_live_pmc and not args.no_cpu_baseline
            and not os.environ.get("CIP_BENCH_PMC_CHILD") and torch.cuda.device_count() > 0):
        # (only in the full default run -- the A/B and profiling invocations pass --no-cpu-baseline.  The children are fresh
        #  subprocesses started before this process makes its first HIP call that creates a context -- so far it has only
        #  counted devices -- and it never execs: each child opens the GPU for itself)
        live_traffic = live_pmc_traffic(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the KKT path has no CPU fallback)")
    # CIP_BENCH_SHARE_GPU=1 (tests only): every rank on cuda:0 and the process group on gloo -- RCCL refuses two ranks on one
    # device -- so that rank != 0's shard generation, the rank-0-alone pass behind the barrier and the reductions run on real
    # kernels on a one-GPU box.  Not a measurement: the ranks share the chip.
    share_gpu = bool(os.environ.get("CIP_BENCH_SHARE_GPU")) and world > 1
    if share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    reduce_device = torch.device("cpu") if share_gpu else device
    dist = None
    if world > 1 or os.environ.get("CIP_BENCH_FORCE_DIST"):     # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)   # nccl == RCCL on ROCm

    import cipkkt
    from cipkkt import workloads
    from cipkkt.batch import run_config5
    lib = cipkkt._lib.load()
    n = args.n
    if args.nbo:
        lib.cip_set_ldlt_outer_block(args.nbo)
    workload = args.workload or ("c5" if world > 1 else "c2")
    os.environ["CIP_BATCH"] = "auto" if args.batch_mode == "lockstep" else "threads"

    def config5(steps, warmup, rank_=rank, world_=world, dist_=None):
        stats, el = run_config5(rank_, world_, dist_, device, steps, warmup, count=64, n=2048, seed=4000,
                                in_flight=args.in_flight, reduce_device=reduce_device)
        return dict(value=stats["n_factor"] * steps / el, unit="KKT solves/s", ms_per_pass=el / steps * 1e3,
                    problems_per_s=64 * steps / el, n_optimal=stats["n_optimal"], n_problems=stats["n_problems"],
                    iters=stats["iters"], n_factor=stats["n_factor"], n_solve=stats["n_solve"],
                    rank_busy_ms_min=stats.get("rank_busy_ms_min"), rank_busy_ms_max=stats.get("rank_busy_ms_max"),
                    batch_mode=args.batch_mode, in_flight_per_gpu=args.in_flight if args.batch_mode == "threads" else None,
                    note="64 dense QPs n=m=2048 (seeds 4000+i), problem i -> rank i mod N, level-1 upload included, "
                         "%s; KKT solves = factorisations" % ("cip_conicip_lockstep (one launch per step for the rank's whole shard)"
                                                              if args.batch_mode == "lockstep" else "cip_conicip_problems (host threads)"))

    if workload == "c5":
        # rank 0 alone on the whole batch first (the 1-GPU point of the scaling curve, measured in THIS run) ...
        c5_one = None
        if world > 1 and rank == 0:
            c5_one = config5(2, 1, 0, 1, None)
        if dist is not None:
            dist.barrier()
        # ... then the sharded passes
        c5 = config5(args.steps, args.warmup, dist_=dist)
        ranks_seen = 1
        shard_sizes = [64]
        if dist is not None:
            one = torch.ones(1, dtype=torch.float64, device=reduce_device)
            dist.all_reduce(one, op=dist.ReduceOp.SUM)
            ranks_seen = int(round(float(one.item())))
            from cipkkt.batch import shard_indices
            sz = torch.zeros(world, dtype=torch.float64, device=reduce_device)       # the shard map as the ranks themselves see it
            sz[rank] = len(shard_indices(64, rank, world))
            dist.all_reduce(sz, op=dist.ReduceOp.SUM)
            shard_sizes = [int(round(x)) for x in sz.cpu().tolist()]
        if rank == 0:
            # roofline of THIS workload's dominant MFMA kernel: one more (untimed) pass of rank 0's shard with HIP events
            # around every trailing-update launch of the lock-step factorisations (launch = all live problems of the shard)
            roof = None
            if args.batch_mode == "lockstep":
                prev_split = lib.cip_set_lockstep_split(1)      # the event profile belongs to the calling thread: one group after the other
                lib.cip_profile_trailing_thread(1)
                from cipkkt.batch import solve_batch
                from cipkkt.workloads import c5_batch
                mine = list(range(rank, 64, world))
                probs = [None] * 64
                for i, pr in zip(mine, c5_batch(64, 2048, 4000, device=device, indices=mine)):
                    probs[i] = pr
                solve_batch(probs, rank=rank, world=world, dist=None, device=device, native=True)
                o3 = (cipkkt._lib.C.c_double * 3)()
                cipkkt._lib.check(lib.cip_profile_thread_get(o3))
                lib.cip_profile_trailing_thread(0)
                lib.cip_set_lockstep_split(prev_split)
                if o3[1] > 0:
                    ach = o3[2] / (o3[1] * 1e-3) / 1e12
                    roof = {"bound": "mfma", "kernel": "LDL' trailing update of the lock-step batch: k_ldlt_trailing_64, "
                                                       "grid.z = problems of rank 0's shard (%d at the start)" % len(mine),
                            "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                            "clock_limited_peak": FP64_MFMA_CLOCK_LIMITED_TFLOPS, "traffic": None,
                            "launches": o3[0], "avg_launch_ms": o3[1] / max(1.0, o3[0]),
                            "algorithmic_flops_per_launch_avg": o3[2] / max(1.0, o3[0]),
                            "note": "HIP events on the launch stream, one untimed extra pass of rank 0's shard; "
                                    "flops = live problems x r(r+1) x outer block per launch"}
            out = {"metric": "KKT solves/sec, batch of 64 independent dense QPs n=2048 (BASELINE config 5), %d GPU(s)" % world,
                   "value": c5["value"], "unit": "KKT solves/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": c5["ms_per_pass"], "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": {"workload": "config 5: 64 x dense QP n=m=2048 p=0 K=[(R,2048)] Q=M'M/n A=I(sparse) b=0 "
                                          "optTol=1e-6, seeds 4000+i (SplitMix64), problem i -> rank i mod %d; "
                                          "step = one pass over the batch" % world,
                              "parallelism": ("problem-per-GPU x%d, lock-step batch per GPU" % world) if args.batch_mode == "lockstep"
                                             else "problem-per-GPU x%d, %d in flight per GPU" % (world, args.in_flight)},
                   "ranks_seen": ranks_seen, "ranks_share_one_gpu": share_gpu, "shard_sizes": shard_sizes,
                   "batch": c5, "roofline": roof}
            if c5_one is not None:
                out["c5_single_gpu"] = c5_one
                out["speedup_vs_c5_single_gpu"] = c5["value"] / c5_one["value"]
            fx = fixture_iters("c5_n2048_seed4000")
            if fx is not None:
                out["iters_cpu"] = sum(p_["Iter"] for p_ in fx["problems"].values())
                out["iters_gpu"] = c5["iters"]
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = c5_cpu_baseline()
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    Q, c_host, A, b, cone_dims = workloads.c2_problem(n, 1234 + rank, device)
    ks = cipkkt.KKTSystem(Q, A, None, cone_dims, route=args.route, device=device)

    # ---- wall-clock to converge (the second half of the metric) + the statistics that define a step
    iters = n_factor = n_solve = None
    converge_s = None
    status = None
    iterates = []
    if not args.no_converge:
        torch.cuda.synchronize()
        # first pass (per-operation driver): warms the library up and records the iterates the step is taken from
        cipkkt.conicIP(Q, c_host, A, b, cone_dims, optTol=1e-6, system=ks, kktsolver=args.route,
                       keep_iterates=iterates, driver="python")
        torch.cuda.synchronize()
        # timed pass: the native loop (cip_conicip), same problem, same handle
        sol = cipkkt.conicIP(Q, c_host, A, b, cone_dims, optTol=1e-6, system=ks, kktsolver=args.route)
        converge_s, iters, n_factor, n_solve, status = sol.wall_s, sol.Iter, sol.n_factor, sol.n_solve, sol.status
        spf = max(1, int(round(n_solve / n_factor)))
        zmid = iterates[max(0, len(iterates) // 2 - 1)]
        v_mid = zmid[n:2 * n].clone()
        s_mid = zmid[2 * n:3 * n].clone()
    else:
        spf = 2
        g = torch.Generator(device=device)
        g.manual_seed(7)
        v_mid = torch.rand(n, generator=g, dtype=torch.float64, device=device) + 0.01
        s_mid = torch.rand(n, generator=g, dtype=torch.float64, device=device) + 0.01
    del iterates

    lam = torch.zeros(n, dtype=torch.float64, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(99)
    rhs = torch.randn(3 * n, generator=g, dtype=torch.float64, device=device)
    dz = torch.zeros(3 * n, dtype=torch.float64, device=device)

    def step():
        ks.set_scaling_from_iterate(v_mid, s_mid, lam)
        ks.factor(check=False)                      # enqueue only, as the native loop does
        for _ in range(spf):
            ks.solve4x4_dev(lam, rhs, dz)

    for _ in range(args.warmup):
        step()
    ks.check_factor()
    # roofline events INSIDE the timed region (contract), sampled: every trailing-update launch of every `stride`-th step is
    # bracketed by a HIP-event pair on the launch stream (the first timed step included).  An event pair costs the chain ~8 us;
    # round 4 timed every step (0.9 % of `value`, `profiling_cost`), the sampled form perturbs a `stride`-th as much.
    prof_stride = max(1, min(args.profile_stride, args.steps))
    ks.profile_trailing(prof_stride)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = ks.profile_get()
    ks.profile_trailing(False)
    ks.check_factor()
    # what the HIP events around every trailing-update launch cost the number above: the same K steps again without them
    # (same session A/B; round-4 review)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed_noprof = time.perf_counter() - t0
    ks.check_factor()

    # separate factor / solve split (untimed region, for the report)
    ks.set_timing(True)
    ks.factor()
    st = ks.stats()
    ks.set_timing(False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        ks.solve4x4_dev(lam, rhs, dz)
    torch.cuda.synchronize()
    solve_ms = (time.perf_counter() - t1) / 5 * 1e3

    pb = None
    if rank == 0 and not args.no_plugin_boundary and not os.environ.get("CIP_BENCH_PMC_CHILD"):
        pb = plugin_boundary(ks, v_mid, s_mid, lam, spf)
    ks.close()
    c5_single = None
    c5_shards = None
    if world == 1 and not args.no_c5 and n == 8192 and args.route == "schur":
        c5_single = config5(2, 1)                     # single-GPU config-5 figure, for cross-checking a SCALE run
        if rank == 0 and not os.environ.get("CIP_BENCH_PMC_CHILD") and args.batch_mode == "lockstep":
            try:
                c5_shards = c5_shard_passes(device, args.in_flight, full=c5_single)
            except Exception as e:                   # a side figure never takes the bench line down
                c5_shards = {"error": repr(e)}
    Qh = Q.cpu().numpy() if (rank == 0 and not args.no_cpu_baseline) else None
    secondary = None
    if (world == 1 and rank == 0 and not args.no_secondary and n == 8192 and args.route == "schur"
            and not os.environ.get("CIP_BENCH_PMC_CHILD")):
        secondary = {}
        try:                                         # the literal 3x3 route on the headline workload (needs Q: before it is dropped)
            secondary["full3x3"] = full3x3_headline(Q, c_host, A, b, cone_dims, device, n)
        except Exception as e:
            secondary["full3x3"] = {"error": repr(e)}
        del Q
        torch.cuda.empty_cache()
        for name in ("c3", "c4"):
            try:
                secondary[name] = secondary_config(name, lib, device)
            except Exception as e:                   # a side figure never takes the bench line down
                secondary[name] = {"error": repr(e)}
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        ach = prof["flops"] / (prof["ms"] * 1e-3) / 1e12 if prof["ms"] > 0 else 0.0
        N = ks.N
        pdir = pmc_dir()
        fx = fixture_iters("c2_n%d_seed1234" % n) if (args.route == "schur" and rank == 0) else None
        out = {
            "metric": "KKT solves/sec + wall-clock to converge, dense QP n=%d, 1 GPU" % n,
            "value": value, "unit": "KKT solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "dense QP n=m=%d p=0 K=[(R,%d)] Q=M'M/n A=I(sparse) b=0 optTol=1e-6; "
                                   "step = NT scaling + assembly + LDL' + %d solve4x4" % (n, n, spf),
                       "route": args.route, "kkt_order": N, "solves_per_factor": spf,
                       "ldlt_outer_block": int(st["nbo"]), "parallelism": "problem-per-GPU x%d" % world},
            "converge": {"wall_s": converge_s, "iters": iters, "status": status, "n_factor": n_factor,
                         "n_solve": n_solve},
            # north-star: "at identical iteration count to convergence" -- the CPU side is the oracle's run on the same
            # SplitMix64 inputs, committed as tests/golden/fullsize_trajectories.json (asserted equal, with mu / alpha per
            # iteration at 1e-6, by tests/test_gpu_configs_full.py)
            "iters_gpu": iters, "iters_cpu": fx["Iter"] if fx else None,
            "iters_cpu_source": "tests/golden/fullsize_trajectories.json (oracle: pivot(kktsolver_2x2), same inputs)" if fx else None,
            "breakdown_ms": {"assemble": st["ms_assemble"], "ldlt_factor": st["ms_ldlt"], "solve4x4": solve_ms,
                             "ldlt_tflops_whole_factor": (N ** 3 / 3.0) / (st["ms_ldlt"] * 1e-3) / 1e12
                             if st["ms_ldlt"] > 0 else None},
            "roofline_solve": {"bound": "hbm", "kernel": "LDL' triangular sweeps inside cip_solve4x4 (k_gemv_t block steps)",
                               "achieved": (8.0 * N * (N + 1) + 16.0 * N * min(1024, ks.Npad)) / (solve_ms * 1e-3) / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": (8.0 * N * (N + 1) + 16.0 * N * min(1024, ks.Npad)) / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "traffic": live_traffic[2] if live_traffic[2] is not None else pmc_solve_traffic(ks.Npad),
                               "traffic_measured_live": live_traffic[2] is not None,
                               "note": "algorithmic bytes of one solve = L read once per sweep (8 N (N+1)) + the 1024-wide block "
                                       "inverses (16 N Bs); time = whole cip_solve4x4 call (cone division, A'/A products, "
                                       "two sweeps), host-timed average of 5"},
            "roofline": {"bound": "mfma",
                         "kernel": "LDL' trailing update: k_ldlt_trailing_64 (64x64 fp64-MFMA tiles, K = outer block)",
                         "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                         "clock_limited_peak": FP64_MFMA_CLOCK_LIMITED_TFLOPS,
                         "frac_of_clock_limited_peak": ach / FP64_MFMA_CLOCK_LIMITED_TFLOPS,
                         "traffic": live_traffic[0] if live_traffic[0] is not None else pmc_traffic_per_launch(),
                         "traffic_measured_live": live_traffic[0] is not None,
                         "traffic_source": (("MEASURED in this run: two child passes of this script under rocprofv3 --pmc (FETCH_SIZE, "
                                             "WRITE_SIZE; %d trailing-update launches each, the children's warm-up step included) before the timed process touched the GPU"
                                             % live_traffic[1]) if live_traffic[0] is not None else
                                            ((os.path.relpath(pdir, ROOT) + "/final_pmc_{fetch,write}.csv: REPLAYED from the committed "
                                              "rocprofv3 --pmc passes of this same command, not measured in this run (live pass: %s)"
                                              % (live_traffic[1],)) if pdir else None)),
                         "traffic_note": "HBM-side bytes per trailing-update launch = (2*FETCH_SIZE + WRITE_SIZE) KiB of "
                                         "k_ldlt_trailing_64, averaged over its launches (separate --pmc passes; null if absent)",
                         "launches": prof["launches"], "avg_launch_ms": prof["ms"] / max(1.0, prof["launches"]),
                         "algorithmic_flops_per_launch_avg": prof["flops"] / max(1.0, prof["launches"])},
        }
        out["profiling_cost"] = {"ms_per_step_with_trailing_events": ms_per_step,
                                 "ms_per_step_without": elapsed_noprof / args.steps * 1e3,
                                 "profiled_steps": (args.steps + prof_stride - 1) // prof_stride, "profile_stride": prof_stride,
                                 "note": "`value` is measured WITH the per-launch HIP events of the roofline inside the timed "
                                         "region (contract): every trailing-update launch of every `profile_stride`-th timed step; "
                                         "the same steps again without any events, same session"}
        if pb is not None:
            out["plugin_boundary"] = pb
        if secondary is not None:
            out["secondary"] = secondary
        out["config"]["seed"] = 1234
        out["config"]["rng"] = "SplitMix64 + Box-Muller (cipkkt/workloads.py), generated in HBM"
        if c5_single is not None:
            out["c5_single_gpu"] = c5_single
        if c5_shards is not None:
            out["c5_shards"] = c5_shards
        if not args.no_cpu_baseline:
            cb = cpu_baseline(Qh, n, spf)
            out["cpu_baseline"] = cb
            if pb is not None and not args.no_converge:
                try:
                    hl = host_loop_through_plugin(Qh, c_host, A, b, cone_dims, args.route)
                    hl["native_loop_wall_s"] = converge_s
                    hl["iters_native"] = iters
                    pb["host_loop"] = hl
                except Exception as e:
                    pb["host_loop"] = {"error": repr(e)}
            out["gpu_over_cpu"] = (value / world) / cb["value"]                       # against the reference-faithful kktsolver_qr leg
            out["gpu_over_strong_cpu"] = (value / world) / cb["strong_cpu_value"]     # against Schur + LAPACK Cholesky: the honest ratio
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
