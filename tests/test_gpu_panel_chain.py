"""The in-launch hand-offs of the blocked LDL's panel chain (diag.hip: k_ldlt_panel, k_ldlt_diag_upd).  Inside those
launches workgroups hand blocks of K to each other through device-memory counters, write-through stores and agent-scope
loads -- so the test that matters is a BIT-FOR-BIT comparison with the three-launch chain running the same arithmetic: a
single stale read anywhere changes bits.  Repeated, under a concurrent memory-streaming load, and -- for the default
schedule -- thousands of times (the opt-in look-ahead schedules of rounds 1-2, whose bit-identity test had to be softened
after one unexplained mismatch, are gone from the library: VERDICT r2 / ADVICE r2)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import cipkkt
    return cipkkt._lib.load()


def _factor(lib, dK0, N, ws, out=None):
    from cipkkt import _lib as L
    dK = dK0.clone() if out is None else out.copy_(dK0)
    info = C.c_int(-1)
    L.check(lib.cip_ldlt_factor_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), C.byref(info)))
    torch.cuda.synchronize()
    assert info.value == 0
    return dK


def _spd(N, seed, quasi=0):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    M = torch.randn(N, N, generator=g, dtype=torch.float64, device="cuda")
    K = M @ M.t() / N + torch.eye(N, dtype=torch.float64, device="cuda")
    if quasi:                                   # [S G'; G 0]: the last `quasi` pivots are negative
        K[N - quasi:, N - quasi:] = 0.0
    return K.contiguous()                       # symmetric: row-major == column-major


@pytest.mark.parametrize("mode", [1, 3], ids=["diag+update", "one-launch-panel"])
@pytest.mark.parametrize("N,quasi", [(128, 0), (256, 0), (384, 38), (896, 0), (1024, 0), (2048, 0), (4608, 512), (8192, 0)])
def test_fused_panel_chain_bitwise_equals_unfused(lib, N, quasi, mode):
    """Mode 1: from the second panel of an outer block on, the diagonal kernel's launch carries the previous panel's in-block
    update and waits, INSIDE the launch, for the three tiles that are its own block (diag.hip: k_ldlt_diag_upd).  Same
    arithmetic in the same order as the three-launch chain: identical bits -- also repeated under a concurrent 1-GiB copy
    load, which is when a missing fence or a stale line would show.  Mode 3: that launch also carries the panel's TRSM,
    which follows the diagonal kernel micro-panel by micro-panel through agent-scope stores, loads and a stage counter
    (diag.hip: k_ldlt_panel) -- same bar.  The small orders are the launch shapes without strips (one block), without
    update tiles (last panel of an outer block) and with a single strip."""
    from cipkkt import _lib as L
    nbytes = C.c_size_t()
    L.check(lib.cip_ldlt_workspace_bytes(N, C.byref(nbytes)))
    ws = torch.zeros(nbytes.value // 8 + 8, dtype=torch.float64, device="cuda")
    K0 = _spd(N, N + quasi + 11, quasi)
    prev = lib.cip_set_ldlt_fused_chain(0)
    # the automatic outer block depends on the chain mode (896 fused from order 4096 on, 512 else) and the block width changes the summation
    # order: pin the production width for both sides of the comparison
    lib.cip_set_ldlt_outer_block(896 if N >= 4096 else 512)
    try:
        ref = _factor(lib, K0, N, ws)
        lib.cip_set_ldlt_fused_chain(mode)
        for rep in range(3):
            got = _factor(lib, K0, N, ws)
            assert torch.equal(torch.tril(got.t()), torch.tril(ref.t())), "repetition %d differs from the unfused chain" % rep
        side = torch.cuda.Stream()
        a = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
        b = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
        for rep in range(3):
            with torch.cuda.stream(side):
                for _ in range(4):
                    b.copy_(a, non_blocking=True)
                    a.copy_(b, non_blocking=True)
            got = _factor(lib, K0, N, ws)
            assert torch.equal(torch.tril(got.t()), torch.tril(ref.t())), "under load, repetition %d differs" % rep
        side.synchronize()
    finally:
        lib.cip_set_ldlt_fused_chain(prev)
        lib.cip_set_ldlt_outer_block(0)


@pytest.mark.parametrize("N,reps", [(1024, 1200), (2048, 700), (4608, 120), (8192, 40)])
def test_default_chain_is_bit_reproducible_over_thousands_of_factorisations(lib, N, reps):
    """The DEFAULT schedule (one launch per panel) under a concurrent copy load: every factor of the same matrix must equal
    the first one bit for bit, and the first one must equal the unfused three-launch chain.  2060 factorisations in all
    (tools/chain_stress.py is the stand-alone form); a mismatch is a failure -- no retry, no warning."""
    from cipkkt import _lib as L
    nbytes = C.c_size_t()
    L.check(lib.cip_ldlt_workspace_bytes(N, C.byref(nbytes)))
    ws = torch.zeros(nbytes.value // 8 + 8, dtype=torch.float64, device="cuda")
    K0 = _spd(N, 5 * N + 1)
    lib.cip_set_ldlt_outer_block(896 if N >= 4096 else 512)
    prev = lib.cip_set_ldlt_fused_chain(0)
    try:
        unfused = torch.tril(_factor(lib, K0, N, ws).t()).clone()
        lib.cip_set_ldlt_fused_chain(3)
        buf = torch.empty_like(K0)
        side = torch.cuda.Stream()
        a = torch.empty(1 << 26, dtype=torch.float64, device="cuda")
        b = torch.empty_like(a)
        bad = 0
        for r in range(reps):
            if r % 3 == 0:
                with torch.cuda.stream(side):
                    b.copy_(a, non_blocking=True)
                    a.copy_(b, non_blocking=True)
            got = _factor(lib, K0, N, ws, out=buf)
            if not torch.equal(torch.tril(got.t()), unfused):
                bad += 1
        side.synchronize()
        assert bad == 0, "%d of %d factorisations differ from the unfused chain" % (bad, reps)
    finally:
        lib.cip_set_ldlt_fused_chain(prev)
        lib.cip_set_ldlt_outer_block(0)


def test_factor_is_asynchronous():
    """cip_factor must not wait for the GPU (include/cipkkt.h): at n = 8192 the call returns in a
    fraction of the factorisation's device time, the pivot flag is resolved later (cip_check_factor / the solves)."""
    import time
    import cipkkt
    from cipkkt import workloads as W
    n = 8192
    Q, c, A, b, K = W.c2_problem(n, seed=7, device="cuda")
    ks = cipkkt.KKTSystem(Q, A, None, K)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    v = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
    s = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
    lam = torch.zeros(n, dtype=torch.float64, device="cuda")
    rhs = torch.randn(3 * n, generator=g, dtype=torch.float64, device="cuda")
    dz = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
    for _ in range(2):
        ks.set_scaling_from_iterate(v, s, lam); ks.factor(); ks.solve4x4_dev(lam, rhs, dz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ks.set_scaling_from_iterate(v, s, lam)
    ks.factor(check=False)
    ks.solve4x4_dev(lam, rhs, dz)                 # speculative (a factorisation of this handle has been verified): no host wait
    t_enqueue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    ks.check_factor()
    assert t_total > 3e-3, t_total                 # the factorisation alone is ~5 ms of device time
    assert t_enqueue < 0.5 * t_total, (t_enqueue, t_total)
    ks.close()


def test_first_solve_after_factor_waits_for_the_pivot_flag():
    """factor -> solve*_dev on a fresh handle must not return rc 0 on a factor that met a bad pivot.  Q = 0 with
    fewer inequality rows than variables makes the Schur block A'(F'F)^-1 A singular although the KKT matrix is not (the
    equalities pin the rest): the FIRST factorisation meets a zero pivot.  The first *_dev solve must resolve the flag
    with a host wait -- the handle switches to the regularised factorisation inside that call -- and either return the
    solution of the system or raise; never rc 0 with a solution of the broken factor."""
    import ctypes as C
    import cipkkt
    rng = np.random.default_rng(2)
    n, m, p = 40, 24, 16
    Q = np.zeros((n, n))
    A = rng.standard_normal((m, n))
    G = rng.standard_normal((p, n))
    ks = cipkkt.KKTSystem(Q, A, G, [("R", m)])
    ks.set_scaling_identity()
    ks.factor(check=False)                         # enqueue only
    f64 = dict(dtype=torch.float64, device="cuda")
    x, y, z = (torch.as_tensor(rng.standard_normal(k), **f64) for k in (n, p, m))
    a, b, c = (torch.zeros(k, **f64) for k in (n, p, m))
    try:
        ks.solve3x3_dev(x, y, z, a, b, c)
    except cipkkt.CipError as e:                   # acceptable: the error surfaces HERE, in the first solve
        assert "pivot" in str(e) or "repeat" in str(e), str(e)
        ks.solve3x3_dev(x, y, z, a, b, c)          # "repeat them": now on the regularised factor
    rel, switched = C.c_double(), C.c_int()
    cipkkt._lib.check(ks.lib.cip_get_regularization(ks.h, C.byref(rel), C.byref(switched)))
    assert switched.value == 1 and rel.value > 0   # resolved by the first solve, not by "a later, unrelated call"
    torch.cuda.synchronize()
    ah, bh, ch = a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()
    r1 = Q @ ah + G.T @ bh - A.T @ ch - x.cpu().numpy()
    r2 = G @ ah - y.cpu().numpy()
    r3 = A @ ah + ch - z.cpu().numpy()             # F = I
    scale = 1 + max(np.linalg.norm(v) for v in (ah, bh, ch))
    assert max(np.linalg.norm(r1), np.linalg.norm(r2), np.linalg.norm(r3)) < 1e-8 * scale
    ks.close()


@pytest.mark.parametrize("n", [4096, 8192])
def test_side_stream_solve_preparation_bitwise(n):
    """Round 4: block inverses + mirror image of the solve blocks whose columns are final run on a side stream beside the last
    panels of the factorisation (ldlt.hip: cip_ldlt_factor).  The factor and every solve must be bit-identical to the serial
    preparation -- over repeated factorisations with solves in between (the next assembly overwrites K: the join matters)."""
    import cipkkt
    from cipkkt import workloads as W
    Q, c, A, b, K = W.c2_problem(n, seed=77, device="cuda")
    ks = cipkkt.KKTSystem(Q, A, None, K)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    outs = {}
    prev = ks.lib.cip_set_ldlt_side_prep(0)
    try:
        for mode in (0, 1, 0, 1):
            ks.lib.cip_set_ldlt_side_prep(mode)
            res = []
            for it in range(4):
                v = torch.rand(n, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(10 + it)) + 0.1
                s = torch.rand(n, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(20 + it)) + 0.1
                ks.set_scaling_from_iterate(v, s)
                ks.factor()
                x = torch.randn(n, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(30 + it))
                z = torch.randn(n, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(40 + it))
                a_, c_ = torch.zeros_like(x), torch.zeros_like(z)
                y0 = torch.zeros(0, dtype=torch.float64, device="cuda")
                ks.solve3x3_dev(x, y0, z, a_, y0, c_)
                ks.solve3x3_dev(z, y0, x, c_, y0, z.clone())       # a second solve on the same factor
                torch.cuda.synchronize()
                res.append((a_.clone(), c_.clone()))
            if mode in outs:
                for (a0, c0), (a1, c1) in zip(outs[mode], res):
                    assert torch.equal(a0, a1) and torch.equal(c0, c1)
            outs[mode] = res
        for (a0, c0), (a1, c1) in zip(outs[0], outs[1]):
            assert torch.equal(a0, a1) and torch.equal(c0, c1), "side-stream preparation changed a solve"
    finally:
        ks.lib.cip_set_ldlt_side_prep(prev)
        ks.close()


# ---------------------------------------------------------------- round 6: an in-launch wait that gives up (a GPU shared with other processes)
def test_a_fused_chain_that_gives_up_falls_back_to_the_three_launch_chain():
    """The fused panel chain waits inside a launch for workgroups of the same launch; the wait is bounded (~1 s) and then the
    factorisation reports that it gave up.  With eight processes on one MI355X the hardware scheduler was seen to keep a launch's
    workgroups apart for longer than that (tests/test_gpu_bench_contract.py::test_eight_ranks_on_one_gpu, round 6).  The library
    then redoes the factorisation with the three-launch chain (no in-launch wait, same bits) and keeps it for the handle; a problem
    of a lock-step group leaves the group and is solved alone.  cip_debug_chain_giveup(n) makes the next n fused factorisations
    report such a give-up: the solve must come out exactly as without it -- one problem, and a lock-step group."""
    import cipkkt
    from cipkkt import _lib as L
    from cipkkt.batch import _solve_problems_native
    from cipkkt.workloads import c5_batch
    lib = L.load()
    dev = torch.device("cuda:0")
    prs = c5_batch(count=6, n=640, seed=5100, device=dev)
    pr = prs[0]
    ref = cipkkt.conicIP(pr["Q"], pr["c"], pr["A"], pr["b"], pr["cone_dims"], optTol=1e-6)
    assert ref.status == "Optimal"
    lib.cip_debug_chain_giveup(1)
    try:
        got = cipkkt.conicIP(pr["Q"], pr["c"], pr["A"], pr["b"], pr["cone_dims"], optTol=1e-6)
        assert lib.cip_debug_chain_giveup(-1) == 0                       # the hook fired
    finally:
        lib.cip_debug_chain_giveup(0)
    assert (got.status, got.Iter, got.n_solve) == (ref.status, ref.Iter, ref.n_solve)
    assert np.array_equal(got.y, ref.y) and np.array_equal(got.v, ref.v)
    # a lock-step group: the first factorisation of the group gives up for every problem -> all leave the group, solved alone
    one = _solve_problems_native(prs, dev, 1, "lockstep")
    lib.cip_debug_chain_giveup(1)
    try:
        two = _solve_problems_native(prs, dev, 1, "lockstep")
        assert lib.cip_debug_chain_giveup(-1) == 0
    finally:
        lib.cip_debug_chain_giveup(0)
    st = (C.c_int * 3)()
    lib.cip_lockstep_stats(st)
    assert st[1] == 6 and st[2] == 6, list(st)                             # six problems, six left their group
    for a, b in zip(two, one):
        assert (a.status, a.Iter) == (b.status, b.Iter) and a.status == "Optimal"
        assert np.allclose(a.y, b.y, rtol=1e-9, atol=1e-12)                # (alone: another solve block than in the group -> rounding)


def test_late_helper_stores_of_the_diagonal_kernel_do_not_change_the_factor(tmp_path):
    """Round 6: the fused panel launch hands its micro-panels to the TRSM strips through ONE stage word.  Until then the serial wave's
    count of micro-panel 7 (which it publishes itself, ahead of the helper waves' last stores) and the helpers' counts of micro-panels
    0 .. 6 ran on the same running number -- and "stage 6 reached" came true without micro-panel 6 whenever a helper's write-through
    stores took a microsecond longer than usual: about one factorisation of order 4096 in 100 000 had a few 64-row strips of one panel
    computed from the previous contents of column block 6 (tests/test_gpu_driver.py::test_dense_qp_2048_properties failed once in a few
    dozen suite runs; profiles/r6/stage_count_defect.txt).  A test build of the library (-DDIAG_DEBUG_SLOW_HELPERS: the helpers' write-back
    of micro-panel 6 held back) must produce the regular build's bits; the same delay on the old counting (-DDIAG_OLD_STAGE_COUNT)
    is what tools/stage_mix_demo.sh shows going wrong."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box: the test build cannot be made")
    sys.path.insert(0, os.path.join(root, "conicip.jl_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("cipkkt_build", os.path.join(root, "conicip.jl_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    mod.build()                                                       # (no-op when the objects are up to date)
    obj = str(tmp_path / "diag_slow.o")
    subprocess.run([hipcc] + [f for f in mod.FLAGS if f != "-Wall"] + ["-w", "-DDIAG_DEBUG_SLOW_HELPERS=4000", "-c", os.path.join(mod.CSRC, "diag.hip"), "-o", obj], check=True)
    objs = [obj if src == "diag.hip" else os.path.join(mod.OBJ, src.replace(".hip", ".o")) for src in mod.SOURCES]
    lib = str(tmp_path / "libcipkkt_slowhelpers.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], check=True)

    def bits(env_lib, n):
        env = dict(os.environ)
        env.pop("CIPKKT_LIB", None)
        if env_lib:
            env["CIPKKT_LIB"] = env_lib
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "_chain_bits.py"), str(n), "3"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("BITS ")][-1]
    for n in (1024, 2048):
        assert bits(lib, n) == bits(None, n), n


@pytest.mark.parametrize("env", [{"CIP_DEBUG_POISON": "63"}, {"CIP_DEBUG_POISON": "255"}, {"CIP_DEBUG_SIDE_DELAY_US": "3000"}],
                         ids=["buffers start as 0x3f", "buffers start as NaN", "side stream 3 ms late"])
def test_bits_do_not_depend_on_what_fresh_memory_holds_or_on_when_the_side_stream_runs(env):
    """The library's debug switches of round 6 as a regression test: every buffer of a handle filled with a byte pattern at creation
    (fresh memory of a fresh process is zero, recycled memory of a long-lived one is not), and every group of the solve preparation's
    side stream started late (whoever reads a group's blocks without having waited for it reads the previous factorisation's) --
    factor + solve of three scalings on both routes must come out with the plain run's bits."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def bits(extra):
        e = dict(os.environ)
        for k in ("CIPKKT_LIB", "CIP_DEBUG_POISON", "CIP_DEBUG_SIDE_DELAY_US"):
            e.pop(k, None)
        e.update(extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "_chain_bits.py"), "2048", "3"], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("BITS ")][-1]
    assert bits(env) == bits({})
