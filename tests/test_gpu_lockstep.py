"""Lock-step batches (csrc/lockstep.hip, `cip_conicip_lockstep`): B problems of one shape advance through the
interior-point loop together, one launch per step with the problem index in blockIdx.z.  The kernels and their
arithmetic are those of the one-problem path, so the bar is BIT-identity with `cip_conicip` on each problem
(iterates, iteration / factorisation / solve counts, status) -- through mixed cones, both routes, dense and CSR A,
problems that finish at different iterations or with different statuses, problems that leave the group because their
factorisation needs the regularised retry, and more than 64 problems (two groups)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import problems as P

pytestmark = pytest.mark.gpu


def _solve(prs, mode, in_flight=4):
    """the thread-pool reference runs with the solve-block limit the lock-step group uses for its handles (chosen from the size of the whole call: 512 up to 8 problems,
    else 256, unless CIP_LOCKSTEP_SOLVE_BLOCK says otherwise): the block size changes the summation order of the
    triangular solves, and the comparison below is bit for bit"""
    from cipkkt import _lib as L
    from cipkkt.batch import _solve_problems_native
    lib = L.load()
    prev = lib.cip_set_solve_block_max(lib.cip_lockstep_solve_block_for(len(prs))) if mode == "threads" else None
    try:
        return _solve_problems_native(prs, torch.device("cuda:0"), in_flight, mode)
    finally:
        if prev is not None:
            lib.cip_set_solve_block_max(prev)


def _as_problem(t, **kw):
    Q, c, A, b, cone_dims, G, d = t[:7]
    return dict(Q=Q.toarray() if sp.issparse(Q) else Q, c=c, A=A, b=b, cone_dims=cone_dims, G=G, d=d, kwargs=dict(kw))


def _assert_identical(a, b):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.status == y.status, (i, x.status, y.status)
        assert (x.Iter, x.n_factor, x.n_solve) == (y.Iter, y.n_factor, y.n_solve), i
        for f in ("y", "w", "v"):
            assert np.array_equal(getattr(x, f), getattr(y, f), equal_nan=True), (i, f)
        for f in ("Mu", "prFeas", "duFeas", "muFeas", "pobj", "dobj"):
            assert getattr(x, f) == getattr(y, f) or (getattr(x, f) != getattr(x, f) and getattr(y, f) != getattr(y, f)), (i, f)


@pytest.mark.parametrize("route", ["schur", "full3x3"])
@pytest.mark.parametrize("dense_A", [True, False], ids=["denseA", "csrA"])
def test_mixed_cones_bit_identical(route, dense_A):
    prs = []
    for seed in range(7):
        t = P.random_mixed(n=40, nq=3, kq=6, p=4, seed=100 + seed, dense_A=dense_A)
        pr = _as_problem(t, kktsolver=route)
        if not dense_A:
            pr["A"] = sp.csr_matrix(pr["A"])
        prs.append(pr)
    one = _solve(prs, "threads", in_flight=1)
    lock = _solve(prs, "lockstep")
    assert all(s.status == "Optimal" for s in one)
    assert len({s.Iter for s in one}) >= 1
    _assert_identical(lock, one)


def test_dense_qps_finishing_at_different_iterations():
    """config-5 family at n = 384: the problems of a group need 8-11 iterations; the ones that are done stop taking part"""
    from cipkkt.workloads import c5_batch
    prs = c5_batch(count=12, n=384, seed=4000)
    # spread the iteration counts: different tolerances are not allowed inside one batch, scale the data instead
    for i, pr in enumerate(prs):
        pr["c"] = pr["c"] * (10.0 ** (i % 4))
    one = _solve(prs, "threads", in_flight=1)
    lock = _solve(prs, "lockstep")
    assert len({s.Iter for s in one}) > 1, "the case is meant to have problems finishing at different iterations"
    _assert_identical(lock, one)
    assert all(s.status == "Optimal" for s in lock)


def test_statuses_differ_inside_one_group():
    """feasible, infeasible and unbounded problems of one shape in one group"""
    n = 10
    prs = []
    for seed in range(3):
        prs.append(_as_problem(P.infeasible_box(n, seed)))                       # y >= 1 and y <= -1
    rng = np.random.default_rng(3)
    for _ in range(3):                                                           # same shape, feasible: -5 <= y <= 5
        h = rng.standard_normal(n)
        H = np.outer(h, h) + 0.1 * np.eye(n)
        A = sp.vstack([sp.identity(n), -sp.identity(n)]).tocsr()
        prs.append(dict(Q=H, c=rng.standard_normal(n), A=A, b=-5 * np.ones(2 * n), cone_dims=[("R", 2 * n)], G=None, d=None, kwargs={}))
    one = _solve(prs, "threads", in_flight=1)
    lock = _solve(prs, "lockstep")
    assert {s.status for s in one} == {"Infeasible", "Optimal"}
    _assert_identical(lock, one)


def test_problems_that_need_the_regularised_factorisation_leave_the_group():
    """LPs (Q = 0): the Schur block A'(F'F)^-1 A of the free-variable part is singular in the static order for some of
    them -> bad pivot -> the problem is solved by the one-problem loop on its own handle; the others stay in lock-step."""
    n = 12
    prs = []
    rng = np.random.default_rng(5)
    for k in range(6):
        if k % 2 == 0:      # LP with a free direction in Q and A (rank-deficient S): needs regularisation
            Q = np.zeros((n, n))
            A = np.zeros((n, n))
            A[:n - 2, :n - 2] = np.eye(n - 2)
            A[n - 2, 0] = 1.0
            A[n - 1, 1] = 1.0
            G = np.zeros((2, n))
            G[0, n - 2] = 1.0
            G[1, n - 1] = 1.0
            d = np.array([1.0, 2.0])
            c = np.concatenate([rng.random(n - 2) + 0.5, [0.0, 0.0]])
        else:
            M = rng.standard_normal((n, n))
            Q = M @ M.T / n + 0.1 * np.eye(n)
            A = np.eye(n)
            G = rng.standard_normal((2, n))
            d = G @ np.ones(n)
            c = rng.standard_normal(n)
        prs.append(dict(Q=Q, c=c, A=A, b=np.zeros(n), cone_dims=[("R", n)], G=G, d=d, kwargs={}))
    one = _solve(prs, "threads", in_flight=1)
    lock = _solve(prs, "lockstep")
    _assert_identical(lock, one)
    assert [s.status for s in lock[1::2]] == ["Optimal"] * 3
    import ctypes as C
    from cipkkt import _lib as L
    st = (C.c_int * 3)()
    L.check(L.load().cip_lockstep_stats(st))
    assert (st[0], st[1]) == (1, 6) and st[2] >= 1, list(st)      # at least one LP met a bad pivot and left the group


def test_more_than_64_problems_two_groups():
    prs = []
    for seed in range(70):
        prs.append(_as_problem(P.random_mixed(n=16, nq=1, kq=4, p=2, seed=300 + seed)))
    one = _solve(prs, "threads", in_flight=4)
    lock = _solve(prs, "lockstep")
    _assert_identical(lock, one)


@pytest.mark.parametrize("count", [16, 40, 70])
def test_two_groups_side_by_side_equal_one_group_after_the_other(count):
    """Round 6: a lock-step call of at least 2 x 8 problems runs as TWO groups side by side, each on its own host thread and stream
    (cip_set_lockstep_split, default 2; lockstep.hip).  Per problem nothing may change -- same kernels, the solve block chosen from the
    size of the whole call -- so the results must equal those of one group after the other bit for bit, and the statistics must add
    up over the groups (70 problems: 2 x 35 side by side against 64 + 6 one after the other)."""
    import ctypes as C
    from cipkkt import _lib as L
    lib = L.load()
    prs = [_as_problem(P.random_mixed(n=24, nq=2, kq=5, p=3, seed=4200 + seed)) for seed in range(count)]
    prev = lib.cip_set_lockstep_split(1)
    try:
        one = _solve(prs, "lockstep")
        st1 = (C.c_int * 3)()
        lib.cip_lockstep_stats(st1)
        lib.cip_set_lockstep_split(2)
        two = _solve(prs, "lockstep")
        st2 = (C.c_int * 3)()
        lib.cip_lockstep_stats(st2)
    finally:
        lib.cip_set_lockstep_split(prev)
    _assert_identical(two, one)
    assert all(s.status == "Optimal" for s in two)
    assert st1[1] == st2[1] == count and st1[2] == st2[2]
    assert st1[0] == (count + 63) // 64 and st2[0] == 2


@pytest.mark.parametrize("count", [1, 2, 65])
def test_groups_of_one_problem(count):
    """a lock-step call of ONE problem, and 65 problems = a full group + a tail group of one (round-4 advisor finding: the
    tail group's max-steps wrote through a null pointer).  A group of one is not a batch for the kernels (no gather buffer):
    its max-steps take the one-problem read-back.  Mixed cones so that m > 0 and every max-step kind runs."""
    prs = [_as_problem(P.random_mixed(n=16, nq=1, kq=4, p=2, seed=800 + seed)) for seed in range(count)]
    one = _solve(prs, "threads", in_flight=4)
    lock = _solve(prs, "lockstep")
    assert all(s.status == "Optimal" for s in lock)
    _assert_identical(lock, one)
    auto = _solve(prs, "auto")                  # cip_conicip_mixed: one bin (a lone problem goes through the thread pool)
    _assert_identical(auto, one)


def test_group_of_one_box_qp_csr():
    """the config-5 family (CSR A = I, R cones only: lazy copy, fused element-wise kernels of solve4x4) as a group of one"""
    from cipkkt.workloads import c5_batch
    prs = c5_batch(count=1, n=384, seed=4100)
    one = _solve(prs, "threads", in_flight=1)
    lock = _solve(prs, "lockstep")
    _assert_identical(lock, one)
    assert lock[0].status == "Optimal"


def _csr_problem(n, m, nnz_extra, seed):
    """R-cone QP behind a CSR A with m >= n rows: an identity part plus `nnz_extra` further entries"""
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    Q = M @ M.T / n + 0.1 * np.eye(n)
    extra = rng.choice((m - n) * (n - 1), size=nnz_extra, replace=False)   # distinct positions away from column 0, which every extra row holds
    rows = list(range(n)) + list(range(n, m)) + [n + int(e) // (n - 1) for e in extra]
    cols = list(range(n)) + [0] * (m - n) + [1 + int(e) % (n - 1) for e in extra]
    vals = [1.0] * n + [0.5] * (m - n) + list(0.3 * rng.standard_normal(nnz_extra))
    A = sp.coo_matrix((vals, (rows, cols)), shape=(m, n)).tocsr()
    return dict(Q=Q, c=rng.standard_normal(n), A=A, b=-np.ones(m), cone_dims=[("R", m)], G=None, d=None, kwargs={})


def test_mixed_batch_csr_same_shape_differing_nnz():
    """CSR problems of one (n, m, p, cones) shape whose A differ in the number of non-zeros do not share a slab layout: the
    number of non-zeros is part of the binning key (host row pointers), so `cip_conicip_mixed` forms one lock-step group per
    count and nothing fails (round-4 advisor finding: the whole batch raised CIP_E_UNSUPPORTED)"""
    import ctypes as C
    from cipkkt import _lib as L
    prs = [_csr_problem(24, 30, e, 40 + i) for i, e in enumerate((5, 9, 5, 9, 5, 13))]
    nnz = [pr["A"].nnz for pr in prs]
    assert nnz[0] == nnz[2] == nnz[4] and nnz[1] == nnz[3] and len(set(nnz)) == 3, nnz
    one = _solve(prs, "threads", in_flight=1)
    mixed = _solve(prs, "auto")
    st = (C.c_int * 3)()
    L.load().cip_lockstep_stats(st)
    assert list(st)[:2] == [2, 5], list(st)           # groups of 3 and 2; the lone one through the thread pool
    assert all(s.status == "Optimal" for s in one), [s.status for s in one]
    _assert_identical(mixed, one)
    with pytest.raises(L.CipError) as ei:             # asked for as ONE lock-step call: refused, nothing written
        _solve(prs, "lockstep")
    assert ei.value.code == L.E_UNSUPPORTED


def test_mixed_batch_device_csr_differing_nnz_falls_back_to_the_thread_pool():
    """the same with the CSR arrays in DEVICE memory (no CIP_FLAG_CSR_HOST): the library cannot read the counts when it forms
    the bins, the lock-step group finds out at the slab-layout check -- and the bin's problems must then be solved by the
    thread pool instead of failing the batch"""
    import ctypes as C
    from cipkkt import _lib as L
    from cipkkt.driver import solution_from_result
    from cipkkt.kkt import make_problem
    lib = L.load()
    dev = torch.device("cuda:0")
    prs = [_csr_problem(24, 30, e, 60 + i) for i, e in enumerate((5, 11, 5))]
    assert prs[0]["A"].nnz == prs[2]["A"].nnz != prs[1]["A"].nnz
    k = len(prs)
    prev = lib.cip_set_solve_block_max(lib.cip_lockstep_solve_block_for(k))
    try:
        one = _solve_problems_native_plain(prs)
        structs = (L.CipProblem * k)()
        keep = []
        for i, pr in enumerate(prs):
            st, kp, _ = make_problem(pr["Q"], pr["A"], None, pr["cone_dims"], "schur", dev)
            csr = pr["A"].tocsr()
            csr.sort_indices()
            dv = [torch.from_numpy(np.ascontiguousarray(x, dtype=dt)).to(dev)
                  for x, dt in ((csr.indptr, np.int32), (csr.indices, np.int32), (csr.data, np.float64))]
            st.A_rowptr, st.A_colind, st.A_val = (C.c_void_p(x.data_ptr()) for x in dv)
            st.flags = L.FLAG_DEVICE_PTRS
            structs[i] = st
            keep += [kp, dv]
        torch.cuda.synchronize()
        vp = C.c_void_p * k
        cs = [np.ascontiguousarray(pr["c"]) for pr in prs]
        bs = [np.ascontiguousarray(pr["b"]) for pr in prs]
        ys = [np.zeros(24) for _ in prs]
        vs = [np.zeros(30) for _ in prs]
        zs = [np.zeros(1) for _ in prs]
        arr = lambda xs: vp(*[x.ctypes.data for x in xs])
        res = (L.CipResult * k)()
        opt = L.CipOptions(1e-6, 0.01, -1.0, -1.0, 3, 100, 0)
        L.check(lib.cip_conicip_mixed(k, structs, arr(cs), arr(bs), arr(zs), C.byref(opt), arr(ys), arr(zs), arr(vs), res, 2))
    finally:
        lib.cip_set_solve_block_max(prev)
    got = [solution_from_result(res[i], ys[i], zs[i][:0], vs[i]) for i in range(k)]
    assert all(s.status == "Optimal" for s in got), [s.status for s in got]
    _assert_identical(got, one)


def _solve_problems_native_plain(prs):
    from cipkkt.batch import _solve_problems_native
    return _solve_problems_native(prs, torch.device("cuda:0"), 1, "threads")


def test_unsupported_batches_fall_back():
    from cipkkt import _lib as L
    a = _as_problem(P.random_mixed(n=16, nq=1, kq=4, p=2, seed=1))
    b = _as_problem(P.random_mixed(n=18, nq=1, kq=4, p=2, seed=2))
    with pytest.raises(L.CipError) as ei:
        _solve([a, b], "lockstep")
    assert ei.value.code == L.E_UNSUPPORTED
    sols = _solve([a, b], "auto")                       # falls back to the thread pool
    assert [s.status for s in sols] == ["Optimal", "Optimal"]
    # S cones of matrix order >= 133 take chip-wide kernels with one workspace: not in lock-step
    r = 133
    k = r * (r + 1) // 2
    big = dict(Q=np.eye(4), c=np.ones(4), A=np.zeros((k, 4)), b=-P_vecm_identity(r), cone_dims=[("S", k)], G=None, d=None, kwargs={})
    with pytest.raises(L.CipError) as ei:
        _solve([big, big], "lockstep")
    assert ei.value.code == L.E_UNSUPPORTED


def test_mixed_shapes_group_by_shape():
    """`cip_conicip_mixed` (what mode "auto" calls): problems of three shapes interleaved -- the two shapes that occur more than
    once advance in lock-step (two groups, 5 + 3 problems), the singleton goes through the thread pool; every result
    bit-identical to the one-problem loop."""
    from cipkkt import _lib as L
    import ctypes as C
    prs = []
    for seed in range(5):
        prs.append(_as_problem(P.random_mixed(n=40, nq=3, kq=6, p=4, seed=500 + seed)))
    for seed in range(3):
        prs.append(_as_problem(P.random_mixed(n=24, nq=2, kq=5, p=3, seed=600 + seed)))
    prs.append(_as_problem(P.random_mixed(n=18, nq=1, kq=4, p=2, seed=700)))
    order = [0, 5, 1, 8, 6, 2, 3, 7, 4]
    prs = [prs[i] for i in order]
    one = _solve(prs, "threads", in_flight=1)
    mixed = _solve(prs, "auto")
    st = (C.c_int * 3)()
    L.load().cip_lockstep_stats(st)
    assert list(st) == [2, 8, 0]
    assert all(s.status == "Optimal" for s in one)
    _assert_identical(mixed, one)


def P_vecm_identity(r):
    from cipkkt.workloads import vecm_identity
    return vecm_identity(r)


def test_s_cones_in_lockstep():
    """small S cones (one workgroup per cone, sdp.hip) follow the batch dimension too: SDPs of one shape, mixed with R and Q
    cones, bit-identical to the one-problem loop"""
    from oracle.cones import vecm
    prs = []
    for seed in range(6):
        rng = np.random.default_rng(900 + seed)
        n, r1, r2 = 12, 6, 9
        k1, k2 = r1 * (r1 + 1) // 2, r2 * (r2 + 1) // 2
        cone_dims = [("S", k1), ("R", 5), ("Q", 4), ("S", k2)]
        m = k1 + 5 + 4 + k2
        M = rng.standard_normal((n, n))
        Q = M @ M.T / n + 0.1 * np.eye(n)
        A = rng.standard_normal((m, n)) * 0.3
        # strictly feasible at y = 0: b = -(interior point of every cone)
        b = -np.concatenate([vecm(np.eye(r1)), np.ones(5), np.array([2.0, 0.3, -0.2, 0.1]), vecm(np.eye(r2))])
        G = rng.standard_normal((2, n))
        prs.append(dict(Q=Q, c=rng.standard_normal(n), A=A, b=b, cone_dims=cone_dims, G=G, d=np.zeros(2), kwargs={}))
    one = _solve(prs, "threads", in_flight=1)
    lock = _solve(prs, "lockstep")
    assert all(s.status == "Optimal" for s in one), [s.status for s in one]
    _assert_identical(lock, one)


def test_mixed_batch_with_small_and_large_s_cones():
    """`cip_conicip_mixed` with S cones: three SDPs of one shape with small S cones (lock-step, one group), two with an S cone of
    matrix order 133 (same shape as each other, but the chip-wide kernels of sdp_large.hip have one workspace: thread pool) and a
    lone QP -- every result bit-identical to the one-problem loop, one lock-step group of three."""
    import ctypes as C
    from cipkkt import _lib as L
    from cipkkt.workloads import c4_sdp
    from oracle.cones import vecm
    prs = []
    for seed in range(3):
        rng = np.random.default_rng(950 + seed)
        n, r1 = 10, 7
        k1 = r1 * (r1 + 1) // 2
        M = rng.standard_normal((n, n))
        prs.append(dict(Q=M @ M.T / n + 0.1 * np.eye(n), c=rng.standard_normal(n), A=rng.standard_normal((k1 + 4, n)) * 0.3,
                        b=-np.concatenate([vecm(np.eye(r1)), np.ones(4)]), cone_dims=[("S", k1), ("R", 4)], G=None, d=None, kwargs={}))
    for seed in (31, 32):
        Q, c, A, b, K, G, d = c4_sdp(r=133, n=16, p=2, seed=seed)
        prs.append(dict(Q=Q, c=c, A=A, b=b, cone_dims=K, G=G, d=d, kwargs={}))
    prs.append(_as_problem(P.random_mixed(n=18, nq=1, kq=4, p=2, seed=701)))
    prs = [prs[i] for i in (3, 0, 5, 1, 4, 2)]
    one = _solve(prs, "threads", in_flight=1)
    mixed = _solve(prs, "auto", in_flight=2)
    st = (C.c_int * 3)()
    L.load().cip_lockstep_stats(st)
    assert list(st) == [1, 3, 0]
    assert all(s.status == "Optimal" for s in one), [s.status for s in one]
    _assert_identical(mixed, one)


@pytest.mark.parametrize("count", [8, 24])
def test_config5_reduced_lockstep_matches_threads(count):
    """BASELINE config 5 at reduced count: 8 / 24 x n = 2048 dense QPs generated in HBM (the per-rank shards at 8 GPUs, and a
    group whose one-launch-per-panel chain is four rounds of the chip: 24 x 42 long-lived workgroups on 256 CUs);
    lock-step == thread pool, bit for bit"""
    from cipkkt.workloads import c5_batch
    prs = c5_batch(count=count, n=2048, seed=4000, device=torch.device("cuda:0"))
    one = _solve(prs, "threads", in_flight=4)
    lock = _solve(prs, "lockstep")
    _assert_identical(lock, one)
    assert all(s.status == "Optimal" for s in lock)


def test_config5_all_64_problems_lockstep_matches_threads():
    """BASELINE config 5 at its stated size: 64 x n = 2048, one full lock-step group, against the thread pool bit for bit"""
    from cipkkt.workloads import c5_batch
    prs = c5_batch(count=64, n=2048, seed=4000, device=torch.device("cuda:0"))
    one = _solve(prs, "threads", in_flight=8)
    lock = _solve(prs, "lockstep")
    _assert_identical(lock, one)
    assert all(s.status == "Optimal" for s in lock)
    assert sum(s.n_factor for s in lock) == 627                 # the count bench.py divides by


def test_lockstep_against_the_oracle():
    """not only against the product's own one-problem loop: a lock-step group of mixed-cone problems against the numpy
    restatement of the reference (same status, iteration count, factorisations; iterates to 1e-6)"""
    from oracle.conicip import conicIP as oracle_conicIP
    from oracle.kktsolvers import kktsolver_2x2, pivot
    prs, refs = [], []
    for seed in range(5):
        Q, c, A, b, cone_dims, G, d, _ = P.random_mixed(n=30, nq=2, kq=5, p=3, seed=700 + seed)
        prs.append(dict(Q=Q, c=c, A=A, b=b, cone_dims=cone_dims, G=G, d=d, kwargs={}))
        refs.append(oracle_conicIP(Q, c, A, b, cone_dims, G, d, kktsolver=pivot(kktsolver_2x2)))
    lock = _solve(prs, "lockstep")
    for got, ref in zip(lock, refs):
        assert got.status == ref.status == "Optimal"
        assert got.Iter == ref.Iter
        for f in ("y", "w", "v"):
            a, bb = getattr(got, f), np.asarray(getattr(ref, f)).reshape(-1)
            assert np.linalg.norm(a - bb) <= 1e-6 * (1.0 + np.linalg.norm(bb)), f
