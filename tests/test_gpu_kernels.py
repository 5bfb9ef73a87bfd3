"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel family of the
KKT path against the oracle / numpy on the same seeded inputs, through the C ABI.
Tolerances are stated per test (fp64; the path is not bit-reproducible against
LAPACK because the elimination order differs, so parity is to rounding-level
backward error, as the reference's own tests compare solvers, runtests.jl:133-135)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import problems as P
from oracle import cones as ocones
from oracle.block import Block, Diagonal, SymWoodbury
from oracle.conicip import make_cone_ops
from oracle.kktsolvers import assemble3x3, kktsolver_2x2, kktsolver_qr, kktsolver_sparse, pivot, schur2x2

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64, device="cuda")


def colmajor_dev(M):
    """device buffer holding M column-major"""
    return torch.as_tensor(np.ascontiguousarray(np.asarray(M).T), dtype=torch.float64, device="cuda")


def from_colmajor(t, rows, cols):
    return t.cpu().numpy().reshape(cols, rows).T


@pytest.fixture(scope="module")
def lib():
    import cipkkt
    return cipkkt._lib.load()


# ----------------------------------------------------------------- MFMA GEMM
@pytest.mark.parametrize("M,N,K,lower", [(128, 128, 16, 0), (256, 128, 32, 0), (384, 384, 256, 1),
                                         (512, 256, 128, 0), (1024, 1024, 256, 1)])
def test_gemm_nt(lib, M, N, K, lower):
    from cipkkt import _lib as L
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((N, K))
    Cm = rng.standard_normal((M, N))
    dA, dB, dC = colmajor_dev(A), colmajor_dev(B), colmajor_dev(Cm)
    L.check(lib.cip_gemm_nt_dev(None, M, N, K, -1.0, dA.data_ptr(), M, dB.data_ptr(), N, dC.data_ptr(), M, lower))
    torch.cuda.synchronize()
    got = from_colmajor(dC, M, N)
    ref = Cm - A @ B.T
    if lower:
        # contract: the lower triangle is updated; tiles strictly above the diagonal are untouched
        # (the strictly-upper part of DIAGONAL tiles is scratch: never referenced by the LDL')
        for bi in range(M // 128):
            for bj in range(N // 128):
                blk = (slice(bi * 128, bi * 128 + 128), slice(bj * 128, bj * 128 + 128))
                if bi > bj:
                    np.testing.assert_allclose(got[blk], ref[blk], rtol=0, atol=1e-11 * K)
                elif bi == bj:
                    np.testing.assert_allclose(np.tril(got[blk]), np.tril(ref[blk]), rtol=0, atol=1e-11 * K)
                else:
                    np.testing.assert_array_equal(got[blk], Cm[blk])
    else:
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11 * K)


def test_gemm_mfma_layout_asymmetric(lib):
    """A = I-like pattern against an asymmetric B catches a transposed C write."""
    from cipkkt import _lib as L
    M = N = 128
    K = 128
    A = np.eye(M, K)
    B = np.arange(N * K, dtype=np.float64).reshape(N, K) / 7.0
    dA, dB = colmajor_dev(A), colmajor_dev(B)
    dC = torch.zeros(M * N, dtype=torch.float64, device="cuda")
    L.check(lib.cip_gemm_nt_dev(None, M, N, K, 1.0, dA.data_ptr(), M, dB.data_ptr(), N, dC.data_ptr(), M, 0))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(from_colmajor(dC, M, N), A @ B.T)


# ----------------------------------------------------------------- LDL'
def _ldlt_roundtrip(lib, Kmat, nbo):
    from cipkkt import _lib as L
    N = Kmat.shape[0]
    lib.cip_set_ldlt_outer_block(nbo)
    nbytes = C.c_size_t()
    L.check(lib.cip_ldlt_workspace_bytes(N, C.byref(nbytes)))
    ws = torch.zeros(nbytes.value // 8 + 8, dtype=torch.float64, device="cuda")
    dK = colmajor_dev(Kmat)
    info = C.c_int(-1)
    L.check(lib.cip_ldlt_factor_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), C.byref(info)))
    assert info.value == 0
    F = from_colmajor(dK, N, N)
    Lf = np.tril(F, -1) + np.eye(N)
    D = np.diag(F).copy()
    rec = (Lf * D[None, :]) @ Lf.T
    err = np.abs(np.tril(rec - Kmat)).max() / np.abs(Kmat).max()
    rng = np.random.default_rng(5)
    x_true = rng.standard_normal(N)
    rhs = Kmat @ x_true
    drhs = dev(rhs)
    L.check(lib.cip_ldlt_solve_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), drhs.data_ptr()))
    torch.cuda.synchronize()
    x = drhs.cpu().numpy()
    res = np.linalg.norm(Kmat @ x - rhs) / (np.linalg.norm(Kmat, 2) * np.linalg.norm(x))
    lib.cip_set_ldlt_outer_block(0)          # back to automatic
    return err, res, D


@pytest.mark.parametrize("N,nbo", [(128, 128), (256, 256), (384, 256), (512, 128), (640, 256), (1024, 512)])
def test_ldlt_spd(lib, N, nbo):
    rng = np.random.default_rng(N)
    M = rng.standard_normal((N, N))
    Kmat = M @ M.T / N + np.eye(N)
    err, res, D = _ldlt_roundtrip(lib, Kmat, nbo)
    assert err < 1e-13, "||L D L' - K|| / ||K|| = %g" % err
    assert res < 1e-14, "normwise backward error of the solve = %g" % res
    assert np.all(D > 0)


def test_ldlt_quasidefinite(lib):
    """[S G'; G 0] with S > 0: n positive then p negative pivots, no pivoting needed."""
    rng = np.random.default_rng(11)
    n, p = 300, 84
    M = rng.standard_normal((n, n))
    S = M @ M.T / n + np.eye(n)
    G = rng.standard_normal((p, n))
    Kmat = np.zeros((384, 384))
    Kmat[:n, :n] = S
    Kmat[n:, :n] = G
    Kmat[:n, n:] = G.T
    err, res, D = _ldlt_roundtrip(lib, Kmat, 256)
    assert err < 1e-12 and res < 1e-13
    assert np.all(D[:n] > 0) and np.all(D[n:] < 0)


@pytest.mark.parametrize("N,bmax", [(384, 128), (640, 128), (1536, 512), (2048, 256), (2048, 512), (3072, 1024)])
def test_ldlt_solve_one_launch_per_block_step(lib, N, bmax):
    """The triangular sweeps with the pre-multiplied neighbour blocks (one launch per block step, cip_set_solve_fused) against the
    two-launch form and against numpy: same backward error, results equal to rounding.  Quasi-definite matrix (both pivot signs)."""
    from cipkkt import _lib as L
    rng = np.random.default_rng(N + bmax)
    n = N - N // 4
    M = rng.standard_normal((n, n))
    Kmat = np.zeros((N, N))
    Kmat[:n, :n] = M @ M.T / n + np.eye(n)
    G = rng.standard_normal((N - n, n))
    Kmat[n:, :n] = G
    Kmat[:n, n:] = G.T
    x_true = rng.standard_normal(N)
    rhs = Kmat @ x_true
    prev_b = lib.cip_set_solve_block_max(bmax)
    prev_f = lib.cip_set_solve_fused(-1)
    out = {}
    try:
        for mode in (0, 2):
            lib.cip_set_solve_fused(mode)
            nbytes = C.c_size_t()
            L.check(lib.cip_ldlt_workspace_bytes(N, C.byref(nbytes)))
            ws = torch.zeros(nbytes.value // 8 + 8, dtype=torch.float64, device="cuda")
            dK = colmajor_dev(Kmat)
            info = C.c_int(-1)
            L.check(lib.cip_ldlt_factor_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), C.byref(info)))
            assert info.value == 0
            xs = []
            for rep in range(2):                      # (twice: the sweeps must leave the prepared blocks alone)
                drhs = dev(rhs)
                L.check(lib.cip_ldlt_solve_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), drhs.data_ptr()))
                torch.cuda.synchronize()
                xs.append(drhs.cpu().numpy())
            np.testing.assert_array_equal(xs[0], xs[1])
            out[mode] = xs[0]
    finally:
        lib.cip_set_solve_fused(prev_f)
        lib.cip_set_solve_block_max(prev_b)
    nK = np.linalg.norm(Kmat, 2)
    for mode, x in out.items():
        res = np.linalg.norm(Kmat @ x - rhs) / (nK * np.linalg.norm(x))
        assert res < 1e-14, "mode %d: normwise backward error %g" % (mode, res)
    assert np.linalg.norm(out[0] - out[2]) <= 1e-11 * np.linalg.norm(out[0])


def test_ldlt_reports_zero_pivot(lib):
    from cipkkt import _lib as L
    N = 128
    Kmat = np.eye(N)
    Kmat[5, 5] = 0.0
    nbytes = C.c_size_t()
    L.check(lib.cip_ldlt_workspace_bytes(N, C.byref(nbytes)))
    ws = torch.zeros(nbytes.value // 8 + 8, dtype=torch.float64, device="cuda")
    dK = colmajor_dev(Kmat)
    info = C.c_int(0)
    L.check(lib.cip_ldlt_factor_dev(None, dK.data_ptr(), N, N, ws.data_ptr(), C.byref(info)))
    assert info.value == 6


# ----------------------------------------------------------------- cone kernels vs oracle
CONESETS = [
    [("R", 5)],
    [("Q", 3)],
    [("Q", 8)] * 5,
    [("R", 7), ("Q", 4), ("R", 3), ("Q", 9)],
    [("R", 5000), ("Q", 700)],
    # sub-wavefront packing of small Q cones (cones.hip): runs of mixed sizes (pack width = next power of two of the
    # run's largest cone), a run split by an R cone, the boundary 64 / 65, more cones than one workgroup holds
    [("Q", 3), ("Q", 8), ("Q", 5), ("R", 7), ("Q", 64), ("Q", 65), ("Q", 2), ("Q", 33)],
    [("Q", 8)] * 70 + [("Q", 2)] * 3,
]


def interior_point(cone_dims, rng):
    xs = []
    for t, k in cone_dims:
        if t == "R":
            xs.append(rng.random(k) + 0.1)
        else:
            x = rng.standard_normal(k)
            x[0] = np.linalg.norm(x[1:]) + rng.random() + 0.1
            xs.append(x)
    return np.concatenate(xs)


def make_system(cone_dims, n=6, p=0, seed=0, sparse=False, route="schur"):
    import cipkkt
    rng = np.random.default_rng(seed)
    m = sum(k for _, k in cone_dims)
    M = rng.standard_normal((n, n))
    Q = M @ M.T / n + 0.5 * np.eye(n)
    A = rng.standard_normal((m, n))
    if sparse:
        A = A * (rng.random((m, n)) < 0.3)
        A[np.arange(m), rng.integers(0, n, m)] = 1.0
        A = sp.csr_matrix(A)
    G = rng.standard_normal((p, n)) if p else None
    return cipkkt.KKTSystem(Q, A, G, cone_dims, route=route), Q, A, G


@pytest.mark.parametrize("cone_dims", CONESETS, ids=[str(i) for i in range(len(CONESETS))])
def test_cone_ops(cone_dims):
    from cipkkt import OP_F, OP_FT, OP_FINV, OP_FINVT
    rng = np.random.default_rng(len(cone_dims))
    ks, *_ = make_system(cone_dims)
    m = ks.m
    maxstep, nt_scaling, cone_div, cone_prod = make_cone_ops(cone_dims)
    v = interior_point(cone_dims, rng)
    s = interior_point(cone_dims, rng)
    dv, dsv = dev(v), dev(s)
    lam = torch.zeros(m, dtype=torch.float64, device="cuda")
    ks.set_scaling_from_iterate(dv, dsv, lam)
    F = nt_scaling(v, s)
    packed_ref = ks.pack_scaling(F)
    np.testing.assert_allclose(ks.get_scaling_packed(), packed_ref, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(lam.cpu().numpy(), F.mul(v), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(lam.cpu().numpy(), F.inv_adjoint().mul(s), rtol=1e-9, atol=1e-10)
    x = rng.standard_normal(m)
    dx = dev(x)
    out = torch.zeros_like(dx)
    for mode, ref in ((OP_F, F.mul(x)), (OP_FT, F.tmul(x)), (OP_FINV, F.inv().mul(x)),
                      (OP_FINVT, F.inv_adjoint().mul(x))):
        ks.apply_F(mode, dx, out)
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-10, atol=1e-11)
    # in-place apply
    y = dx.clone()
    ks.apply_F(OP_F, y, y)
    np.testing.assert_allclose(y.cpu().numpy(), F.mul(x), rtol=1e-10, atol=1e-11)
    # Jordan product / division
    yv = interior_point(cone_dims, rng)
    dy = dev(yv)
    ks.cone_prod(dx, dy, out)
    np.testing.assert_allclose(out.cpu().numpy(), cone_prod(x, yv), rtol=1e-12, atol=1e-12)
    ks.cone_div(dx, dy, out)
    np.testing.assert_allclose(out.cpu().numpy(), cone_div(x, yv), rtol=1e-10, atol=1e-11)
    # max step, both variants
    d = rng.standard_normal(m)
    for scale in (1.0, 1.0 / 0.99):
        got = ks.maxstep(dv, dev(d), scale)
        ref = maxstep(v, d * scale)
        assert got == pytest.approx(ref, rel=1e-10) or (np.isinf(ref) and np.isinf(got))
    assert ks.maxstep(dv, None) == 0.0
    xo = x.copy()
    assert ks.maxstep(dev(xo), None) == pytest.approx(maxstep(xo, None), rel=1e-12)
    # identity
    e = torch.zeros(m, dtype=torch.float64, device="cuda")
    ks.cone_identity(e)
    from oracle.conicip import cone_identity
    np.testing.assert_array_equal(e.cpu().numpy(), cone_identity(cone_dims)[0])
    # identity scaling: F = I
    ks.set_scaling_identity()
    ks.apply_F(OP_FINV, dx, out)
    np.testing.assert_allclose(out.cpu().numpy(), x, rtol=1e-15, atol=0)
    ks.close()


# ----------------------------------------------------------------- assembly + factor + solve3x3
ASM_CASES = [
    dict(cone_dims=[("R", 9)], n=9, p=0),
    dict(cone_dims=[("R", 30), ("Q", 6), ("Q", 4)], n=17, p=3),
    dict(cone_dims=[("Q", 8)] * 20, n=150, p=10),
    dict(cone_dims=[("R", 100), ("Q", 40)], n=130, p=7),
    dict(cone_dims=[("Q", 5), ("Q", 16), ("R", 4), ("Q", 65), ("Q", 3)] + [("Q", 8)] * 40, n=90, p=4),
    # large-k second-order cone with a CSR A: the rank-1 column of the Schur complement (SURVEY 8f rank 3;
    # the reference's benchmark/profile.jl single SOC n = 500)
    dict(cone_dims=[("Q", 501)], n=500, p=0),
]


def oracle_F(cone_dims, rng):
    _, nt_scaling, _, _ = make_cone_ops(cone_dims)
    v = interior_point(cone_dims, rng)
    s = interior_point(cone_dims, rng)
    return nt_scaling(v, s)


@pytest.mark.parametrize("case", ASM_CASES, ids=[str(i) for i in range(len(ASM_CASES))])
@pytest.mark.parametrize("sparse", [False, True], ids=["denseA", "csrA"])
@pytest.mark.parametrize("route", ["schur", "full3x3"])
def test_assembly_factor_solve(case, sparse, route):
    rng = np.random.default_rng(42)
    cone_dims, n, p = case["cone_dims"], case["n"], case["p"]
    ks, Q, A, G = make_system(cone_dims, n=n, p=p, seed=3, sparse=sparse, route=route)
    m = ks.m
    F = oracle_F(cone_dims, rng)
    ks.set_scaling_packed(ks.pack_scaling(F, F.inv_adjoint()))
    ks.assemble_only()
    Kd = ks.kkt_matrix()
    Gd = np.zeros((0, n)) if G is None else G
    if route == "schur":
        ref = schur2x2(Q, A, Gd, F)
        N = n + p
        got = np.tril(Kd[:N, :N])
        np.testing.assert_allclose(got, np.tril(ref), rtol=1e-10, atol=1e-10 * np.abs(ref).max())
    else:
        Z = assemble3x3(Q, A, Gd, F)             # [Q G' -A'; G 0 0; A 0 F'F]  order (y, w, v)
        N = n + p + m
        # device order is (v, y, w), symmetrised: [-F'F -A 0; -A' Q G'; 0 G 0]
        perm = np.concatenate([np.arange(n + p, n + p + m), np.arange(n + p)])
        Zs = Z.copy()
        Zs[n + p:, :] *= -1.0                    # negate the third block row -> symmetric
        ref = Zs[np.ix_(perm, perm)]
        np.testing.assert_allclose(ref, ref.T, atol=1e-12)
        np.testing.assert_allclose(np.tril(Kd[:N, :N]), np.tril(ref), rtol=1e-10, atol=1e-10 * np.abs(ref).max())
    # padding is an identity block
    np.testing.assert_array_equal(np.tril(Kd[N:, :]), np.tril(np.eye(Kd.shape[0])[N:, :]))
    # factor + level-3 solve against all three reference solvers (oracle)
    ks.factor(check=True)
    x, y, z = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    a, b, c = ks.solve3x3(x, y, z)
    Z = assemble3x3(Q, A, Gd, F)
    sol = np.concatenate([a, b, c])
    rhs = np.concatenate([x, y, z])
    berr = np.linalg.norm(Z @ sol - rhs) / (np.linalg.norm(Z, 2) * np.linalg.norm(sol) + np.linalg.norm(rhs))
    assert berr < 1e-12, "backward error %g" % berr
    for kk in (kktsolver_qr, kktsolver_sparse, pivot(kktsolver_2x2)):
        ra, rb, rc = kk(Q, A, Gd, cone_dims)(F, F.inv_adjoint())(x, y, z)
        ref = np.concatenate([ra, rb, rc])
        assert np.linalg.norm(sol - ref) / np.linalg.norm(ref) < 1e-8
    ks.close()


def test_plugin_closure_matches_reference_interface():
    """kktsolver_hip(Q,A,G,cone_dims)(F,F_invT)(x,y,z) -- same call shape as the reference's solvers."""
    import cipkkt
    rng = np.random.default_rng(9)
    Q, c, A, b, cone_dims, G, d, _ = P.random_mixed(n=40, nq=3, kq=6, p=4)
    F = oracle_F(cone_dims, rng)
    n, m, p = Q.shape[0], A.shape[0], G.shape[0]
    x, y, z = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    ref = np.concatenate(kktsolver_qr(Q, A, G, cone_dims)(F, F.inv_adjoint())(x, y, z))
    for solver in (cipkkt.kktsolver_hip, cipkkt.kktsolver_hip_full3x3):
        gen = solver(Q, A, G, cone_dims)
        got = np.concatenate(gen(F, F.inv_adjoint())(x, y, z))
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-9
        # a second level-2 call with another scaling reuses level 1
        F2 = oracle_F(cone_dims, rng)
        got2 = np.concatenate(gen(F2, F2.inv_adjoint())(x, y, z))
        ref2 = np.concatenate(kktsolver_qr(Q, A, G, cone_dims)(F2, F2.inv_adjoint())(x, y, z))
        assert np.linalg.norm(got2 - ref2) / np.linalg.norm(ref2) < 1e-9
        gen.system.close()


def test_solve4x4_matches_oracle():
    from oracle.conicip import V4
    rng = np.random.default_rng(21)
    Q, c, A, b, cone_dims, G, d, _ = P.random_mixed(n=30, nq=2, kq=5, p=3)
    import cipkkt
    ks = cipkkt.KKTSystem(Q, A, G, cone_dims)
    n, m, p = ks.n, ks.m, ks.p
    maxstep, nt_scaling, cone_div, cone_prod = make_cone_ops(cone_dims)
    v, s = interior_point(cone_dims, rng), interior_point(cone_dims, rng)
    F = nt_scaling(v, s)
    lam = F.mul(v)
    r = rng.standard_normal(n + p + 2 * m)
    # oracle solve4x4 (src/ConicIP.jl:684-692)
    s3 = kktsolver_qr(Q, A, G, cone_dims)(F, F.inv_adjoint())
    q = cone_div(r[n + p + m:], lam)
    t1 = F.tmul(q)
    dy, dw, dv = s3(r[:n], r[n:n + p], r[n + p:n + p + m] + t1)
    ds = t1 - F.tmul(F.mul(dv))
    ref = np.concatenate([dy, dw, dv, ds])
    dlam = torch.zeros(m, dtype=torch.float64, device="cuda")
    ks.set_scaling_from_iterate(dev(v), dev(s), dlam)
    ks.factor(check=True)
    dz = torch.zeros(n + p + 2 * m, dtype=torch.float64, device="cuda")
    ks.solve4x4_dev(dlam, dev(r), dz)
    got = dz.cpu().numpy()
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-9
    ks.close()


@pytest.mark.parametrize("sparse,n,m,p", [(True, 300, 300, 0), (True, 257, 400, 9), (False, 200, 330, 7), (True, 2048, 2048, 0)])
def test_solve4x4_fused_r_path_bitwise(sparse, n, m, p):
    """All cones R: cip_solve4x4 runs ONE element-wise kernel in front of the triangular sweeps and one behind them
    (vecops.hip: k_s4_pre_r / k_s4_post_r) instead of 17 small launches.  Same operations on every element in the same
    order (no contraction across what used to be kernel boundaries): the step must equal the unfused path's bit for bit
    (CIP_S4_FUSED=0, read when a handle's first solve4x4 runs), and the oracle's to rounding."""
    import os
    import cipkkt
    from oracle.block import Block, Diagonal
    rng = np.random.default_rng(n + m + p)
    M = rng.standard_normal((n, n))
    Q = M.T @ M / n + 0.05 * np.eye(n)
    if sparse:
        import scipy.sparse as sp
        A = sp.random(m, n, density=0.02, random_state=3, format="csr") + (sp.identity(n, format="csr") if m == n else
                                                                              sp.vstack([sp.identity(n), sp.csr_matrix((m - n, n))]) if m > n else 0)
        A = sp.csr_matrix(A)
        Ad = A.toarray()
    else:
        A = rng.standard_normal((m, n)) / np.sqrt(n)
        Ad = A
    G = rng.standard_normal((p, n)) if p else None
    K = [("R", m // 3), ("R", m - m // 3)]
    v, sv = rng.random(m) + 0.05, rng.random(m) + 0.05
    r = rng.standard_normal(n + p + 2 * m)
    outs = []
    for fused in ("0", "1"):
        os.environ["CIP_S4_FUSED"] = fused
        ks = cipkkt.KKTSystem(Q, A, G, K)
        lam = torch.zeros(m, dtype=torch.float64, device="cuda")
        ks.set_scaling_from_iterate(dev(v), dev(sv), lam)
        ks.factor(check=True)
        dz = torch.zeros(n + p + 2 * m, dtype=torch.float64, device="cuda")
        ks.solve4x4_dev(lam, dev(r), dz)
        torch.cuda.synchronize()
        outs.append(dz.cpu().numpy())
        ks.close()
    os.environ.pop("CIP_S4_FUSED")
    np.testing.assert_array_equal(outs[0], outs[1])
    # and it is the reference's solve4x4 (src/ConicIP.jl:684-692)
    f = np.sqrt(sv / v)
    lamh = f * v
    t1 = f * (r[n + p + m:] / lamh)
    Gd = G if p else np.zeros((0, n))
    Kmat = np.block([[Q, Gd.T, -Ad.T], [Gd, np.zeros((p, p)), np.zeros((p, m))], [Ad, np.zeros((m, p)), np.diag(f * f)]])
    sol = np.linalg.solve(Kmat, np.concatenate([r[:n], r[n:n + p], r[n + p:n + p + m] + t1]))
    dv = sol[n + p:]
    ref = np.concatenate([sol[:n + p], dv, t1 - f * (f * dv)])
    assert np.linalg.norm(outs[1] - ref) / np.linalg.norm(ref) < 1e-9


@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_lazy_copy_of_Q_gives_the_same_factor_and_step(n):
    """Box-QP family (A = I as CSR, R cones, p = 0, n a multiple of 128): cip_factor copies only the first outer block's columns
    of Q into K and the first trailing update reads the rest of its C operand from Q (assemble.hip / EPI_LAZYC).  The factor
    and a solve4x4 step must equal the eager copy's bit for bit."""
    import cipkkt
    from cipkkt import workloads as W
    Q, c, A, b, K = W.c2_problem(n, seed=77, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    v = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
    s = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
    r = torch.randn(3 * n, generator=g, dtype=torch.float64, device="cuda")
    outs, mats = [], []
    for lazy in (0, 1):
        ks = cipkkt.KKTSystem(Q, A, None, K)
        prev = ks.lib.cip_set_lazy_copy(lazy)
        try:
            lam = torch.zeros(n, dtype=torch.float64, device="cuda")
            ks.set_scaling_from_iterate(v, s, lam)
            ks.factor(check=True)
            dz = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
            ks.solve4x4_dev(lam, r, dz)
            torch.cuda.synchronize()
            outs.append(dz.cpu().numpy())
            mats.append(np.tril(ks.kkt_matrix()))
        finally:
            ks.lib.cip_set_lazy_copy(prev)
            ks.close()
    np.testing.assert_array_equal(mats[0], mats[1])
    np.testing.assert_array_equal(outs[0], outs[1])


def test_factor_graph_follows_the_lazy_copy_state_and_the_regularisation(monkeypatch):
    """with CIP_GRAPH=1 the recorded factorisation bakes in whether the first trailing update reads its C
    operand from Q (lazy copy) or from K; a replay under the other state would read stale data.  One handle, graph replay on:
    lazy -> eager -> lazy -> regularised (the eager assembly + a shifted diagonal) -- after every switch the factor and a
    solve4x4 step must equal those of a fresh handle WITHOUT graphs in the same state, bit for bit."""
    import cipkkt
    from cipkkt import workloads as W
    n = 1024
    Q, c, A, b, K = W.c2_problem(n, seed=78, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    v = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
    s = torch.rand(n, generator=g, dtype=torch.float64, device="cuda") + 0.05
    r = torch.randn(3 * n, generator=g, dtype=torch.float64, device="cuda")

    def step(ks):
        lam = torch.zeros(n, dtype=torch.float64, device="cuda")
        ks.set_scaling_from_iterate(v, s, lam)
        ks.factor(check=True)
        dz = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
        ks.solve4x4_dev(lam, r, dz)
        torch.cuda.synchronize()
        return np.tril(ks.kkt_matrix()), dz.cpu().numpy()

    def reference(lazy, reg):
        monkeypatch.delenv("CIP_GRAPH", raising=False)
        ks = cipkkt.KKTSystem(Q, A, None, K)
        prev = ks.lib.cip_set_lazy_copy(lazy)
        try:
            if reg:
                cipkkt._lib.check(ks.lib.cip_set_regularization(ks.h, 1e-12, 1))
            return step(ks)
        finally:
            ks.lib.cip_set_lazy_copy(prev)
            ks.close()

    refs = {(lz, rg): reference(lz, rg) for lz, rg in ((1, 0), (0, 0), (0, 1))}
    monkeypatch.setenv("CIP_GRAPH", "1")
    ks = cipkkt.KKTSystem(Q, A, None, K)
    prev = ks.lib.cip_set_lazy_copy(1)
    try:
        for lz, rg in ((1, 0), (0, 0), (1, 0), (1, 0), (0, 0)):
            ks.lib.cip_set_lazy_copy(lz)
            Kf, dz = step(ks)
            np.testing.assert_array_equal(Kf, refs[(lz, rg)][0], err_msg="factor, lazy=%d" % lz)
            np.testing.assert_array_equal(dz, refs[(lz, rg)][1], err_msg="step, lazy=%d" % lz)
        cipkkt._lib.check(ks.lib.cip_set_regularization(ks.h, 1e-12, 1))      # regularised: the assembly copies eagerly whatever the knob
        ks.lib.cip_set_lazy_copy(1)
        Kf, dz = step(ks)
        np.testing.assert_array_equal(Kf, refs[(0, 1)][0], err_msg="regularised factor under graph replay")
        np.testing.assert_array_equal(dz, refs[(0, 1)][1])
    finally:
        ks.lib.cip_set_lazy_copy(prev)
        ks.close()


def test_gemv_and_dots():
    from cipkkt import MAT_A, MAT_G, MAT_Q
    rng = np.random.default_rng(2)
    for sparse in (False, True):
        ks, Q, A, G = make_system([("R", 33), ("Q", 10)], n=21, p=5, seed=8, sparse=sparse)
        Ad = A.toarray() if sparse else A
        x, w, v = rng.standard_normal(21), rng.standard_normal(5), rng.standard_normal(43)
        out_n = dev(rng.standard_normal(21))
        o0 = out_n.cpu().numpy().copy()
        ks.gemv(MAT_Q, 0, 2.0, dev(x), 0.5, out_n)
        np.testing.assert_allclose(out_n.cpu().numpy(), 2 * Q @ x + 0.5 * o0, rtol=1e-12, atol=1e-12)
        out_m = torch.zeros(43, dtype=torch.float64, device="cuda")
        ks.gemv(MAT_A, 0, 1.0, dev(x), 0.0, out_m)
        np.testing.assert_allclose(out_m.cpu().numpy(), Ad @ x, rtol=1e-12, atol=1e-12)
        ks.gemv(MAT_A, 1, -1.0, dev(v), 0.0, out_n)
        np.testing.assert_allclose(out_n.cpu().numpy(), -Ad.T @ v, rtol=1e-12, atol=1e-12)
        out_p = torch.zeros(5, dtype=torch.float64, device="cuda")
        ks.gemv(MAT_G, 0, 1.0, dev(x), 0.0, out_p)
        np.testing.assert_allclose(out_p.cpu().numpy(), G @ x, rtol=1e-12, atol=1e-12)
        ks.gemv(MAT_G, 1, 1.0, dev(w), 1.0, out_n)
        np.testing.assert_allclose(out_n.cpu().numpy(), -Ad.T @ v + G.T @ w, rtol=1e-12, atol=1e-12)
        big = rng.standard_normal(100003)
        d = ks.dots([(dev(x), dev(x)), (dev(big), dev(big)), (dev(v), dev(v)[:0])])
        np.testing.assert_allclose(d, [x @ x, big @ big, 0.0], rtol=1e-13)
        ks.close()


def test_plain_c_abi_sequence_as_the_julia_shim_calls_it(lib):
    """cip_create -> cip_set_scaling_packed -> cip_factor -> cip_solve3x3 -> cip_destroy with HOST pointers only,
    exactly the sequence of the ccall shim in INTEGRATION.md (levels 1-3 of src/ConicIP.jl:667,682,688)."""
    from cipkkt import _lib as L
    rng = np.random.default_rng(77)
    cone_dims = [("R", 6), ("Q", 5), ("R", 2)]
    n, p = 11, 3
    m = sum(k for _, k in cone_dims)
    M = rng.standard_normal((n, n))
    Q = np.asfortranarray(M @ M.T / n + np.eye(n))
    A = np.asfortranarray(rng.standard_normal((m, n)))
    G = np.asfortranarray(rng.standard_normal((p, n)))
    F = oracle_F(cone_dims, rng)
    ctype = (C.c_int * 3)(0, 1, 0)
    cdim = (C.c_int * 3)(6, 5, 2)
    h = C.c_void_p()
    L.check(lib.cip_create(n, m, p, 3, ctype, cdim, Q.ctypes.data, A.ctypes.data, G.ctypes.data, 0, C.byref(h)))
    assert lib.cip_scaling_packed_len(h) == 6 + (1 + 5) + 2
    packed = np.concatenate([F.Blocks[0].diag, [-F.Blocks[1].A[0]], F.Blocks[1].B[:, 0], F.Blocks[2].diag])
    L.check(lib.cip_set_scaling_packed(h, packed.ctypes.data))
    L.check(lib.cip_factor(h))
    L.check(lib.cip_check_factor(h))
    x, y, z = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    a, b, c = np.empty(n), np.empty(p), np.empty(m)
    L.check(lib.cip_solve3x3(h, x.ctypes.data, y.ctypes.data, z.ctypes.data, a.ctypes.data, b.ctypes.data, c.ctypes.data))
    # the defining equations (src/ConicIP.jl:443-447)
    FtF = F.square().matrix()
    assert np.linalg.norm(Q @ a + G.T @ b - A.T @ c - x) < 1e-10
    assert np.linalg.norm(G @ a - y) < 1e-10
    assert np.linalg.norm(A @ a + FtF @ c - z) < 1e-9
    # errors are reported, not swallowed
    assert lib.cip_solve3x3(None, x.ctypes.data, y.ctypes.data, z.ctypes.data, a.ctypes.data, b.ctypes.data,
                            c.ctypes.data) != 0
    bad = C.c_void_p()
    rc = lib.cip_create(n, m, p, 3, ctype, (C.c_int * 3)(6, 5, 3), Q.ctypes.data, A.ctypes.data, G.ctypes.data, 0,
                        C.byref(bad))
    assert rc != 0 and b"cone_dims" in lib.cip_last_error()
    L.check(lib.cip_destroy(h))


def test_singular_schur_block_is_regularised_and_refined():
    """Q = 0 and free variables that only the equalities pin: S = A'(F'F)^-1 A is singular although the KKT matrix is
    not, so the static-order LDL' meets a zero pivot.  The handle switches to the regularised factorisation and
    solve3x3 refines against the true operator: the answer must match the reference-faithful null-space QR solver."""
    import cipkkt
    rng = np.random.default_rng(12)
    n, p, k = 30, 12, 14                                  # 14 bounded variables, 16 free ones, 12 equalities
    Q = np.zeros((n, n))
    A = np.zeros((k, n))
    A[np.arange(k), np.arange(k)] = 1.0
    G = rng.standard_normal((p, n))
    cone_dims = [("R", k)]
    _, nt_scaling, _, _ = make_cone_ops(cone_dims)
    F = nt_scaling(rng.random(k) + 0.1, rng.random(k) + 0.1)
    x, y, z = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(k)
    # (the 16 free directions need 16 <= ... pinned: G has 12 rows, so add curvature on 4 of them to keep K non-singular)
    Q[k:k + 4, k:k + 4] = np.eye(4)
    ref = np.concatenate(kktsolver_qr(Q, A, G, cone_dims)(F, F.inv_adjoint())(x, y, z))
    for route in ("schur", "full3x3"):
        gen = cipkkt.kktsolver_hip(Q, A, G, cone_dims) if route == "schur" else cipkkt.kktsolver_hip_full3x3(Q, A, G, cone_dims)
        got = np.concatenate(gen(F, F.inv_adjoint())(x, y, z))
        rel = C.c_double()
        times = C.c_int()
        gen.system.lib.cip_get_regularization(gen.system.h, C.byref(rel), C.byref(times))
        assert times.value == 1 and rel.value > 0, "the factorisation should have switched to the regularised form"
        np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-9)
        gen.system.close()


def test_update_problem_reuses_the_handle():
    """cip_update_problem (level 1 again on an existing handle: what the batch workers do between problems) must give
    exactly what a fresh handle gives, and refuse another shape."""
    import cipkkt
    from cipkkt import _lib as L
    from cipkkt.kkt import make_problem
    rng = np.random.default_rng(31)
    n, m = 96, 96
    K = [("R", 60), ("Q", 36)]

    def problem(seed):
        r = np.random.default_rng(seed)
        M = r.standard_normal((n, n))
        return M @ M.T / n + np.eye(n), sp.identity(m, format="csr") * (1.0 + 0.1 * seed), r.standard_normal((3, n))

    Q1, A1, G1 = problem(1)
    Q2, A2, G2 = problem(2)
    ks = cipkkt.KKTSystem(Q1, A1, G1, K)
    pr, keep, _ = make_problem(Q2, A2, G2, K, "schur", ks.device)
    torch.cuda.synchronize()
    L.check(ks.lib.cip_update_problem(ks.h, C.byref(pr)))
    fresh = cipkkt.KKTSystem(Q2, A2, G2, K)
    v, s = interior_point(K, rng), interior_point(K, rng)
    x, y, z = rng.standard_normal(n), rng.standard_normal(3), rng.standard_normal(m)
    outs = []
    for sysm in (ks, fresh):
        sysm.set_scaling_from_iterate(dev(v), dev(s))
        sysm.factor()
        outs.append(np.concatenate(sysm.solve3x3(x, y, z)))
    np.testing.assert_array_equal(outs[0], outs[1])
    pr_bad, keep2, _ = make_problem(Q2[:50, :50], sp.identity(50, format="csr"), None, [("R", 50)], "schur", ks.device)
    assert ks.lib.cip_update_problem(ks.h, C.byref(pr_bad)) == -1
    ks.close(); fresh.close()


def test_single_entry_and_tiny_cones():
    """Q cones of dimension 1 and 2, an R cone of one element, beside larger ones (pack width 1, 2, ...)."""
    import cipkkt
    K = [("Q", 1), ("Q", 1), ("R", 1), ("Q", 2), ("Q", 1), ("Q", 7)]
    rng = np.random.default_rng(5)
    m = sum(k for _, k in K)
    ks, Q, A, G = make_system(K, n=5)
    maxstep, nt_scaling, cone_div, cone_prod = make_cone_ops(K)
    v, s = interior_point(K, rng), interior_point(K, rng)
    lam = torch.zeros(m, dtype=torch.float64, device="cuda")
    ks.set_scaling_from_iterate(dev(v), dev(s), lam)
    F = nt_scaling(v, s)
    np.testing.assert_allclose(lam.cpu().numpy(), F.mul(v), rtol=1e-12, atol=1e-14)
    x = rng.standard_normal(m)
    out = torch.zeros(m, dtype=torch.float64, device="cuda")
    ks.apply_F(cipkkt.OP_FINVT, dev(x), out)
    np.testing.assert_allclose(out.cpu().numpy(), F.inv_adjoint().mul(x), rtol=1e-11, atol=1e-13)
    ks.cone_div(dev(x), dev(s), out)
    np.testing.assert_allclose(out.cpu().numpy(), cone_div(x, s), rtol=1e-11, atol=1e-13)
    d = rng.standard_normal(m)
    assert ks.maxstep(dev(v), dev(d)) == pytest.approx(maxstep(v, d), rel=1e-11)
    ks.close()


def test_symmetric_matvec_from_lower_tiles(lib):
    """Q x through cip_gemv_dev for n a multiple of 128 (>= 2048): the two-launch form that reads only the tiles on and below
    the diagonal (vecops.hip: k_symv_tiles / k_symv_reduce) against numpy, with alpha / beta, twice (deterministic)."""
    import cipkkt
    from cipkkt import _lib as L
    n = 2176
    rng = np.random.default_rng(8)
    M = rng.standard_normal((n, n))
    Q = M @ M.T / n + np.eye(n)
    A = np.eye(n)
    ks = cipkkt.KKTSystem(Q, A, None, [("R", n)])
    x = rng.standard_normal(n)
    y0 = rng.standard_normal(n)
    dx = dev(x)
    outs = []
    for rep in range(2):
        dy = dev(y0)
        L.check(lib.cip_gemv_dev(ks.h, L.MAT_Q, 0, -0.75, dx.data_ptr(), 0.5, dy.data_ptr()))
        torch.cuda.synchronize()
        outs.append(dy.cpu().numpy())
    want = -0.75 * (Q @ x) + 0.5 * y0
    assert np.abs(outs[0] - want).max() <= 1e-12 * (1 + np.abs(want).max())
    assert np.array_equal(outs[0], outs[1])
    ks.close()
