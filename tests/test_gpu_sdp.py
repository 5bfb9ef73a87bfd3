"""GPU parity for the semidefinite cone: scaling / operator / Jordan algebra / max-step kernels
against the oracle (restatement of src/ConicIP.jl:35-40, :69, :196-210, :272-303, :347-360) and the
reference's SDP known-answer test (test/runtests.jl:527-552) through the product driver."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import problems as P
from oracle import cones as oc
from oracle.block import VecCongurance
from oracle.conicip import conicIP as oracle_conicIP, make_cone_ops

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64, device="cuda")


def psd_vec(r, rng, shift=0.5):
    M = rng.standard_normal((r, r))
    return oc.vecm(M @ M.T / r + shift * np.eye(r))


def interior(cone_dims, rng):
    out = []
    for t, k in cone_dims:
        if t == "R":
            out.append(rng.random(k) + 0.1)
        elif t == "Q":
            x = rng.standard_normal(k)
            x[0] = np.linalg.norm(x[1:]) + 0.3
            out.append(x)
        else:
            out.append(psd_vec(oc.ord_(np.zeros(k)), rng))
    return np.concatenate(out)


@pytest.mark.parametrize("cone_dims", [[("S", 6)], [("S", 21)], [("S", 15), ("S", 3)],
                                       [("R", 5), ("Q", 4), ("S", 10)], [("S", 465)], [("S", 2080)], [("S", 5050), ("S", 6)],
                                       [("S", 8256)], [("S", 11325)], [("S", 20100), ("S", 10)], [("S", 32896)], [("S", 45150)],
                                       [("S", 205120)], [("S", 605550), ("S", 3)]],
                         ids=["r3", "r6", "r5+r2", "mixed", "r30", "r64", "r100+r3", "r128", "r150", "r200+r4", "r256", "r300",
                              "r640", "r1100+r2"])        # (r640: the padded-1024 path; r1100: padded order 2048, round 5.  r = 1000 + 3 (67 s of oracle time, in the suite until
                                                          #  round 5) and r = 1600 + 2 (7 minutes) pass the same assertions and are left out for the suite's run time)
def test_sdp_cone_ops(cone_dims):
    import cipkkt
    from cipkkt import OP_F, OP_FT, OP_FINV, OP_FINVT
    rng = np.random.default_rng(17)
    m = sum(k for _, k in cone_dims)
    n = 4
    ks = cipkkt.KKTSystem(np.eye(n), rng.standard_normal((m, n)), None, cone_dims)
    maxstep, nt_scaling, cone_div, cone_prod = make_cone_ops(cone_dims)
    v, s = interior(cone_dims, rng), interior(cone_dims, rng)
    lam = torch.zeros(m, dtype=torch.float64, device="cuda")
    ks.set_scaling_from_iterate(dev(v), dev(s), lam)
    F = nt_scaling(v, s)
    lam_h = lam.cpu().numpy()
    # lambda is unique up to the ordering of the eigenvalues inside each S block: compare F'F-invariant quantities
    off = 0
    for (t, k), blk in zip(cone_dims, F.Blocks):
        ref = F.mul(v)[off:off + k]
        if t == "S":
            np.testing.assert_allclose(np.sort(np.linalg.eigvalsh(oc.mat(lam_h[off:off + k]))),
                                       np.sort(np.linalg.eigvalsh(oc.mat(ref))), rtol=1e-9, atol=1e-11)
        else:
            np.testing.assert_allclose(lam_h[off:off + k], ref, rtol=1e-10, atol=1e-12)
        off += k
    x = rng.standard_normal(m)
    dx = dev(x)
    t1 = torch.zeros_like(dx)
    t2 = torch.zeros_like(dx)
    # F'F x is invariant under the column ambiguity of R
    ks.apply_F(OP_F, dx, t1)
    ks.apply_F(OP_FT, t1, t2)
    np.testing.assert_allclose(t2.cpu().numpy(), F.tmul(F.mul(x)), rtol=1e-8, atol=1e-9)
    # F^-1 F = I, F^-T F' = I, and F v = F^-T s (the defining property, src/ConicIP.jl:591-592)
    ks.apply_F(OP_FINV, t1, t2)
    np.testing.assert_allclose(t2.cpu().numpy(), x, rtol=1e-8, atol=1e-9)
    ks.apply_F(OP_FT, dx, t1)
    ks.apply_F(OP_FINVT, t1, t2)
    np.testing.assert_allclose(t2.cpu().numpy(), x, rtol=1e-8, atol=1e-9)
    ks.apply_F(OP_FINVT, dev(s), t1)
    np.testing.assert_allclose(t1.cpu().numpy(), lam_h, rtol=1e-7, atol=1e-8)
    # Jordan product / division
    y = interior(cone_dims, rng)
    ks.cone_prod(dx, dev(y), t1)
    np.testing.assert_allclose(t1.cpu().numpy(), cone_prod(x, y), rtol=1e-11, atol=1e-11)
    ks.cone_div(dx, dev(y), t1)
    np.testing.assert_allclose(t1.cpu().numpy(), cone_div(x, y), rtol=1e-8, atol=1e-9)
    # max step
    d = rng.standard_normal(m)
    for scale in (1.0, 1.0 / 0.99):
        got, ref = ks.maxstep(dev(v), dev(d), scale), maxstep(v, d * scale)
        assert got == pytest.approx(ref, rel=1e-8)
    assert ks.maxstep(dev(v), None) == 0.0
    assert ks.maxstep(dx, None) == pytest.approx(maxstep(x, None), rel=1e-9)
    # identity element and identity scaling
    e = torch.zeros(m, dtype=torch.float64, device="cuda")
    ks.cone_identity(e)
    from oracle.conicip import cone_identity
    np.testing.assert_array_equal(e.cpu().numpy(), cone_identity(cone_dims)[0])
    ks.set_scaling_identity()
    ks.apply_F(OP_F, dx, t1)
    np.testing.assert_allclose(t1.cpu().numpy(), x, rtol=1e-14, atol=1e-15)
    ks.close()


def test_maxstep_sdc_infinite():
    """test/runtests.jl:79-82: X = -I is not PD -> Inf."""
    import cipkkt
    ks = cipkkt.KKTSystem(np.eye(2), np.zeros((6, 2)), None, [("S", 6)])
    assert ks.maxstep(dev(oc.vecm(-np.eye(3))), dev(oc.vecm(np.eye(3)))) == np.inf
    ks.close()


@pytest.mark.parametrize("route", ["schur", "full3x3"])
def test_kkt_solve_with_s_cone(route):
    """level-2/3 with a VecCongurance block handed over in packed form (R, inv(R))."""
    import cipkkt
    from oracle.kktsolvers import kktsolver_qr
    rng = np.random.default_rng(4)
    cone_dims = [("R", 4), ("S", 10), ("Q", 3)]
    m = 17
    n, p = 9, 2
    M = rng.standard_normal((n, n))
    Q = M @ M.T / n + np.eye(n)
    A = rng.standard_normal((m, n))
    G = rng.standard_normal((p, n))
    _, nt_scaling, _, _ = make_cone_ops(cone_dims)
    F = nt_scaling(interior(cone_dims, rng), interior(cone_dims, rng))
    x, y, z = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    ref = np.concatenate(kktsolver_qr(Q, A, G, cone_dims)(F, F.inv_adjoint())(x, y, z))
    solver = cipkkt.kktsolver_hip if route == "schur" else cipkkt.kktsolver_hip_full3x3
    gen = solver(Q, A, G, cone_dims)
    got = np.concatenate(gen(F, F.inv_adjoint())(x, y, z))
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-9
    gen.system.close()


@pytest.mark.parametrize("route", ["schur", "full3x3"])
def test_psd_projection_kat(route):
    """test/runtests.jl:527-552: projection of diag(1,1,1,-1,-1,-1) onto the PSD cone."""
    import cipkkt
    Q, c, A, b, K, G, d, y = P.psd_projection()
    sol = cipkkt.conicIP(Q, c, A, b, K, G, d, optTol=1e-7, kktsolver=route)
    ref = oracle_conicIP(Q, c, A, b, K, G, d, optTol=1e-7)
    assert sol.status == "Optimal"
    assert np.abs(oc.mat(sol.y) - oc.mat(y)).max() < 1e-3            # the reference's assertion
    assert sol.Iter == ref.Iter == 6                                  # the pinned Iter (:547)
    np.testing.assert_allclose(sol.y, ref.y, rtol=1e-6, atol=1e-8)
    for key, val in dict(prFeas=4.2341217602756234e-16, Mu=3.4583513329836624e-10,
                         muFeas=1.48267911727847e-9, duFeas=4.2341217602756234e-16).items():
        assert abs(getattr(sol, key) - val) < 1e-3


def test_mixed_rqs_problem():
    """benchmark/profile.jl:116-131 pattern (R + Q + S cones, A = I) against the oracle."""
    import cipkkt
    rng = np.random.default_rng(42)
    n_r, n_q, k_s = 50, 21, 5
    n_s = k_s * (k_s + 1) // 2
    n = m = n_r + n_q + n_s
    Q = sp.identity(n, format="csr")
    c = rng.standard_normal(n)
    A = sp.identity(m, format="csr")
    b = np.concatenate([np.zeros(n_r), [-1.0], np.zeros(n_q - 1), np.zeros(n_s)])
    K = [("R", n_r), ("Q", n_q), ("S", n_s)]
    ref = oracle_conicIP(Q, c, A, b, K, optTol=1e-7)
    sol = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-7)
    assert sol.status == ref.status == "Optimal"
    assert sol.Iter == ref.Iter
    np.testing.assert_allclose(sol.y, ref.y, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("r", [133, 200, 256])
def test_lanczos_maxstep_against_lapack_on_hard_spectra(r):
    """Round 3: at orders 133..256 the max-step takes its ONE eigenvalue from Lanczos with full reorthogonalisation
    (sdp_large.hip: k_lg_lanczos1) instead of a full tridiagonalisation.  Spectra chosen against it: a tight cluster at the
    wanted end, a dense edge (semicircle: the slowest case, close to a complete tridiagonalisation), an exactly repeated extreme
    eigenvalue, one isolated outlier, a multiple of the identity (breakdown at the first step), a rank-one perturbation of it;
    a negative definite and a zero direction (no bound: Inf);
    both variants of maxstep_sdc (src/ConicIP.jl:272-303) against LAPACK at 1e-10, and against the library's own
    tridiagonalisation + Sturm path (cip_set_sdp_lanczos(0))."""
    import cipkkt
    from cipkkt import _lib
    lib = _lib.load()
    k = r * (r + 1) // 2
    ks = cipkkt.KKTSystem(np.eye(2), np.zeros((k, 2)), None, [("S", k)])
    rng = np.random.default_rng(r)
    Qm, _ = np.linalg.qr(rng.standard_normal((r, r)))
    def sym(lam):
        M = (Qm * lam) @ Qm.T
        return 0.5 * (M + M.T)
    spectra = {
        "cluster at the top": np.concatenate([1.0 + 0.3 * rng.random(r - 4), 2.0 + 1e-9 * np.arange(4)]),
        "cluster at the bottom": np.concatenate([1.0 + 0.3 * rng.random(r - 4), 0.05 + 1e-10 * np.arange(4)]),
        "semicircle": None,
        "repeated extreme": np.concatenate([np.linspace(0.5, 1.5, r - 3), [2.5, 2.5, 2.5]]),
        "outlier": np.concatenate([np.linspace(0.9, 1.1, r - 1), [40.0]]),
        "identity": np.full(r, 3.0),
        "identity + rank one": None,
        "negative definite": -(0.5 + rng.random(r)),                   # no step bound: Inf (src/ConicIP.jl:288-290)
        "zero": np.zeros(r),
    }
    for name, lam in spectra.items():
        if name == "semicircle":
            G = rng.standard_normal((r, r)); D = (G + G.T) / np.sqrt(2 * r)
        elif name == "identity + rank one":
            u = rng.standard_normal(r); D = 3.0 * np.eye(r) - 2.0 * np.outer(u, u) / (u @ u)
        else:
            D = sym(lam)
        X = sym(0.5 + rng.random(r))                                   # the point: positive definite
        x, d = dev(oc.vecm(X)), dev(oc.vecm(D))
        with np.errstate(divide="ignore"):                             # the zero direction: 1 / 0 = Inf, as the reference's
            ref_d, ref_n = oc.maxstep_sdc(oc.vecm(X), oc.vecm(D)), oc.maxstep_sdc(oc.vecm(D - 2.0 * np.eye(r)), None)
        dn = dev(oc.vecm(D - 2.0 * np.eye(r)))
        prev = lib.cip_set_sdp_lanczos(1)
        got_d, got_n = ks.maxstep(x, d), ks.maxstep(dn, None)
        lib.cip_set_sdp_lanczos(0)
        tri_d, tri_n = ks.maxstep(x, d), ks.maxstep(dn, None)
        lib.cip_set_sdp_lanczos(2)                                     # Lanczos + inertia certificate: same verdicts, no fallback needed here
        cer_d, cer_n = ks.maxstep(x, d), ks.maxstep(dn, None)
        lib.cip_set_sdp_lanczos(prev)
        assert (cer_d, cer_n) == (got_d, got_n), (name, cer_d, got_d, cer_n, got_n)
        for got, tri, ref, what in ((got_d, tri_d, ref_d, "maxstep(x, d)"), (got_n, tri_n, ref_n, "maxstep(x, nothing)")):
            if np.isinf(ref):
                assert np.isinf(got) and np.isinf(tri), (name, what)
            else:
                assert got == pytest.approx(ref, rel=1e-10, abs=1e-12), (name, what, got, ref)
                assert got == pytest.approx(tri, rel=1e-10, abs=1e-12), (name, what, got, tri)
    import ctypes as C
    nfb = C.c_int(-1)
    _lib.check(lib.cip_sdp_lanczos_fallbacks(ks.h, C.byref(nfb)))
    assert nfb.value == 0, nfb.value
    ks.close()


@pytest.mark.parametrize("r", [133, 256])
def test_lanczos_inertia_certificate_catches_a_start_vector_without_the_extreme_direction(r):
    """the Lanczos max-step starts from a FIXED vector v1; a direction whose extreme eigenvector is orthogonal to v1 is
    invisible to the recurrence (until rounding brings it in), and the stop test -- a converged Ritz pair -- passes on the second
    eigenvalue.  Such a direction is built here (v1 is cos(0.7 i + 0.3) + 1 / (1 + i), sdp_large.hip: k_lg_lanczos1): top eigenvalue
    2.05 along u orthogonal to v1, 2.0 next, the rest in [0.2, 1].  Mode 2 (cip_set_sdp_lanczos(2)) must return LAPACK's answer,
    through the certificate's fallback when plain Lanczos is fooled; mode 1's answer is recorded, not asserted (rounding may or may
    not rescue it -- that is the point of the certificate)."""
    import ctypes as C
    import cipkkt
    from cipkkt import _lib
    lib = _lib.load()
    k = r * (r + 1) // 2
    ks = cipkkt.KKTSystem(np.eye(2), np.zeros((k, 2)), None, [("S", k)])
    rng = np.random.default_rng(7 * r)
    i = np.arange(r)
    v1 = np.cos(0.7 * i + 0.3) + 1.0 / (1.0 + i)
    v1 /= np.linalg.norm(v1)
    u = rng.standard_normal(r); u -= (u @ v1) * v1; u /= np.linalg.norm(u)
    # an orthonormal basis whose first vector is u: the other eigenvectors span u's complement (v1 lies in it)
    Qm, _ = np.linalg.qr(np.column_stack([u, rng.standard_normal((r, r - 1))]))
    Qm[:, 0] = u
    lam = np.concatenate([[2.05, 2.0], np.linspace(0.2, 1.0, r - 2)])
    D = (Qm * lam) @ Qm.T
    D = 0.5 * (D + D.T)
    x, d = dev(oc.vecm(np.eye(r))), dev(oc.vecm(D))
    ref = oc.maxstep_sdc(oc.vecm(np.eye(r)), oc.vecm(D))
    assert ref == pytest.approx(1.0 / np.linalg.eigvalsh(D)[-1], rel=1e-12)
    prev = lib.cip_set_sdp_lanczos(1)
    plain = ks.maxstep(x, d)
    lib.cip_set_sdp_lanczos(2)
    certified = ks.maxstep(x, d)
    lib.cip_set_sdp_lanczos(prev)
    nfb = C.c_int(-1)
    _lib.check(lib.cip_sdp_lanczos_fallbacks(ks.h, C.byref(nfb)))
    print("r = %d: LAPACK %.12f, plain Lanczos %.12f, certified %.12f, fallbacks %d" % (r, ref, plain, certified, nfb.value))
    assert certified == pytest.approx(ref, rel=1e-10)
    if abs(plain - ref) > 1e-8 * ref:                                # plain Lanczos was fooled: then the certificate must have fired
        assert nfb.value >= 1
    # (measured: rounding brings u into the recurrence and plain Lanczos finds 2.05 as well -- the fallback is not reached that way.)
    # Self-test mode 3 puts the certificate's bound on the wrong side of theta: every certificate fails, the verdict must come from
    # the gated tridiagonalisation + Sturm kernels and agree with LAPACK, for both variants and for the two sides of a pair
    before = nfb.value
    lib.cip_set_sdp_lanczos(3)
    X = (Qm * (0.5 + rng.random(r))) @ Qm.T; X = 0.5 * (X + X.T)
    xs = dev(oc.vecm(X))
    forced = ks.maxstep(xs, d)
    forced_n = ks.maxstep(dev(oc.vecm(D - 2.02 * np.eye(r))), None)
    pair = ks.maxstep_pair(xs, d, x, d) if r <= 256 else None
    lib.cip_set_sdp_lanczos(prev)
    _lib.check(lib.cip_sdp_lanczos_fallbacks(ks.h, C.byref(nfb)))
    assert nfb.value - before == (4 if pair else 2), (nfb.value, before)
    assert forced == pytest.approx(oc.maxstep_sdc(oc.vecm(X), oc.vecm(D)), rel=1e-10)
    assert forced_n == pytest.approx(oc.maxstep_sdc(oc.vecm(D - 2.02 * np.eye(r)), None), rel=1e-10)
    if pair:
        assert pair[0] == pytest.approx(forced, rel=1e-12) and pair[1] == pytest.approx(ref, rel=1e-10)
    ks.close()


def _nt_products(ks, v, s, x, k):
    """set the scaling from (v, s) and return (packed scaling, F'F x, lambda): F'F x and lambda are invariant under the orthogonal
    freedom of the factor R (src/ConicIP.jl:37-40, :69, :735)"""
    from cipkkt import _lib as L
    lam = torch.zeros(k, dtype=torch.float64, device="cuda")
    ks.set_scaling_from_iterate(v, s, lam)
    F = ks.get_scaling_packed()
    y = torch.zeros_like(x)
    ks.apply_F(L.OP_F, x, y)
    z = torch.zeros_like(x)
    ks.apply_F(L.OP_FT, y, z)
    return F, z.cpu().numpy(), lam.cpu().numpy().copy()


@pytest.mark.parametrize("r", [200, 300, 700, 2048])        # (2048: the largest order the library takes -- include/cipkkt.h)
def test_large_s_cone_nt_scaling_is_reproducible_and_has_the_oracles_invariants(r):
    """The NT scaling of a large S cone (one-sided Jacobi, one launch per phase -- the only form since round 6: the persistent
    kernel with in-launch block hand-offs, measured to come out with other bits once in 800 ... 40000 scalings, and its switch are
    gone) twice from the same iterate on two fresh handles: same bits.  And against the oracle's nestod_sdc on the quantities that
    do not depend on the factor's orthogonal freedom: lambda (the singular values, as vecm of a diagonal matrix) and F'F x
    -- src/ConicIP.jl:196-210, :37-40."""
    import cipkkt
    k = r * (r + 1) // 2
    cone_dims = [("S", k)]
    rng = np.random.default_rng(r)
    vh, sh = interior(cone_dims, rng), interior(cone_dims, rng)
    v, s = dev(vh), dev(sh)
    xh = rng.standard_normal(k)
    x = dev(xh)
    out = []
    for rep in range(2):
        ks = cipkkt.KKTSystem(np.eye(2), np.zeros((k, 2)), None, cone_dims)
        out.append(_nt_products(ks, v, s, x, k))
        ks.close()
    (Fa, za, la), (Fb, zb, lb) = out
    np.testing.assert_array_equal(Fa, Fb)
    np.testing.assert_array_equal(la, lb)
    np.testing.assert_array_equal(za, zb)
    R = oc.nestod_sdc(vh, sh)                                  # F x = vecm(R' mat(x) R), F'F x = vecm(P mat(x) P), P = R R'
    lam_o = oc.vecm(R.T @ oc.mat(vh) @ R)
    Pm = R @ R.T
    ftf_o = oc.vecm(Pm @ oc.mat(xh) @ Pm)
    # lambda = vecm(diag(sigma)): the oracle's singular values come out in LAPACK's order, the Jacobi's in column order -> compare sorted
    dg = np.array([oc_vidx(i, r) for i in range(r)])
    np.testing.assert_allclose(np.sort(la[dg]), np.sort(np.asarray(lam_o)[dg]), rtol=1e-10)
    np.testing.assert_allclose(za, ftf_o, rtol=1e-9, atol=1e-9 * np.abs(ftf_o).max())
    off = np.ones(k, dtype=bool); off[dg] = False
    assert np.abs(la[off]).max() <= 1e-10 * np.abs(la[dg]).max()


def oc_vidx(i, r):
    """index of the diagonal entry (i, i) in the vecm layout (upper triangle by rows: src/ConicIP.jl:101-104)"""
    return i * r - i * (i - 1) // 2


def test_warm_start_of_the_jacobi_is_not_taken_from_a_scaling_that_left_the_cone():
    """Round-5 advisor finding: cip_set_scaling_from_iterate on an iterate OUTSIDE the cone (a Cholesky pivot <= 0) left NaN in the
    stored right singular vectors, and the next scaling of a perfectly interior iterate on the same handle was warm-started from
    them: NaN out, clean flag.  Now the warm start is dropped on the host when the Cholesky flags are raised and gated on the device
    (clean flags, finite positive singular values, |V'V - I| <= 1e-6: sdp_large.hip k_lg_warm_gate): the scaling behind a flagged one
    has exactly the bits of a cold run on a fresh handle -- src/ConicIP.jl:196-210."""
    import cipkkt
    r = 150
    k = r * (r + 1) // 2
    cone_dims = [("S", k)]
    rng = np.random.default_rng(77)
    vh, sh = interior(cone_dims, rng), interior(cone_dims, rng)
    x = dev(rng.standard_normal(k))
    bad = vh.copy()
    bad[[oc_vidx(i, r) for i in range(r)]] -= 50.0           # far outside the cone
    ks = cipkkt.KKTSystem(np.eye(2), np.zeros((k, 2)), None, cone_dims)
    _nt_products(ks, dev(vh), dev(sh), x, k)                  # a valid scaling: V stored
    lam = torch.zeros(k, dtype=torch.float64, device="cuda")
    ks.set_scaling_from_iterate(dev(bad), dev(sh), lam)       # flagged: must not poison what follows
    got = _nt_products(ks, dev(vh), dev(sh), x, k)
    ks.close()
    fresh = cipkkt.KKTSystem(np.eye(2), np.zeros((k, 2)), None, cone_dims)
    ref = _nt_products(fresh, dev(vh), dev(sh), x, k)
    fresh.close()
    for g, f in zip(got, ref):
        assert np.all(np.isfinite(g))
        np.testing.assert_array_equal(g, f)


@pytest.mark.parametrize("cone_dims", [[("R", 700)], [("Q", 40), ("R", 9), ("S", 6)], [("S", 200 * 201 // 2)],
                                       [("R", 5), ("S", 150 * 151 // 2), ("S", 10)]])
def test_maxstep_pair_equals_two_calls(cone_dims):
    """cip_maxstep_pair_dev -- the pair the loop asks for together (src/ConicIP.jl:708-709, :881-882, :927-928); with a large S
    cone its two sides run on two streams in the NT scaling's idle buffers -- returns exactly what two cip_maxstep_dev calls do."""
    import cipkkt
    m = sum(k for _, k in cone_dims)
    ks = cipkkt.KKTSystem(np.eye(2), np.zeros((m, 2)), None, cone_dims)
    rng = np.random.default_rng(m)
    for rep in range(3):
        v, s = dev(interior(cone_dims, rng)), dev(interior(cone_dims, rng))
        dv, ds = dev(rng.standard_normal(m)), dev(rng.standard_normal(m))
        for scale in (1.0, 1.0 / 0.99):
            assert ks.maxstep_pair(v, dv, s, ds, scale) == (ks.maxstep(v, dv, scale), ks.maxstep(s, ds, scale))
        assert ks.maxstep_pair(dv, None, ds, None) == (ks.maxstep(dv, None), ks.maxstep(ds, None))
    ks.close()


@pytest.mark.parametrize("r,n,p,seed", [(133, 64, 4, 919850), (192, 40, 4, 790518), (256, 24, 0, 639913), (640, 12, 0, 31337), (1100, 8, 0, 4242)])
def test_large_s_cone_programs_walk_the_oracles_trajectory(r, n, p, seed):
    """Config 4's family on the large-cone path (orders 133..256: Lanczos max-step, the two max-steps of a pair side by side;
    round 4: order 640 on the order-1024 workspaces -- 64 Jacobi workgroups with 16 elements per lane, tridiagonalisation in
    16-column slabs; round 5: order 1100 on the order-2048 workspaces -- Jacobi with 32 elements per lane, one launch per phase, the
    max-step's tridiagonalisation with the matrix in global memory; the reference has no order limit, src/ConicIP.jl:196-210) against the oracle with the exact block
    elimination: same status, same iteration count, iterates at 1e-8 (tools/fuzz_sdp.py draws more of them)."""
    import cipkkt
    from cipkkt import workloads as W
    from oracle import kktsolvers as ok
    prob = W.c4_sdp(r=r, n=n, p=p, seed=seed)
    ref = oracle_conicIP(*prob, optTol=1e-6, kktsolver=ok.kktsolver_schur_exact)
    got = cipkkt.conicIP(*prob, optTol=1e-6)
    assert got.status == ref.status == "Optimal" and got.Iter == ref.Iter
    np.testing.assert_allclose(got.y, ref.y, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(got.v, ref.v, rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("csr", [False, True], ids=["denseA", "csrA"])
def test_update_problem_drops_the_cached_column_images_of_a_large_cone(csr):
    """The large-cone path keeps mat(a_i) of every column of A from one factorisation to the next (sdp_large.hip: built once per
    upload).  cip_update_problem with another A on the same handle -- what the batch workers do between problems -- must give exactly
    what a fresh handle gives: factor with A1 first (the images exist), update to A2, factor + solve, compare bit for bit."""
    import ctypes as C
    import cipkkt
    import scipy.sparse as sp
    from cipkkt import _lib as L
    from cipkkt.kkt import make_problem
    r, n = 140, 24
    k = r * (r + 1) // 2
    K = [("R", 5), ("S", k)]
    m = 5 + k
    rng = np.random.default_rng(77)

    mask = np.random.default_rng(5).random((m, n)) < 0.3          # one sparsity pattern: a CSR handle is re-loaded with the same nnz

    def problem(seed):
        g = np.random.default_rng(seed)
        M = g.standard_normal((n, n))
        A = (1.0 + g.random((m, n))) * mask
        return M @ M.T / n + np.eye(n), (sp.csr_matrix(A) if csr else A)

    (Q1, A1), (Q2, A2) = problem(1), problem(2)
    ks = cipkkt.KKTSystem(Q1, A1, None, K)
    v, s = dev(interior(K, rng)), dev(interior(K, rng))
    ks.set_scaling_from_iterate(v, s)
    ks.factor()
    pr, keep, _ = make_problem(Q2, A2, None, K, "schur", ks.device)
    torch.cuda.synchronize()
    L.check(ks.lib.cip_update_problem(ks.h, C.byref(pr)))
    fresh = cipkkt.KKTSystem(Q2, A2, None, K)
    x, z = rng.standard_normal(n), rng.standard_normal(m)
    outs = []
    for sysm in (ks, fresh):
        sysm.set_scaling_from_iterate(v, s)
        sysm.factor()
        outs.append(np.concatenate(sysm.solve3x3(x, np.zeros(0), z)))
    np.testing.assert_array_equal(outs[0], outs[1])
    ks.close(); fresh.close()


def test_large_cone_schur_scaling_in_chunks_equals_the_one_batch_path(monkeypatch):
    """A'F^-1 of a large S cone: by default all n columns go through ONE pair of batched GEMM launches from the cached mat(a_i)
    images; when they do not fit the workspace (CIP_LG_CHUNK forces it here: 16 columns per batch, n = 40 -> three batches, no
    image cache) the same kernels run batch by batch.  Same Schur matrix bit for bit, same solve."""
    import cipkkt
    r, n = 150, 40
    k = r * (r + 1) // 2
    K = [("R", 3), ("S", k)]
    m = 3 + k
    rng = np.random.default_rng(123)
    M = rng.standard_normal((n, n))
    Q, A = M @ M.T / n + np.eye(n), rng.standard_normal((m, n)) / np.sqrt(n)
    v, s = dev(interior(K, rng)), dev(interior(K, rng))
    x, z = rng.standard_normal(n), rng.standard_normal(m)
    outs = []
    for chunk in (None, "16"):
        if chunk:
            monkeypatch.setenv("CIP_LG_CHUNK", chunk)
        ks = cipkkt.KKTSystem(Q, A, None, K)
        ks.set_scaling_from_iterate(v, s)
        ks.factor()
        outs.append((ks.kkt_matrix(), np.concatenate(ks.solve3x3(x, np.zeros(0), z))))
        ks.close()
    np.testing.assert_array_equal(np.tril(outs[0][0]), np.tril(outs[1][0]))
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


# ---- CSR A together with S cones on the device (round 4: no host-side densification)
def test_reference_sparse_psd_projection_without_densification():
    """test/runtests.jl:527-552: project onto the PSD cone with A = sparse identity (6 x 6), one ("S", 6) cone -- through
    the CSR entry of the C ABI (cip_problem.A_rowptr ...): the library expands the S cone's rows on the device.  Same
    iteration count as the oracle (the reference pins Iter == 6 for its data; here: equal to the oracle's) and the analytic
    answer diag(1, 1, 1, 0, 0, 0) of the reference's own problem."""
    import cipkkt
    import scipy.sparse as sp
    from oracle.conicip import conicIP as oracle_conicIP
    import problems as P
    Q, c, A, b, K = P.psd_projection()[:5]
    want = P.psd_projection()[7]
    As = sp.csr_matrix(A)
    ref = oracle_conicIP(Q, c, np.asarray(As.todense()), b, K, optTol=1e-7)
    for route in ("schur", "full3x3"):
        ks = cipkkt.KKTSystem(Q, As, None, K, route=route)
        assert ks.A_sparse, "CSR A with an S cone must reach the library as CSR"
        got = cipkkt.conicIP(Q, c, As, b, K, optTol=1e-7, system=ks, kktsolver=route)
        assert got.status == ref.status == "Optimal" and got.Iter == ref.Iter
        np.testing.assert_allclose(got.y, ref.y, rtol=1e-6, atol=1e-7)
        assert np.abs(got.y - want).max() < 1e-3                  # `tol` of test/runtests.jl:13
        ks.close()


@pytest.mark.parametrize("route", ["schur", "full3x3"])
def test_csr_A_with_mixed_R_Q_S_cones_equals_dense_A(route):
    """A random program with R, Q and two S cones (one of them interleaved between the others): the CSR upload -- O(nnz)
    Schur rows for R / Q, congruences on the expanded S rows + one GEMM -- walks the trajectory of the dense upload and of
    the oracle; the assembled Schur matrix agrees with the dense path's to 1e-12."""
    import cipkkt
    import scipy.sparse as sp
    from oracle.conicip import conicIP as oracle_conicIP
    from oracle import cones as oc
    rng = np.random.default_rng(42)
    n, p = 30, 3
    K = [("R", 12), ("S", 10), ("Q", 5), ("S", 21), ("R", 4)]
    m = sum(k for _, k in K)
    A = np.where(rng.random((m, n)) < 0.3, rng.standard_normal((m, n)), 0.0)
    xs = []
    for t, k in K:                                              # b = A y0 - s0 with s0 strictly inside the cone, y0 = 1
        if t == "R":
            xs.append(rng.random(k) + 0.5)
        elif t == "Q":
            x = rng.standard_normal(k); x[0] = np.linalg.norm(x[1:]) + 1.0; xs.append(x)
        else:
            r = int(round((np.sqrt(1 + 8 * k) - 1) / 2)); M = rng.standard_normal((r, r)); xs.append(oc.vecm(M @ M.T / r + np.eye(r)))
    s0 = np.concatenate(xs)
    b = A @ np.ones(n) - s0
    G = rng.standard_normal((p, n)); d = G @ np.ones(n)
    Mq = rng.standard_normal((n, n)); Q = Mq.T @ Mq / n + 0.1 * np.eye(n); c = rng.standard_normal(n)
    ref = oracle_conicIP(Q, c, A, b, K, G, d, optTol=1e-7)
    dense = cipkkt.conicIP(Q, c, A, b, K, G, d, optTol=1e-7, kktsolver=route)
    csr = cipkkt.conicIP(Q, c, sp.csr_matrix(A), b, K, G, d, optTol=1e-7, kktsolver=route)
    assert ref.status == dense.status == csr.status == "Optimal"
    assert ref.Iter == dense.Iter == csr.Iter
    np.testing.assert_allclose(csr.y, dense.y, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(csr.y, ref.y, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(csr.v, ref.v, rtol=1e-5, atol=1e-6)
    if route == "schur":
        kd = cipkkt.KKTSystem(Q, A, G, K); kc = cipkkt.KKTSystem(Q, sp.csr_matrix(A), G, K)
        import torch
        v, s = torch.as_tensor(s0 * 1.3, device="cuda"), torch.as_tensor(s0, device="cuda")
        for ks in (kd, kc):
            ks.set_scaling_from_iterate(v, s); ks.assemble_only()
        Kd, Kc = kd.kkt_matrix(), kc.kkt_matrix()
        N = n + p
        np.testing.assert_allclose(np.tril(Kc[:N, :N]), np.tril(Kd[:N, :N]), rtol=1e-12, atol=1e-12 * np.abs(np.tril(Kd[:N, :N])).max())   # (outside the lower triangle of the order-N block the buffer is undefined)
        kd.close(); kc.close()


# ------------------------------------------------------------------ the edges of the S-cone envelope (round-4 review, item 7)
def _sdp_problem(r, n=4, extra_small=0):
    k = r * (r + 1) // 2
    from cipkkt.workloads import vecm_identity
    return dict(Q=np.eye(n), c=np.ones(n), A=np.zeros((k, n)), b=-vecm_identity(r), cone_dims=[("S", k)], G=None, d=None, kwargs={})


def test_s_cone_above_order_2048_is_refused_cleanly():
    """the reference has no limit on the matrix order (src/ConicIP.jl:196-210); this library stops at 2048 (1024 until round 5's second
    session: orders 1025..2048 run the NT scaling's Jacobi with 32 elements per lane, one launch per phase, and the max-step's
    tridiagonalisation with the matrix in global memory, two launches per column).  A larger cone must
    be REFUSED at level 1 -- CIP_E_UNSUPPORTED, a message naming the cone, no handle -- not mis-handled.  Only the cone table
    is inspected before the refusal (no 4 GB upload of A)."""
    import ctypes as C
    from cipkkt import _lib as L
    lib = L.load()
    r = 2049
    k = r * (r + 1) // 2
    ct = (C.c_int * 1)(L.CONE_S if hasattr(L, "CONE_S") else 2)
    cd = (C.c_int * 1)(k)
    Q = np.eye(2, order="F")
    A = np.zeros((1, 1))                                   # never read: the cone table is checked first
    pr = L.CipProblem()
    pr.n, pr.m, pr.p, pr.ncones = 2, k, 0, 1
    pr.cone_type, pr.cone_dim = ct, cd
    pr.Q, pr.ldq = C.c_void_p(Q.ctypes.data), 2
    pr.A, pr.lda = C.c_void_p(A.ctypes.data), k
    pr.G, pr.ldg = None, 1
    pr.route, pr.flags = L.ROUTE_SCHUR, 0
    h = C.c_void_p()
    rc = lib.cip_create_ex(C.byref(pr), C.byref(h))
    assert rc == L.E_UNSUPPORTED, rc
    assert not h.value                                     # nothing to destroy
    assert b"2049" in lib.cip_last_error() and b"2048" in lib.cip_last_error()


def test_more_than_1024_large_s_cones_are_refused_cleanly():
    """1025 S cones of order 133 (the chip-wide kernels keep one set of padded matrices per large cone, at most 1024 -- 8 until
    round 5, 64 in its first session): CIP_E_UNSUPPORTED from level 1, no handle; fourteen (refused until round 5) are accepted AND solved"""
    import cipkkt
    from cipkkt import _lib as L
    from cipkkt.workloads import vecm_identity
    r = 133
    k = r * (r + 1) // 2
    n = 3

    def build(count):
        rng = np.random.default_rng(count)
        A = rng.standard_normal((count * k, n)) * 0.01
        return np.eye(n), A, [("S", k)] * count

    Q, A, K = build(1025)
    with pytest.raises(L.CipError) as ei:
        cipkkt.KKTSystem(Q, A, None, K)
    assert ei.value.code == L.E_UNSUPPORTED and "S cones" in str(ei.value)
    # 14 large cones: more than the 12 per-cone division gates a 16-int flag buffer had room for (round 5's buffer: cones 13.. wrote
    # their gates past the allocation); the whole loop -- NT scaling, division by lambda in every solve4x4, max-steps -- against the oracle
    ncone = 14
    Q, A, K = build(ncone)
    b = -np.concatenate([vecm_identity(r)] * ncone)
    c = np.array([1.0, -0.5, 0.25])
    sol = cipkkt.conicIP(Q, c, A, b, K, optTol=1e-6)
    assert sol.status == "Optimal", sol.status
    # optimality of the program: dual residual Q y - c - A'v = 0, complementarity v's = 0 (s = A y - b)
    y, v = sol.y, sol.v
    sres = A @ y - b
    assert np.linalg.norm(Q @ y - c - A.T @ v) <= 1e-5 * (1 + np.linalg.norm(v))
    assert abs(v @ sres) <= 1e-4 * (1 + abs(sol.pobj))
    from oracle import kktsolvers as ok
    ref = oracle_conicIP(Q, c, A, b, K, None, None, optTol=1e-6, kktsolver=ok.kktsolver_schur_exact)
    assert ref.status == "Optimal" and sol.Iter == ref.Iter, (sol.Iter, ref.Iter)
    np.testing.assert_allclose(sol.y, ref.y, rtol=1e-7, atol=1e-8)


def test_lockstep_refusal_of_large_s_cones_writes_nothing_and_mixed_routes_them():
    """a lock-step call on problems with a chip-wide S cone answers CIP_E_UNSUPPORTED and leaves the caller's result and
    solution buffers untouched; cip_conicip_mixed solves the same problems through the thread pool"""
    import ctypes as C
    from cipkkt import _lib as L
    from cipkkt.kkt import make_problem
    from cipkkt.workloads import c4_sdp
    lib = L.load()
    dev = torch.device("cuda:0")
    prs = []
    for seed in (41, 42):
        Q, c, A, b, K, G, d = c4_sdp(r=133, n=12, p=2, seed=seed)
        prs.append(dict(Q=Q, c=c, A=A, b=b, cone_dims=K, G=G, d=d))
    k = len(prs)
    structs = (L.CipProblem * k)()
    keep = []
    for i, pr in enumerate(prs):
        st, kp, _ = make_problem(pr["Q"], pr["A"], pr["G"], pr["cone_dims"], "schur", dev)
        structs[i] = st
        keep.append(kp)
    torch.cuda.synchronize()
    vp = C.c_void_p * k
    m, n, p = prs[0]["A"].shape[0], 12, 2
    cs = [np.ascontiguousarray(pr["c"]) for pr in prs]
    bs = [np.ascontiguousarray(pr["b"]) for pr in prs]
    ds = [np.ascontiguousarray(pr["d"]) for pr in prs]
    sentinel = 12345.678
    ys = [np.full(n, sentinel) for _ in prs]
    ws = [np.full(p, sentinel) for _ in prs]
    vs = [np.full(m, sentinel) for _ in prs]
    arr = lambda xs: vp(*[x.ctypes.data for x in xs])
    res = (L.CipResult * k)()
    for i in range(k):
        res[i].status, res[i].iter = 77, 88
    opt = L.CipOptions(1e-6, 0.01, -1.0, -1.0, 3, 100, 0)
    rc = lib.cip_conicip_lockstep(k, structs, arr(cs), arr(bs), arr(ds), C.byref(opt), arr(ys), arr(ws), arr(vs), res)
    assert rc == L.E_UNSUPPORTED
    assert all(np.all(y == sentinel) for y in ys) and all(np.all(v == sentinel) for v in vs) and all(np.all(w == sentinel) for w in ws)
    assert all(res[i].status == 77 and res[i].iter == 88 for i in range(k))
    L.check(lib.cip_conicip_mixed(k, structs, arr(cs), arr(bs), arr(ds), C.byref(opt), arr(ys), arr(ws), arr(vs), res, 2))
    st3 = (C.c_int * 3)()
    lib.cip_lockstep_stats(st3)
    assert list(st3) == [0, 0, 0]                          # no lock-step group was formed: both went through the thread pool
    assert all(res[i].status == 1 for i in range(k)), [res[i].status for i in range(k)]       # CIP_STATUS_OPTIMAL
    assert not any(np.any(y == sentinel) for y in ys)
