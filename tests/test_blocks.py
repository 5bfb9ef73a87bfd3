"""Host side of the plugin boundary without a GPU: the packer accepts what the reference hands a plugin
(src/ConicIP.jl:704-706: Block([Diagonal(ones(k)) ...]) for EVERY cone type; :189-192, :208, :598 inside the loop),
and the Block objects rebuilt from a packed scaling for user plugins carry the reference's fields and algebra
(test/runtests.jl:27-88)."""
import numpy as np
import pytest

from cipkkt import blocks as B
from cipkkt.kkt import pack_scaling
from oracle import cones as oc
from oracle.block import identity_block
from oracle.conicip import make_cone_ops

K = [("R", 5), ("Q", 4), ("S", 6), ("Q", 3), ("S", 10)]


def interior(rng, K):
    xs = []
    for t, k in K:
        if t == "R":
            xs.append(rng.random(k) + 0.1)
        elif t == "Q":
            x = rng.standard_normal(k)
            x[0] = np.linalg.norm(x[1:]) + 0.5
            xs.append(x)
        else:
            r = oc.ord_(np.zeros(k))
            M = rng.standard_normal((r, r))
            xs.append(oc.vecm(M @ M.T + 0.5 * np.eye(r)))
    return np.concatenate(xs)


def test_identity_block_is_accepted_for_every_cone_type():
    sizes = [k for _, k in K]
    pk = pack_scaling(K, identity_block(sizes), identity_block(sizes))
    F, FiT = B.blocks_from_packed(K, pk)
    m = sum(sizes)
    assert np.abs(F.matrix() - np.eye(m)).max() < 1e-15 and np.abs(FiT.matrix() - np.eye(m)).max() < 1e-15


def test_uniform_diagonal_scalings_and_rejection_of_nonuniform():
    from oracle.block import Block, Diagonal
    F = Block([Diagonal(np.full(5, 2.0)), Diagonal(np.full(4, 3.0)), Diagonal(np.full(6, 0.25)), Diagonal(np.ones(3)),
               Diagonal(np.full(10, 7.0))])
    Fh, _ = B.blocks_from_packed(K, pack_scaling(K, F, F.inv_adjoint()))
    np.testing.assert_allclose(Fh.matrix(), F.matrix(), atol=1e-14)
    bad = Block([Diagonal(np.ones(5)), Diagonal(np.array([1.0, 2, 1, 1]))] + F.Blocks[2:])
    with pytest.raises(ValueError):
        pack_scaling(K, bad, bad)


def test_round_trip_of_nt_scalings_and_block_algebra():
    rng = np.random.default_rng(0)
    _, nt_scaling, _, _ = make_cone_ops(K)
    F = nt_scaling(interior(rng, K), interior(rng, K))
    Fh, FiTh = B.blocks_from_packed(K, pack_scaling(K, F, F.inv_adjoint()))
    x = rng.standard_normal(Fh.size)
    np.testing.assert_allclose(Fh.mul(x), F.mul(x), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(Fh.tmul(x), F.tmul(x), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(FiTh.mul(x), F.inv_adjoint().mul(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(Fh.inv().mul(Fh.mul(x)), x, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose((Fh.T * Fh).matrix(), F.square().matrix(), rtol=1e-10, atol=1e-11)
    # the reference's element fields
    assert Fh[0].diag.shape == (5,) and Fh[1].A.diag[0] < 0 < Fh[1].A.diag[1] and Fh[1].B.shape == (4, 1)
    assert Fh[2].R.shape == (3, 3)
    # F[1]*F[1], inv(...).diag -- what the reference's box-QP plugin does (test/runtests.jl:104)
    v = (Fh[0] * Fh[0]).inv().diag
    np.testing.assert_allclose(v, 1.0 / F.Blocks[0].diag ** 2, rtol=1e-14)
