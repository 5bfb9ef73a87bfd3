"""The portable generator behind every synthetic input (cipkkt/workloads.py): SplitMix64 known answers, numpy and
torch streams identical, config builders shaped as BASELINE.md section 3 says."""
import numpy as np
import pytest
import torch

from cipkkt import workloads as W


def test_splitmix64_known_answers():
    # reference output of SplitMix64 (Vigna) for seed 1234567
    got = W.splitmix64_np(1234567, np.arange(3, dtype=np.uint64))
    assert [int(x) for x in got] == [6457827717110365317, 3203168211198807973, 9817491932198370423]


def test_numpy_and_torch_streams_agree():
    for seed in (0, 1234, 2 ** 63 + 5):
        a = W.splitmix64_np(seed, np.arange(1000, 1100, dtype=np.uint64))
        b = W.splitmix64_torch(seed, 100, "cpu", offset=1000).numpy().view(np.uint64)
        assert np.array_equal(a, b)
        assert np.array_equal(W.uniform_np(seed, 257, offset=3), W.uniform_torch(seed, 257, "cpu", offset=3).numpy())
    x, y = W.randn_np(5, 300, 70), W.randn_torch(5, 300, 70, device="cpu").numpy()
    assert np.abs(x - y).max() < 1e-14
    assert abs(x.mean()) < 0.02 and abs(x.std() - 1) < 0.02


def test_config_builders_shapes():
    Q, c, A, b, K = W.c1_readme_boxqp(n=200, seed=3)
    assert Q.shape == (200, 200) and np.allclose(Q, Q.T) and K == [("R", 200)] and np.all(c == 1) and A.shape == (200, 200)
    assert np.linalg.eigvalsh(Q).min() > -1e-9
    Q, c, A, b, K, G, d = W.c3_socp(n=64, ncones=8, kq=8, p=4, seed=1)
    assert A.shape == (64, 64) and b[0] == -1 and b[1] == 0 and len(K) == 8 and G.shape == (4, 64)
    Q, c, A, b, K, G, d = W.c4_sdp(r=6, n=10, p=2, seed=1)
    assert K == [("S", 21)] and A.shape == (21, 10)
    from oracle.cones import vecm
    assert np.array_equal(-b, vecm(np.eye(6)))
    pr = W.c5_batch(count=3, n=16, seed=9)
    assert len(pr) == 3 and not np.allclose(pr[0]["Q"], pr[1]["Q"])
    Q2, c2 = W.c2_dense_qp(16, 9)
    assert np.array_equal(pr[0]["Q"], Q2) and np.array_equal(pr[0]["c"], c2)


@pytest.mark.gpu
def test_device_stream_matches_host():
    a = W.randn_np(77, 64, 33)
    b = W.randn_torch(77, 64, 33, device="cuda").cpu().numpy()
    assert np.abs(a - b).max() < 1e-13
    Qh, ch = W.c2_dense_qp(128, 5)
    Qd, cd = W.c2_dense_qp(128, 5, device="cuda")
    np.testing.assert_allclose(Qd.cpu().numpy(), Qh, rtol=0, atol=1e-12)
    np.testing.assert_allclose(cd.cpu().numpy(), ch, rtol=0, atol=1e-13)
