"""Synthetic inputs of the BASELINE.json configurations, from a portable counter-based generator.

Julia's `Random.seed!` streams cannot be reproduced outside Julia, so every synthetic input of the
bench and of the full-size tests comes from SplitMix64 (counter -> 64-bit word -> uniform ->
Box-Muller), implemented twice with identical integer streams: numpy (`*_np`, host) and torch
integer ops (device: the n = 8192 matrices are generated in HBM, nothing large crosses PCIe).
The uniforms are identical bit for bit; the normals agree to the last ulp or two of the device's
log / cos (tests/test_workloads.py).  Seeds are stated in every bench line.

Configs (SURVEY 8d / BASELINE.md section 3):
  C1  README box-QP (README.md:56-65): n = 1000, Q = B'B with B 10 %-dense N(0,1), c = 1, A = I, b = 0
  C2  dense QP (headline): n = m = 8192, Q = M'M/n, c ~ N(0,1), A = I, b = 0
  C3  SOCP: n = 4096, 512 x ("Q", 8), dense A, head rows b = -1, equality block G (p = 512), d = 0
  C4  SDP: one S cone of matrix order r (k = r(r+1)/2), n variables, b = -vecm(I), G p x n, d = 0
  C5  64 x C2-style with n = 2048, seeds base + i
"""
import numpy as np

_M64 = (1 << 64) - 1
_GAMMA = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB


# ------------------------------------------------------------------ numpy (host) stream
def splitmix64_np(seed, idx):
    """word idx of the SplitMix64 stream `seed` (idx: uint64 array)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed & _M64) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(_GAMMA))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        return z ^ (z >> np.uint64(31))


def uniform_np(seed, count, offset=0):
    """`count` uniforms in (0, 1): ((word >> 11) + 0.5) * 2^-53."""
    w = splitmix64_np(seed, np.arange(offset, offset + count, dtype=np.uint64))
    return ((w >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def randn_np(seed, *shape):
    """N(0,1) by Box-Muller on the word pairs (2i, 2i+1); row-major fill of `shape`."""
    count = int(np.prod(shape)) if shape else 1
    u = uniform_np(seed, 2 * count)
    z = np.sqrt(-2.0 * np.log(u[0::2])) * np.cos(2.0 * np.pi * u[1::2])
    return z.reshape(shape)


# ------------------------------------------------------------------ torch (device) stream
def _to_i64(c):
    return c - (1 << 64) if c >= (1 << 63) else c


def _lsr(z, k):
    """logical shift right of an int64 tensor holding uint64 bit patterns."""
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64_torch(seed, count, device, offset=0):
    import torch
    idx = torch.arange(offset + 1, offset + count + 1, dtype=torch.int64, device=device)
    z = idx * _to_i64(_GAMMA) + _to_i64(seed & _M64)          # wraps modulo 2^64 (two's complement)
    z = (z ^ _lsr(z, 30)) * _to_i64(_C1)
    z = (z ^ _lsr(z, 27)) * _to_i64(_C2)
    return z ^ _lsr(z, 31)


def uniform_torch(seed, count, device, offset=0):
    import torch
    w = splitmix64_torch(seed, count, device, offset)
    return (_lsr(w, 11).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)


def randn_torch(seed, *shape, device="cuda"):
    import math
    import torch
    count = int(np.prod(shape)) if shape else 1
    out = torch.empty(count, dtype=torch.float64, device=device)
    step = 1 << 24                                         # bounded temporaries (n = 8192: 2^26 normals)
    for o in range(0, count, step):
        c = min(step, count - o)
        u = uniform_torch(seed, 2 * c, device, offset=2 * o)
        out[o:o + c] = torch.sqrt(-2.0 * torch.log(u[0::2])) * torch.cos((2.0 * math.pi) * u[1::2])
    return out.reshape(shape)


# ------------------------------------------------------------------ config builders
def c2_dense_qp(n, seed, device=None):
    """(Q, c): Q = M'M/n symmetrised, c ~ N(0,1).  device = None -> numpy arrays, else torch tensors on it
    (the matrix product is problem set-up, not part of any timed path)."""
    if device is None:
        M = randn_np(seed, n, n)
        Q = M.T @ M / n
        return 0.5 * (Q + Q.T), randn_np(seed + 1000003, n)
    M = randn_torch(seed, n, n, device=device)
    Q = (M.t() @ M) / n
    del M
    return 0.5 * (Q + Q.t()), randn_torch(seed + 1000003, n, device=device)


def c2_problem(n, seed, device=None):
    """(Q, c, A, b, cone_dims) of config 2 / 5: A = I (sparse), b = 0, one R cone; c always a numpy vector."""
    import scipy.sparse as sp
    Q, c = c2_dense_qp(n, seed, device)
    if device is not None:
        c = c.cpu().numpy()
    return Q, c, sp.identity(n, format="csr"), np.zeros(n), [("R", n)]


def c1_readme_boxqp(n=1000, seed=42, density=0.1):
    """README.md:56-65: Q = sprandn(n,n,0.1)' * sprandn(n,n,0.1) (the same B twice), c = ones, A = I, b = 0."""
    import scipy.sparse as sp
    mask = uniform_np(seed, n * n).reshape(n, n) < density
    B = np.where(mask, randn_np(seed + 7, n, n), 0.0)
    Q = B.T @ B
    return 0.5 * (Q + Q.T), np.ones(n), sp.identity(n, format="csr"), np.zeros(n), [("R", n)]


def c3_socp(n=4096, ncones=512, kq=8, p=512, seed=11):
    """Config 3: ncones x ("Q", kq), dense A / sqrt(n), head rows b = -1 (strictly feasible at y = 0,
    the pattern of benchmark/profile.jl:53-69), Q = I, equality block G (p x n), d = 0."""
    m = ncones * kq
    A = randn_np(seed, m, n) / np.sqrt(n)
    b = np.zeros(m)
    b[::kq] = -1.0
    G = randn_np(seed + 1, p, n)
    return np.eye(n), randn_np(seed + 2, n), A, b, [("Q", kq)] * ncones, G, np.zeros(p)


def soc_single(n=500, seed=42, dense=False):
    """The reference's "single large SOC" benchmark problem (benchmark/profile.jl:43-52; benchmark/report.md:57-59: 6
    iterations on its draw): Q = I, c ~ N(0,1), A = [0; I] (sparse unless `dense`), b = [-1; 0], one ("Q", n+1) cone, i.e.
    minimise 1/2 |y|^2 - c'y over the unit ball -- solution c / max(1, |c|)."""
    import scipy.sparse as sp
    A = sp.vstack([sp.csr_matrix((1, n)), sp.identity(n, format="csr")], format="csr")
    b = np.zeros(n + 1)
    b[0] = -1.0
    return (sp.identity(n, format="csr") if not dense else np.eye(n)), randn_np(seed, n), (A.toarray() if dense else A), b, \
        [("Q", n + 1)]


def soc_many_small(n=500, k=250, seed=42, density=0.1):
    """The reference's "many small SOCs" benchmark problem (benchmark/profile.jl:54-69; report.md:60-62: 9 iterations on
    its draw): k cones ("Q", 3), A = sprandn(3k, n, 0.1), head rows b = -1, Q = I."""
    import scipy.sparse as sp
    m = 3 * k
    mask = uniform_np(seed, m * n).reshape(m, n) < density
    A = sp.csr_matrix(np.where(mask, randn_np(seed + 7, m, n), 0.0))
    b = np.zeros(m)
    b[::3] = -1.0
    return sp.identity(n, format="csr"), randn_np(seed + 1, n), A, b, [("Q", 3)] * k


def soc_large_dense(n=4096, seed=21):
    """One ("Q", n+1) cone behind a DENSE A: head row zero with b = -1 (the bound), tail rows N(0,1)/sqrt(n), b = 0 --
    |A_tail y| <= 1, strictly feasible at y = 0; Q = I.  The large-SOC case of SURVEY 8(f3) (src/kktsolvers.jl:60-131,
    :192-240 treat such blocks specially): on the device the Q cone is two O(mn) passes beside the m n^2 SYRK."""
    A = randn_np(seed, n + 1, n) / np.sqrt(n)
    A[0, :] = 0.0
    b = np.zeros(n + 1)
    b[0] = -1.0
    return np.eye(n), randn_np(seed + 2, n), A, b, [("Q", n + 1)]


def vecm_identity(r):
    """vecm(I_r) (src/ConicIP.jl:128-151): ones at the diagonal positions of the row-major upper triangle."""
    e = np.zeros(r * (r + 1) // 2)
    e[np.cumsum(np.concatenate([[0], np.arange(r, 1, -1)])).astype(int)] = 1.0
    return e


def c4_sdp(r=256, n=1024, p=16, seed=5):
    """Config 4: ("S", 256) is not a legal cone spec (256 is not triangular, src/ConicIP.jl:85); the reading used
    everywhere in this repo is matrix order r = 256, i.e. ("S", 32896).  A is k x n / sqrt(n), b = -vecm(I)
    (strictly feasible at y = 0), Q = I, G p x n, d = 0."""
    k = r * (r + 1) // 2
    A = randn_np(seed, k, n) / np.sqrt(n)
    return np.eye(n), randn_np(seed + 2, n), A, -vecm_identity(r), [("S", k)], randn_np(seed + 1, p, n), np.zeros(p)


def c5_batch(count=64, n=2048, seed=4000, device=None, indices=None):
    """Config 5: problems i = 0..count-1 (or `indices`), seeds seed + i, as dicts for cipkkt.batch.solve_batch."""
    out = []
    for i in (range(count) if indices is None else indices):
        Q, c, A, b, K = c2_problem(n, seed + i, device)
        out.append(dict(Q=Q, c=c, A=A, b=b, cone_dims=K, kwargs=dict(optTol=1e-6)))
    return out
