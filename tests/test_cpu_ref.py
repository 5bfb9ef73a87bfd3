"""The contract of the C ABI, exercised WITHOUT a GPU: oracle/cpu_ref/cipkkt_cpu.cpp implements the plugin levels of
include/cipkkt.h on the host (literal 3x3 system + LU, the reference's kktsolver_sparse algebra), and the SAME ctypes
table (cipkkt/_lib.py SIGNATURES), the SAME packer (cipkkt.pack_scaling) and the SAME plain-C program
(tests/c_abi/solve_qp.c) that drive libcipkkt.so on the GPU drive it here.  What this pins in the build container:
argument order and meaning, column-major / CSR conventions, the packed-scaling layout per cone type, the identity
scaling of the initial point, the 2x2 form, and the error codes.  The CPU library is test infrastructure (never loaded
by the product; `tests/test_abi_symbols.py::test_product_does_not_import_oracle`)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

import problems as P
from oracle.conicip import make_cone_ops
from oracle.kktsolvers import kktsolver_2x2, kktsolver_qr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "oracle", "cpu_ref", "cipkkt_cpu.cpp")
OUT = os.path.join(ROOT, "oracle", "cpu_ref", "_build")
pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")



def _header_codes():
    import re
    txt = open(os.path.join(ROOT, "include", "cipkkt.h")).read()
    return {k: int(v) for k, v in re.findall(r"#define\s+(CIP_E_\w+)\s+(-?\d+)", txt)}


_CODES = _header_codes()
E_INVALID, E_NOTFACTORED, E_SINGULAR, E_UNSUPPORTED = (_CODES[k] for k in ("CIP_E_INVALID", "CIP_E_NOTFACTORED", "CIP_E_SINGULAR", "CIP_E_UNSUPPORTED"))


@pytest.fixture(scope="module")
def cpu():
    from cipkkt import _lib as L
    os.makedirs(OUT, exist_ok=True)
    so = os.environ.get("CIP_CPU_REF_SO")        # the -fsanitize=address,undefined build (oracle/cpu_ref/build_asan.py)
    if not so:
        so = os.path.join(OUT, "libcipkkt_cpu.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(SRC):
            subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, SRC], check=True)
    lib = C.CDLL(so)
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def test_error_codes_match_the_header(cpu):
    from cipkkt import _lib as L
    assert L.E_SINGULAR == E_SINGULAR and len({E_INVALID, E_NOTFACTORED, E_SINGULAR, E_UNSUPPORTED}) == 4


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class CpuSystem:
    """host-pointer twin of cipkkt.KKTSystem over the CPU library"""

    def __init__(self, lib, Q, A, G, cone_dims, route=0, csr=False):
        from cipkkt import _lib as L
        from cipkkt.kkt import _CONE_CODE
        self.lib, self.cone_dims = lib, cone_dims
        self.n, self.m, self.p = Q.shape[0], A.shape[0], (G.shape[0] if G is not None else 0)
        pr = L.CipProblem()
        pr.n, pr.m, pr.p, pr.ncones = self.n, self.m, self.p, len(cone_dims)
        self._ct = (C.c_int * len(cone_dims))(*[_CONE_CODE[t] for t, _ in cone_dims])
        self._cd = (C.c_int * len(cone_dims))(*[k for _, k in cone_dims])
        pr.cone_type, pr.cone_dim = self._ct, self._cd
        self._Q = np.asfortranarray(Q, dtype=np.float64)
        pr.Q, pr.ldq = _ptr(self._Q), self.n
        if csr:
            a = sp.csr_matrix(A)
            self._A = (a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64))
            pr.A_rowptr, pr.A_colind, pr.A_val = (_ptr(x) for x in self._A)
        else:
            self._A = np.asfortranarray(A.toarray() if sp.issparse(A) else A, dtype=np.float64)
            pr.A, pr.lda = _ptr(self._A), self.m
        if self.p:
            self._G = np.asfortranarray(G, dtype=np.float64)
            pr.G, pr.ldg = _ptr(self._G), self.p
        pr.route, pr.flags = route, 0
        h = C.c_void_p()
        rc = lib.cip_create_ex(C.byref(pr), C.byref(h))
        assert rc == 0, lib.cip_last_error()
        self.h = h

    def solve3x3(self, x, y, z):
        a, b, c = np.zeros(self.n), np.zeros(self.p), np.zeros(self.m)
        rc = self.lib.cip_solve3x3(self.h, _ptr(x), _ptr(y), _ptr(z), _ptr(a), _ptr(b), _ptr(c))
        return rc, a, b, c

    def close(self):
        self.lib.cip_destroy(self.h)


def _interior(cone_dims, rng):
    xs = []
    for t, k in cone_dims:
        if t == "R":
            xs.append(rng.random(k) + 0.1)
        elif t == "Q":
            x = rng.standard_normal(k)
            x[0] = np.linalg.norm(x[1:]) + rng.random() + 0.1
            xs.append(x)
        else:
            from oracle.cones import vecm
            r = int(round((np.sqrt(1 + 8 * k) - 1) / 2))
            M = rng.standard_normal((r, r))
            xs.append(vecm(M @ M.T + 0.5 * np.eye(r)))
    return np.concatenate(xs)


CASES = [
    dict(cone_dims=[("R", 7)], n=5, p=0),
    dict(cone_dims=[("R", 4), ("Q", 5), ("Q", 3)], n=6, p=2),
    dict(cone_dims=[("S", 6), ("R", 3), ("Q", 4), ("S", 10)], n=8, p=1),
]


@pytest.mark.parametrize("case", CASES, ids=["R", "RQQ", "SRQS"])
@pytest.mark.parametrize("csr", [False, True], ids=["denseA", "csrA"])
def test_three_levels_against_the_oracle(cpu, case, csr):
    """level 1 (create) -> level 2 (packed NT scaling + factor) -> level 3 (solve3x3), vs the oracle's kktsolver_qr
    restatement (src/kktsolvers.jl:18-58) called the way the reference calls a kktsolver."""
    import cipkkt.kkt
    rng = np.random.default_rng(5)
    cone_dims, n, p = case["cone_dims"], case["n"], case["p"]
    m = sum(k for _, k in cone_dims)
    M = rng.standard_normal((n, n))
    Q = M @ M.T / n + 0.5 * np.eye(n)
    A = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.6)
    G = rng.standard_normal((p, n))
    ks = CpuSystem(cpu, Q, A, G if p else None, cone_dims, csr=csr)
    _, nt_scaling, _, _ = make_cone_ops(cone_dims)
    F = nt_scaling(_interior(cone_dims, rng), _interior(cone_dims, rng))
    packed = cipkkt.kkt.pack_scaling(cone_dims, F, F.inv_adjoint())
    assert packed.size == cpu.cip_scaling_packed_len(ks.h)
    # solving before factoring is an error with the documented code
    x, y, z = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    assert cpu.cip_set_scaling_packed(ks.h, _ptr(packed)) == 0
    assert ks.solve3x3(x, y, z)[0] == E_NOTFACTORED
    assert cpu.cip_factor(ks.h) == 0 and cpu.cip_check_factor(ks.h) == 0
    rc, a, b, c = ks.solve3x3(x, y, z)
    assert rc == 0
    ra, rb, rc_ = kktsolver_qr(Q, A, G, cone_dims)(F, F.inv_adjoint())(x, y, z)
    ref = np.concatenate([ra, rb, rc_])
    got = np.concatenate([a, b, c])
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-10
    # round trip of the packed scaling
    back = np.zeros_like(packed)
    assert cpu.cip_get_scaling_packed(ks.h, _ptr(back)) == 0 and np.array_equal(back, packed)
    # 2x2 form (src/ConicIP.jl:450-466): [Q + A'(F'F)^-1 A, G'; G, 0]
    dy, dw = np.zeros(n), np.zeros(p)
    assert cpu.cip_solve2x2(ks.h, _ptr(x), _ptr(y), _ptr(dy), _ptr(dw)) == 0
    r2y, r2w = kktsolver_2x2(Q, A, G, cone_dims)(F, F.inv_adjoint())(x, y)
    ref2 = np.concatenate([r2y, r2w])
    assert np.linalg.norm(np.concatenate([dy, dw]) - ref2) / np.linalg.norm(ref2) < 1e-10
    ks.close()


def test_identity_scaling_is_the_initial_point_block(cpu):
    """cip_set_scaling_identity == pack_scaling(Block([Diagonal(ones(k)) ...])) for every cone type
    (the reference's initial-point call, src/ConicIP.jl:704-706)."""
    import cipkkt.kkt
    from oracle.block import identity_block
    cone_dims = [("R", 3), ("Q", 4), ("S", 6)]
    rng = np.random.default_rng(2)
    n, m = 5, 13
    M = rng.standard_normal((n, n))
    Q = M @ M.T + np.eye(n)
    A = rng.standard_normal((m, n))
    ks = CpuSystem(cpu, Q, A, None, cone_dims)
    ident = np.zeros(cpu.cip_scaling_packed_len(ks.h))
    assert cpu.cip_get_scaling_packed(ks.h, _ptr(ident)) == 0            # a fresh handle holds F = I
    Iblk = identity_block([k for _, k in cone_dims])
    assert np.array_equal(ident, cipkkt.kkt.pack_scaling(cone_dims, Iblk, Iblk))
    assert cpu.cip_factor(ks.h) == 0
    x, z = rng.standard_normal(n), rng.standard_normal(m)
    rc, a, _, c = ks.solve3x3(x, np.zeros(0), z)
    assert rc == 0
    K = np.block([[Q, -A.T], [A, np.eye(m)]])
    assert np.linalg.norm(K @ np.concatenate([a, c]) - np.concatenate([x, z])) < 1e-10
    ks.close()


def test_invalid_arguments(cpu):
    from cipkkt import _lib as L
    Q = np.eye(3)
    A = np.ones((4, 3))
    h = C.c_void_p()
    ct = (C.c_int * 1)(L.CONE_S)
    cd = (C.c_int * 1)(4)                                  # 4 is not a triangular number (src/ConicIP.jl:85)
    Qf, Af = np.asfortranarray(Q), np.asfortranarray(A)
    assert cpu.cip_create(3, 4, 0, 1, ct, cd, _ptr(Qf), _ptr(Af), None, 0, C.byref(h)) == E_INVALID
    assert b"triangular" in cpu.cip_last_error()
    ct = (C.c_int * 1)(L.CONE_R)
    cd = (C.c_int * 1)(3)                                  # cone dims do not cover A's rows
    assert cpu.cip_create(3, 4, 0, 1, ct, cd, _ptr(Qf), _ptr(Af), None, 0, C.byref(h)) == E_INVALID
    cd = (C.c_int * 1)(4)
    assert cpu.cip_create(3, 4, 0, 1, ct, cd, _ptr(Qf), None, None, 0, C.byref(h)) == E_INVALID       # A missing
    assert cpu.cip_create(3, 4, 0, 1, ct, cd, _ptr(Qf), _ptr(Af), None, 0, C.byref(h)) == 0
    assert cpu.cip_conicip(h, None, None, None, None, None, None, None, None, None, 0) == E_UNSUPPORTED
    cpu.cip_destroy(h)


def test_singular_system_is_reported(cpu):
    """Q = 0 and a zero column of A: a structurally singular KKT matrix -> CIP_E_SINGULAR at check / solve time."""
    from cipkkt import _lib as L
    n, m = 3, 2
    Qf = np.zeros((n, n), order="F")
    Af = np.asfortranarray(np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]))
    ct, cd = (C.c_int * 1)(L.CONE_R), (C.c_int * 1)(m)
    h = C.c_void_p()
    assert cpu.cip_create(n, m, 0, 1, ct, cd, _ptr(Qf), _ptr(Af), None, 0, C.byref(h)) == 0
    assert cpu.cip_factor(h) == 0
    assert cpu.cip_check_factor(h) == E_SINGULAR
    cpu.cip_destroy(h)


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_program_plugin_levels_on_the_cpu_reference(cpu, tmp_path):
    """tests/c_abi/solve_qp.c (levels 1-3) compiled against include/cipkkt.h and linked with the CPU reference."""
    exe = str(tmp_path / "solve_qp_cpu")
    subprocess.run(["gcc", "-std=c99", "-O1", "-DCIP_PLUGIN_LEVELS_ONLY", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "solve_qp.c"), "-L", OUT, "-lcipkkt_cpu", "-lm",
                    "-Wl,-rpath," + OUT, "-o", exe], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "solve3x3 residuals" in r.stdout
