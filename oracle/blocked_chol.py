"""TEST / BASELINE INFRASTRUCTURE (never imported by the product): a right-looking blocked Cholesky on the host BLAS, used by
bench.py's `cpu_baseline` leg as a second "strong CPU" contender beside LAPACK's potrf.

Why: on a many-core host OpenBLAS' `dpotrf` is threading-starved (545 GFLOP/s on a box whose `dgemm` reaches 1 339, round-4
review) -- its panel factorisation serialises.  A right-looking blocked factorisation spends all but O(n^2 nb) of its n^3/3 flops
in `dsyrk` / `dgemm`, which do scale.  Algebra: the Schur + Cholesky route of the GPU path (reference role:
src/kktsolvers.jl:281-338 with a symmetric factorisation in place of UMFPACK's LU).

The BLAS / LAPACK routines are called IN PLACE on sub-blocks of one column-major matrix (leading dimension = n), through the
function pointers scipy exports in `scipy.linalg.cython_blas.__pyx_capi__` -- scipy's Python wrappers (`scipy.linalg.blas.dsyrk`)
take no leading dimension and would copy every sub-block.
"""
import ctypes as C

import numpy as np


def _fptr(mod, name, argtypes):
    cap = mod.__pyx_capi__[name]
    api = C.pythonapi
    api.PyCapsule_GetName.restype = C.c_char_p
    api.PyCapsule_GetName.argtypes = [C.py_object]
    api.PyCapsule_GetPointer.restype = C.c_void_p
    api.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    addr = api.PyCapsule_GetPointer(cap, api.PyCapsule_GetName(cap))
    return C.CFUNCTYPE(None, *argtypes)(addr)


_cp, _ip, _dp = C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
_fn = {}


def _load():
    if _fn:
        return _fn
    import scipy.linalg.cython_blas as cb
    import scipy.linalg.cython_lapack as cl
    vp = C.c_void_p
    _fn["dgemm"] = _fptr(cb, "dgemm", [_cp, _cp, _ip, _ip, _ip, _dp, vp, _ip, vp, _ip, _dp, vp, _ip])
    _fn["dsyrk"] = _fptr(cb, "dsyrk", [_cp, _cp, _ip, _ip, _dp, vp, _ip, _dp, vp, _ip])
    _fn["dtrsm"] = _fptr(cb, "dtrsm", [_cp, _cp, _cp, _cp, _ip, _ip, _dp, vp, _ip, vp, _ip])
    _fn["dpotrf"] = _fptr(cl, "dpotrf", [_cp, _ip, vp, _ip, _ip])
    return _fn


def _i(x):
    return C.byref(C.c_int(x))


def _d(x):
    return C.byref(C.c_double(x))


def blocked_cholesky(S, nb=512, trailing="syrk"):
    """In place: the lower triangle of the column-major (F-ordered) SPD matrix S becomes its Cholesky factor L (S = L L').
    Right-looking, block `nb`: dpotrf on the diagonal block, dtrsm for the panel below it, then the trailing update as ONE
    dsyrk (trailing="syrk") or as one dgemm per block column of the lower triangle (trailing="gemm": more, smaller calls, but
    dgemm threads better than dsyrk in some BLAS builds).  Returns 0, or LAPACK's info of the failing diagonal block."""
    f = _load()
    assert S.flags.f_contiguous and S.dtype == np.float64 and S.shape[0] == S.shape[1]
    n = S.shape[0]
    base = S.ctypes.data
    at = lambda i, j: C.c_void_p(base + 8 * (i + j * n))
    info = C.c_int(0)
    for k in range(0, n, nb):
        kb = min(nb, n - k)
        f["dpotrf"](b"L", _i(kb), at(k, k), _i(n), C.byref(info))
        if info.value:
            return info.value + k
        r = n - k - kb
        if r <= 0:
            break
        # panel: A21 <- A21 L11^-T
        f["dtrsm"](b"R", b"L", b"T", b"N", _i(r), _i(kb), _d(1.0), at(k, k), _i(n), at(k + kb, k), _i(n))
        if trailing == "syrk":
            f["dsyrk"](b"L", b"N", _i(r), _i(kb), _d(-1.0), at(k + kb, k), _i(n), _d(1.0), at(k + kb, k + kb), _i(n))
        else:
            for j in range(k + kb, n, nb):
                jb = min(nb, n - j)
                # C[j:, j:j+jb] -= A[j:, k:k+kb] A[j:j+jb, k:k+kb]'   (the block column from its diagonal block down)
                f["dgemm"](b"N", b"T", _i(n - j), _i(jb), _i(kb), _d(-1.0), at(j, k), _i(n), at(j, k), _i(n), _d(1.0), at(j, j), _i(n))
    return 0


def cholesky_solve(L, rhs):
    """x with L L' x = rhs (L: the lower triangle left by blocked_cholesky); two dtrsm calls on a copy of rhs."""
    f = _load()
    n = L.shape[0]
    x = np.array(rhs, dtype=np.float64, order="F").reshape(n, -1, order="F")
    k = x.shape[1]
    p = C.c_void_p(x.ctypes.data)
    f["dtrsm"](b"L", b"L", b"N", b"N", _i(n), _i(k), _d(1.0), C.c_void_p(L.ctypes.data), _i(n), p, _i(n))
    f["dtrsm"](b"L", b"L", b"T", b"N", _i(n), _i(k), _d(1.0), C.c_void_p(L.ctypes.data), _i(n), p, _i(n))
    return x.reshape(np.shape(rhs), order="F")
