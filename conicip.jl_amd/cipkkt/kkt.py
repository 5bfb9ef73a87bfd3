"""Host-side mirror of the reference's KKT-solver plugin interface, bound to the HIP
library through the C ABI (include/cipkkt.h).

Reference interface (src/ConicIP.jl:432-466, :667, :682, :688;
docs/src/guides/kkt_solvers.md:84-115):

    solve3x3gen = kktsolver(Q, A, G, cone_dims)     # level 1, once
    solve3x3    = solve3x3gen(F, F_invT)            # level 2, per iteration
    (a, b, c)   = solve3x3(x, y, z)                 # level 3, per right-hand side

``kktsolver_hip`` is that closure trio; ``KKTSystem`` is the object behind it and
also exposes the device-pointer entry points used by the device-resident driver
(``cipkkt.driver.conicIP``).  PyTorch is used for device memory only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

_CONE_CODE = {"R": L.CONE_R, "Q": L.CONE_Q, "S": L.CONE_S}


def _ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def _is_sparse(M):
    return hasattr(M, "tocsr") and not isinstance(M, np.ndarray)


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("cipkkt: no HIP device visible -- the KKT path is GPU-only (no CPU fallback)")


def pack_scaling(cone_dims, F, FinvT=None):
    """Read the packed scaling off Block elements shaped like the reference's
    (src/ConicIP.jl:189-192 SymWoodbury(.A.diag,.B,.D); :208 VecCongurance(.R); :598 Diagonal(.diag))."""
    blocks = F.Blocks if hasattr(F, "Blocks") else list(F)
    iblocks = (FinvT.Blocks if hasattr(FinvT, "Blocks") else list(FinvT)) if FinvT is not None else [None] * len(blocks)
    out = []
    for (t, k), blk, iblk in zip(cone_dims, blocks, iblocks):
        # The reference's initial point is computed with F = F^-T = Block([Diagonal(ones(k)) ...]) for EVERY cone
        # type (src/ConicIP.jl:704-706): a (uniform) Diagonal element must be accepted for Q and S cones too.
        uniform = None
        if t != "R" and hasattr(blk, "diag") and not hasattr(blk, "B") and not hasattr(blk, "R"):
            dg = np.asarray(blk.diag, dtype=np.float64).reshape(-1)
            if dg.size != k or not np.all(dg == dg[0]) or not dg[0] > 0:
                raise ValueError("a Diagonal scaling element of a %s cone must be a positive multiple of the identity" % t)
            uniform = float(dg[0])
        if t == "R":
            out.append(np.asarray(blk.diag, dtype=np.float64).reshape(k))
        elif t == "Q":
            if uniform is not None:
                # d I = diag(-beta, beta, ...) + w w'  with beta = d, w = sqrt(2 d) e1
                w = np.zeros(k)
                w[0] = np.sqrt(2.0 * uniform)
                out.append(np.concatenate([[uniform], w]))
                continue
            Ad = blk.A.diag if hasattr(blk.A, "diag") else blk.A
            Ad = np.asarray(Ad, dtype=np.float64)
            B = np.asarray(blk.B, dtype=np.float64).reshape(k, -1)
            D = np.asarray(blk.D, dtype=np.float64).reshape(B.shape[1], B.shape[1])
            if B.shape[1] != 1:
                raise ValueError("Q-cone scaling must be rank one (diag + w w')")
            out.append(np.concatenate([[-Ad[0]], B[:, 0] * np.sqrt(D[0, 0])]))
        else:
            if uniform is not None:
                # vecm(R' X R) = d vecm(X)  <=>  R = sqrt(d) I
                r = int(round((np.sqrt(1 + 8 * k) - 1) / 2))
                R = np.sqrt(uniform) * np.eye(r)
                out.append(np.concatenate([R.reshape(-1), (R / uniform).reshape(-1)]))
                continue
            R = np.asarray(blk.R, dtype=np.float64)
            Rinv = np.asarray(iblk.R, dtype=np.float64).T if iblk is not None and hasattr(iblk, "R") else np.linalg.inv(R)
            out.append(np.concatenate([R.reshape(-1, order="F"), Rinv.reshape(-1, order="F")]))
    return np.ascontiguousarray(np.concatenate(out)) if out else np.zeros(0)



def make_problem(Q, A, G, cone_dims, route, device):
    """(cip_problem, keep-alive list, A_is_sparse): every matrix handed over as a DEVICE pointer.  Host arrays are
    uploaded in whatever (row-major) layout they have and re-laid out column-major by a device transpose --
    numpy.asfortranarray of a 2048 x 2048 Q alone cost 33 ms of the 42 ms level 1 took, cip_create_ex itself 3 ms.
    The staging copies run on torch's current stream: synchronise it before handing the struct to the library."""
    n = Q.shape[0]
    m = A.shape[0] if A is not None else 0
    p = G.shape[0] if G is not None else 0
    keep = []
    pr = L.CipProblem()
    pr.n, pr.m, pr.p, pr.ncones = n, m, p, len(cone_dims)
    ct = (C.c_int * max(1, len(cone_dims)))(*[_CONE_CODE[t] for t, _ in cone_dims])
    cdm = (C.c_int * max(1, len(cone_dims)))(*[k for _, k in cone_dims])
    keep += [ct, cdm]
    pr.cone_type, pr.cone_dim = ct, cdm

    def dense(M, rows, cols):
        """column-major fp64 device buffer"""
        if not isinstance(M, torch.Tensor):
            M = torch.from_numpy(np.ascontiguousarray(M.toarray() if _is_sparse(M) else M, dtype=np.float64))
        Mt = M.to(dtype=torch.float64, device=device).reshape(rows, cols).t().contiguous()
        keep.append(Mt)            # row-major of M' == column-major of M
        return C.c_void_p(Mt.data_ptr())

    pr.Q, pr.ldq = dense(Q, n, n), n
    a_sparse = False
    csr_host = False
    if m > 0 and _is_sparse(A):          # (S cones too, round 4: the library expands their rows on the device)
        csr = A.tocsr()
        csr.sort_indices()
        # the CSR arrays stay on the host (CIP_FLAG_CSR_HOST): the library builds the CSR of A' there and uploads both
        rp_h, ci_h, av_h = (np.ascontiguousarray(x, dtype=dt)
                            for x, dt in ((csr.indptr, np.int32), (csr.indices, np.int32), (csr.data, np.float64)))
        keep += [rp_h, ci_h, av_h]
        pr.A_rowptr, pr.A_colind, pr.A_val = (C.c_void_p(x.ctypes.data) for x in (rp_h, ci_h, av_h))
        pr.A = None
        a_sparse = True
        csr_host = True
    else:
        pr.A, pr.lda = (dense(A, m, n) if m > 0 else None), max(m, 1)
    pr.G, pr.ldg = (dense(G, p, n) if p > 0 else None), max(p, 1)
    pr.route = L.ROUTE_SCHUR if route in ("schur", L.ROUTE_SCHUR) else L.ROUTE_FULL3X3
    pr.flags = L.FLAG_DEVICE_PTRS | (L.FLAG_CSR_HOST if csr_host else 0)
    return pr, keep, a_sparse


class KKTSystem:
    """Level-1 object: problem matrices resident in HBM, cone layout, workspaces.
    ≙ what `kktsolver(Q,A,G,cone_dims)` captures (src/kktsolvers.jl:18-28, :180-190, :281-285)."""

    def __init__(self, Q, A, G, cone_dims, route="schur", device=None):
        _require_gpu()
        self.lib = L.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.cone_dims = [(str(t), int(k)) for t, k in cone_dims]
        n = Q.shape[0]
        if Q.shape[0] != Q.shape[1]:
            raise ValueError("Q is not square")                       # src/ConicIP.jl:538
        m = A.shape[0] if A is not None else 0
        if G is None:
            G = np.zeros((0, n))
        p = G.shape[0]
        if (m > 0 and A.shape[1] != n) or (p > 0 and G.shape[1] != n):
            raise ValueError("Inconsistency in inequalities/objective")  # :540-542
        if sum(k for _, k in self.cone_dims) != m:
            raise ValueError("cone_dims do not cover the rows of A")
        self.n, self.m, self.p = n, m, p
        self.route = L.ROUTE_SCHUR if route in ("schur", L.ROUTE_SCHUR) else L.ROUTE_FULL3X3

        pr, keep, self.A_sparse = make_problem(Q, A, G, self.cone_dims, self.route, self.device)
        h = C.c_void_p()
        torch.cuda.current_stream(self.device).synchronize()    # staging transposes ran on torch's current stream
        L.check(self.lib.cip_create_ex(C.byref(pr), C.byref(h)))
        self.h = h
        N, Np = C.c_int(), C.c_int()
        L.check(self.lib.cip_kkt_order(self.h, C.byref(N), C.byref(Np)))
        self.N, self.Npad = N.value, Np.value
        self.scaling_len = int(self.lib.cip_scaling_packed_len(self.h))
        del keep

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.cip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- level 2
    def pack_scaling(self, F, FinvT=None):
        return pack_scaling(self.cone_dims, F, FinvT)

    def set_scaling_packed(self, packed):
        packed = np.ascontiguousarray(packed, dtype=np.float64)
        if packed.size != self.scaling_len:
            raise ValueError("packed scaling has %d entries, expected %d" % (packed.size, self.scaling_len))
        L.check(self.lib.cip_set_scaling_packed(self.h, _ptr(packed)))

    def get_scaling_packed(self):
        out = np.zeros(self.scaling_len)
        L.check(self.lib.cip_get_scaling_packed(self.h, _ptr(out)))
        return out

    def set_scaling_identity(self):
        L.check(self.lib.cip_set_scaling_identity(self.h))

    def set_scaling_from_iterate(self, v, s, lam_out=None):
        L.check(self.lib.cip_set_scaling_from_iterate_dev(self.h, _ptr(v), _ptr(s), _ptr(lam_out)))

    def assemble_only(self):
        L.check(self.lib.cip_assemble_only(self.h))

    def factor(self, check=True):
        """Level 2.  check=True (default) waits for the factorisation and raises on a bad pivot; check=False only enqueues
        it (the native loop's use: the flag is resolved by the next solve, see include/cipkkt.h: cip_factor)."""
        L.check(self.lib.cip_factor(self.h))
        if check:
            self.check_factor()

    def check_factor(self):
        L.check(self.lib.cip_check_factor(self.h))

    # ---------------------------------------------------------------- level 3
    def solve3x3(self, x, y, z):
        """Host-pointer ABI (what the Julia ccall shim uses); returns fresh arrays,
        as the reference requires of level 3 (src/ConicIP.jl:690)."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.n)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(self.p)
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(self.m)
        a, b, c = np.empty(self.n), np.empty(self.p), np.empty(self.m)
        L.check(self.lib.cip_solve3x3(self.h, _ptr(x), _ptr(y), _ptr(z), _ptr(a), _ptr(b), _ptr(c)))
        return a, b, c

    def solve2x2(self, y, w):
        """The 2x2 form (src/ConicIP.jl:450-466): [Q + A'(F'F)^-1 A, G'; G, 0][dy; dw] = [y; w]; fresh host arrays."""
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(self.n)
        w = np.ascontiguousarray(w, dtype=np.float64).reshape(self.p)
        dy, dw = np.empty(self.n), np.empty(self.p)
        L.check(self.lib.cip_solve2x2(self.h, _ptr(y), _ptr(w), _ptr(dy), _ptr(dw)))
        return dy, dw

    def solve2x2_dev(self, y, w, dy, dw):
        L.check(self.lib.cip_solve2x2_dev(self.h, _ptr(y), _ptr(w), _ptr(dy), _ptr(dw)))

    def solve3x3_dev(self, x, y, z, a, b, c):
        L.check(self.lib.cip_solve3x3_dev(self.h, _ptr(x), _ptr(y), _ptr(z), _ptr(a), _ptr(b), _ptr(c)))

    def solve4x4_dev(self, lam, r, dz):
        L.check(self.lib.cip_solve4x4_dev(self.h, _ptr(lam), _ptr(r), _ptr(dz)))

    # ---------------------------------------------------------------- cone / vector helpers (device tensors)
    def apply_F(self, mode, x, out):
        L.check(self.lib.cip_apply_F_dev(self.h, mode, _ptr(x), _ptr(out)))

    def cone_prod(self, x, y, out):
        L.check(self.lib.cip_cone_prod_dev(self.h, _ptr(x), _ptr(y), _ptr(out)))

    def cone_div(self, x, y, out):
        L.check(self.lib.cip_cone_div_dev(self.h, _ptr(x), _ptr(y), _ptr(out)))

    def maxstep(self, x, d=None, scale=1.0):
        out = C.c_double()
        L.check(self.lib.cip_maxstep_dev(self.h, _ptr(x), _ptr(d), float(scale), C.byref(out)))
        return out.value

    def maxstep_pair(self, x1, d1, x2, d2, scale=1.0):
        """(maxstep(x1, d1 * scale), maxstep(x2, d2 * scale)) with one wait (src/ConicIP.jl:708-709, :881-882, :927-928)."""
        out = (C.c_double * 2)()
        L.check(self.lib.cip_maxstep_pair_dev(self.h, _ptr(x1), _ptr(d1), _ptr(x2), _ptr(d2), float(scale), out))
        return out[0], out[1]

    def cone_identity(self, e):
        L.check(self.lib.cip_cone_identity_dev(self.h, _ptr(e)))

    def gemv(self, which, trans, alpha, x, beta, y):
        L.check(self.lib.cip_gemv_dev(self.h, which, int(trans), float(alpha), _ptr(x), float(beta), _ptr(y)))

    def dots(self, pairs):
        k = len(pairs)
        xs = (C.c_void_p * k)(*[p[0].data_ptr() for p in pairs])
        ys = (C.c_void_p * k)(*[p[1].data_ptr() for p in pairs])
        ln = (C.c_int * k)(*[min(p[0].numel(), p[1].numel()) for p in pairs])
        out = (C.c_double * k)()
        L.check(self.lib.cip_dots_dev(self.h, k, xs, ys, ln, out))
        return list(out)

    def axpby(self, alpha, x, beta, y):
        L.check(self.lib.cip_axpby_dev(self.h, y.numel(), float(alpha), _ptr(x), float(beta), _ptr(y)))

    # ---------------------------------------------------------------- introspection
    def kkt_matrix(self):
        K = np.empty((self.Npad, self.Npad), order="F")
        L.check(self.lib.cip_get_kkt_matrix(self.h, _ptr(K)))
        return K

    def stats(self):
        out = (C.c_double * 8)()
        L.check(self.lib.cip_stats(self.h, out))
        return dict(n_factor=out[0], n_solve=out[1], ms_assemble=out[2], ms_ldlt=out[3], flops_ldlt=out[4],
                    nbo=out[5], N=out[6], Npad=out[7])

    def health(self):
        """(relative static regularisation in force, times it was switched on, factorisations redone on the three-launch chain)"""
        rel = C.c_double(0.0)
        k = C.c_int(0)
        L.check(self.lib.cip_get_regularization(self.h, C.byref(rel), C.byref(k)))
        return dict(reg_rel=rel.value, n_regularized=k.value, n_chain_fallbacks=int(self.lib.cip_get_chain_fallbacks(self.h)))

    def profile_trailing(self, on):
        """HIP events around the trailing-update launches: True / 1 = every factorisation, k > 1 = every k-th, False = off."""
        L.check(self.lib.cip_profile_trailing(self.h, int(on)))

    def profile_get(self):
        out = (C.c_double * 3)()
        L.check(self.lib.cip_profile_get(self.h, out))
        return dict(launches=out[0], ms=out[1], flops=out[2])

    def set_timing(self, on):
        L.check(self.lib.cip_set_timing(self.h, int(bool(on))))

    def set_stream(self, stream):
        L.check(self.lib.cip_set_stream(self.h, C.c_void_p(stream)))


def kktsolver_hip(Q, A, G, cone_dims, route="schur", device=None):
    """The reference's 3-level plugin closure, backed by the HIP library.

        solve3x3gen = kktsolver_hip(Q, A, G, cone_dims)
        solve3x3    = solve3x3gen(F, F_invT)
        a, b, c     = solve3x3(x, y, z)

    solves [Q G' -A'; G 0 0; A 0 F'F][a;b;c] = [x;y;z] (src/ConicIP.jl:443-447).
    F / F_invT are Block-like objects with the reference's element fields."""
    sysm = KKTSystem(Q, A, G, cone_dims, route=route, device=device)

    def solve3x3gen(F, FinvT=None):
        sysm.set_scaling_packed(sysm.pack_scaling(F, FinvT))
        sysm.factor()

        def solve3x3(x, y, z):
            return sysm.solve3x3(x, y, z)

        return solve3x3

    solve3x3gen.system = sysm
    return solve3x3gen


def kktsolver_2x2_hip(Q, A, G, cone_dims, device=None):
    """The reference's 2x2 plugin form (src/ConicIP.jl:450-466; src/kktsolvers.jl:281-310) on the device:

        solve2x2gen = kktsolver_2x2_hip(Q, A, G, cone_dims)
        solve2x2    = solve2x2gen(F, F_invT)
        dy, dw      = solve2x2(y, w)          # [Q + A'(F'F)^-1 A, G'; G, 0][dy; dw] = [y; w]

    to be wrapped by `pivot` exactly as `pivot(ConicIP.kktsolver_2x2)`."""
    sysm = KKTSystem(Q, A, G, cone_dims, route="schur", device=device)

    def solve2x2gen(F, FinvT=None):
        sysm.set_scaling_packed(sysm.pack_scaling(F, FinvT))
        sysm.factor()
        return lambda y, w: sysm.solve2x2(y, w)

    solve2x2gen.system = sysm
    return solve2x2gen


def pivot(kktsolver_2x2):
    """`pivot` of the reference (src/kktsolvers.jl:316-349): wraps a 2x2 solver -- `kktsolver_2x2_hip` or any
    user-written one with the same three-level shape -- into the 3x3 plugin interface by eliminating the third
    block row:  t = F^-T(F^-T v);  (dy, dw) = solve2x2(y + A't, w);  dv = t - F^-T(F^-T(A dy)).
    F^-T is the host Block the caller hands over (its `mul`), A any matrix with `@`/`.T`."""

    def kktsolver(Q, A, G, cone_dims):
        solve2x2gen = kktsolver_2x2(Q, A, G, cone_dims)
        At = A.T

        def solve3x3gen(F, FinvT):
            solve2x2 = solve2x2gen(F, FinvT)

            def solve3x3(y, w, v):
                t1 = FinvT.mul(FinvT.mul(np.asarray(v, dtype=np.float64)))          # :326
                dy, dw = solve2x2(np.asarray(y, dtype=np.float64) + At @ t1, w)     # :327
                t1 = t1 - FinvT.mul(FinvT.mul(A @ dy))                              # :328
                return np.asarray(dy), np.asarray(dw), t1

            return solve3x3

        if hasattr(solve2x2gen, "system"):
            solve3x3gen.system = solve2x2gen.system
        return solve3x3gen

    return kktsolver


def kktsolver_hip_full3x3(Q, A, G, cone_dims, device=None):
    """Same interface, literal 3x3 assembly route (src/kktsolvers.jl:254-256)."""
    return kktsolver_hip(Q, A, G, cone_dims, route="full3x3", device=device)
