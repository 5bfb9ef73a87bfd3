"""Host-side Block objects handed to USER-SUPPLIED kktsolver plugins.

The reference passes its plugins a `Block` of per-cone operators (src/blockmatrices.jl:35-131)
whose elements are `Diagonal` (R cone, src/ConicIP.jl:598), `SymWoodbury` (Q cone, :189-192) and
`VecCongurance` (S cone, :208); a plugin reads their fields (`.diag`, `.A.diag/.B/.D`, `.R`) and
uses a little algebra on them (`F[1]*F[1]`, `inv(...)`, `F'`, `F*x` -- test/runtests.jl:102-110,
docs/src/guides/kkt_solvers.md).  When `cipkkt.conicIP` drives a plugin that is a Python callable,
it rebuilds exactly those objects from the packed scaling the device computed
(`cip_get_scaling_packed`) so that a plugin written against the reference's interface runs
unchanged.  The HIP solver itself never touches these classes: it reads the packed form.

Elements expose: `.size`, `mul(x)` / `@`, `tmul(x)`, `inv()`, `adjoint()` / `.T`, `matrix()`,
and `*` between elements of the same kind (src/blockmatrices.jl:173-200, src/ConicIP.jl:69-83).
"""
import numpy as np


class Diagonal:
    """LinearAlgebra.Diagonal: field `diag`."""

    def __init__(self, diag):
        self.diag = np.asarray(diag, dtype=np.float64).reshape(-1)

    @property
    def size(self):
        return self.diag.size

    def mul(self, x):
        x = np.asarray(x, dtype=np.float64)
        return self.diag * x if x.ndim == 1 else self.diag[:, None] * x

    tmul = mul

    def inv(self):
        return Diagonal(1.0 / self.diag)

    def adjoint(self):
        return self

    def matrix(self):
        return np.diag(self.diag)

    def __mul__(self, other):
        if isinstance(other, Diagonal):
            return Diagonal(self.diag * other.diag)
        return self.mul(other)

    __matmul__ = __mul__
    T = property(adjoint)


class SymWoodbury:
    """WoodburyMatrices.SymWoodbury(A, B, D) = A + B D B' with A diagonal (fields `A`, `B`, `D`)."""

    def __init__(self, A, B, D):
        self.A = A if isinstance(A, Diagonal) else Diagonal(A)
        B = np.asarray(B, dtype=np.float64)
        self.B = B.reshape(B.shape[0], -1)
        self.D = np.asarray(D, dtype=np.float64).reshape(self.B.shape[1], self.B.shape[1])

    @property
    def size(self):
        return self.A.size

    def mul(self, x):
        x = np.asarray(x, dtype=np.float64)
        return self.A.mul(x) + self.B @ (self.D @ (self.B.T @ x))

    tmul = mul                       # symmetric

    def inv(self):
        # (A + B D B')^-1 = A^-1 - A^-1 B (D^-1 + B'A^-1 B)^-1 B'A^-1
        ai = 1.0 / self.A.diag
        AiB = ai[:, None] * self.B
        cap = np.linalg.inv(np.linalg.inv(self.D) + self.B.T @ AiB)
        return SymWoodbury(Diagonal(ai), AiB, -cap)

    def adjoint(self):
        return self

    def matrix(self):
        return np.diag(self.A.diag) + self.B @ self.D @ self.B.T

    def __mul__(self, other):
        if isinstance(other, SymWoodbury):
            return Dense(self.matrix() @ other.matrix())
        return self.mul(other)

    __matmul__ = __mul__
    T = property(adjoint)


class VecCongurance:
    """x -> vecm(R' mat(x) R)  (src/ConicIP.jl:35-40, :69-83): field `R`."""

    def __init__(self, R):
        self.R = np.asarray(R, dtype=np.float64)

    @property
    def size(self):
        r = self.R.shape[0]
        return r * (r + 1) // 2

    @staticmethod
    def _mat(x):
        k = x.size
        r = int(round((np.sqrt(1 + 8 * k) - 1) / 2))
        Z = np.zeros((r, r))
        iu = np.triu_indices(r)
        Z[iu] = x / np.sqrt(2.0)
        Z = Z + Z.T
        Z[np.diag_indices(r)] = x[np.cumsum(np.concatenate([[0], np.arange(r, 1, -1)]))]
        return Z

    @staticmethod
    def _vecm(Z):
        r = Z.shape[0]
        iu = np.triu_indices(r)
        out = Z[iu] * np.sqrt(2.0)
        out[np.cumsum(np.concatenate([[0], np.arange(r, 1, -1)]))] = np.diag(Z)
        return out

    def _apply(self, R, x):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 2:
            return np.stack([self._apply(R, x[:, j]) for j in range(x.shape[1])], axis=1)
        return self._vecm(R.T @ self._mat(x) @ R)

    def mul(self, x):
        return self._apply(self.R, x)

    def tmul(self, x):
        return self._apply(self.R.T, x)

    def inv(self):
        return VecCongurance(np.linalg.inv(self.R))

    def adjoint(self):
        return VecCongurance(self.R.T)

    def matrix(self):
        return self.mul(np.eye(self.size))

    def __mul__(self, other):
        if isinstance(other, VecCongurance):
            return VecCongurance(other.R @ self.R)       # W1*W2 = VecCongurance(W2.R*W1.R)  (:75)
        return self.mul(other)

    __matmul__ = __mul__
    T = property(adjoint)


class Dense:
    """A plain matrix element (result of products that leave the structured classes)."""

    def __init__(self, M):
        self.M = np.asarray(M, dtype=np.float64)

    @property
    def size(self):
        return self.M.shape[0]

    def mul(self, x):
        return self.M @ x

    def tmul(self, x):
        return self.M.T @ x

    def inv(self):
        return Dense(np.linalg.inv(self.M))

    def adjoint(self):
        return Dense(self.M.T)

    def matrix(self):
        return self.M

    def __mul__(self, other):
        return Dense(self.M @ other.matrix()) if hasattr(other, "matrix") else self.mul(other)

    __matmul__ = __mul__
    T = property(adjoint)


class Block:
    """Block-diagonal operator (src/blockmatrices.jl:35-131): `F[i]`, `F*x`, `F'`, `inv(F)`, `F*G`."""

    def __init__(self, blocks):
        self.Blocks = list(blocks)

    def __getitem__(self, i):
        return self.Blocks[i]

    def __len__(self):
        return len(self.Blocks)

    def __iter__(self):
        return iter(self.Blocks)

    @property
    def size(self):
        return sum(b.size for b in self.Blocks)

    def _each(self, fn, x):
        x = np.asarray(x, dtype=np.float64)
        out = np.empty_like(x)
        o = 0
        for b in self.Blocks:
            out[o:o + b.size] = fn(b, x[o:o + b.size])
            o += b.size
        return out

    def mul(self, x):
        return self._each(lambda b, v: b.mul(v), x)

    def tmul(self, x):
        return self._each(lambda b, v: b.tmul(v), x)

    def inv(self):
        return Block([b.inv() for b in self.Blocks])

    def adjoint(self):
        return Block([b.adjoint() for b in self.Blocks])

    def inv_adjoint(self):
        return Block([b.inv().adjoint() for b in self.Blocks])

    def matrix(self):
        m = self.size
        M = np.zeros((m, m))
        o = 0
        for b in self.Blocks:
            M[o:o + b.size, o:o + b.size] = b.matrix()
            o += b.size
        return M

    def __mul__(self, other):
        if isinstance(other, Block):
            return Block([a * b for a, b in zip(self.Blocks, other.Blocks)])
        return self.mul(other)

    __matmul__ = __mul__
    T = property(adjoint)


def blocks_from_packed(cone_dims, packed):
    """(F, F^-T) as Blocks with the reference's element types, from the packed scaling of the device
    (layout: include/cipkkt.h, level 2)."""
    packed = np.asarray(packed, dtype=np.float64)
    F, FiT = [], []
    o = 0
    for t, k in cone_dims:
        if t == "R":
            d = packed[o:o + k].copy()
            o += k
            F.append(Diagonal(d))
            FiT.append(Diagonal(1.0 / d))
        elif t == "Q":
            beta, w = packed[o], packed[o + 1:o + 1 + k].copy()
            o += 1 + k
            J = np.full(k, beta)
            J[0] = -beta
            blk = SymWoodbury(Diagonal(J), w, 1.0)        # src/ConicIP.jl:189-192
            F.append(blk)
            FiT.append(blk.inv())                         # symmetric: inv == inv-adjoint
        else:
            r = int(round((np.sqrt(1 + 8 * k) - 1) / 2))
            R = packed[o:o + r * r].reshape(r, r, order="F").copy()
            Ri = packed[o + r * r:o + 2 * r * r].reshape(r, r, order="F").copy()
            o += 2 * r * r
            F.append(VecCongurance(R))
            FiT.append(VecCongurance(Ri.T))               # F^-T[i] = VecCongurance(inv(R)')
    return Block(F), Block(FiT)
