"""ORACLE (test infrastructure, NOT product code).

numpy restatement of the reference's block-diagonal operator algebra
(src/blockmatrices.jl) and of the block element types it holds:
``Diagonal`` (LinearAlgebra), ``SymWoodbury`` (WoodburyMatrices.jl 0.5.6,
third-party, pinned in docs/Manifest.toml:507-511 -- restated here from its
published definition A + B*D*B' and the Woodbury identity) and
``VecCongurance`` (src/ConicIP.jl:35-40, :69-83).
"""
import numpy as np

from .cones import mat, vecm


class Diagonal:
    """LinearAlgebra.Diagonal as used at src/ConicIP.jl:598, :704."""

    def __init__(self, diag):
        self.diag = np.asarray(diag, dtype=np.float64)

    def size(self):
        return len(self.diag)

    def mul(self, x):
        return self.diag * x if x.ndim == 1 else self.diag[:, None] * x

    def tmul(self, x):
        return self.mul(x)

    def inv(self):
        return Diagonal(1.0 / self.diag)

    def adjoint(self):
        return self

    def square(self):
        """F'F for a diagonal block."""
        return Diagonal(self.diag * self.diag)

    def matrix(self):
        return np.diag(self.diag)


class SymWoodbury:
    """A + B D B' with A diagonal (vector ``A``), B (k x r), D (r x r).
    Call sites: src/ConicIP.jl:192 (construction, r=1, D=1.0);
    src/blockmatrices.jl:135-141 (densify), :183-200 (inv / adjoint / *)."""

    def __init__(self, A, B, D):
        self.A = np.asarray(A, dtype=np.float64)
        B = np.asarray(B, dtype=np.float64)
        self.B = B.reshape(len(self.A), -1)
        D = np.asarray(D, dtype=np.float64)
        self.D = D.reshape(self.B.shape[1], self.B.shape[1])

    def size(self):
        return len(self.A)

    def mul(self, x):
        if x.ndim == 1:
            return self.A * x + self.B @ (self.D @ (self.B.T @ x))
        return self.A[:, None] * x + self.B @ (self.D @ (self.B.T @ x))

    def tmul(self, x):
        return self.mul(x)            # symmetric

    def inv(self):
        """Woodbury identity: (A + B D B')^-1 = A^-1 - A^-1 B (D^-1 + B'A^-1 B)^-1 B' A^-1."""
        Ai = 1.0 / self.A
        AiB = Ai[:, None] * self.B
        C = np.linalg.inv(self.D) + self.B.T @ AiB
        return SymWoodbury(Ai, AiB, -np.linalg.inv(C))

    def adjoint(self):
        return self

    def square(self):
        """(A + B D B')^2 = A^2 + Z D' Z', Z = [AB, B], D' = [[0, D],[D, D B'B D]]
        (what F'F evaluates to for a symmetric block; src/kktsolvers.jl:195,209,252)."""
        AB = self.A[:, None] * self.B
        Z = np.hstack([AB, self.B])
        r = self.B.shape[1]
        Dp = np.zeros((2 * r, 2 * r))
        Dp[:r, r:] = self.D
        Dp[r:, :r] = self.D
        Dp[r:, r:] = self.D @ (self.B.T @ self.B) @ self.D
        return SymWoodbury(self.A * self.A, Z, Dp)

    def matrix(self):
        return np.diag(self.A) + self.B @ self.D @ self.B.T


class VecCongurance:
    """src/ConicIP.jl:35-40, :69-83 -- x -> vecm(R' mat(x) R)."""

    def __init__(self, R):
        self.R = np.asarray(R, dtype=np.float64)

    def size(self):
        r = self.R.shape[0]
        return int(round(r * (r + 1) / 2))

    def _apply(self, R, x):
        if x.ndim == 1:
            return vecm(R.T @ mat(x) @ R)
        return np.stack([vecm(R.T @ mat(x[:, j]) @ R) for j in range(x.shape[1])], axis=1)

    def mul(self, x):
        return self._apply(self.R, x)

    def tmul(self, x):
        return self._apply(self.R.T, x)

    def inv(self):
        return VecCongurance(np.linalg.inv(self.R))

    def adjoint(self):
        return VecCongurance(self.R.T)

    def compose(self, other):
        """W1*W2 = VecCongurance(W2.R*W1.R) (src/ConicIP.jl:40)."""
        return VecCongurance(other.R @ self.R)

    def square(self):
        """F'F = adjoint(F)*F -> VecCongurance(F.R * F.R')."""
        return self.adjoint().compose(self)

    def matrix(self):
        """src/ConicIP.jl:71-79 -- apply to identity columns."""
        n = self.size()
        return self.mul(np.eye(n))


class Dense:
    """Plain Matrix block (src/blockmatrices.jl:54)."""

    def __init__(self, M):
        self.M = np.asarray(M, dtype=np.float64)

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

    def square(self):
        return Dense(self.M.T @ self.M)

    def matrix(self):
        return self.M


class Block:
    """src/blockmatrices.jl:35-218."""

    def __init__(self, blocks):
        self.Blocks = list(blocks)

    def size(self):
        return sum(b.size() for b in self.Blocks)

    def block_idx(self):
        """src/blockmatrices.jl:62-76 (0-based slices here)."""
        out, c = [], 0
        for b in self.Blocks:
            out.append(slice(c, c + b.size()))
            c += b.size()
        return out

    def _bcast(self, fn, x):
        """broadcastf(op, A, x) src/blockmatrices.jl:107-131."""
        y = np.empty_like(x, dtype=np.float64)
        for I, b in zip(self.block_idx(), self.Blocks):
            y[I] = fn(b, x[I])
        return y

    def mul(self, x):          # F*x   (:173,:176)
        return self._bcast(lambda b, xi: b.mul(xi), np.asarray(x, dtype=np.float64))

    def tmul(self, x):         # F'*x  (:174,:177)
        return self._bcast(lambda b, xi: b.tmul(xi), np.asarray(x, dtype=np.float64))

    def inv(self):             # :183
        return Block([b.inv() for b in self.Blocks])

    def adjoint(self):         # :185
        return Block([b.adjoint() for b in self.Blocks])

    def inv_adjoint(self):     # :193-198
        return Block([b.inv().adjoint() for b in self.Blocks])

    def square(self):          # F'F  (:200)
        return Block([b.square() for b in self.Blocks])

    def matrix(self):          # :162-170
        n = self.size()
        O = np.zeros((n, n))
        for I, b in zip(self.block_idx(), self.Blocks):
            O[I, I] = b.matrix()
        return O


def identity_block(block_sizes):
    """Block([Diagonal(ones(i)) for i = block_sizes]) src/ConicIP.jl:704."""
    return Block([Diagonal(np.ones(k)) for k in block_sizes])
