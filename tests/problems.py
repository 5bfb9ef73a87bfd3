"""Deterministic problem instances taken from the reference's own test-suite and
docs (inputs are formulaic, so no data file is needed).  Each builder returns
(Q, c, A, b, cone_dims, G, d, expect) with ``expect`` = the analytic answer the
reference asserts.  Citations: /root/reference/test/runtests.jl and docs."""
import numpy as np
import scipy.sparse as sp


def sphere(n=2):
    """test/runtests.jl:137-166 -- projection of ones(n) onto the unit sphere."""
    H = sp.identity(n, format="csr")
    a = np.ones(n)
    A = sp.vstack([sp.csr_matrix((1, n)), sp.identity(n)]).tocsr()
    b = np.concatenate([[-1.0], np.zeros(n)])
    return H, H @ a, A, b, [("Q", n + 1)], None, None, a / np.linalg.norm(a)


def combined(n=10):
    """test/runtests.jl:168-206 -- R + Q cones."""
    H = sp.identity(n, format="csr")
    c = np.arange(1.0, n + 1)
    A = sp.vstack([sp.identity(n), sp.csr_matrix((1, n)), sp.identity(n)]).tocsr()
    b = np.concatenate([np.zeros(n), [-1.0], np.zeros(n)])
    y = np.maximum(0, c)
    return H, H @ c, A, b, [("R", n), ("Q", n + 1)], None, None, y / np.linalg.norm(y)


def simplex(n=10):
    """test/runtests.jl:208-244 -- projection onto the simplex (R + equality)."""
    H = np.eye(n)
    c = np.arange(1.0, n + 1)
    A = sp.identity(n, format="csr")
    b = np.zeros(n)
    G = np.ones((1, n))
    d = np.array([1.0])
    y = np.zeros(n)
    y[-1] = 1
    return H, H @ c, A, b, [("R", n)], G, d, y


def psd_projection():
    """test/runtests.jl:527-552 -- projection of diag(1,1,1,-1,-1,-1) onto PSD."""
    from oracle.cones import vecm
    n = 21
    H = np.eye(n)
    c = vecm(np.diag([1.0, 1, 1, -1, -1, -1]))
    A = sp.identity(n, format="csr")
    b = np.zeros(n)
    return H, c, A, b, [("S", n)], None, None, vecm(np.diag([1.0, 1, 1, 0, 0, 0]))


def soc_direct():
    """test/runtests.jl:554-590 -- min 1/2||x||^2 + 1'x, ||x[1:3]||<=1, x>=0 -> 0."""
    n = 4
    Q = sp.identity(n, format="csr")
    c = -np.ones(n)
    A_soc = sp.vstack([sp.csr_matrix((1, n)), sp.identity(n, format="csr")[:3, :]])
    A = sp.vstack([A_soc, sp.identity(n)]).tocsr()
    b = np.concatenate([[-1.0], np.zeros(3), np.zeros(n)])
    return Q, c, A, b, [("Q", 4), ("R", n)], None, None, np.zeros(n)


def box_qp(n=1000):
    """test/runtests.jl:90-131 -- box-constrained QP, H = I/2, c = 1:n."""
    H = 0.5 * sp.identity(n, format="csr")
    c = np.arange(1.0, n + 1)
    A = sp.vstack([sp.identity(n), -sp.identity(n)]).tocsr()
    b = -np.ones(2 * n)
    return H, H @ c, A, b, [("R", 2 * n)], None, None, np.ones(n)


def unbounded(n=10):
    """test/runtests.jl:487-505."""
    H = np.zeros((n, n))
    c = np.arange(1.0, n + 1)
    A = sp.identity(n, format="csr")
    b = np.zeros(n)
    return H, c, A, b, [("R", n)], None, None, None


def infeasible_box(n=10, seed=0):
    """test/runtests.jl:441-460 (status is data-independent: y>=1 and y<=-1)."""
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n)
    H = np.outer(h, h)
    c = np.arange(1.0, n + 1)
    A = sp.vstack([sp.identity(n), -sp.identity(n)]).tocsr()
    b = np.ones(2 * n)
    return H, H @ c, A, b, [("R", 2 * n)], None, None, None


def infeasible_eq(n=10, seed=0):
    """test/runtests.jl:462-485 (y>=0 with y1 = -1)."""
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n)
    H = np.outer(h, h)
    c = np.arange(1.0, n + 1)
    A = sp.identity(n, format="csr")
    b = np.zeros(n)
    G = np.zeros((1, n))
    G[0, 0] = 1
    d = np.array([-1.0])
    return H, H @ c, A, b, [("R", n)], G, d, None


def lp_doc():
    """docs/src/tutorials/lp.jl:21-41 -- y2 = 4."""
    n = 5
    Q = sp.csr_matrix((n, n))
    c = np.array([2.0, 3.0, 1.0, 1.0, 1.0])
    A = sp.identity(n, format="csr")
    b = np.zeros(n)
    G = np.ones((1, n))
    d = np.array([4.0])
    return Q, c, A, b, [("R", n)], G, d, np.array([0, 4.0, 0, 0, 0])


def random_mixed(n=40, nq=3, kq=6, p=4, seed=1, dense_A=True):
    """Synthetic strictly feasible R+Q+equality problem in the pattern of
    benchmark/profile.jl:95-114 (feasible at y = ones)."""
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    Q = M.T @ M / n + 0.1 * np.eye(n)
    c = rng.standard_normal(n)
    A_r = np.eye(n)
    blocks = [A_r]
    bs = [np.zeros(n)]
    cone_dims = [("R", n)]
    for _ in range(nq):
        Aq = rng.standard_normal((kq, n)) * 0.2
        Aq[0, :] = 0
        blocks.append(Aq)
        bq = np.zeros(kq)
        bq[0] = -(np.linalg.norm(Aq[1:] @ np.ones(n)) + 1.0)
        bs.append(bq)
        cone_dims.append(("Q", kq))
    A = np.vstack(blocks)
    b = np.concatenate(bs)
    G = rng.standard_normal((p, n))
    d = G @ np.ones(n)
    if not dense_A:
        A = sp.csr_matrix(A)
    return Q, c, A, b, cone_dims, G, d, None


# ------------------------------------------------------------------------------------------------------------
# "Miles's counterexamples" (test/runtests.jl:592-651).  The numeric data is the reference's own fixture
# (test/testdata.jl:109-150, extracted to tests/golden/miles_problems.json by tests/golden/make_miles_fixture.py);
# the conversion from the MathProgBase conic form to ConicIP's form restates what the reference's test helper
# documents (test/testdata.jl:5-15):   MPB:  min c'x  s.t.  b - Ax in K_con,  x in K_var
#                                      here: min 1/2 y'Qy - c'y  s.t.  Ay - b in K,  Gy = d
def miles_problem(k):
    import json
    import os
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "miles_problems.json")))[
        "miles_problem_%d" % k]
    c, b = np.array(rec["c"]), np.array(rec["b"])
    A = sp.csr_matrix((rec["V"], (np.array(rec["I"]) - 1, np.array(rec["J"]) - 1)), shape=(b.size, c.size))
    cones = lambda lst: [(t, np.array(idx) - 1) for t, idx in lst]            # 1-based -> 0-based
    return c, A, b, cones(rec["con_cones"]), cones(rec["var_cones"])


def mpb_to_conicip(c, A, b, con_cones, var_cones):
    """(Q, c, A, b, cone_dims, G, d) of the ConicIP form.  Zero cones become equality rows; NonPos rows are kept
    (b - Ax <= 0  <=>  Ax - b >= 0); NonNeg / SOC / SDP rows are negated (b - Ax in K  <=>  (-A)x - (-b) in K);
    variable cones add rows nA * I (nA = ||A||_F, the reference's conditioning scale) with zero right-hand side;
    the objective sign flips because ConicIP minimises -c'y."""
    A = sp.csr_matrix(A)
    n = c.size
    nA = np.sqrt(A.multiply(A).sum())
    eq = [idx for t, idx in con_cones if t == "Zero"]
    rows, rhs, dims = [], [], []
    kind = {"NonPos": ("R", 1.0), "NonNeg": ("R", -1.0), "SOC": ("Q", -1.0), "SDP": ("S", -1.0)}
    for t, idx in con_cones:
        if t == "Zero":
            continue
        cone, sign = kind[t]
        rows.append(sign * A[idx, :])
        rhs.append(sign * b[idx])
        dims.append((cone, len(idx)))
    vkind = {"NonNeg": ("R", 1.0), "NonPos": ("R", -1.0), "SOC": ("Q", 1.0), "SDP": ("S", 1.0)}
    for t, idx in var_cones:
        if t == "Free":
            continue
        cone, sign = vkind[t]
        rows.append(sp.csr_matrix((np.full(len(idx), sign * nA), (np.arange(len(idx)), idx)), shape=(len(idx), n)))
        rhs.append(np.zeros(len(idx)))
        dims.append((cone, len(idx)))
    Ai = sp.vstack(rows, format="csr") if rows else sp.csr_matrix((0, n))
    bi = np.concatenate(rhs) if rhs else np.zeros(0)
    if eq:
        e = np.concatenate(eq)
        G, d = A[e, :], b[e]
    else:
        G, d = sp.csr_matrix((0, n)), np.zeros(0)
    return sp.csr_matrix((n, n)), -c, Ai, bi, dims, G, d


def random_degenerate(seed):
    """A small conic program with the structures that stress the static-order LDL' and the pre-solve: rank-deficient Q
    (down to Q = 0), free variables, SOC / small PSD blocks, redundant equality rows, objective scales 1e-2..1e2.
    Returns (Q, c, A, b, cone_dims, G, d); about two thirds are solvable, the rest unbounded or infeasible."""
    from oracle import cones as oc
    rng = np.random.default_rng(seed)
    n = int(rng.integers(4, 30))
    rankq = int(rng.integers(0, n + 1))
    B = rng.standard_normal((rankq, n))
    Q = B.T @ B if rankq else np.zeros((n, n))
    c = rng.standard_normal(n) * (10.0 ** rng.integers(-2, 3))
    cone_dims, rows, rhs = [], [], []
    nb = int(rng.integers(0, n + 1))                              # bounded variables (the rest are free)
    if nb:
        A0 = np.zeros((nb, n))
        A0[np.arange(nb), rng.permutation(n)[:nb]] = 1.0
        rows.append(A0); rhs.append(-rng.random(nb)); cone_dims.append(("R", nb))
    for _ in range(int(rng.integers(0, 3))):
        k = int(rng.integers(2, 6))
        b = rng.standard_normal(k)
        b[0] = -abs(b[0]) - np.linalg.norm(b[1:]) - 1
        rows.append(rng.standard_normal((k, n))); rhs.append(b); cone_dims.append(("Q", k))
    if rng.random() < 0.3:                                        # a small semidefinite block: mat(A y - b) >= 0
        r_ = int(rng.integers(2, 5))
        k = r_ * (r_ + 1) // 2
        rows.append(rng.standard_normal((k, n)) * 0.3); rhs.append(-oc.vecm(np.eye(r_) * (1 + rng.random())))
        cone_dims.append(("S", k))
    if not rows:
        rows.append(np.eye(n)[:1]); rhs.append(np.array([-1.0])); cone_dims.append(("R", 1))
    A, b = np.vstack(rows), np.concatenate(rhs)
    p = int(rng.integers(0, max(1, n // 2)))
    G = rng.standard_normal((p, n))
    d = G @ rng.standard_normal(n)
    if p and rng.random() < 0.3:                                  # a redundant (consistent) equality row
        G, d = np.vstack([G, G[:1]]), np.concatenate([d, d[:1]])
    return Q, c, A, b, cone_dims, G, d
