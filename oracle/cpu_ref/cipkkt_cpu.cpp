// ORACLE-side test infrastructure, NOT product code: a CPU implementation of the plugin levels of include/cipkkt.h.
//
// Purpose ("identical ABI implemented by cpu_ref"): the contract of the drop-in boundary -- argument
// meaning, packed-scaling layout, error codes, ownership -- can be exercised in the GPU-less build container: the same
// ctypes table and the same plain-C program (tests/c_abi/solve_qp.c) bind to this library instead of libcipkkt.so.
// Only tests/ may load it (tests/test_cpu_ref.py builds it into oracle/cpu_ref/_build/, which is git-ignored).
//
// Algebra: the literal 3x3 system the reference documents for a kktsolver (src/ConicIP.jl:443-447) and assembles in
// kktsolver_sparse (src/kktsolvers.jl:254-256),
//     [ Q  G' -A' ] [a]   [x]
//     [ G  0   0  ] [b] = [y]            F'F dense from the packed scaling (R: d^2; Q: (diag(-beta,beta,..) + w w')^2;
//     [ A  0  F'F ] [c]   [z]            S: columns of x -> vecm(R (R' mat(x) R) R'), src/ConicIP.jl:35-40)
// factored by Gaussian elimination with partial pivoting (the reference: UMFPACK lu, :257).  Level 1-3, the 2x2 form and
// the identity scaling are implemented; every device-pointer / loop / batch entry point returns CIP_E_UNSUPPORTED.
#include "../../include/cipkkt.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

static thread_local char g_err[512] = "";
static void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct cip_handle {
    int n, m, p, route;
    std::vector<int> ctype, cdim;
    std::vector<double> Q, A, G;          // column-major n x n, m x n, p x n
    std::vector<double> scal;             // packed scaling
    std::vector<double> LU;               // N x N, N = n + p + m
    std::vector<int> piv;
    bool factored = false;
    int singular_col = 0;
};

static size_t scal_len(const cip_handle *h) {
    size_t s = 0;
    for (size_t c = 0; c < h->ctype.size(); ++c) {
        const int k = h->cdim[c];
        if (h->ctype[c] == CIP_CONE_R) s += k;
        else if (h->ctype[c] == CIP_CONE_Q) s += 1 + k;
        else { const int r = (int)llround((sqrt(1.0 + 8.0 * k) - 1.0) / 2.0); s += 2 * (size_t)r * r; }
    }
    return s;
}
static int svec_index(int i, int j, int r) { return i * r - i * (i - 1) / 2 + (j - i); }   // i <= j

// out (k x k, column-major) = F'F of one cone
static void cone_ftf(int type, int k, const double *sc, std::vector<double> &out) {
    out.assign((size_t)k * k, 0.0);
    if (type == CIP_CONE_R) {
        for (int i = 0; i < k; ++i) out[i + (size_t)i * k] = sc[i] * sc[i];
    } else if (type == CIP_CONE_Q) {
        std::vector<double> F((size_t)k * k, 0.0);
        const double beta = sc[0];
        const double *w = sc + 1;
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < k; ++i) F[i + (size_t)j * k] = w[i] * w[j] + (i == j ? (i == 0 ? -beta : beta) : 0.0);
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < k; ++i) {
                double s = 0;
                for (int l = 0; l < k; ++l) s += F[l + (size_t)i * k] * F[l + (size_t)j * k];     // F symmetric
                out[i + (size_t)j * k] = s;
            }
    } else {
        const int r = (int)llround((sqrt(1.0 + 8.0 * k) - 1.0) / 2.0);
        const double *R = sc;
        const double s2 = sqrt(2.0);
        std::vector<double> X((size_t)r * r), T((size_t)r * r), Y((size_t)r * r);
        auto congr = [&](bool transposed) {      // X <- P' X P with P = R (transposed = false) or R' (true)
            auto P = [&](int i, int j) { return transposed ? R[j + (size_t)i * r] : R[i + (size_t)j * r]; };
            for (int j = 0; j < r; ++j) for (int i = 0; i < r; ++i) { double s = 0; for (int l = 0; l < r; ++l) s += X[i + (size_t)l * r] * P(l, j); T[i + (size_t)j * r] = s; }
            for (int j = 0; j < r; ++j) for (int i = 0; i < r; ++i) { double s = 0; for (int l = 0; l < r; ++l) s += P(l, i) * T[l + (size_t)j * r]; Y[i + (size_t)j * r] = s; }
            X = Y;
        };
        for (int c = 0; c < k; ++c) {
            // unit vector e_c -> mat -> F -> F' -> vecm
            std::fill(X.begin(), X.end(), 0.0);
            for (int i = 0; i < r; ++i) for (int j = i; j < r; ++j) if (svec_index(i, j, r) == c) { const double v = (i == j) ? 1.0 : 1.0 / s2; X[i + (size_t)j * r] = v; X[j + (size_t)i * r] = v; }
            congr(false);
            congr(true);
            for (int i = 0; i < r; ++i) for (int j = i; j < r; ++j) out[svec_index(i, j, r) + (size_t)c * k] = (i == j) ? X[i + (size_t)j * r] : s2 * X[i + (size_t)j * r];
        }
    }
}

extern "C" const char *cip_last_error(void) { return g_err; }

extern "C" int cip_create_ex(const cip_problem *pr, cip_handle **out) {
    if (!pr || !out) { set_error("NULL argument"); return CIP_E_INVALID; }
    *out = nullptr;
    if (pr->flags & CIP_FLAG_DEVICE_PTRS) { set_error("cpu reference: device pointers"); return CIP_E_UNSUPPORTED; }
    const int n = pr->n, m = pr->m, p = pr->p;
    if (n <= 0 || m < 0 || p < 0 || pr->ncones < 0) { set_error("bad dimensions n=%d m=%d p=%d", n, m, p); return CIP_E_INVALID; }
    if (!pr->Q) { set_error("Q is NULL"); return CIP_E_INVALID; }
    if (m > 0 && !pr->A && !(pr->A_rowptr && pr->A_colind && pr->A_val)) { set_error("A is NULL"); return CIP_E_INVALID; }
    if (p > 0 && !pr->G) { set_error("G is NULL"); return CIP_E_INVALID; }
    if (pr->route != CIP_ROUTE_SCHUR && pr->route != CIP_ROUTE_FULL3X3) { set_error("bad route"); return CIP_E_INVALID; }
    cip_handle *h = new cip_handle();
    h->n = n; h->m = m; h->p = p; h->route = pr->route;
    int off = 0;
    for (int c = 0; c < pr->ncones; ++c) {
        const int t = pr->cone_type[c], k = pr->cone_dim[c];
        if (k <= 0) { set_error("cone %d has dimension %d", c, k); delete h; return CIP_E_INVALID; }
        if (t == CIP_CONE_S) {
            const int r = (int)llround((sqrt(1.0 + 8.0 * k) - 1.0) / 2.0);
            if (r * (r + 1) / 2 != k) { set_error("S cone %d: %d is not a triangular number", c, k); delete h; return CIP_E_INVALID; }
        } else if (t != CIP_CONE_R && t != CIP_CONE_Q) { set_error("cone %d: unknown type %d", c, t); delete h; return CIP_E_INVALID; }
        h->ctype.push_back(t); h->cdim.push_back(k);
        off += k;
    }
    if (off != m) { set_error("cone_dims cover %d rows but A has %d", off, m); delete h; return CIP_E_INVALID; }
    const int ldq = pr->ldq > 0 ? pr->ldq : n, lda = pr->lda > 0 ? pr->lda : (m > 0 ? m : 1), ldg = pr->ldg > 0 ? pr->ldg : (p > 0 ? p : 1);
    h->Q.resize((size_t)n * n); h->A.assign((size_t)m * n, 0.0); h->G.resize((size_t)p * n);
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) h->Q[i + (size_t)j * n] = pr->Q[i + (size_t)j * ldq];
    if (m > 0) {
        if (pr->A) { for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) h->A[i + (size_t)j * m] = pr->A[i + (size_t)j * lda]; }
        else {
            if (pr->A_rowptr[0] != 0) { set_error("bad CSR row pointer"); delete h; return CIP_E_INVALID; }
            for (int r = 0; r < m; ++r) for (int q = pr->A_rowptr[r]; q < pr->A_rowptr[r + 1]; ++q) {
                if (pr->A_colind[q] < 0 || pr->A_colind[q] >= n) { set_error("CSR column index out of range"); delete h; return CIP_E_INVALID; }
                h->A[r + (size_t)pr->A_colind[q] * m] += pr->A_val[q];
            }
        }
    }
    for (int j = 0; j < n; ++j) for (int i = 0; i < p; ++i) h->G[i + (size_t)j * p] = pr->G[i + (size_t)j * ldg];
    h->scal.assign(scal_len(h), 0.0);
    *out = h;
    return cip_set_scaling_identity(h);
}
extern "C" int cip_create(int n, int m, int p, int ncones, const int *cone_type, const int *cone_dim, const double *Q, const double *A,
                          const double *G, int route, cip_handle **out) {
    cip_problem pr;
    memset(&pr, 0, sizeof(pr));
    pr.n = n; pr.m = m; pr.p = p; pr.ncones = ncones; pr.cone_type = cone_type; pr.cone_dim = cone_dim;
    pr.Q = Q; pr.ldq = n; pr.A = A; pr.lda = m; pr.G = G; pr.ldg = p; pr.route = route;
    if (m > 0 && !A) { set_error("A is NULL"); return CIP_E_INVALID; }
    return cip_create_ex(&pr, out);
}
extern "C" int cip_destroy(cip_handle *h) { delete h; return 0; }
extern "C" size_t cip_scaling_packed_len(const cip_handle *h) { return h ? scal_len(h) : 0; }
extern "C" int cip_set_scaling_packed(cip_handle *h, const double *packedF) {
    if (!h || !packedF) { set_error("NULL argument"); return CIP_E_INVALID; }
    memcpy(h->scal.data(), packedF, sizeof(double) * h->scal.size());
    h->factored = false;
    return 0;
}
extern "C" int cip_get_scaling_packed(cip_handle *h, double *packedF) {
    if (!h || !packedF) { set_error("NULL argument"); return CIP_E_INVALID; }
    memcpy(packedF, h->scal.data(), sizeof(double) * h->scal.size());
    return 0;
}
extern "C" int cip_set_scaling_identity(cip_handle *h) {
    if (!h) return CIP_E_INVALID;
    size_t o = 0;
    for (size_t c = 0; c < h->ctype.size(); ++c) {
        const int k = h->cdim[c];
        if (h->ctype[c] == CIP_CONE_R) { for (int i = 0; i < k; ++i) h->scal[o + i] = 1.0; o += k; }
        else if (h->ctype[c] == CIP_CONE_Q) { h->scal[o] = 1.0; h->scal[o + 1] = sqrt(2.0); for (int i = 1; i < k; ++i) h->scal[o + 1 + i] = 0.0; o += 1 + k; }
        else {
            const int r = (int)llround((sqrt(1.0 + 8.0 * k) - 1.0) / 2.0);
            for (int e = 0; e < 2 * r * r; ++e) h->scal[o + e] = 0.0;
            for (int i = 0; i < r; ++i) { h->scal[o + i + (size_t)i * r] = 1.0; h->scal[o + (size_t)r * r + i + (size_t)i * r] = 1.0; }
            o += 2 * (size_t)r * r;
        }
    }
    h->factored = false;
    return 0;
}
extern "C" int cip_kkt_order(const cip_handle *h, int *N, int *Np) {
    if (!h) return CIP_E_INVALID;
    const int order = h->n + h->p + h->m;
    if (N) *N = order;
    if (Np) *Np = order;
    return 0;
}
extern "C" int cip_factor(cip_handle *h) {
    if (!h) return CIP_E_INVALID;
    const int n = h->n, m = h->m, p = h->p, N = n + p + m;
    std::vector<double> &K = h->LU;
    K.assign((size_t)N * N, 0.0);
    auto at = [&](int i, int j) -> double & { return K[i + (size_t)j * N]; };
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) at(i, j) = h->Q[i + (size_t)j * n];
    for (int j = 0; j < n; ++j) for (int i = 0; i < p; ++i) { at(n + i, j) = h->G[i + (size_t)j * p]; at(j, n + i) = h->G[i + (size_t)j * p]; }
    for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) { at(n + p + i, j) = h->A[i + (size_t)j * m]; at(j, n + p + i) = -h->A[i + (size_t)j * m]; }
    size_t so = 0;
    int off = 0;
    std::vector<double> B;
    for (size_t c = 0; c < h->ctype.size(); ++c) {
        const int k = h->cdim[c];
        cone_ftf(h->ctype[c], k, h->scal.data() + so, B);
        for (int j = 0; j < k; ++j) for (int i = 0; i < k; ++i) at(n + p + off + i, n + p + off + j) = B[i + (size_t)j * k];
        so += (h->ctype[c] == CIP_CONE_R) ? k : (h->ctype[c] == CIP_CONE_Q ? 1 + k : 2 * (size_t)llround((sqrt(1.0 + 8.0 * k) - 1.0) / 2.0) * llround((sqrt(1.0 + 8.0 * k) - 1.0) / 2.0));
        off += k;
    }
    h->piv.assign(N, 0);
    h->singular_col = 0;
    for (int c = 0; c < N; ++c) {
        int pr = c;
        double best = fabs(at(c, c));
        for (int i = c + 1; i < N; ++i) if (fabs(at(i, c)) > best) { best = fabs(at(i, c)); pr = i; }
        h->piv[c] = pr;
        if (!(best > 0.0) || !std::isfinite(best)) { if (!h->singular_col) h->singular_col = c + 1; continue; }
        if (pr != c) for (int j = 0; j < N; ++j) std::swap(at(c, j), at(pr, j));
        const double d = 1.0 / at(c, c);
        for (int i = c + 1; i < N; ++i) at(i, c) *= d;
        for (int j = c + 1; j < N; ++j) { const double u = at(c, j); if (u != 0.0) for (int i = c + 1; i < N; ++i) at(i, j) -= at(i, c) * u; }
    }
    h->factored = true;
    return 0;
}
extern "C" int cip_check_factor(cip_handle *h) {
    if (!h) return CIP_E_INVALID;
    if (!h->factored) { set_error("no factorisation"); return CIP_E_NOTFACTORED; }
    if (h->singular_col) { set_error("LU: zero or non-finite pivot at column %d", h->singular_col); return CIP_E_SINGULAR; }
    return 0;
}
static int lu_solve(cip_handle *h, std::vector<double> &r) {
    const int N = h->n + h->p + h->m;
    const std::vector<double> &K = h->LU;
    for (int c = 0; c < N; ++c) if (h->piv[c] != c) std::swap(r[c], r[h->piv[c]]);      // P b first: whole rows were swapped
    for (int c = 0; c < N; ++c) for (int i = c + 1; i < N; ++i) r[i] -= K[i + (size_t)c * N] * r[c];
    for (int c = N - 1; c >= 0; --c) { r[c] /= K[c + (size_t)c * N]; for (int i = 0; i < c; ++i) r[i] -= K[i + (size_t)c * N] * r[c]; }
    return 0;
}
extern "C" int cip_solve3x3(cip_handle *h, const double *x, const double *y, const double *z, double *a, double *b, double *c) {
    if (!h) return CIP_E_INVALID;
    if (!h->factored) { set_error("cip_solve3x3: no factorisation (call cip_factor first)"); return CIP_E_NOTFACTORED; }
    if (h->singular_col) { set_error("LU: zero or non-finite pivot at column %d", h->singular_col); return CIP_E_SINGULAR; }
    const int n = h->n, m = h->m, p = h->p;
    std::vector<double> r((size_t)n + p + m);
    for (int i = 0; i < n; ++i) r[i] = x[i];
    for (int i = 0; i < p; ++i) r[n + i] = y[i];
    for (int i = 0; i < m; ++i) r[n + p + i] = z ? z[i] : 0.0;
    lu_solve(h, r);
    for (int i = 0; i < n; ++i) a[i] = r[i];
    for (int i = 0; i < p; ++i) b[i] = r[n + i];
    if (c) for (int i = 0; i < m; ++i) c[i] = r[n + p + i];
    return 0;
}
// [Q + A'(F'F)^-1 A, G'; G, 0][dy; dw] = [y; w]  ==  the first two components of the 3x3 solve with z = 0
extern "C" int cip_solve2x2(cip_handle *h, const double *y, const double *w, double *dy, double *dw) {
    if (!h) return CIP_E_INVALID;
    if (h->route != CIP_ROUTE_SCHUR) { set_error("cip_solve2x2: needs the Schur route"); return CIP_E_UNSUPPORTED; }
    return cip_solve3x3(h, y, w, nullptr, dy, dw, nullptr);
}

// ---- everything that needs the device: not part of the CPU reference
#define UNSUP(name, ...) extern "C" int name(__VA_ARGS__) { set_error(#name ": not implemented by the CPU reference"); return CIP_E_UNSUPPORTED; }
UNSUP(cip_update_problem, cip_handle *, const cip_problem *)
UNSUP(cip_set_stream, cip_handle *, void *)
UNSUP(cip_set_scaling_from_iterate_dev, cip_handle *, const double *, const double *, double *)
UNSUP(cip_set_regularization, cip_handle *, double, int)
UNSUP(cip_get_regularization, cip_handle *, double *, int *)
UNSUP(cip_solve3x3_dev, cip_handle *, const double *, const double *, const double *, double *, double *, double *)
UNSUP(cip_solve2x2_dev, cip_handle *, const double *, const double *, double *, double *)
UNSUP(cip_solve4x4_dev, cip_handle *, const double *, const double *, double *)
UNSUP(cip_apply_F_dev, cip_handle *, int, const double *, double *)
UNSUP(cip_cone_prod_dev, cip_handle *, const double *, const double *, double *)
UNSUP(cip_cone_div_dev, cip_handle *, const double *, const double *, double *)
UNSUP(cip_maxstep_dev, cip_handle *, const double *, const double *, double, double *)
UNSUP(cip_cone_identity_dev, cip_handle *, double *)
UNSUP(cip_gemv_dev, cip_handle *, int, int, double, const double *, double, double *)
UNSUP(cip_dots_dev, cip_handle *, int, const double *const *, const double *const *, const int *, double *)
UNSUP(cip_axpby_dev, cip_handle *, int, double, const double *, double, double *)
UNSUP(cip_conicip, cip_handle *, const double *, const double *, const double *, const cip_options *, double *, double *, double *, cip_result *, double *, int)
