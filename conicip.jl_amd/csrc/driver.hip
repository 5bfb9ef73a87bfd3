// Native interior-point loop: the Mehrotra predictor-corrector iteration of the reference's `conicIP`
// (src/ConicIP.jl:468-939) driven from C++ through the library's own device entry points, every vector
// resident in HBM.  The host sees only scalars (residual norms, mu, step lengths), exactly as the Python
// driver (cipkkt/driver.py) does -- this is the same loop without the interpreter between the launches
// (measured, warm library: dense QP n = 8192 0.111 -> 0.087 s to converge, n = 2048 20.2 -> 18.4 ms; the loop is
// GPU-bound either way).  The two loops issue the same kernels in the same order and agree to the last bit
// (tests/test_gpu_driver.py).
//
// Quirks of the reference are kept: a factorisation also happens in the terminating
// iteration (:737 precedes :786), rPr ignores the equality residual (:765), norm(v4x1) is the sum of the
// block 2-norms (:61), the returned (y, w, v) is the last iterate.
#include "cip_handle.h"
#include "../../include/cipkkt.h"
#include <chrono>
#include <cmath>
#include <vector>

namespace {

inline double jlmax(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }   // Julia max propagates NaN
inline double jlmax(double a, double b, double c) { return jlmax(jlmax(a, b), c); }
inline double nrm(double x2) { return x2 >= 0 ? std::sqrt(x2) : NAN; }

struct Vec4 {          // (y[n], w[p], v[m], s[m]) stored contiguously
    double *base = nullptr, *y = nullptr, *w = nullptr, *v = nullptr, *s = nullptr;
};

struct Driver {
    cip_handle *h;
    int n, m, p, NT;
    double *next = nullptr;      // bump pointer into h->drv (one allocation per handle, kept for later calls)
    int rc = 0;

    static size_t pad(size_t c) { return (c + 31) & ~(size_t)31; }      // 256-byte alignment of every vector
    size_t total() const {
        return 9 * pad(NT) + pad(n) + pad(m) + pad(p) + 5 * pad(m) + 2 * pad(n) + pad(m) + pad(p) + 32;
    }
    int init() {
        if (!h->drv) {
            void *ptr = nullptr;
            if (hipMalloc(&ptr, sizeof(double) * total()) != hipSuccess) { rc = CIP_E_HIP; return rc; }
            h->drv = (double *)ptr;
        }
        if (hipMemsetAsync(h->drv, 0, sizeof(double) * total(), h->stream) != hipSuccess) { rc = CIP_E_HIP; return rc; }
        next = h->drv;
        return 0;
    }
    double *dalloc(size_t count) {
        double *ptr = next;
        next += pad(count);
        return ptr;
    }
    Vec4 vec4() {
        Vec4 v;
        v.base = dalloc(NT);
        v.y = v.base; v.w = v.y + n; v.v = v.w + p; v.s = v.v + m;
        return v;
    }

    // y <- alpha x + beta y
    int axpby(int len, double alpha, const double *x, double beta, double *y) { return len > 0 ? cip_axpby_dev(h, len, alpha, x, beta, y) : 0; }
    int copy(int len, const double *x, double *y) { return axpby(len, 1.0, x, 0.0, y); }

    // out.y = Q x.y + G' x.w - A' x.v ; out.w = G x.y ; out.v = A x.y - x.s     (:747-749, :912-914)
    int kkt_apply(const Vec4 &x, Vec4 &out) {
        int e = cip_gemv_dev(h, CIP_MAT_Q, 0, 1.0, x.y, 0.0, out.y);
        if (p > 0) {
            e |= cip_gemv_dev(h, CIP_MAT_G, 1, 1.0, x.w, 1.0, out.y);
            e |= cip_gemv_dev(h, CIP_MAT_G, 0, 1.0, x.y, 0.0, out.w);
        }
        if (m > 0) {
            e |= cip_gemv_dev(h, CIP_MAT_A, 1, -1.0, x.v, 1.0, out.y);
            e |= cip_gemv_dev(h, CIP_MAT_A, 0, 1.0, x.y, 0.0, out.v);
            e |= axpby(m, -1.0, x.s, 1.0, out.v);
        }
        return e;
    }
};

}   // namespace

extern "C" int cip_conicip(cip_handle *h, const double *c_host, const double *b_host, const double *d_host,
                           const cip_options *opt_in, double *y_out, double *w_out, double *v_out, cip_result *res,
                           double *trace, int trace_cap) {
    if (!h || !res || !c_host || (h->m > 0 && !b_host) || (h->p > 0 && !d_host) || !y_out || (h->p > 0 && !w_out) ||
        (h->m > 0 && !v_out)) { cip_set_error("cip_conicip: null argument"); return CIP_E_INVALID; }
    const auto t_start = std::chrono::steady_clock::now();
    cip_options o;
    o.optTol = 1e-6; o.DTB = 0.01; o.infeasTol = -1.0; o.refinementThreshold = -1.0;
    o.maxRefinementSteps = 3; o.maxIters = 100; o.verbose = 0;                       // src/ConicIP.jl:498-509
    if (opt_in) o = *opt_in;
    if (o.infeasTol < 0) o.infeasTol = o.optTol;
    if (o.refinementThreshold < 0) o.refinementThreshold = o.optTol / 1e7;
    CIP_HIP_CHECK(hipSetDevice(h->device));

    Driver D{h, h->n, h->m, h->p, h->n + h->p + 2 * h->m};
    const int n = D.n, m = D.m, p = D.p;
    if (D.init()) { cip_set_error("cip_conicip: device allocation failed"); return CIP_E_HIP; }
    double *c_d = D.dalloc(n), *b_d = D.dalloc(m), *d_d = D.dalloc(p);
    Vec4 z = D.vec4(), r0 = D.vec4(), rleft = D.vec4(), r = D.vec4(), daff = D.vec4(), dz = D.vec4(), dzr = D.vec4(),
         rIr = D.vec4(), rkkt = D.vec4();
    double *e = D.dalloc(m), *lam = D.dalloc(m), *mb1 = D.dalloc(m), *mb2 = D.dalloc(m), *mb3 = D.dalloc(m);
    double *Qy = D.dalloc(n), *pinf = D.dalloc(n), *Ays = D.dalloc(m), *Gy = D.dalloc(p);
    CIP_HIP_CHECK(hipMemcpyAsync(c_d, c_host, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    if (m > 0) CIP_HIP_CHECK(hipMemcpyAsync(b_d, b_host, sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
    if (p > 0) CIP_HIP_CHECK(hipMemcpyAsync(d_d, d_host, sizeof(double) * p, hipMemcpyHostToDevice, h->stream));
    double normc = 0, normb = 0, normd = -INFINITY;
    for (int i = 0; i < n; ++i) normc += c_host[i] * c_host[i];
    normc = std::sqrt(normc);
    for (int i = 0; i < m; ++i) normb += b_host[i] * b_host[i];
    normb = std::sqrt(normb);
    if (p > 0) { normd = 0; for (int i = 0; i < p; ++i) normd += d_host[i] * d_host[i]; normd = std::sqrt(normd); }

    // conedim (:547-552) and e (:559-565)
    double conedim = 0;
    for (const ConeDesc &cd : h->h_cones) conedim += cd.type == CIP_CONE_R ? cd.dim : (cd.type == CIP_CONE_Q ? 1 : cd.r);
    int rc;
#define CK(x) do { if ((rc = (x)) != 0) return rc; } while (0)
    if (m > 0) CK(cip_cone_identity_dev(h, e));

    int n_factor = 0, n_solve = 0;
    *res = cip_result{};
    res->prFeas = res->duFeas = res->muFeas = INFINITY; res->pobj = INFINITY; res->dobj = -INFINITY;
    double optBest = INFINITY;

    auto finish = [&](int status) -> int {
        CIP_HIP_CHECK(hipMemcpyAsync(y_out, z.y, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
        if (p > 0) CIP_HIP_CHECK(hipMemcpyAsync(w_out, z.w, sizeof(double) * p, hipMemcpyDeviceToHost, h->stream));
        if (m > 0) CIP_HIP_CHECK(hipMemcpyAsync(v_out, z.v, sizeof(double) * m, hipMemcpyDeviceToHost, h->stream));
        CIP_HIP_CHECK(hipStreamSynchronize(h->stream));
        res->status = status; res->n_factor = n_factor; res->n_solve = n_solve;
        res->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        return 0;
    };

    // ---------------------------------------------------------------- initial point (:704-713)
    CK(cip_set_scaling_identity(h));
    CK(cip_factor(h)); ++n_factor;
    CK(D.copy(n, c_d, r0.y)); CK(D.copy(p, d_d, r0.w)); CK(D.copy(m, b_d, r0.v));
    if (m > 0) CIP_HIP_CHECK(hipMemsetAsync(r0.s, 0, sizeof(double) * m, h->stream));
    // initial point: one wait (LPs meet their first bad pivot here)
    if ((rc = cip_factor_resolve(h, 1)) != 0) { if (rc == CIP_E_SINGULAR) return finish(CIP_STATUS_ERROR); return rc; }
    CK(cip_solve4x4_dev(h, e, r0.base, z.base)); ++n_solve;
    if (m > 0) {
        double a_v, a_s;
        CK(cip_maxstep_dev(h, z.v, nullptr, 1.0, &a_v));
        CK(cip_maxstep_dev(h, z.s, nullptr, 1.0, &a_s));
        CK(D.axpby(m, -a_v, e, 1.0, z.v));
        CK(D.axpby(m, -a_s, e, 1.0, z.s));
    }

    struct IterRange { IterRange() { cip_range_push("cip:iteration"); } ~IterRange() { cip_range_pop(); } };
    for (int Iter = 1; Iter <= o.maxIters; ++Iter) {                                   // :730
        IterRange iter_range;
        if (m > 0) CK(cip_set_scaling_from_iterate_dev(h, z.v, z.s, lam));             // :732-735 (F, lambda = F v)
        CK(cip_factor(h)); ++n_factor;                                                 // :737 -> :682
        if (m > 0) CK(cip_cone_prod_dev(h, lam, lam, rleft.s));                        // :746
        CK(D.kkt_apply(z, rleft));                                                     // :747-750
        // pieces needed by the certificates
        CK(cip_gemv_dev(h, CIP_MAT_Q, 0, 1.0, z.y, 0.0, Qy));
        CIP_HIP_CHECK(hipMemsetAsync(pinf, 0, sizeof(double) * n, h->stream));
        if (p > 0) { CK(cip_gemv_dev(h, CIP_MAT_G, 1, 1.0, z.w, 0.0, pinf)); CK(D.copy(p, rleft.w, Gy)); }
        if (m > 0) { CK(cip_gemv_dev(h, CIP_MAT_A, 1, -1.0, z.v, 1.0, pinf)); CK(D.copy(m, rleft.v, Ays)); }
        // r0 = rleft - (c, d, b, 0)   (:753)
        CK(D.copy(D.NT, rleft.base, r0.base));
        CK(D.axpby(n, -1.0, c_d, 1.0, r0.y));
        CK(D.axpby(p, -1.0, d_d, 1.0, r0.w));
        CK(D.axpby(m, -1.0, b_d, 1.0, r0.v));

        const double *px[16] = {z.v, c_d, r0.y, r0.v, r0.s, z.y, z.w, z.v, d_d, b_d, pinf, z.y, z.v, Ays, Gy, Qy};
        const double *py[16] = {z.s, z.y, r0.y, r0.v, r0.s, Qy, r0.w, r0.v, z.w, z.v, pinf, z.y, z.v, Ays, Gy, Qy};
        const int ln[16] = {m, n, n, m, m, n, p, m, p, m, n, n, m, m, p, n};
        double dt[16];
        CK(cip_dots_dev(h, 16, px, py, ln, dt));
        // the stream has just been drained: the pivot flag of this iteration's factorisation is in.  A dead (zero /
        // non-finite) pivot even after regularisation is where the reference's LU hands back NaNs and the loop ends
        // with :Error at its next residual check (src/ConicIP.jl:870-873)
        if ((rc = cip_factor_resolve(h, 1)) != 0) { if (rc == CIP_E_SINGULAR) return finish(CIP_STATUS_ERROR); return rc; }
        const double mubar = dt[0], cTy = dt[1], r0y2 = dt[2], r0v2 = dt[3], r0s2 = dt[4], yQy = dt[5], wr0w = dt[6],
                     vr0v = dt[7], dTw = dt[8], bTv = dt[9], pinf2 = dt[10], yy = dt[11], vv = dt[12], ays2 = dt[13],
                     gy2 = dt[14], qy2 = dt[15];
        const double mu = conedim > 0 ? mubar / conedim : NAN;                         // :756-757
        const double rDu = nrm(r0y2) / (1 + normc);                                    // :764
        const double rPr = (m > 0 ? nrm(r0v2) : 0.0) / (1 + normb);                    // :765
        const double rCp = (m > 0 ? nrm(r0s2) : 0.0) / (1 + std::fabs(cTy));           // :766
        const double worst = jlmax(rDu, rPr, rCp);
        if (worst < optBest) {                                                         // :768-773
            res->iter = Iter; res->mu = mu; res->duFeas = rDu; res->prFeas = rPr; res->muFeas = rCp;
            optBest = worst;
        }
        const double pobj = 0.5 * yQy - cTy;                                           // :775
        const double dobj = pobj + wr0w + vr0v - mubar;                                // :776
        res->pobj = pobj; res->dobj = dobj;
        double *tr = (trace && Iter <= trace_cap) ? trace + (size_t)(Iter - 1) * CIP_TRACE_COLS : nullptr;
        if (tr) { tr[0] = Iter; tr[1] = mu; tr[2] = rDu; tr[3] = rPr; tr[4] = rCp; tr[5] = pobj; tr[6] = dobj; tr[7] = NAN; tr[8] = NAN; }
        res->trace_rows = tr ? Iter : res->trace_rows;
        if (o.verbose) printf(" %6d | %-8.1e %-8.1e %-8.1e | % -8.1e % -8.1e\n", Iter, rDu, rPr, rCp, pobj, dobj);

        int status = CIP_STATUS_NONE;
        if (worst < o.optTol) status = CIP_STATUS_OPTIMAL;                             // :786
        if (!(p == 0 && m == 0)) {                                                     // :790
            const double dTy_bTv = dTw - bTv;                                          // :808
            double p_infeas = NAN;
            if (dTy_bTv < 0) {
                const double p_unscaled = nrm(pinf2);                                  // :810
                const double den = nrm(yy) + (m > 0 ? nrm(vv) : 0.0);
                const double p_cvx = den != 0 ? p_unscaled / den : INFINITY;           // :811
                const double p_ecos = p_unscaled / (std::fmax(1.0, normc) * std::fabs(dTy_bTv));   // :812
                p_infeas = jlmax(p_cvx, p_ecos);
            }
            if (p_infeas < o.infeasTol) {                                              // :815-818
                CK(finish(CIP_STATUS_INFEASIBLE));
                for (int i = 0; i < n; ++i) y_out[i] = NAN;
                for (int i = 0; i < p; ++i) w_out[i] /= -dTy_bTv;
                for (int i = 0; i < m; ++i) v_out[i] /= -dTy_bTv;
                return 0;
            }
            const double d1 = m == 0 ? -INFINITY : nrm(ays2);                          // :839
            const double d2 = p == 0 ? -INFINITY : nrm(gy2);                           // :840
            const double d3 = nrm(qy2);                                                // :841
            double d_infeas = NAN;
            if (cTy > 0) {
                const double d_cvx = jlmax(d1 / std::fmax(1.0, normb), d2 / std::fmax(1.0, normd), d3 / std::fmax(1.0, normc)) / std::fabs(cTy);   // :843
                const double ny = nrm(yy);
                const double d_ecos = ny != 0 ? jlmax(d1, d2, d3) / ny : INFINITY;     // :844
                d_infeas = std::fabs(jlmax(d_cvx, d_ecos));
            }
            if (d_infeas < o.infeasTol) {                                              // :847-850
                CK(finish(CIP_STATUS_UNBOUNDED));
                for (int i = 0; i < n; ++i) y_out[i] /= std::fabs(cTy);
                for (int i = 0; i < m; ++i) v_out[i] = NAN;
                for (int i = 0; i < p; ++i) w_out[i] = NAN;
                return 0;
            }
        }
        if (status != CIP_STATUS_NONE) return finish(status);                          // :867
        if (!(std::isfinite(mu) && std::isfinite(rDu) && std::isfinite(rPr) && std::isfinite(rCp))) return finish(CIP_STATUS_ERROR);   // :870-873

        // ------------------------------------------------------------ predictor (:879-887)
        CK(cip_solve4x4_dev(h, lam, r0.base, daff.base)); ++n_solve;
        double a_aff = 1.0, sigma = 0.0;
        if (m > 0) {
            double a1, a2;
            CK(cip_maxstep_dev(h, z.v, daff.v, 1.0, &a1));
            CK(cip_maxstep_dev(h, z.s, daff.s, 1.0, &a2));
            a_aff = std::fmin(std::fmin(a1, 1.0), a2);
            const double *qx[4] = {z.v, z.v, daff.v, daff.v};
            const double *qy[4] = {z.s, daff.s, z.s, daff.s};
            const int ql[4] = {m, m, m, m};
            double q4[4];
            CK(cip_dots_dev(h, 4, qx, qy, ql, q4));
            const double rho = (q4[0] - a_aff * q4[1] - a_aff * q4[2] + a_aff * a_aff * q4[3]) / mubar;   // fts :162-163, :886
            const double cl = std::fmax(0.0, std::fmin(1.0, rho));
            sigma = std::pow(cl, 3.0);          // as the Python driver's `** 3` (the two loops agree to the last bit)
        }

        // ------------------------------------------------------------ corrector (:893-901)
        CK(D.copy(D.NT, r0.base, r.base));
        if (m > 0) {
            CK(cip_apply_F_dev(h, CIP_OP_FINVT, daff.s, mb1));                         // F^-T d_aff.s
            CK(cip_apply_F_dev(h, CIP_OP_F, daff.v, mb2));                             // F d_aff.v
            CK(cip_cone_prod_dev(h, mb1, mb2, mb3));
            // lc = -(mb3 - sigma mu e) ; r.s = rleft.s - lc
            CK(D.axpby(m, 1.0, mb3, 1.0, r.s));
            CK(D.axpby(m, -sigma * mu, e, 1.0, r.s));
        }

        // ------------------------------------------------------------ Newton step + refinement (:907-921)
        CK(cip_solve4x4_dev(h, lam, r.base, dz.base)); ++n_solve;
        for (int it = 0; it < o.maxRefinementSteps; ++it) {
            CK(D.kkt_apply(dz, rkkt));
            if (m > 0) {
                CK(cip_apply_F_dev(h, CIP_OP_F, dz.v, mb1));
                CK(cip_cone_prod_dev(h, lam, mb1, mb2));
                CK(cip_apply_F_dev(h, CIP_OP_FINVT, dz.s, mb1));
                CK(cip_cone_prod_dev(h, lam, mb1, mb3));
                CK(D.copy(m, mb2, rkkt.s));
                CK(D.axpby(m, 1.0, mb3, 1.0, rkkt.s));
            }
            CK(D.copy(D.NT, r.base, rIr.base));
            CK(D.axpby(D.NT, -1.0, rkkt.base, 1.0, rIr.base));
            const double *nx[4] = {rIr.y, rIr.w, rIr.v, rIr.s};
            const int nl[4] = {n, p, m, m};
            double n2[4];
            CK(cip_dots_dev(h, 4, nx, nx, nl, n2));
            const double rnorm = (nrm(n2[0]) + (p > 0 ? nrm(n2[1]) : 0.0) + (m > 0 ? nrm(n2[2]) + nrm(n2[3]) : 0.0)) / (n + 2 * m);   // :917 (norm(v4x1) :61)
            if (rnorm < o.refinementThreshold) break;
            CK(cip_solve4x4_dev(h, lam, rIr.base, dzr.base)); ++n_solve;
            CK(D.axpby(D.NT, 1.0, dzr.base, 1.0, dz.base));                            // :920
        }

        // ------------------------------------------------------------ step (:927-932)
        double alpha = 1.0;
        if (m > 0) {
            double a_v, a_s;
            CK(cip_maxstep_dev(h, z.v, dz.v, 1.0 / (1.0 - o.DTB), &a_v));
            CK(cip_maxstep_dev(h, z.s, dz.s, 1.0 / (1.0 - o.DTB), &a_s));
            alpha = std::fmin(std::fmin(a_v, 1.0), std::fmin(a_s, 1.0));
        }
        CK(D.axpby(D.NT, -alpha, dz.base, 1.0, z.base));
        if (tr) { tr[7] = alpha; tr[8] = sigma; }
    }
    return finish(CIP_STATUS_ABANDONED);                                               // :936
#undef CK
}
