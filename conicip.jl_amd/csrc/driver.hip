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
#include "cip_driver.h"
#include <chrono>
#include <vector>

using namespace cipdrv;

size_t cip_driver_bytes(const cip_handle *h) { return sizeof(double) * driver_doubles(h->n, h->m, h->p); }

namespace {

struct Driver {
    cip_handle *h;
    int n, m, p, NT;
    int rc = 0;

    int init() {
        if (!h->drv) {
            void *ptr = nullptr;
            if (cip_handle_alloc(h, &ptr, cip_driver_bytes(h)) != 0) { rc = CIP_E_HIP; return rc; }
            h->drv = (double *)ptr;
        }
        if (hipMemsetAsync(h->drv, 0, cip_driver_bytes(h), h->stream) != hipSuccess) { rc = CIP_E_HIP; return rc; }
        return 0;
    }

    // y <- alpha x + beta y
    int axpby(int len, double alpha, const double *x, double beta, double *y) { return len > 0 ? cip_axpby_dev(h, len, alpha, x, beta, y) : 0; }
    int copy(int len, const double *x, double *y) { return axpby(len, 1.0, x, 0.0, y); }

    // out.y = Q x.y + G' x.w - A' x.v ; out.w = G x.y ; out.v = A x.y - x.s     (:747-749, :912-914)
    // Qx != NULL: Q x.y has already been computed (the certificates need it on its own): copied instead of a second
    // pass over Q -- the same bits, 8 n^2 bytes of HBM traffic less per iteration
    int kkt_apply(const Vec4 &x, Vec4 &out, const double *Qx = nullptr) {
        int e = Qx ? copy(n, Qx, out.y) : cip_gemv_dev(h, CIP_MAT_Q, 0, 1.0, x.y, 0.0, out.y);
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
    const cip_options o = resolve_options(opt_in);
    CIP_HIP_CHECK(hipSetDevice(h->device));

    Driver D{h, h->n, h->m, h->p, h->n + h->p + 2 * h->m};
    const int n = D.n, m = D.m, p = D.p;
    if (D.init()) { cip_set_error("cip_conicip: device allocation failed"); return CIP_E_HIP; }
    Vectors V;
    V.carve(h->drv, n, m, p);
    double *c_d = V.c_d, *b_d = V.b_d, *d_d = V.d_d;
    Vec4 &z = V.z, &r0 = V.r0, &rleft = V.rleft, &r = V.r, &daff = V.daff, &dz = V.dz, &dzr = V.dzr, &rIr = V.rIr, &rkkt = V.rkkt;
    double *e = V.e, *lam = V.lam, *mb1 = V.mb1, *mb2 = V.mb2, *mb3 = V.mb3;
    double *Qy = V.Qy, *pinf = V.pinf, *Ays = V.Ays, *Gy = V.Gy;
    CIP_HIP_CHECK(hipMemcpyAsync(c_d, c_host, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    if (m > 0) CIP_HIP_CHECK(hipMemcpyAsync(b_d, b_host, sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
    if (p > 0) CIP_HIP_CHECK(hipMemcpyAsync(d_d, d_host, sizeof(double) * p, hipMemcpyHostToDevice, h->stream));
    const Norms nm = host_norms(n, m, p, c_host, b_host, d_host);
    const double conedim = cone_degree(h);                                          // (:547-552); e (:559-565) below
    CipHostScratch hs;
    if (cip_host_scratch(&hs)) { cip_set_error("cip_conicip: host scratch"); return CIP_E_HIP; }
    constexpr int MS_SLOT = 64;                     // slots of the host-mapped scratch the deferred max-step pairs use (the dot products use 0 .. 31)
    const double *f = cip_loop_all_r(h);            // diag F when every cone is an R cone (the loop's cone operations are then fused into its vector kernels), else NULL
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
        double a2[2];
        CK(cip_maxstep_pair_dev(h, z.v, nullptr, z.s, nullptr, 1.0, a2));
        const double a_v = a2[0], a_s = a2[1];
        CK(D.axpby(m, -a_v, e, 1.0, z.v));
        CK(D.axpby(m, -a_s, e, 1.0, z.s));
    }

    struct IterRange { IterRange() { cip_range_push("cip:iteration"); } ~IterRange() { cip_range_pop(); } };
    for (int Iter = 1; Iter <= o.maxIters; ++Iter) {                                   // :730
        IterRange iter_range;
        if (m > 0) CK(cip_set_scaling_from_iterate_dev(h, z.v, z.s, lam));             // :732-735 (F, lambda = F v)
        CK(cip_factor(h)); ++n_factor;                                                 // :737 -> :682
        // (round 5) the element-wise part of :746-753 is one kernel (vecops.hip: k_loop_resid) behind the mat-vecs; with R cones
        // only (f != NULL) lam o lam is formed there too
        if (m > 0 && !f) CK(cip_cone_prod_dev(h, lam, lam, rleft.s));                  // :746
        CK(cip_gemv_dev(h, CIP_MAT_Q, 0, 1.0, z.y, 0.0, Qy));                          // needed by the certificates
        CK(D.copy(n, Qy, rleft.y));                                                    // :747-750: Q y + G'w - A'v, G y, A y (- s: in the kernel)
        if (p > 0) {
            CK(cip_gemv_dev(h, CIP_MAT_G, 1, 1.0, z.w, 1.0, rleft.y));
            CK(cip_gemv_dev(h, CIP_MAT_G, 0, 1.0, z.y, 0.0, rleft.w));
            CK(cip_gemv_dev(h, CIP_MAT_G, 1, 1.0, z.w, 0.0, pinf));
        }
        if (m > 0) {
            CK(cip_gemv_dev(h, CIP_MAT_A, 1, -1.0, z.v, 1.0, rleft.y));
            CK(cip_gemv_dev(h, CIP_MAT_A, 0, 1.0, z.y, 0.0, rleft.v));
            CK(cip_gemv_dev(h, CIP_MAT_A, 1, -1.0, z.v, p > 0 ? 1.0 : 0.0, pinf));     // pinf = G'w - A'v (a zero start when p == 0: the same bits)
        } else if (p == 0) CIP_HIP_CHECK(hipMemsetAsync(pinf, 0, sizeof(double) * n, h->stream));
        // rleft.v -= s, [rleft.s = lam o lam], Gy = rleft.w, Ays = rleft.v, r0 = rleft - (c, d, b, 0)   (:753)
        CK(cip_loop_resid(h->stream, n, m, p, rleft.base, z.s, c_d, d_d, b_d, lam, f, r0.base, Gy, Ays));

        const double *px[16] = {z.v, c_d, r0.y, r0.v, r0.s, z.y, z.w, z.v, d_d, b_d, pinf, z.y, z.v, Ays, Gy, Qy};
        const double *py[16] = {z.s, z.y, r0.y, r0.v, r0.s, Qy, r0.w, r0.v, z.w, z.v, pinf, z.y, z.v, Ays, Gy, Qy};
        const int ln[16] = {m, n, n, m, m, n, p, m, p, m, n, n, m, m, p, n};
        double dt[16];
        CK(cip_dots_dev(h, 16, px, py, ln, dt));
        // the stream has just been drained: the pivot flag of this iteration's factorisation is in.  A dead (zero /
        // non-finite) pivot even after regularisation is where the reference's LU hands back NaNs and the loop ends
        // with :Error at its next residual check (src/ConicIP.jl:870-873)
        if ((rc = cip_factor_resolve(h, 1)) != 0) { if (rc == CIP_E_SINGULAR) return finish(CIP_STATUS_ERROR); return rc; }
        IterDots dd;
        for (int i = 0; i < 16; ++i) dd.v[i] = dt[i];
        double *tr = (trace && Iter <= trace_cap) ? trace + (size_t)(Iter - 1) * CIP_TRACE_COLS : nullptr;
        const IterOutcome oc = evaluate_iteration(dd, nm, conedim, m, p, o, Iter, res, optBest, tr);
        const double mubar = oc.mubar, mu = oc.mu;
        if (oc.status != CIP_STATUS_NONE) {
            CK(finish(oc.status));
            apply_certificate(oc, n, m, p, y_out, w_out, v_out);
            return 0;
        }

        // ------------------------------------------------------------ predictor (:879-887)
        CK(cip_solve4x4_dev(h, lam, r0.base, daff.base)); ++n_solve;
        double a_aff = 1.0, sigma = 0.0;
        if (m > 0) {
            // (round 5) one host round trip for the pair of max-steps and the four dot products: the minima are left in the host-mapped
            // scratch without a wait and read behind the dots' wait (as the lock-step loop does since round 4)
            CK(cip_cones_maxstep2(h->stream, h->cs, z.v, daff.v, z.s, daff.s, 1.0, nullptr, MS_SLOT));
            const double *qx[4] = {z.v, z.v, daff.v, daff.v};
            const double *qy[4] = {z.s, daff.s, z.s, daff.s};
            const int ql[4] = {m, m, m, m};
            double q4[4];
            CK(cip_dots_dev(h, 4, qx, qy, ql, q4));
            const double a1 = hs.host[MS_SLOT], a2 = hs.host[MS_SLOT + 1];
            a_aff = std::fmin(std::fmin(a1, 1.0), a2);
            const double rho = (q4[0] - a_aff * q4[1] - a_aff * q4[2] + a_aff * a_aff * q4[3]) / mubar;   // fts :162-163, :886
            const double cl = std::fmax(0.0, std::fmin(1.0, rho));
            sigma = std::pow(cl, 3.0);          // as the Python driver's `** 3` (the two loops agree to the last bit)
        }

        // ------------------------------------------------------------ corrector (:893-901)
        // r = r0 ; r.s += (F^-T d_aff.s) o (F d_aff.v) - sigma mu e     -- one kernel (vecops.hip: k_loop_corr); the cone operations in
        // front of it unless every cone is an R cone
        if (m > 0 && !f) {
            CK(cip_apply_F_dev(h, CIP_OP_FINVT, daff.s, mb1));                         // F^-T d_aff.s
            CK(cip_apply_F_dev(h, CIP_OP_F, daff.v, mb2));                             // F d_aff.v
            CK(cip_cone_prod_dev(h, mb1, mb2, mb3));
        }
        if (m > 0) { const double sm = sigma * mu; CK(cip_loop_corr(h->stream, n, m, p, r0.base, daff.base, mb3, e, f, &sm, r.base)); }
        else CK(D.copy(D.NT, r0.base, r.base));

        // ------------------------------------------------------------ Newton step + refinement (:907-921)
        CK(cip_solve4x4_dev(h, lam, r.base, dz.base)); ++n_solve;
        bool step_known = false;
        double step_a[2] = {0.0, 0.0};
        for (int it = 0; it < o.maxRefinementSteps; ++it) {
            // rkkt = K dz (mat-vecs), then rkkt.v -= dz.s, rkkt.s = lam o (F dz.v) + lam o (F^-T dz.s), rIr = r - rkkt: one kernel
            // (vecops.hip: k_loop_refine)
            CK(cip_gemv_dev(h, CIP_MAT_Q, 0, 1.0, dz.y, 0.0, rkkt.y));
            if (p > 0) {
                CK(cip_gemv_dev(h, CIP_MAT_G, 1, 1.0, dz.w, 1.0, rkkt.y));
                CK(cip_gemv_dev(h, CIP_MAT_G, 0, 1.0, dz.y, 0.0, rkkt.w));
            }
            if (m > 0) {
                CK(cip_gemv_dev(h, CIP_MAT_A, 1, -1.0, dz.v, 1.0, rkkt.y));
                CK(cip_gemv_dev(h, CIP_MAT_A, 0, 1.0, dz.y, 0.0, rkkt.v));
                if (!f) {
                    CK(cip_apply_F_dev(h, CIP_OP_F, dz.v, mb1));
                    CK(cip_cone_prod_dev(h, lam, mb1, mb2));
                    CK(cip_apply_F_dev(h, CIP_OP_FINVT, dz.s, mb1));
                    CK(cip_cone_prod_dev(h, lam, mb1, mb3));
                }
            }
            CK(cip_loop_refine(h->stream, n, m, p, rkkt.base, dz.base, r.base, lam, mb2, mb3, f, rIr.base));
            const double *nx[4] = {rIr.y, rIr.w, rIr.v, rIr.s};
            const int nl[4] = {n, p, m, m};
            double n2[4];
            // the step's two max-steps ride on the first refinement test's read-back: when no refinement is asked for -- the usual
            // case -- dz is final and the iteration has saved a host round trip; otherwise they are taken again behind the loop
            const bool spec = it == 0 && m > 0;
            if (spec) CK(cip_cones_maxstep2(h->stream, h->cs, z.v, dz.v, z.s, dz.s, 1.0 / (1.0 - o.DTB), nullptr, MS_SLOT));
            CK(cip_dots_dev(h, 4, nx, nx, nl, n2));
            const double rnorm = (nrm(n2[0]) + (p > 0 ? nrm(n2[1]) : 0.0) + (m > 0 ? nrm(n2[2]) + nrm(n2[3]) : 0.0)) / (n + 2 * m);   // :917 (norm(v4x1) :61)
            if (rnorm < o.refinementThreshold) {
                if (spec) { step_known = true; step_a[0] = hs.host[MS_SLOT]; step_a[1] = hs.host[MS_SLOT + 1]; }
                break;
            }
            CK(cip_solve4x4_dev(h, lam, rIr.base, dzr.base)); ++n_solve;
            CK(D.axpby(D.NT, 1.0, dzr.base, 1.0, dz.base));                            // :920
        }

        // ------------------------------------------------------------ step (:927-932)
        double alpha = 1.0;
        if (m > 0) {
            if (!step_known) CK(cip_maxstep_pair_dev(h, z.v, dz.v, z.s, dz.s, 1.0 / (1.0 - o.DTB), step_a));
            const double a_v = step_a[0], a_s = step_a[1];
            alpha = std::fmin(std::fmin(a_v, 1.0), std::fmin(a_s, 1.0));
        }
        CK(D.axpby(D.NT, -alpha, dz.base, 1.0, z.base));
        if (tr) { tr[7] = alpha; tr[8] = sigma; }
    }
    return finish(CIP_STATUS_ABANDONED);                                               // :936
#undef CK
}
