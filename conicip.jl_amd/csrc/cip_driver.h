// Host-side pieces shared by the two native interior-point loops (driver.hip: one problem; lockstep.hip: a lock-step
// batch): the vector layout of the loop inside one device allocation, and the per-iteration scalar logic of
// src/ConicIP.jl:756-873 (residuals, best-iterate bookkeeping, objective values, stopping tests, certificates).
#pragma once
#include "cip_handle.h"
#include "../../include/cipkkt.h"
#include <cmath>

namespace cipdrv {

inline double jlmax(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }   // Julia max propagates NaN
inline double jlmax(double a, double b, double c) { return jlmax(jlmax(a, b), c); }
inline double nrm(double x2) { return x2 >= 0 ? std::sqrt(x2) : NAN; }

struct Vec4 {          // (y[n], w[p], v[m], s[m]) stored contiguously
    double *base = nullptr, *y = nullptr, *w = nullptr, *v = nullptr, *s = nullptr;
};

inline size_t pad32(size_t c) { return (c + 31) & ~(size_t)31; }      // 256-byte alignment of every vector
inline size_t driver_doubles(int n, int m, int p) {
    const size_t NT = (size_t)n + p + 2 * (size_t)m;
    return 9 * pad32(NT) + pad32(n) + pad32(m) + pad32(p) + 5 * pad32(m) + 2 * pad32(n) + pad32(m) + pad32(p) + 32;
}

// every vector of the loop, carved out of h->drv in a fixed order
struct Vectors {
    double *c_d, *b_d, *d_d;
    Vec4 z, r0, rleft, r, daff, dz, dzr, rIr, rkkt;
    double *e, *lam, *mb1, *mb2, *mb3, *Qy, *pinf, *Ays, *Gy;
    void carve(double *base, int n, int m, int p) {
        double *next = base;
        auto dalloc = [&](size_t count) { double *ptr = next; next += pad32(count); return ptr; };
        const size_t NT = (size_t)n + p + 2 * (size_t)m;
        auto vec4 = [&]() { Vec4 v; v.base = dalloc(NT); v.y = v.base; v.w = v.y + n; v.v = v.w + p; v.s = v.v + m; return v; };
        c_d = dalloc(n); b_d = dalloc(m); d_d = dalloc(p);
        z = vec4(); r0 = vec4(); rleft = vec4(); r = vec4(); daff = vec4(); dz = vec4(); dzr = vec4(); rIr = vec4(); rkkt = vec4();
        e = dalloc(m); lam = dalloc(m); mb1 = dalloc(m); mb2 = dalloc(m); mb3 = dalloc(m);
        Qy = dalloc(n); pinf = dalloc(n); Ays = dalloc(m); Gy = dalloc(p);
    }
};

struct Norms { double normc = 0, normb = 0, normd = -INFINITY; };
inline Norms host_norms(int n, int m, int p, const double *c, const double *b, const double *d) {
    Norms nm;
    for (int i = 0; i < n; ++i) nm.normc += c[i] * c[i];
    nm.normc = std::sqrt(nm.normc);
    for (int i = 0; i < m; ++i) nm.normb += b[i] * b[i];
    nm.normb = std::sqrt(nm.normb);
    if (p > 0) { nm.normd = 0; for (int i = 0; i < p; ++i) nm.normd += d[i] * d[i]; nm.normd = std::sqrt(nm.normd); }
    return nm;
}

// the 16 dot products one iteration reads back, in this order (driver.hip / lockstep.hip fill the pointer tables alike)
struct IterDots { double v[16]; };

struct IterOutcome {
    int status = CIP_STATUS_NONE;     // NONE: keep iterating
    double mu = NAN, mubar = NAN;
    double scale = NAN;               // Infeasible: (w, v) /= scale, y = NaN ; Unbounded: y /= scale, (w, v) = NaN
};

// src/ConicIP.jl:756-873 on the host: updates res (best iterate, objectives) and the trace row, returns what the loop
// does next.  Quirks kept: rPr ignores the equality residual (:765); the primal-infeasibility
// certificate is tested before the dual one and before :Optimal is acted upon.
inline IterOutcome evaluate_iteration(const IterDots &dd, const Norms &nm, double conedim, int m, int p, const cip_options &o,
                                      int Iter, cip_result *res, double &optBest, double *tr) {
    const double *dt = dd.v;
    IterOutcome out;
    const double mubar = dt[0], cTy = dt[1], r0y2 = dt[2], r0v2 = dt[3], r0s2 = dt[4], yQy = dt[5], wr0w = dt[6],
                 vr0v = dt[7], dTw = dt[8], bTv = dt[9], pinf2 = dt[10], yy = dt[11], vv = dt[12], ays2 = dt[13],
                 gy2 = dt[14], qy2 = dt[15];
    const double mu = conedim > 0 ? mubar / conedim : NAN;                         // :756-757
    out.mu = mu; out.mubar = mubar;
    const double rDu = nrm(r0y2) / (1 + nm.normc);                                 // :764
    const double rPr = (m > 0 ? nrm(r0v2) : 0.0) / (1 + nm.normb);                 // :765
    const double rCp = (m > 0 ? nrm(r0s2) : 0.0) / (1 + std::fabs(cTy));           // :766
    const double worst = jlmax(rDu, rPr, rCp);
    if (worst < optBest) {                                                         // :768-773
        res->iter = Iter; res->mu = mu; res->duFeas = rDu; res->prFeas = rPr; res->muFeas = rCp;
        optBest = worst;
    }
    const double pobj = 0.5 * yQy - cTy;                                           // :775
    const double dobj = pobj + wr0w + vr0v - mubar;                                // :776
    res->pobj = pobj; res->dobj = dobj;
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
            const double p_ecos = p_unscaled / (std::fmax(1.0, nm.normc) * std::fabs(dTy_bTv));   // :812
            p_infeas = jlmax(p_cvx, p_ecos);
        }
        if (p_infeas < o.infeasTol) {                                              // :815-818
            out.status = CIP_STATUS_INFEASIBLE; out.scale = -dTy_bTv;
            return out;
        }
        const double d1 = m == 0 ? -INFINITY : nrm(ays2);                          // :839
        const double d2 = p == 0 ? -INFINITY : nrm(gy2);                           // :840
        const double d3 = nrm(qy2);                                                // :841
        double d_infeas = NAN;
        if (cTy > 0) {
            const double d_cvx = jlmax(d1 / std::fmax(1.0, nm.normb), d2 / std::fmax(1.0, nm.normd), d3 / std::fmax(1.0, nm.normc)) / std::fabs(cTy);   // :843
            const double ny = nrm(yy);
            const double d_ecos = ny != 0 ? jlmax(d1, d2, d3) / ny : INFINITY;     // :844
            d_infeas = std::fabs(jlmax(d_cvx, d_ecos));
        }
        if (d_infeas < o.infeasTol) {                                              // :847-850
            out.status = CIP_STATUS_UNBOUNDED; out.scale = std::fabs(cTy);
            return out;
        }
    }
    if (status != CIP_STATUS_NONE) { out.status = status; return out; }            // :867
    if (!(std::isfinite(mu) && std::isfinite(rDu) && std::isfinite(rPr) && std::isfinite(rCp))) out.status = CIP_STATUS_ERROR;   // :870-873
    return out;
}

// what the certificates do to the returned iterate (:816-817, :848-849)
inline void apply_certificate(const IterOutcome &oc, int n, int m, int p, double *y, double *w, double *v) {
    if (oc.status == CIP_STATUS_INFEASIBLE) {
        for (int i = 0; i < n; ++i) y[i] = NAN;
        for (int i = 0; i < p; ++i) w[i] /= oc.scale;
        for (int i = 0; i < m; ++i) v[i] /= oc.scale;
    } else if (oc.status == CIP_STATUS_UNBOUNDED) {
        for (int i = 0; i < n; ++i) y[i] /= oc.scale;
        for (int i = 0; i < m; ++i) v[i] = NAN;
        for (int i = 0; i < p; ++i) w[i] = NAN;
    }
}

inline cip_options resolve_options(const cip_options *opt_in) {
    cip_options o;
    o.optTol = 1e-6; o.DTB = 0.01; o.infeasTol = -1.0; o.refinementThreshold = -1.0;
    o.maxRefinementSteps = 3; o.maxIters = 100; o.verbose = 0;                       // src/ConicIP.jl:498-509
    if (opt_in) o = *opt_in;
    if (o.infeasTol < 0) o.infeasTol = o.optTol;
    if (o.refinementThreshold < 0) o.refinementThreshold = o.optTol / 1e7;
    return o;
}

inline double cone_degree(const cip_handle *h) {      // conedim (:547-552)
    double conedim = 0;
    for (const ConeDesc &cd : h->h_cones) conedim += cd.type == CIP_CONE_R ? cd.dim : (cd.type == CIP_CONE_Q ? 1 : cd.r);
    return conedim;
}

}   // namespace cipdrv
