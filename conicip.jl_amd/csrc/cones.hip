// Per-cone kernels over the block-diagonal Nesterov-Todd scaling (R and Q cones here;
// S-cone kernels live in sdp.hip).  Device counterparts of
//   nt_scaling / nestod_soc         src/ConicIP.jl:589-605, :165-194
//   Block * x, Block' * x, inv      src/blockmatrices.jl:173-200  (+ WoodburyMatrices SymWoodbury algebra)
//   cone_prod! / cone_div!          src/ConicIP.jl:607-665, :305-345
//   maxstep                         src/ConicIP.jl:212-270, :571-587
//   cone identity e                 src/ConicIP.jl:559-565
//
// One workgroup (256 threads = 4 wave64) per work item.  A work item is a <= 2048-element chunk of an R cone, one Q cone
// of dimension > 64 (all 256 threads: wave shuffles + one LDS hop per reduction), or a PACK of consecutive Q cones of
// dimension <= 64: each cone gets a lane segment of width W = the next power of two (256 / W cones per workgroup, e.g.
// 32 cones of BASELINE config 3's ("Q", 8) per workgroup, 8 per wavefront), its reductions are xor-shuffles inside the
// segment and no barrier is needed at all.  (The first version spent a 256-thread workgroup on every Q(8): 8 of 256
// lanes busy, 512 workgroups per call for config 3; now 16.)
//
// Packed scaling storage per cone (== what the Julia shim reads off the Block elements):
//   R: diag(F) (k)      Q: beta, w (1+k) with F = diag(-beta,beta,..) + w w'      S: R, inv(R)
#include "cip_internal.h"
#include "../../include/cipkkt.h"
#include <math.h>

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
// block-wide sum of up to 3 values at once (256 threads); result broadcast to all threads
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *sh /*>=12*/) {
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[w] = a; sh[4 + w] = b; sh[8 + w] = c; }
    __syncthreads();
    a = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    b = (sh[4] + sh[5]) + (sh[6] + sh[7]);
    c = (sh[8] + sh[9]) + (sh[10] + sh[11]);
}
__device__ __forceinline__ double block_min(double a, double *sh) {
    a = wave_min(a);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = a;
    __syncthreads();
    return fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
}

// the lanes that work on one Q cone: the whole workgroup (T = 256) or a segment of a wavefront (T = pack width)
struct QTeam { ConeDesc cd; int T, tl, slot; bool active; };
__device__ __forceinline__ QTeam q_team(const ConeDesc *cones, const WorkItem &it, const ConeDesc &first) {
    QTeam t;
    t.cd = first; t.T = 256; t.tl = threadIdx.x; t.slot = it.slot; t.active = true;
    if (it.width) {
        const int seg = threadIdx.x / it.width;
        t.T = it.width; t.tl = threadIdx.x - seg * it.width; t.active = seg < it.len; t.slot = it.slot + seg;
        t.cd = cones[it.cone + (t.active ? seg : 0)];
    }
    return t;
}
__device__ __forceinline__ void team_sum3(double &a, double &b, double &c, int T, double *sh) {
    if (T == 256) { block_sum3(a, b, c, sh); return; }
    for (int o = T >> 1; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
}
__device__ __forceinline__ void team_sync(int T) { if (T == 256) __syncthreads(); }   // a segment lives inside one wavefront

// ------------------------------------------------------------------ NT scaling
__global__ __launch_bounds__(256) void k_nt_scaling(const ConeDesc *cones, const WorkItem *items, const double *v,
                                                     const double *s, double *scal, double *lambda, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, v, s, scal, lambda);
    __shared__ double sh[12];
    const WorkItem it = items[blockIdx.x];
    const ConeDesc cd = cones[it.cone];
    const int tid = threadIdx.x;
    if (cd.type == CIP_CONE_R) {
        for (int e = it.start + tid; e < it.start + it.len; e += 256) {
            const double vi = v[cd.off + e], si = s[cd.off + e];
            const double d = sqrt(si / vi);                        // src/ConicIP.jl:598
            scal[cd.soff + e] = d;
            if (lambda) lambda[cd.off + e] = d * vi;
        }
    } else if (cd.type == CIP_CONE_Q) {
        // nestod_soc(z = v block, s = s block)  src/ConicIP.jl:165-194
        const QTeam tm = q_team(cones, it, cd);
        const ConeDesc qc = tm.cd;
        const int T = tm.T, tl = tm.tl;
        const double *z = v + qc.off, *sv = s + qc.off;
        const int k = qc.dim;
        double zz = 0, ss = 0, zs = 0;
        for (int e = 1 + tl; e < k; e += T) { zz += z[e] * z[e]; ss += sv[e] * sv[e]; zs += z[e] * sv[e]; }
        team_sum3(zz, ss, zs, T, sh);
        const double z0 = z[0], s0 = sv[0];
        const double qfz = z0 * z0 - zz, qfs = s0 * s0 - ss;       // QF(.)
        const double beta = sqrt(sqrt(qfs / qfz));                 // (QF(s)/QF(z))^(1/4)
        const double rz = 1.0 / sqrt(qfz), rs = 1.0 / sqrt(qfs);
        const double zdots = (z0 * s0 + zs) * rz * rs;             // zbar . sbar
        const double gamma = sqrt((1.0 + zdots) * 0.5);
        const double h = 1.0 / (2.0 * gamma);
        const double wb0 = h * (s0 * rs + z0 * rz);                // wbar_1
        const double c = sqrt(beta / (wb0 + 1.0));                 // sqrt(2 beta)/sqrt(2 w[1])
        // w = c * (wbar + e1), wbar_t = h (sbar_t - zbar_t)
        // w . v (v == z):  c * (wbar.z + z0),  wbar.z = h * (s.z * rs + QF(z) * rz)
        const double wdotz = c * (h * ((z0 * s0 + zs) * rs + qfz * rz) + z0);
        if (tm.active) {
            if (tl == 0) {
                scal[qc.soff] = beta;
                const double w0 = c * (wb0 + 1.0);
                scal[qc.soff + 1] = w0;
                if (lambda) lambda[qc.off] = -beta * z0 + w0 * wdotz;
            }
            for (int e = 1 + tl; e < k; e += T) {
                const double we = c * h * (sv[e] * rs - z[e] * rz);
                scal[qc.soff + 1 + e] = we;
                if (lambda) lambda[qc.off + e] = beta * z[e] + we * wdotz;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_identity_scaling(const ConeDesc *cones, const WorkItem *items, double *scal, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO1(cb, scal);
    const WorkItem it = items[blockIdx.x];
    const ConeDesc cd = cones[it.cone];
    const int tid = threadIdx.x;
    if (cd.type == CIP_CONE_R) {
        for (int e = it.start + tid; e < it.start + it.len; e += 256) scal[cd.soff + e] = 1.0;
    } else if (cd.type == CIP_CONE_Q) {
        // I = diag(-beta, beta, ...) + w w' with beta = 1, w = sqrt(2) e1
        const QTeam tm = q_team(cones, it, cd);
        if (tm.active) {
            if (tm.tl == 0) { scal[tm.cd.soff] = 1.0; scal[tm.cd.soff + 1] = sqrt(2.0); }
            for (int e = 1 + tm.tl; e < tm.cd.dim; e += tm.T) scal[tm.cd.soff + 1 + e] = 0.0;
        }
    } else {
        const int r = cd.r;
        for (int e = tid; e < r * r; e += 256) {
            const double v = ((e % r) == (e / r)) ? 1.0 : 0.0;
            scal[cd.soff + e] = v;
            scal[cd.soff + r * r + e] = v;
        }
    }
}

// ------------------------------------------------------------------ F, F', F^-1, F^-T on a vector
__global__ __launch_bounds__(256) void k_apply(const ConeDesc *cones, const WorkItem *items, const double *scal,
                                                int mode, const double *x, double *out, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO3(cb, scal, x, out);
    __shared__ double sh[12];
    const WorkItem it = items[blockIdx.x];
    const ConeDesc cd = cones[it.cone];
    const int tid = threadIdx.x;
    const bool inv = (mode == CIP_OP_FINV || mode == CIP_OP_FINVT);
    if (cd.type == CIP_CONE_R) {
        for (int e = it.start + tid; e < it.start + it.len; e += 256) {
            const double d = scal[cd.soff + e];
            out[cd.off + e] = inv ? x[cd.off + e] / d : x[cd.off + e] * d;
        }
    } else if (cd.type == CIP_CONE_Q) {
        const QTeam tm = q_team(cones, it, cd);
        const ConeDesc qc = tm.cd;
        const int T = tm.T, tl = tm.tl;
        const int k = qc.dim;
        const double beta = scal[qc.soff];
        const double *w = scal + qc.soff + 1;
        const double *xb = x + qc.off;
        double *ob = out + qc.off;
        double wx = 0, d1 = 0, d2 = 0;
        for (int e = 1 + tl; e < k; e += T) wx += w[e] * xb[e];
        const double w0 = w[0], x0 = xb[0];      // read before the barriers: out may alias x
        team_sum3(wx, d1, d2, T, sh);
        if (!tm.active) return;
        if (!inv) {
            // F x = -beta J x + w (w.x)
            const double t = w0 * x0 + wx;
            if (tl == 0) ob[0] = -beta * x0 + w0 * t;
            for (int e = 1 + tl; e < k; e += T) ob[e] = beta * xb[e] + w[e] * t;
        } else {
            // F^-1 x = ( (Jw) (Jw.x)/beta - J x ) / beta
            const double t = (w0 * x0 - wx) / beta;
            const double ib = 1.0 / beta;
            if (tl == 0) ob[0] = (w0 * t - x0) * ib;
            for (int e = 1 + tl; e < k; e += T) ob[e] = (xb[e] - w[e] * t) * ib;
        }
    }
}

// Wt[i, off + e] = (F^-T a_i)_e,  a_i = At[i, off:off+k]   (thread per row i of At, coalesced along i)
// The packs of small Q cones; the R chunks and the large Q cones have kernels of their own below (grids over THEIR items only:
// an empty workgroup costs ~50 ns of dispatch, and config 3's 512 x Q(8) ran 130 -> 330 us when 15 of 16 workgroups of a
// common grid exited at once)
#define SCALE_AT_SPLIT 16
__global__ __launch_bounds__(256) void k_scale_At(const ConeDesc *cones, const WorkItem *items, const int *packq, const double *scal,
                                                   int n, const double *At, long ldat, double *Wt, long ldwt, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO3(cb, scal, At, Wt);
    const WorkItem it = items[packq[blockIdx.y]];         // grid.y = the packs of small Q cones only
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int ncone = it.len;                             // a pack of small cones: this thread's row through each of them
    for (int q = 0; q < ncone; ++q) {
        const ConeDesc qc = cones[it.cone + q];
        const int k = qc.dim;
        const double beta = scal[qc.soff];
        const double *w = scal + qc.soff + 1;
        const double *ap = At + i + (long)qc.off * ldat;
        double *wp = Wt + i + (long)qc.off * ldwt;
        double t = w[0] * ap[0];
        for (int e = 1; e < k; ++e) t -= w[e] * ap[(long)e * ldat];
        t /= beta;
        const double ib = 1.0 / beta;
        wp[0] = (w[0] * t - ap[0]) * ib;
        for (int e = 1; e < k; ++e) wp[(long)e * ldwt] = (ap[(long)e * ldat] - w[e] * t) * ib;
    }
}
// R chunks (up to 2048 rows of A each): grid.x = row blocks x SCALE_AT_SPLIT, a chunk's columns dealt to SCALE_AT_SPLIT
// workgroups per row block (round 4: one thread walked all of them -- one R cone of 4097 rows behind n = 4096 was 48
// workgroups and 0.7 ms); grid.y = the R items
__global__ __launch_bounds__(256) void k_scale_At_r(const ConeDesc *cones, const WorkItem *items, const int *ritems, const double *scal,
                                                     int n, const double *At, long ldat, double *Wt, long ldwt, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO3(cb, scal, At, Wt);
    const WorkItem it = items[ritems[blockIdx.y]];
    const ConeDesc cd = cones[it.cone];
    const int nxb = gridDim.x / SCALE_AT_SPLIT, sub = blockIdx.x / nxb;
    const int i = (blockIdx.x % nxb) * 256 + threadIdx.x;
    if (i >= n) return;
    const int per = (it.len + SCALE_AT_SPLIT - 1) / SCALE_AT_SPLIT;
    const int e0 = it.start + sub * per, e1 = min(it.start + it.len, e0 + per);
    for (int e = e0; e < e1; ++e) {
        const long c = cd.off + e;
        Wt[i + c * ldwt] = At[i + c * ldat] / scal[cd.soff + e];
    }
}

// The same for a Q cone of dimension > 64 (round 4; SURVEY 8(f3): the reference lifts / densifies such blocks,
// src/kktsolvers.jl:60-131).  F^-T a = (a - J w t) / beta... with t = (w0 a_0 - sum_e w_e a_e) / beta is two O(k) passes per row
// of A'; one thread per row walked them alone (k = 4097, n = 4096: 16 workgroups, 1.8 ms beside a 2.3-ms SYRK).  Here a
// workgroup takes 16 rows and spreads the cone's entries over 16 thread columns (e = c, c + 16, ..): 16 partial dot products
// per row, summed in a fixed order through LDS, then the element-wise pass on the same split -- n / 16 workgroups, the
// second pass re-reads what the first brought into L2.  Deterministic (no atomics); O(k n) beside the SYRK's O(k n^2).
__global__ __launch_bounds__(256) void k_scale_At_qbig(const ConeDesc *cones, const WorkItem *items, const int *bigq, const double *scal,
                                                        int n, const double *At, long ldat, double *Wt, long ldwt, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO3(cb, scal, At, Wt);
    const WorkItem it = items[bigq[blockIdx.y]];              // grid.y = the large Q cones (an empty workgroup costs ~50 ns of dispatch)
    const ConeDesc cd = cones[it.cone];
    __shared__ double part[16][17];
    const int r = threadIdx.x & 15, c = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + r;
    const bool live = i < n;
    const int k = cd.dim;
    const double beta = scal[cd.soff], ib = 1.0 / beta;
    const double *w = scal + cd.soff + 1;
    const double *ap = At + (live ? i : 0) + (long)cd.off * ldat;
    double acc = 0.0;
    for (int e = c; e < k; e += 16) {
        const double a = ap[(long)e * ldat];
        acc += (e == 0 ? w[0] : -w[e]) * a;
    }
    part[r][c] = acc;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part[r][q];
    t *= ib;
    if (!live) return;
    double *wp = Wt + i + (long)cd.off * ldwt;
    for (int e = c; e < k; e += 16) {
        const double a = ap[(long)e * ldat];
        wp[(long)e * ldwt] = (e == 0 ? (w[0] * t - a) : (a - w[e] * t)) * ib;
    }
}

// ------------------------------------------------------------------ Jordan product / division
__global__ __launch_bounds__(256) void k_cone_prod(const ConeDesc *cones, const WorkItem *items, const double *x,
                                                    const double *y, double *out, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO3(cb, x, y, out);
    __shared__ double sh[12];
    const WorkItem it = items[blockIdx.x];
    const ConeDesc cd = cones[it.cone];
    const int tid = threadIdx.x;
    if (cd.type == CIP_CONE_R) {                                   // xrp! src/ConicIP.jl:311-315
        for (int e = it.start + tid; e < it.start + it.len; e += 256)
            out[cd.off + e] = x[cd.off + e] * y[cd.off + e];
    } else if (cd.type == CIP_CONE_Q) {                            // xsoc! :340-345
        const QTeam tm = q_team(cones, it, cd);
        const int T = tm.T, tl = tm.tl;
        const int k = tm.cd.dim;
        const double *xb = x + tm.cd.off, *yb = y + tm.cd.off;
        double *ob = out + tm.cd.off;
        double xy = 0, d1 = 0, d2 = 0;
        for (int e = 1 + tl; e < k; e += T) xy += xb[e] * yb[e];
        team_sum3(xy, d1, d2, T, sh);
        const double x0 = xb[0], y0 = yb[0];                       // out may alias an input: heads read before any write
        team_sync(T);                                              // (a pack segment lives in one wavefront: program order)
        if (!tm.active) return;
        if (tl == 0) ob[0] = x0 * y0 + xy;
        for (int e = 1 + tl; e < k; e += T) ob[e] = x0 * yb[e] + y0 * xb[e];
    }
}

// out = x (./) y : solves y o out = x   (cone_div!(o, x, y) src/ConicIP.jl:622-635)
__global__ __launch_bounds__(256) void k_cone_div(const ConeDesc *cones, const WorkItem *items, const double *x,
                                                   const double *y, double *out, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO3(cb, x, y, out);
    __shared__ double sh[12];
    const WorkItem it = items[blockIdx.x];
    const ConeDesc cd = cones[it.cone];
    const int tid = threadIdx.x;
    if (cd.type == CIP_CONE_R) {                                   // drp! :305-309
        for (int e = it.start + tid; e < it.start + it.len; e += 256)
            out[cd.off + e] = x[cd.off + e] / y[cd.off + e];
    } else if (cd.type == CIP_CONE_Q) {                            // dsoc! :317-338 (arrow inverse)
        const QTeam tm = q_team(cones, it, cd);
        const int T = tm.T, tl = tm.tl;
        const int k = tm.cd.dim;
        const double *xb = x + tm.cd.off, *yb = y + tm.cd.off;
        double *ob = out + tm.cd.off;
        double yy = 0, yx = 0, d2 = 0;
        for (int e = 1 + tl; e < k; e += T) { yy += yb[e] * yb[e]; yx += yb[e] * xb[e]; }
        team_sum3(yy, yx, d2, T, sh);
        const double y1 = yb[0], x1 = xb[0];
        const double alpha = y1 * y1 - yy;
        const double b1 = (-x1 / alpha) + yx / (y1 * alpha);
        const double b2 = 1.0 / y1;
        team_sync(T);
        if (!tm.active) return;
        if (tl == 0) ob[0] = (y1 * x1 - yx) / alpha;
        for (int e = 1 + tl; e < k; e += T) ob[e] = yb[e] * b1 + xb[e] * b2;
    }
}

// ------------------------------------------------------------------ max step
// partial[item] = largest alpha with x - alpha*(scale*d) in the cone (d != NULL), or the
// `nothing` variant (distance into the cone, <= 0) when d == NULL.
__global__ __launch_bounds__(256) void k_maxstep(const ConeDesc *cones, const WorkItem *items, const double *x,
                                                  const double *d, double scale, double *partial, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO3(cb, x, d, partial);
    __shared__ double sh[12];
    const WorkItem it = items[blockIdx.x];
    const ConeDesc cd = cones[it.cone];
    const int tid = threadIdx.x;
    const double INF = __builtin_inf();
    if (cd.type == CIP_CONE_R) {
        double mn = INF;
        if (d) {                                                   // maxstep_rp :212-225
            for (int e = it.start + tid; e < it.start + it.len; e += 256) {
                const double de = d[cd.off + e] * scale;
                if (de > 0) mn = fmin(mn, x[cd.off + e] / de);
            }
            mn = block_min(mn, sh);
        } else {                                                   // :227-240
            for (int e = it.start + tid; e < it.start + it.len; e += 256) mn = fmin(mn, x[cd.off + e]);
            mn = block_min(mn, sh);
            mn = (mn > 0) ? 0.0 : -1.0 + mn;
        }
        if (tid == 0) partial[it.slot] = mn;
    } else if (cd.type == CIP_CONE_Q) {
        const QTeam tm = q_team(cones, it, cd);
        const int T = tm.T, tl = tm.tl;
        const int k = tm.cd.dim;
        const double *xb = x + tm.cd.off;
        if (!d) {                                                  // maxstep_soc(x, nothing) :264-270
            double xx = 0, a1 = 0, a2 = 0;
            for (int e = 1 + tl; e < k; e += T) xx += xb[e] * xb[e];
            team_sum3(xx, a1, a2, T, sh);
            const double a = sqrt(xx) - xb[0];
            if (tl == 0 && tm.active) partial[tm.slot] = (a < 0) ? 0.0 : -1.0 - a;
        } else {                                                   // maxstep_soc(x, d) :242-262
            const double *db = d + tm.cd.off;
            double xx = 0, xd = 0, a2 = 0;
            for (int e = 1 + tl; e < k; e += T) { xx += xb[e] * xb[e]; xd += xb[e] * (-scale * db[e]); }
            team_sum3(xx, xd, a2, T, sh);
            const double x0 = xb[0], d0 = -scale * db[0];
            const double gam = x0 * x0 - xx;                       // Q(x,x)
            const double rg = 1.0 / sqrt(gam);
            const double bet = (x0 * d0 - xd) * rg;                // Q(xbar, d)
            const double rho1 = bet * rg;
            const double mu = (bet + d0) / (x0 * rg + 1.0);
            double r2 = 0, b1 = 0, b2 = 0;
            for (int e = 1 + tl; e < k; e += T) {
                const double t = (-scale * db[e]) - mu * xb[e] * rg;
                r2 += t * t;
            }
            team_sum3(r2, b1, b2, T, sh);
            const double alpha = sqrt(r2) * rg - rho1;
            if (tl == 0 && tm.active) partial[tm.slot] = (alpha < 0) ? INF : 1.0 / alpha;
        }
    } else {
        if (tid == 0) partial[it.slot] = INF;         // S cone: written by k_sdp_maxstep afterwards
    }
}

__global__ __launch_bounds__(256) void k_min_reduce(const double *partial, int n, double *out, double *gather, int gslot, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO2(cb, partial, out);
    __shared__ double sh[12];
    double mn = __builtin_inf();
    int nan_seen = 0;
    for (int e = threadIdx.x; e < n; e += 256) {
        const double p = partial[e];
        if (p != p) nan_seen = 1; else mn = fmin(mn, p);
    }
    // NaN-propagating block min (Julia's min propagates NaN, src/ConicIP.jl:582)
    const double m2 = block_min(mn, sh);
    const int bad = __syncthreads_or(nan_seen);
    if (threadIdx.x == 0) {
        out[0] = bad ? __builtin_nan("") : m2;
        if (gather) gather[blockIdx.z * CIP_GATHER + gslot] = out[0];        // lock-step batch: one read-back for all problems
    }
}

__global__ __launch_bounds__(256) void k_cone_identity(const ConeDesc *cones, const WorkItem *items, double *e, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO1(cb, e);
    const WorkItem it = items[blockIdx.x];
    const ConeDesc cd = cones[it.cone];
    const int tid = threadIdx.x;
    if (cd.type == CIP_CONE_R) {
        for (int q = it.start + tid; q < it.start + it.len; q += 256) e[cd.off + q] = 1.0;
    } else if (cd.type == CIP_CONE_Q) {
        const QTeam tm = q_team(cones, it, cd);
        if (tm.active) for (int q = tm.tl; q < tm.cd.dim; q += tm.T) e[tm.cd.off + q] = (q == 0) ? 1.0 : 0.0;
    } else {
        // vecm(I): row-major upper triangle, diagonal entries at positions i*r - i(i-1)/2
        const int r = cd.r;
        for (int q = tid; q < cd.dim; q += 256) e[cd.off + q] = 0.0;
        __syncthreads();
        for (int i = tid; i < r; i += 256) e[cd.off + i * r - i * (i - 1) / 2] = 1.0;
    }
}

// ------------------------------------------------------------------ host launchers
int cip_sdp_nt_scaling(hipStream_t s, const ConeSet &cs, const double *v, const double *sv, double *lambda);
int cip_sdp_apply(hipStream_t s, const ConeSet &cs, int mode, const double *x, double *out);
int cip_sdp_prod(hipStream_t s, const ConeSet &cs, const double *x, const double *y, double *out);
int cip_sdp_div(hipStream_t s, const ConeSet &cs, const double *x, const double *y, double *out);
int cip_sdp_maxstep(hipStream_t s, const ConeSet &cs, const double *x, const double *d, double scale, double *partial);
int cip_sdp_scale_At(hipStream_t s, const ConeSet &cs, int n, const double *At, long ldat, double *Wt, long ldwt);

int cip_cones_nt_scaling(hipStream_t s, const ConeSet &cs, const double *v, const double *sv, double *lambda) {
    if (cs.nitems == 0) return 0;
    cip_launch_b(k_nt_scaling, dim3(cs.nitems), dim3(256), 0, s, cs.d_cones, cs.d_items, v, sv, cs.d_scal, lambda);
    CIP_HIP_CHECK(hipGetLastError());
    if (cs.has_S) return cip_sdp_nt_scaling(s, cs, v, sv, lambda);
    return 0;
}
int cip_cones_identity_scaling(hipStream_t s, const ConeSet &cs) {
    if (cs.nitems == 0) return 0;
    cip_launch_b(k_identity_scaling, dim3(cs.nitems), dim3(256), 0, s, cs.d_cones, cs.d_items, cs.d_scal);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_cones_apply(hipStream_t s, const ConeSet &cs, int mode, const double *x, double *out) {
    if (cs.nitems == 0) return 0;
    cip_launch_b(k_apply, dim3(cs.nitems), dim3(256), 0, s, cs.d_cones, cs.d_items, cs.d_scal, mode, x, out);
    CIP_HIP_CHECK(hipGetLastError());
    if (cs.has_S) return cip_sdp_apply(s, cs, mode, x, out);
    return 0;
}
int cip_cones_prod(hipStream_t s, const ConeSet &cs, const double *x, const double *y, double *out) {
    if (cs.nitems == 0) return 0;
    cip_launch_b(k_cone_prod, dim3(cs.nitems), dim3(256), 0, s, cs.d_cones, cs.d_items, x, y, out);
    CIP_HIP_CHECK(hipGetLastError());
    if (cs.has_S) return cip_sdp_prod(s, cs, x, y, out);
    return 0;
}
int cip_cones_div(hipStream_t s, const ConeSet &cs, const double *x, const double *y, double *out) {
    if (cs.nitems == 0) return 0;
    cip_launch_b(k_cone_div, dim3(cs.nitems), dim3(256), 0, s, cs.d_cones, cs.d_items, x, y, out);
    CIP_HIP_CHECK(hipGetLastError());
    if (cs.has_S) return cip_sdp_div(s, cs, x, y, out);
    return 0;
}
// defer_slot >= 0 (lock-step batches only): the minima go to slot `defer_slot` of every problem's row of the gather buffer and
// the call returns without a read-back -- they ride on the NEXT read-back of that buffer (lockstep.hip pairs the v- and
// s-side max-steps and the dot products that follow them: one host round trip instead of three); alpha_host is not touched
int cip_cones_maxstep(hipStream_t s, const ConeSet &cs, const double *x, const double *d, double scale, double *alpha_host, int defer_slot) {
    if (cs.nitems == 0) { if (alpha_host) for (int z = 0; z < (cip_tl_bz.B > 1 ? cip_tl_bz.B : 1); ++z) alpha_host[z] = __builtin_inf(); return 0; }
    if (!alpha_host && !(cip_tl_bz.B > 1 && defer_slot >= 0)) { cip_set_error("cip_cones_maxstep: no destination for the minimum"); return CIP_E_INVALID; }
    cip_launch_b(k_maxstep, dim3(cs.nitems), dim3(256), 0, s, cs.d_cones, cs.d_items, x, d, scale, cs.d_partial);
    CIP_HIP_CHECK(hipGetLastError());
    if (cs.has_S) { int rc = cip_sdp_maxstep(s, cs, x, d, scale, cs.d_partial); if (rc) return rc; }
    const CipBatchCtx &bc = cip_tl_bz;
    CipHostScratch hs;
    int rc;
    if ((rc = cip_host_scratch(&hs))) return rc;
    const bool defer = bc.B > 1 && defer_slot >= 0;
    cip_launch_b(k_min_reduce, dim3(1), dim3(256), 0, s, (const double *)cs.d_partial, cs.nslots, bc.B > 1 ? cs.d_scalar : hs.dev,
                 bc.B > 1 ? bc.gather_dev : (double *)nullptr, defer ? defer_slot : 0);
    CIP_HIP_CHECK(hipGetLastError());
    if (defer) return 0;
    if (bc.B > 1) {                          // alpha_host: B values
        CIP_HIP_CHECK(hipMemcpyAsync(bc.gather_host, bc.gather_dev, sizeof(double) * bc.B * CIP_GATHER, hipMemcpyDeviceToHost, s));
        if ((rc = cip_wait(s))) return rc;
        for (int z = 0; z < bc.B; ++z) alpha_host[z] = bc.gather_host[z * CIP_GATHER];
        return 0;
    }
    if ((rc = cip_wait(s))) return rc;       // the minimum went straight into the host-mapped scratch
    alpha_host[0] = hs.host[0];
    return 0;
}
// defer_slot >= 0 (one problem only; round 5): the two minima go to slots defer_slot, defer_slot + 1 of the calling thread's
// host-mapped scratch (cip_host_scratch) and the call returns WITHOUT waiting -- the native loop reads them behind the wait of the
// dot products it enqueues next (one host round trip instead of two); alpha_host2 is not touched
int cip_cones_maxstep2(hipStream_t s, const ConeSet &cs, const double *x1, const double *d1, const double *x2, const double *d2,
                       double scale, double *alpha_host2, int defer_slot) {
    if (defer_slot >= 0 && (cs.nitems == 0 || cip_in_batch())) { cip_set_error("cip_cones_maxstep2: deferred form needs cones and one problem"); return CIP_E_INVALID; }
    if (cs.nitems == 0 || cip_in_batch()) {                        // (a lock-step group gathers per problem: two plain calls)
        const int B = cip_tl_bz.B > 1 ? cip_tl_bz.B : 1;
        int rc = cip_cones_maxstep(s, cs, x1, d1, scale, alpha_host2);
        if (rc) return rc;
        return cip_cones_maxstep(s, cs, x2, d2, scale, alpha_host2 + B);
    }
    double *p1 = cs.d_partial, *p2 = cs.d_partial + cs.nslots + 1;
    cip_launch_b(k_maxstep, dim3(cs.nitems), dim3(256), 0, s, cs.d_cones, cs.d_items, x1, d1, scale, p1);
    cip_launch_b(k_maxstep, dim3(cs.nitems), dim3(256), 0, s, cs.d_cones, cs.d_items, x2, d2, scale, p2);
    CIP_HIP_CHECK(hipGetLastError());
    if (cs.has_S) { const int rc = cip_sdp_maxstep2(s, cs, x1, d1, p1, x2, d2, p2, scale); if (rc) return rc; }
    CipHostScratch hs;
    int rc;
    if ((rc = cip_host_scratch(&hs))) return rc;
    const int so = defer_slot >= 0 ? defer_slot : 0;
    cip_launch_b(k_min_reduce, dim3(1), dim3(256), 0, s, (const double *)p1, cs.nslots, hs.dev + so, (double *)nullptr, 0);
    cip_launch_b(k_min_reduce, dim3(1), dim3(256), 0, s, (const double *)p2, cs.nslots, hs.dev + so + 1, (double *)nullptr, 0);
    CIP_HIP_CHECK(hipGetLastError());
    if (defer_slot >= 0) return 0;
    if ((rc = cip_wait(s))) return rc;       // both minima went straight into the host-mapped scratch
    alpha_host2[0] = hs.host[0]; alpha_host2[1] = hs.host[1];
    return 0;
}
int cip_cones_identity(hipStream_t s, const ConeSet &cs, double *e) {
    if (cs.nitems == 0) return 0;
    cip_launch_b(k_cone_identity, dim3(cs.nitems), dim3(256), 0, s, cs.d_cones, cs.d_items, e);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_cones_scale_At(hipStream_t s, const ConeSet &cs, int n, const double *At, long ldat, double *Wt, long ldwt) {
    if (cs.nitems == 0 || n == 0) return 0;
    if (cs.npackq > 0)
        cip_launch_b(k_scale_At, dim3((n + 255) / 256, cs.npackq), dim3(256), 0, s, cs.d_cones, cs.d_items, (const int *)cs.d_packq, cs.d_scal, n, At, ldat, Wt, ldwt);
    if (cs.nritems > 0)
        cip_launch_b(k_scale_At_r, dim3(((n + 255) / 256) * SCALE_AT_SPLIT, cs.nritems), dim3(256), 0, s, cs.d_cones, cs.d_items,
                           (const int *)cs.d_ritems, cs.d_scal, n, At, ldat, Wt, ldwt);
    if (cs.nbigq > 0)
        cip_launch_b(k_scale_At_qbig, dim3((n + 15) / 16, cs.nbigq), dim3(256), 0, s, cs.d_cones, cs.d_items, (const int *)cs.d_bigq,
                           cs.d_scal, n, At, ldat, Wt, ldwt);
    CIP_HIP_CHECK(hipGetLastError());
    if (cs.has_S) return cip_sdp_scale_At(s, cs, n, At, ldat, Wt, ldwt);
    return 0;
}
