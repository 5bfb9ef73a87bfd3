// TEST INFRASTRUCTURE: drives the product's host control plane (libcipkkt built host-only under ASan + UBSan or TSan, linked against
// tests/hostsan/fake_hip.cpp) through the C ABI of include/cipkkt.h:
//   1. the plugin levels on one handle (create, identity scaling, factor, check, host-pointer solves, packed scaling, destroy), dense
//      and CSR A, p = 0 and p > 0, both routes;
//   2. the native loop (cip_conicip) on a handle, with a trace buffer;
//   3. cip_conicip_mixed on a batch built to hit every branch of the binning: 65 problems of one shape (a lock-step group of 64 and a
//      group of ONE -- round 4's NULL write lived there), CSR problems of equal shape but different nnz, problems with p > 0, a
//      problem with a chip-wide S cone (lock-step refuses it: thread pool), and a batch holding an S cone beyond the envelope
//      (refused at level 1: the call must fail cleanly and free everything);
//   4. the same entry points from several caller threads at once (thread-local batch contexts, the cached arena, the pool);
//   5. the stand-alone LDL' entry points with a caller-owned workspace in both solve modes.
// At the end every "device" allocation must have been returned (the fake runtime counts them).
#include "cipkkt.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" void fake_hip_stats(long *launches, long *emulated, long *live_bytes, long *live_allocs);

#define REQUIRE(cond) do { if (!(cond)) { fprintf(stderr, "drive: %s:%d: %s failed (last error: %s)\n", __FILE__, __LINE__, #cond, cip_last_error()); exit(2); } } while (0)

struct Prob {
    int n, m, p;
    std::vector<int> ctype, cdim;
    std::vector<double> Q, A, G, c, b, d, Av;
    std::vector<int> rp, ci;
    bool csr = false;
    int route = CIP_ROUTE_SCHUR;
    cip_problem desc() const {
        cip_problem pr;
        memset(&pr, 0, sizeof(pr));
        pr.n = n; pr.m = m; pr.p = p; pr.ncones = (int)ctype.size();
        pr.cone_type = ctype.data(); pr.cone_dim = cdim.data();
        pr.Q = Q.data(); pr.ldq = n;
        if (csr) { pr.A = nullptr; pr.A_rowptr = rp.data(); pr.A_colind = ci.data(); pr.A_val = Av.data(); }
        else { pr.A = A.data(); pr.lda = m; }
        pr.G = p > 0 ? G.data() : nullptr; pr.ldg = p > 0 ? p : 1;
        pr.route = route; pr.flags = 0;
        return pr;
    }
};
static thread_local unsigned long long g_rng = 88172645463325252ull;
static double rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (double)(g_rng % 20001) / 10000.0 - 1.0; }

// n variables, cones given as (type, dim) pairs; density < 1 -> CSR with roughly that share of entries (at least the diagonal)
static Prob make(int n, int p, std::vector<std::pair<int, int>> cones, double density, int route = CIP_ROUTE_SCHUR) {
    Prob P;
    P.n = n; P.p = p; P.route = route;
    P.m = 0;
    for (auto &c : cones) { P.ctype.push_back(c.first); P.cdim.push_back(c.second); P.m += c.second; }
    P.Q.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) P.Q[i + (size_t)i * n] = 2.0 + 0.1 * i;
    P.c.resize(n); for (auto &x : P.c) x = rnd();
    P.b.assign(P.m, -1.0);
    P.d.assign(p, 0.0);
    P.G.resize((size_t)p * n); for (auto &x : P.G) x = rnd();
    if (density >= 1.0) {
        P.A.resize((size_t)P.m * n); for (auto &x : P.A) x = 0.3 * rnd();
    } else {
        P.csr = true;
        P.rp.push_back(0);
        for (int i = 0; i < P.m; ++i) {
            for (int j = 0; j < n; ++j)
                if (j == i % n || (rnd() + 1.0) * 0.5 < density) { P.ci.push_back(j); P.Av.push_back(0.3 * rnd() + (j == i % n ? 1.0 : 0.0)); }
            P.rp.push_back((int)P.ci.size());
        }
    }
    return P;
}

static cip_options opts() {
    cip_options o;
    o.optTol = 1e-6; o.DTB = 0.01; o.infeasTol = -1.0; o.refinementThreshold = -1.0;
    o.maxRefinementSteps = 3; o.maxIters = 100; o.verbose = 0;
    return o;
}

static void plugin_levels(const Prob &P) {
    cip_problem pr = P.desc();
    cip_handle *h = nullptr;
    REQUIRE(cip_create_ex(&pr, &h) == CIP_OK && h);
    int N = 0, Np = 0;
    REQUIRE(cip_kkt_order(h, &N, &Np) == CIP_OK && N > 0 && Np >= N && Np % 128 == 0);
    REQUIRE(cip_set_scaling_identity(h) == CIP_OK);
    REQUIRE(cip_factor(h) == CIP_OK);
    REQUIRE(cip_check_factor(h) == CIP_OK);
    std::vector<double> x(P.n, 1.0), y(P.p > 0 ? P.p : 1, 0.5), z(P.m, 0.25), dx(P.n), dy(P.p > 0 ? P.p : 1), dz(P.m);
    REQUIRE(cip_solve3x3(h, x.data(), y.data(), z.data(), dx.data(), dy.data(), dz.data()) == CIP_OK);
    if (P.route == CIP_ROUTE_SCHUR) REQUIRE(cip_solve2x2(h, x.data(), y.data(), dx.data(), dy.data()) == CIP_OK);
    else REQUIRE(cip_solve2x2(h, x.data(), y.data(), dx.data(), dy.data()) == CIP_E_UNSUPPORTED);
    const size_t len = cip_scaling_packed_len(h);
    std::vector<double> F(len ? len : 1);
    REQUIRE(cip_get_scaling_packed(h, F.data()) == CIP_OK);
    REQUIRE(cip_set_scaling_packed(h, F.data()) == CIP_OK);
    REQUIRE(cip_set_timing(h, 1) == CIP_OK);
    REQUIRE(cip_factor(h) == CIP_OK);
    double st[8];
    REQUIRE(cip_stats(h, st) == CIP_OK);
    REQUIRE(cip_set_timing(h, 0) == CIP_OK);
    std::vector<double> K((size_t)Np * Np);
    REQUIRE(cip_get_kkt_matrix(h, K.data()) == CIP_OK);
    REQUIRE(cip_update_problem(h, &pr) == CIP_OK);
    // the native loop on the same handle, with a trace
    cip_options o = opts();
    cip_result res;
    std::vector<double> yy(P.n), ww(P.p > 0 ? P.p : 1), vv(P.m), trace(CIP_TRACE_COLS * 8);
    REQUIRE(cip_conicip(h, P.c.data(), P.b.data(), P.p > 0 ? P.d.data() : nullptr, &o, yy.data(), ww.data(), vv.data(), &res, trace.data(), 8) == CIP_OK);
    REQUIRE(res.status >= CIP_STATUS_OPTIMAL && res.status <= CIP_STATUS_ERROR && res.n_factor >= 1);
    REQUIRE(cip_destroy(h) == CIP_OK);
}

struct Batch {
    std::vector<Prob> P;
    std::vector<cip_problem> desc;
    std::vector<const double *> c, b, d;
    std::vector<std::vector<double>> y, w, v;
    std::vector<double *> yp, wp, vp;
    std::vector<cip_result> res;
    void finish() {
        const size_t k = P.size();
        desc.resize(k); c.resize(k); b.resize(k); d.resize(k); y.resize(k); w.resize(k); v.resize(k); yp.resize(k); wp.resize(k); vp.resize(k); res.resize(k);
        for (size_t i = 0; i < k; ++i) {
            desc[i] = P[i].desc(); c[i] = P[i].c.data(); b[i] = P[i].b.data(); d[i] = P[i].p > 0 ? P[i].d.data() : nullptr;
            y[i].assign(P[i].n, 777.0); w[i].assign(P[i].p > 0 ? P[i].p : 1, 777.0); v[i].assign(P[i].m, 777.0);
            yp[i] = y[i].data(); wp[i] = w[i].data(); vp[i] = v[i].data();
            memset(&res[i], 0, sizeof(cip_result));
        }
    }
};

static Batch mixed_batch(int same_shape, bool with_large_s) {
    Batch B;
    for (int i = 0; i < same_shape; ++i) B.P.push_back(make(8, 0, {{CIP_CONE_R, 8}}, 1.0));                 // 65 -> groups of 64 and 1
    for (int i = 0; i < 3; ++i) B.P.push_back(make(10, 0, {{CIP_CONE_R, 6}, {CIP_CONE_Q, 4}}, 0.2 + 0.25 * i));   // equal shapes, different nnz
    for (int i = 0; i < 2; ++i) B.P.push_back(make(12, 3, {{CIP_CONE_R, 5}, {CIP_CONE_Q, 5}, {CIP_CONE_S, 6}}, 1.0));   // p > 0, a small S cone
    B.P.push_back(make(9, 0, {{CIP_CONE_R, 9}}, 1.0, CIP_ROUTE_FULL3X3));                                    // a bin of one: thread pool
    if (with_large_s) B.P.push_back(make(4, 0, {{CIP_CONE_S, 133 * 134 / 2}}, 1.0));                          // chip-wide S cone: lock-step refuses it
    B.finish();
    return B;
}

static void run_mixed(int same_shape, bool with_large_s, int in_flight) {
    Batch B = mixed_batch(same_shape, with_large_s);
    cip_options o = opts();
    const int k = (int)B.P.size();
    REQUIRE(cip_conicip_mixed(k, B.desc.data(), B.c.data(), B.b.data(), B.d.data(), &o, B.yp.data(), B.wp.data(), B.vp.data(), B.res.data(), in_flight) == CIP_OK);
    for (int i = 0; i < k; ++i) {
        REQUIRE(B.res[i].status >= CIP_STATUS_OPTIMAL && B.res[i].status <= CIP_STATUS_ERROR);
        REQUIRE(B.y[i][0] != 777.0);                               // every problem's solution was written
    }
    int st[3];
    REQUIRE(cip_lockstep_stats(st) == CIP_OK && st[1] >= same_shape);
    // the same problems through the two other batch entry points
    REQUIRE(cip_conicip_problems(k, B.desc.data(), B.c.data(), B.b.data(), B.d.data(), &o, B.yp.data(), B.wp.data(), B.vp.data(), B.res.data(), in_flight) == CIP_OK);
    cip_batch *bt = nullptr;
    REQUIRE(cip_batch_create(k, B.desc.data(), &bt) == CIP_OK && cip_batch_size(bt) == k);
    REQUIRE(cip_batch_conicip(bt, B.c.data(), B.b.data(), B.d.data(), &o, B.yp.data(), B.wp.data(), B.vp.data(), B.res.data(), in_flight) == CIP_OK);
    REQUIRE(cip_batch_destroy(bt) == CIP_OK);
}

static void refused_batch(void) {
    // an S cone of matrix order 2049 (vectorised length 2049 * 2050 / 2): refused at level 1 from the cone table alone -- A is never read
    Batch B;
    B.P.push_back(make(8, 0, {{CIP_CONE_R, 8}}, 1.0));
    B.P.push_back(make(8, 0, {{CIP_CONE_R, 8}}, 1.0));
    B.finish();
    Prob big;
    big.n = 2; big.p = 0; big.m = 2049 * 2050 / 2;
    big.ctype = {CIP_CONE_S}; big.cdim = {big.m};
    big.Q = {1.0, 0.0, 0.0, 1.0}; big.A.assign(4, 0.0); big.c = {1.0, 1.0}; big.b.assign(4, 0.0);
    cip_problem pd = big.desc();
    cip_handle *h = nullptr;
    REQUIRE(cip_create_ex(&pd, &h) == CIP_E_UNSUPPORTED && h == nullptr);
    std::vector<cip_problem> desc = B.desc;
    desc.push_back(pd);
    std::vector<const double *> c = B.c, b = B.b, d = B.d;
    c.push_back(big.c.data()); b.push_back(big.b.data()); d.push_back(nullptr);
    std::vector<double> y3(2, 777.0), w3(1, 777.0), v3(4, 777.0);
    std::vector<double *> yp = B.yp, wp = B.wp, vp = B.vp;
    yp.push_back(y3.data()); wp.push_back(w3.data()); vp.push_back(v3.data());
    std::vector<cip_result> res(3);
    cip_options o = opts();
    const int rc = cip_conicip_mixed(3, desc.data(), c.data(), b.data(), d.data(), &o, yp.data(), wp.data(), vp.data(), res.data(), 2);
    REQUIRE(rc != CIP_OK);                                         // the refusal surfaces; nothing leaks (checked at the end)
    REQUIRE(y3[0] == 777.0);
}

static void standalone_ldlt(void) {
    const int N = 384;
    size_t bytes = 0;
    for (int mode = 0; mode <= 2; mode += 2) {
        const int prev = cip_set_solve_fused(mode);
        const int pb = cip_set_solve_block_max(128);
        REQUIRE(cip_ldlt_workspace_bytes(N, &bytes) == CIP_OK && bytes > 0);
        std::vector<double> K((size_t)N * N, 0.0), rhs(N, 1.0);
        for (int i = 0; i < N; ++i) K[i + (size_t)i * N] = 1.0;
        std::vector<char> ws(bytes);                               // caller-owned workspace of EXACTLY the advertised size
        int info = -1;
        REQUIRE(cip_ldlt_factor_dev(nullptr, K.data(), N, N, ws.data(), &info) == CIP_OK);
        REQUIRE(cip_ldlt_solve_dev(nullptr, K.data(), N, N, ws.data(), rhs.data()) == CIP_OK);
        cip_set_solve_block_max(pb);
        cip_set_solve_fused(prev);
    }
}

int main(int argc, char **argv) {
    const int nthreads = argc > 1 ? atoi(argv[1]) : 3;
    plugin_levels(make(8, 0, {{CIP_CONE_R, 8}}, 1.0));
    plugin_levels(make(10, 2, {{CIP_CONE_R, 4}, {CIP_CONE_Q, 6}}, 0.4));
    plugin_levels(make(12, 3, {{CIP_CONE_R, 5}, {CIP_CONE_Q, 5}, {CIP_CONE_S, 6}}, 1.0, CIP_ROUTE_FULL3X3));
    plugin_levels(make(4, 0, {{CIP_CONE_S, 133 * 134 / 2}}, 1.0));
    standalone_ldlt();
    run_mixed(65, true, 3);
    run_mixed(1, false, 1);
    refused_batch();
    // concurrent callers: every thread its own batches, all through the same process-wide state
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([t] {
            run_mixed(3 + 31 * t, t == 0, 2 + t);
            plugin_levels(make(8 + t, t % 2, {{CIP_CONE_R, 4}, {CIP_CONE_Q, 4 + t}}, t == 1 ? 0.5 : 1.0));
            REQUIRE(cip_release_cached_memory() == CIP_OK);
        });
    for (auto &x : th) x.join();
    REQUIRE(cip_release_cached_memory() == CIP_OK);
    long launches, emulated, live_bytes, live_allocs;
    fake_hip_stats(&launches, &emulated, &live_bytes, &live_allocs);
    printf("drive: %ld launches (%ld emulated), %ld device allocations / %ld bytes still live\n", launches, emulated, live_allocs, live_bytes);
    return 0;
}
