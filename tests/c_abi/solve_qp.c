/* The C ABI from plain C (no Python, no torch, no HIP headers): the three plugin levels of the reference
 * (src/ConicIP.jl:667, :682, :688) on a small box-constrained QP, then the whole interior-point loop
 * (cip_conicip).  Built and run by tests/test_c_abi_program.py:
 *     gcc -std=c99 -I include tests/c_abi/solve_qp.c -L conicip.jl_amd/cipkkt -lcipkkt -lm
 * Problem:  min 1/2 y'Qy - c'y  s.t.  y >= 0   (A = I, b = 0, K = R^n), Q = tridiag(-1, 4, -1).
 * Checks:   the KKT residuals of the level-3 solve, the optimality conditions of the final iterate.
 * With -DCIP_PLUGIN_LEVELS_ONLY the program stops after level 3: that build is linked against the CPU reference of
 * the ABI (oracle/cpu_ref, tests/test_cpu_ref.py), which implements the plugin levels only. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "cipkkt.h"

#define N 300
#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, cip_last_error()); return 2; } } while (0)

int main(void) {
    const int n = N, m = N, p = 0;
    double *Q = calloc((size_t)n * n, sizeof(double)), *A = calloc((size_t)m * n, sizeof(double));
    double c[N], b[N], F[N], x[N], z[N], a[N], cc[N], y[N], v[N];
    for (int i = 0; i < n; ++i) {
        Q[i + (size_t)i * n] = 4.0;
        if (i + 1 < n) { Q[i + 1 + (size_t)i * n] = -1.0; Q[i + (size_t)(i + 1) * n] = -1.0; }
        A[i + (size_t)i * n] = 1.0;
        c[i] = sin(0.37 * i) * 3.0;           /* mixed signs: some bounds active at the optimum */
        b[i] = 0.0;
        F[i] = 0.5 + 0.01 * i;                /* a diagonal NT scaling (R cone) */
        x[i] = cos(0.11 * i); z[i] = 1.0 / (1.0 + i);
    }
    int cone_type[1] = {CIP_CONE_R}, cone_dim[1] = {N};
    cip_handle *h = NULL;
    CHECK(cip_create(n, m, p, 1, cone_type, cone_dim, Q, A, NULL, CIP_ROUTE_SCHUR, &h));     /* level 1 */
    CHECK(cip_set_scaling_packed(h, F));                                                     /* level 2 */
    CHECK(cip_factor(h));
    CHECK(cip_check_factor(h));
    CHECK(cip_solve3x3(h, x, NULL, z, a, NULL, cc));                                         /* level 3 */
    /* residuals of  Q a - A' c = x ,  A a + F'F c = z */
    double r1 = 0, r2 = 0;
    for (int i = 0; i < n; ++i) {
        double qa = 4.0 * a[i] - (i > 0 ? a[i - 1] : 0.0) - (i + 1 < n ? a[i + 1] : 0.0);
        r1 = fmax(r1, fabs(qa - cc[i] - x[i]));
        r2 = fmax(r2, fabs(a[i] + F[i] * F[i] * cc[i] - z[i]));
    }
    printf("solve3x3 residuals %.3e %.3e\n", r1, r2);
    if (!(r1 < 1e-10 && r2 < 1e-10)) return 1;
#ifdef CIP_PLUGIN_LEVELS_ONLY
    (void)c; (void)b; (void)y; (void)v;
    CHECK(cip_destroy(h));
    free(Q); free(A);
    return 0;
#else
    cip_options opt = {1e-8, 0.01, -1.0, -1.0, 3, 100, 0};
    cip_result res;
    CHECK(cip_conicip(h, c, b, NULL, &opt, y, NULL, v, &res, NULL, 0));
    /* optimality: y >= 0, v >= 0, Qy - c - v = 0, y.v = 0 */
    double feas = 0, stat = 0, comp = 0;
    for (int i = 0; i < n; ++i) {
        double qy = 4.0 * y[i] - (i > 0 ? y[i - 1] : 0.0) - (i + 1 < n ? y[i + 1] : 0.0);
        feas = fmax(feas, fmax(-y[i], -v[i]));
        stat = fmax(stat, fabs(qy - c[i] - v[i]));
        comp = fmax(comp, fabs(y[i] * v[i]));
    }
    printf("conicip status %d iter %d  feas %.2e stat %.2e comp %.2e  (%d factorisations, %d solves, %.1f ms)\n",
           res.status, res.iter, feas, stat, comp, res.n_factor, res.n_solve, 1e3 * res.wall_s);
    CHECK(cip_destroy(h));
    free(Q); free(A);
    return (res.status == CIP_STATUS_OPTIMAL && feas < 1e-6 && stat < 1e-6 && comp < 1e-5) ? 0 : 1;
#endif
}
