/* dq_callback.c -- a COMPILED host residual, as a user of nonlin's vecfcn interface would supply it
 * (src/nonlin_multi_eqn_mult_var.f90:14-25: fcn(x, f, args)), behind the C callback signature of include/nonlin_hip.h
 * (nlh_vecfcn).  Used by bench.py's `mode_h` rows and tests/test_gpu_host_callback.py: the literal drop-in case -- the
 * solver runs on the GPU, the user's function stays a host procedure and is called n + 1 times per Jacobian, in the
 * reference's order, on the calling thread.
 *
 * The function is the dense-quadratic family of SURVEY.md 8(d): r_i = (u_i + gamma u_i u_i) - b_i, u_i = sum_j A(i,j) x_j
 * accumulated in ascending j, one multiply and one add per term (build with -ffp-contract=off).
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -o libdq_callback.so dq_callback.c
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    int32_t m, n;
    const double *A;        /* column-major m x n */
    const double *b;
    double gamma;
    int64_t ncalls;         /* how often the solver called back (the reference counts only fcn_count of these) */
    double *u;              /* m doubles of scratch owned by the caller */
} dq_user_ctx;

void dq_user_fcn(void *ctx, int32_t n, const double *x, int32_t m, double *f)
{
    dq_user_ctx *c = (dq_user_ctx *)ctx;
    double *u = c->u;
    c->ncalls += 1;
    for (int32_t i = 0; i < m; ++i) u[i] = 0.0;
    for (int32_t j = 0; j < n; ++j) {                 /* every u_i receives its terms in ascending j */
        const double xj = x[j];
        const double *col = c->A + (size_t)j * (size_t)m;
        for (int32_t i = 0; i < m; ++i) u[i] = u[i] + col[i] * xj;
    }
    for (int32_t i = 0; i < m; ++i) f[i] = (u[i] + c->gamma * u[i] * u[i]) - c->b[i];
}

/* The analytic Jacobian of the same family as a compiled jacobianfcn (src/nonlin_multi_eqn_mult_var.f90:27-38; C shape
 * nlh_jacfcn): jac(i,j) = (1 + 2 gamma u_i) A(i,j), column-major with leading dimension m -- BASELINE config 3's
 * "analytic Jacobian callback" taken literally (newton_solver, n = 1024: 8 MB per call, host to device every iteration). */
void dq_user_jac(void *ctx, int32_t n, const double *x, int32_t m, double *jac)
{
    dq_user_ctx *c = (dq_user_ctx *)ctx;
    double *u = c->u;
    for (int32_t i = 0; i < m; ++i) u[i] = 0.0;
    for (int32_t j = 0; j < n; ++j) {
        const double xj = x[j];
        const double *col = c->A + (size_t)j * (size_t)m;
        for (int32_t i = 0; i < m; ++i) u[i] = u[i] + col[i] * xj;
    }
    for (int32_t i = 0; i < m; ++i) u[i] = 1.0 + 2.0 * c->gamma * u[i];
    for (int32_t j = 0; j < n; ++j) {
        const double *col = c->A + (size_t)j * (size_t)m;
        double *out = jac + (size_t)j * (size_t)m;
        for (int32_t i = 0; i < m; ++i) out[i] = u[i] * col[i];
    }
}
