/*
 * nonlin_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded, bug-compatible restatement of the hot path of
 * jchristopherson/nonlin v2.2.0 (Fortran):
 *     vecfcn_helper%jacobian   src/nonlin_multi_eqn_mult_var.f90:198-277
 *     least_squares_solver     src/nonlin_least_squares.f90:118-791
 *     newton_solver            src/nonlin_solve.f90:452-638
 *     line_search (mimo)       src/nonlin_linesearch.f90:152-326, 495-572
 *     test_convergence         src/nonlin_helper.f90:36-124
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or load this.  The product (nonlin_amd/, include/) never does.
 *
 * PINNING STATUS: pinned against the reference's own known answers for this
 * path (README.md:165-171 ten-digit LM coefficients; tests/nonlin_test_solve.f90
 * test_least_squares_1/2/4, test_newton_1..4; tests/nonlin_test_jacobian.f90;
 * examples/nonlin_newton_solve_jacobian.f90) and against the reference outputs
 * recorded in SURVEY.md section 6 / BASELINE.md section 2 (hex x of the 21x4 fit,
 * iteration/eval counts of six runs).  The reference itself cannot be built
 * here (every module needs the un-vendored `linalg`/`linalg_errors`), so there
 * is no oracle/_ref.  The LU inside newton_solver lives in that un-vendored
 * dependency (jchristopherson/linalg -> LAPACK dgetrf/dgetrs, unpinned):
 * nlo_lu_factor/nlo_lu_solve restate the published partial-pivoting algorithm
 * and are PARITY-UNPINNED at the bit level.
 *
 * Arithmetic conventions (frozen): IEEE binary64, no FMA contraction
 * (-ffp-contract=off), dot_product = left-to-right sequential sum,
 * norm2(x) = the flang runtime's scaled algorithm (amdflang 22 / ROCm 7.2), checked
 * bit-for-bit against amdflang's NORM2 (tests/golden/norm2_flang.json).
 */
#ifndef NONLIN_ORACLE_H
#define NONLIN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* NL_* codes: src/nonlin_error_handling.f90:10-38.  The three LA_* aliases come
 * from the un-vendored linalg_errors module (105/104/106 in linalg 1.x/2.x). */
#define NLO_NO_ERROR                     0
#define NLO_INVALID_INPUT_ERROR        201
#define NLO_ARRAY_SIZE_ERROR           202
#define NLO_OUT_OF_MEMORY_ERROR        105
#define NLO_INVALID_OPERATION_ERROR    104
#define NLO_CONVERGENCE_ERROR          106
#define NLO_DIVERGENT_BEHAVIOR_ERROR   206
#define NLO_SPURIOUS_CONVERGENCE_ERROR 207
#define NLO_TOLERANCE_TOO_SMALL_ERROR  208
#define NLO_UNDEFINED_FUNCTION_ERROR   211
#define NLO_UNDERDEFINED_PROBLEM_ERROR 212

/* vecfcn / jacobianfcn (src/nonlin_multi_eqn_mult_var.f90:14-38) flattened to C:
 * f has m entries; jac is column-major m-by-n with leading dimension m. */
typedef void (*nlo_vecfcn)(void *ctx, int32_t n, const double *x, int32_t m, double *f);
typedef void (*nlo_jacfcn)(void *ctx, int32_t n, const double *x, int32_t m, double *jac);

/* iteration_behavior: src/nonlin_types.f90:8-29 (logicals as int32 0/1). */
typedef struct {
    int32_t iter_count, fcn_count, jacobian_count, gradient_count;
    int32_t converge_on_fcn, converge_on_chng, converge_on_zero_diff;
} nlo_iteration_behavior;

/* equation_solver + least_squares_solver + line_search_solver + line_search
 * configuration (defaults: src/nonlin_multi_eqn_mult_var.f90:69-77,
 * src/nonlin_least_squares.f90:25, src/nonlin_solve.f90:30,
 * src/nonlin_linesearch.f90:35-53). */
typedef struct {
    int32_t max_evals;       /* 100   */
    double  ftol;            /* 1e-8  */
    double  xtol;            /* 1e-12 */
    double  gtol;            /* 1e-12 */
    int32_t print_status;    /* 0     */
    double  factor;          /* 100, clamped to [0.1,100] by the setter */
    int32_t use_line_search; /* 1     */
    int32_t ls_max_evals;    /* 100   */
    double  ls_alpha;        /* 1e-4  */
    double  ls_factor;       /* 0.1   */
} nlo_options;

void nlo_default_options(nlo_options *o);

/* Optional per-evaluation trace: every x the solver hands to fcn, in order. */
typedef struct {
    int32_t capacity;   /* number of x-vectors the buffer can hold */
    int32_t count;      /* number recorded (may exceed capacity; extra dropped) */
    double *xs;         /* capacity * n doubles */
} nlo_trace;

double nlo_norm2(int32_t n, const double *x);
/* NORM2 is a processor-dependent intrinsic: 0 = the flang runtime's algorithm (default: the
 * reference as built here), 1 = sqrt(sequential sum of squares).  Mode 1 exists only to show
 * how far "the reference" moves between Fortran processors (tests/test_oracle.py). */
void nlo_set_norm2_mode(int mode);
/* test-only: 1 = MINPACK's lmpar lines at :531 / :552 instead of the reference's (see nonlin_oracle.c); loop-entry counter */
void nlo_set_lmpar_minpack(int on);
long nlo_lmpar_loop_entries(int reset);
double nlo_dot(int32_t n, const double *x, const double *y);

/* vfh_jac_fcn (src/nonlin_multi_eqn_mult_var.f90:198-277).  fv may be NULL. */
int nlo_fd_jacobian(nlo_vecfcn fcn, nlo_jacfcn jac_or_null, void *ctx,
                    int32_t m, int32_t n, double *x, const double *fv, double *jac);

/* lmfactor / lmsolve / lmpar (src/nonlin_least_squares.f90:569-667, 670-791, 394-566). */
void nlo_lmfactor(int32_t m, int32_t n, double *a, int32_t lda, int32_t pivot,
                  int32_t *ipvt, double *rdiag, double *acnorm, double *wa);
void nlo_lmsolve(int32_t n, double *r, int32_t ldr, const int32_t *ipvt,
                 const double *diag, const double *qtb, double *x, double *sdiag,
                 double *wa);
void nlo_lmpar(int32_t m, int32_t n, double *r, int32_t ldr, const int32_t *ipvt,
               const double *diag, const double *qtb, double delta, double *par,
               double *x, double *sdiag, double *wa1, double *wa2 /* length m */);

/* lss_solve (src/nonlin_least_squares.f90:118-391).  Returns 0 or the code the
 * reference would `error stop` with; ib is always filled. */
int nlo_lm_solve(const nlo_options *opt, nlo_vecfcn fcn, nlo_jacfcn jac_or_null,
                 void *ctx, int32_t m, int32_t n, double *x, double *fvec,
                 nlo_iteration_behavior *ib);

/* TEST-ONLY: a caller-supplied factorisation in place of lmfactor + Q^T f inside nlo_lm_solve (NULL restores the
 * reference's).  See nonlin_oracle.c; used by tests/golden/make_fast_policy_study.py and nothing else. */
typedef void (*nlo_factor_hook)(void *hctx, int32_t m, int32_t n, double *jac, int32_t lda, const double *fvec,
                                int32_t *jpvt, double *rdiag, double *acnorm, double *qtf, double *wa4);
void nlo_set_factor_hook(nlo_factor_hook hook, void *hctx);

/* Partial-pivot LU stand-in for linalg's lu_factor/solve_lu
 * (call sites src/nonlin_solve.f90:570,577).  PARITY-UNPINNED. */
int  nlo_lu_factor(int32_t n, double *a, int32_t lda, int32_t *ipvt);
void nlo_lu_solve(int32_t n, const double *lu, int32_t lda, const int32_t *ipvt, double *b);

/* Line search pieces (src/nonlin_linesearch.f90:495-551, 554-572, 152-326). */
double nlo_min_backtrack_search(int32_t mode, double f0, double f, double f1,
                                double alam, double alam1, double slope);
void nlo_limit_search_vector(int32_t n, double *x, double lim);
int nlo_line_search(const nlo_options *opt, nlo_vecfcn fcn, void *ctx, int32_t m,
                    int32_t n, const double *xold, const double *grad,
                    const double *dir, double *x, double *fvec, double fold,
                    double *fx, nlo_iteration_behavior *ib);

/* test_convergence (src/nonlin_helper.f90:36-124). */
void nlo_test_convergence(int32_t n, int32_t m, const double *x, const double *xo,
                          const double *f, const double *g, int32_t lg, double xtol,
                          double ftol, double gtol, int32_t *c, int32_t *cx,
                          int32_t *cf, int32_t *cg, double *xnorm, double *fnorm);

/* ns_solve (src/nonlin_solve.f90:452-638). */
int nlo_newton_solve(const nlo_options *opt, nlo_vecfcn fcn, nlo_jacfcn jac_or_null,
                     void *ctx, int32_t n, double *x, double *fvec,
                     nlo_iteration_behavior *ib);

/* Dense kernels behind qns_solve (third-party linalg in the reference; restated, see .c). */
void nlo_givens(double f, double g, double *c, double *s, double *r);
void nlo_qr_factor_full(int32_t n, const double *a, double *q, double *r);
void nlo_qr_rank1_update(int32_t n, double *q, double *r, double *u, const double *v);
void nlo_solve_upper(int32_t n, const double *r, double *x);

/* qns_solve (src/nonlin_solve.f90:156-427).  jdelta = quasi_newton_solver%m_jDelta (default 5). */
int nlo_quasi_newton_solve(const nlo_options *opt, int32_t jdelta, nlo_vecfcn fcn,
                           nlo_jacfcn jac_or_null, void *ctx, int32_t n, double *x,
                           double *fvec, nlo_iteration_behavior *ib);

/* constrained_least_squares_solver (src/nonlin_least_squares.f90:33-74, 793-1403). */
void nlo_qr_factor_rhs(int32_t m, int32_t n, double *a, double *f);
double nlo_alpha_box(int32_t n, const double *x, const double *p, const double *xl, const double *xu);
void nlo_coleman_li_scaling(int32_t n, const double *x, const double *xl, const double *xu, double *s);
void nlo_dogleg(int32_t m, int32_t n, double delta, const double *x, const double *f, const double *jac,
                const double *r, const double *qtf, const double *s, const double *xl, const double *xu,
                double *p, double *g, double *Jp, double *prered, double *work);
int nlo_cls_solve(const nlo_options *opt, double delta0, double stepscale0, const double *xl,
                  const double *xu, nlo_vecfcn fcn, nlo_jacfcn jac_or_null, void *ctx, int32_t m,
                  int32_t n, double *x, double *fvec, nlo_iteration_behavior *ib);

/* polynomial%fit / fit_thru_zero / evaluate (src/nonlin_polynomials.f90:146-268). */
int nlo_poly_fit(int32_t npts, int32_t order, const double *x, const double *y, int32_t thru_zero, double *coef);
double nlo_poly_eval(int32_t order, const double *c, double x);

/* fcnnvar_helper%gradient and bfgs%solve (src/nonlin_multi_var.f90:182-246, src/nonlin_optimize.f90:557-770). */
typedef double (*nlo_fcnnvar)(void *ctx, int32_t n, const double *x);
typedef void (*nlo_gradfcn)(void *ctx, int32_t n, const double *x, double *g);
void nlo_fd_gradient(nlo_fcnnvar fcn, nlo_gradfcn grad_or_null, void *ctx, int32_t n, double *x,
                     const double *fv_or_null, double *g);
void nlo_rtr(int32_t n, const double *r, double *b);
void nlo_symv(int32_t n, const double *b, const double *x, double *y);
void nlo_chol_update(int32_t n, double *r, double *u);
int nlo_chol_downdate(int32_t n, double *r, double *u);
int nlo_chol_factor_upper(int32_t n, const double *b, double *r);
void nlo_solve_cholesky_upper(int32_t n, const double *r, double *x);
int nlo_bfgs_solve(const nlo_options *opt, nlo_fcnnvar fcn, nlo_gradfcn grad_or_null, void *ctx, int32_t n,
                   double *x, double *fout, nlo_iteration_behavior *ib);

/* ---- Synthetic "dense-quadratic" residual family ----
 * u_i = sum_j A(i,j) x_j (j ascending, one multiply + one add per term, no FMA)
 * r_i = (u_i + gamma*u_i*u_i) - b_i ;  J(i,j) = (1 + 2*gamma*u_i) * A(i,j).   */
typedef struct {
    int32_t m, n;
    const double *A;   /* column-major m-by-n */
    const double *b;   /* m */
    double gamma;
    int64_t ncalls;    /* incremented by nlo_dq_fcn */
    nlo_trace *trace;  /* optional */
} nlo_dq_problem;

void nlo_dq_fcn(void *ctx /* nlo_dq_problem* */, int32_t n, const double *x, int32_t m, double *f);
void nlo_dq_jac(void *ctx /* nlo_dq_problem* */, int32_t n, const double *x, int32_t m, double *jac);

/* splitmix64 generator, draw order exactly as SURVEY.md section 8(d):
 * A (column-major, (2U-1)/sqrt(n); square => A += 2I), x_true, b = model(x_true)
 * then b_i += sigma(2U-1), then x0 = x_true + s(2U-1). */
void nlo_dq_generate(uint64_t seed, int32_t m, int32_t n, double gamma, double sigma,
                     double spread, int32_t square_shift, double *A, double *b,
                     double *x_true, double *x0);

/* Convenience drivers used by tests / the CPU baseline: generate + solve. */
int nlo_dq_lm_solve(const nlo_options *opt, const nlo_dq_problem *p, double *x,
                    double *fvec, nlo_iteration_behavior *ib);
int nlo_dq_newton_solve(const nlo_options *opt, const nlo_dq_problem *p, int32_t analytic,
                        double *x, double *fvec, nlo_iteration_behavior *ib);
int nlo_dq_quasi_newton_solve(const nlo_options *opt, int32_t jdelta, const nlo_dq_problem *p,
                              int32_t analytic, double *x, double *fvec, nlo_iteration_behavior *ib);
int nlo_dq_cls_solve(const nlo_options *opt, double delta0, double stepscale0, const double *xl, const double *xu,
                     const nlo_dq_problem *p, double *x, double *fvec, nlo_iteration_behavior *ib);
int nlo_dq_bfgs_solve(const nlo_options *opt, const nlo_dq_problem *p, double *x, double *fout,
                      nlo_iteration_behavior *ib);

#ifdef __cplusplus
}
#endif
#endif
