/*
 * nonlin_hip.h -- C ABI of libnonlin_hip.so: the MI355X (gfx950) implementation of
 * nonlin's Jacobian-evaluate + linear-solve inner loop.
 *
 * This is the drop-in boundary.  The reference (jchristopherson/nonlin v2.2.0) is
 * pure Fortran with no C interface; each entry point below replaces one
 * type-bound procedure of the reference and is what a Fortran `bind(C)`
 * interface (nonlin_amd/fortran/) or any other FFI binds to.
 * Signatures use plain pointers, int32_t sizes and doubles only.
 *
 * Conventions
 *   - all matrices are column-major (Fortran order), fp64; integers are int32;
 *     Fortran LOGICALs cross the boundary as int32 0/1;
 *   - "host" entry points take HOST pointers and host callbacks, block until
 *     the result is back in the caller's arrays, and return 0 or the NL_* code
 *     the reference would `error stop` with (the Fortran shim performs the stop);
 *   - "dq" (device-model) entry points take DEVICE pointers (inputs already
 *     resident in HBM), run on the handle's HIP stream and are batched over
 *     independent problems;
 *   - nothing here falls back to a CPU implementation: without a GPU every
 *     compute entry point returns NLH_ERR_NO_DEVICE.
 */
#ifndef NONLIN_HIP_H
#define NONLIN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: src/nonlin_error_handling.f90:10-38 --------------------- */
#define NLH_NO_ERROR                     0
#define NLH_INVALID_INPUT_ERROR        201   /* NL_INVALID_INPUT_ERROR  (:12) */
#define NLH_ARRAY_SIZE_ERROR           202   /* NL_ARRAY_SIZE_ERROR     (:14) */
#define NLH_OUT_OF_MEMORY_ERROR        105   /* = LA_OUT_OF_MEMORY_ERROR (:16), linalg_errors un-vendored */
#define NLH_INVALID_OPERATION_ERROR    104   /* = LA_INVALID_OPERATION_ERROR (:18) */
#define NLH_CONVERGENCE_ERROR          106   /* = LA_CONVERGENCE_ERROR  (:21) */
#define NLH_DIVERGENT_BEHAVIOR_ERROR   206   /* (:23) */
#define NLH_SPURIOUS_CONVERGENCE_ERROR 207   /* (:25) */
#define NLH_TOLERANCE_TOO_SMALL_ERROR  208   /* (:27) */
#define NLH_UNDEFINED_FUNCTION_ERROR   211   /* (:34) */
#define NLH_UNDERDEFINED_PROBLEM_ERROR 212   /* (:37) */
/* Size limits of this implementation (the reference has none); an entry point given a larger problem returns
 * NLH_ARRAY_SIZE_ERROR and touches nothing: quasi-Newton and BFGS n <= 8192 (columns per thread of the single-workgroup
 * rotation kernels); least squares under the opt-in NLH_FACTOR_AUTO / NLH_FACTOR_QR policies n <= 3000 (n-vectors in LDS).
 * Least squares under the default NLH_FACTOR_EXACT policy takes any n <= m (beyond 3000 columns lmpar's n-vectors live in
 * global memory); that is verified at n = 3008 -- the solver's own kernels have no bound in n, but the BUILT-IN
 * dense-quadratic family (the bench / test residual of nlh_dq_*, not part of the reference) keeps a point's x in LDS and
 * stops at n = 20000 (NLH_ARRAY_SIZE_ERROR from its launcher); a user's device function has whatever bound its own kernels
 * have.  Row counts are not limited: polynomial fits and bounded least squares beyond 18000 rows keep the
 * Householder reflector in global memory instead of LDS.
 * The number of problems of a batch is NOT limited: the lock-step drivers carry the problem index in a grid dimension
 * that holds 65535, and a larger batch is solved in slices of 65535 problems, one after the other, inside the entry point
 * (independent problems: the same bits). */
/* library-level failures (not reference codes) */
#define NLH_ERR_NO_DEVICE             -1
#define NLH_ERR_HIP                   -2
#define NLH_ERR_BAD_HANDLE            -3

/* ---- iteration_behavior: src/nonlin_types.f90:8-29 ------------------------ */
typedef struct nlh_iteration_behavior {
    int32_t iter_count;
    int32_t fcn_count;
    int32_t jacobian_count;
    int32_t gradient_count;
    int32_t converge_on_fcn;        /* logical */
    int32_t converge_on_chng;       /* logical */
    int32_t converge_on_zero_diff;  /* logical */
} nlh_iteration_behavior;

/* ---- solver configuration -------------------------------------------------
 * equation_solver   src/nonlin_multi_eqn_mult_var.f90:67-91 (defaults :69-77)
 * least_squares_solver%m_factor   src/nonlin_least_squares.f90:25, clamp :108-114
 * line_search_solver%m_useLineSearch   src/nonlin_solve.f90:30
 * line_search   src/nonlin_linesearch.f90:35-53                              */
#define NLH_FACTOR_AUTO 0  /* J^T J + pivoted Cholesky; Householder QR when the Gauss-Newton
                              step is rejected or the Gram matrix is ill-conditioned.  Opt-in: within 1e-10 of the
                              reference with exact counts on zero-residual problems (tests/test_gpu_auto_policy.py),
                              at the forward-difference noise level (~1e-7) where a residual remains */
#define NLH_FACTOR_QR   1  /* always the reference's pivoted Householder QR (lmfactor), parallel reductions */
#define NLH_FACTOR_EXACT 2 /* lmfactor/lmpar with every reduction in the reference's operation order
                              (sequential dot products, flang NORM2): bit-identical to the CPU path */
typedef struct nlh_options {
    int32_t max_evals;        /* 100   */
    double  ftol;             /* 1e-8  */
    double  xtol;             /* 1e-12 */
    double  gtol;             /* 1e-12 */
    int32_t print_status;     /* 0     */
    double  factor;           /* 100; setters clamp to [0.1, 100] */
    int32_t use_line_search;  /* 1     */
    int32_t ls_max_evals;     /* 100   */
    double  ls_alpha;         /* 1e-4  */
    double  ls_factor;        /* 0.1   */
    int32_t factor_policy;    /* NLH_FACTOR_EXACT */
    double  ne_pivot_tol;     /* 1e-4: Cholesky pivot / column-norm^2 below this => QR */
    int32_t fuse_fd;          /* 1.  Device-model solves only: 1 = the kernel that evaluates the n perturbed
                                 residuals also forms jac(:,j) = (f_j - f0)/h_j (:274) in its epilogue and the
                                 residual panel is never written; same operations per element, same bits */
    int32_t sub_batches;      /* 0.  Batched device-model LM solves: number of sub-batches kept in flight on private
                                 streams (0 = automatic: nprob / 128, at most 3, and two halves for 32 to 255 problems of
                                 m n >= 65536 elements -- smaller problems have no latency-bound pass to hide and stay one
                                 batch; measurements in nlh_lm.hip, lm_sub_batches; 1 = one lock-step batch).  Results do
                                 not depend on it */
} nlh_options;

void nlh_default_options(nlh_options *opts);

/* print_status (src/nonlin_helper.f90:17-33) as text: the block the solvers print between outer iterations when
 * print_status is set -- a blank line, "Iteration: <I0>", "Function Evaluations: <I0>", "Jacobian Evaluations: <I0>"
 * (only when > 0), "Change in Variable: <E10.3>", "Residual: <E10.3>".  snprintf semantics: returns the length needed. */
int nlh_format_status(int32_t iter, int32_t nfeval, int32_t njaceval, double xnorm, double fnorm, char *buf, int32_t len);

/* ---- user callbacks: vecfcn / jacobianfcn (src/nonlin_multi_eqn_mult_var.f90:14-38)
 * flattened to C.  The Fortran shim passes bind(C) trampolines; ctx carries the
 * vecfcn_helper and the optional class(*) args.  jac is column-major, ld = m. */
/* fcnnvar / gradientfcn (src/nonlin_multi_var.f90:14-27) flattened to C. */
typedef double (*nlh_fcnnvar)(void *ctx, int32_t n, const double *x);
typedef void (*nlh_gradfcn)(void *ctx, int32_t n, const double *x, double *g);
typedef void (*nlh_vecfcn)(void *ctx, int32_t n, const double *x, int32_t m, double *f);
typedef void (*nlh_jacfcn)(void *ctx, int32_t n, const double *x, int32_t m, double *jac);

/* ---- handle: owns a HIP stream reference, device workspaces (cached per shape)
 * and per-kernel HIP-event timers.  Not thread-safe; use one per thread. ---- */
typedef struct nlh_handle nlh_handle;
int  nlh_create(nlh_handle **h, int32_t device, void *hip_stream /* NULL => the default (null) stream */);
void nlh_destroy(nlh_handle *h);
int  nlh_device_count(void);             /* 0 when no GPU is visible */
const char *nlh_last_error(const nlh_handle *h);
const char *nlh_version(void);

/* ===========================================================================
 * Host-callback ("mode H") drop-in entry points: HOST pointers.
 * ======================================================================== */

/* vecfcn_helper%jacobian -- vfh_jac_fcn, src/nonlin_multi_eqn_mult_var.f90:198-277.
 * jacfcn != NULL: forwards to it.  Otherwise evaluates fcn at x + h_j e_j on the
 * host (in the reference's order, x perturbed in place and restored), uploads the
 * m-by-n residual panel and forms jac(:,j) = (f_j - f0)/h_j on the GPU.
 * fv may be NULL (then f0 = fcn(x) is evaluated first, :257-259). */
int nlh_fd_jacobian(nlh_handle *h, int32_t m, int32_t n, nlh_vecfcn fcn, nlh_jacfcn jacfcn,
                    void *ctx, double *x, const double *fv, double *jac);

/* least_squares_solver%solve -- lss_solve, src/nonlin_least_squares.f90:118-391. */
int nlh_lm_solve(nlh_handle *h, const nlh_options *opts, int32_t m, int32_t n,
                 nlh_vecfcn fcn, nlh_jacfcn jacfcn, void *ctx,
                 double *x, double *fvec, nlh_iteration_behavior *ib);

/* newton_solver%solve -- ns_solve, src/nonlin_solve.f90:452-638 (LU step: :570,577). */
int nlh_newton_solve(nlh_handle *h, const nlh_options *opts, int32_t n,
                     nlh_vecfcn fcn, nlh_jacfcn jacfcn, void *ctx,
                     double *x, double *fvec, nlh_iteration_behavior *ib);

/* quasi_newton_solver%solve -- qns_solve, src/nonlin_solve.f90:156-427 (Broyden's method; QR of the
 * Jacobian at :289, rank-one QR update at :307, triangular solve at :327).  jdelta =
 * quasi_newton_solver%m_jDelta (get/set_jacobian_interval, :429-447; default 5, :51). */
int nlh_quasi_newton_solve(nlh_handle *h, const nlh_options *opts, int32_t jdelta, int32_t n,
                           nlh_vecfcn fcn, nlh_jacfcn jacfcn, void *ctx,
                           double *x, double *fvec, nlh_iteration_behavior *ib);

/* constrained_least_squares_solver%solve -- cls_solve, src/nonlin_least_squares.f90:938-1176 (bounded
 * trust-region dog-leg: qr_factor :1047, coleman_li_scaling :1050, dogleg :1053/1301-1403, Armijo
 * fallback :1088-1123).  delta0 = get_trust_region_radius() (default 1, :60), stepscale0 =
 * get_step_scaling_factor() (default 1, :61); xl / xu = get_lower_limits() / get_upper_limits()
 * ([n] host arrays, NULL = unbounded, :999-1009). */
int nlh_cls_solve(nlh_handle *h, const nlh_options *opts, double delta0, double stepscale0,
                  const double *xl, const double *xu, int32_t m, int32_t n,
                  nlh_vecfcn fcn, nlh_jacfcn jacfcn, void *ctx,
                  double *x, double *fvec, nlh_iteration_behavior *ib);

/* fcnnvar_helper%gradient -- fnh_grad_fcn, src/nonlin_multi_var.f90:182-246.  gradfcn != NULL: the user's gradient.
 * Otherwise forward differences: g_j = (f(x + h_j e_j) - f(x)) / h_j, h_j = sqrt(eps) |x_j| (sqrt(eps) when x_j = 0),
 * callbacks in ascending j on the calling thread; x is perturbed and restored; fv = NULL => f(x) is evaluated first.
 * Host arrays; needs no handle (n + 1 calls of a host function and nothing else). */
int nlh_fd_gradient(int32_t n, nlh_fcnnvar fcn, nlh_gradfcn gradfcn, void *ctx, double *x, const double *fv, double *g);

/* bfgs%solve -- bfgs_solve, src/nonlin_optimize.f90:557-770, with fcnnvar_helper%gradient
 * (src/nonlin_multi_var.f90:182-246; gradfcn = NULL => forward differences) and ls_search_miso
 * (src/nonlin_linesearch.f90:329-492).  opts->max_evals = get_max_fcn_evals() (500, :46),
 * opts->gtol = get_tolerance() (1e-12, :47), opts->xtol = get_var_tolerance() (1e-12,
 * src/nonlin_optimize.f90:47), use_line_search / ls_* as for newton.  fout may be NULL.
 * ib->gradient_count is filled, ib->jacobian_count = 0. */
int nlh_bfgs_solve(nlh_handle *h, const nlh_options *opts, int32_t n, nlh_fcnnvar fcn,
                   nlh_gradfcn gradfcn, void *ctx, double *x, double *fout,
                   nlh_iteration_behavior *ib);

/* ===========================================================================
 * Device-model ("mode D") batched entry points: DEVICE pointers.
 * Residual family "dense-quadratic" (SURVEY.md 8(d)), evaluated on the GPU with
 * the exact per-row operation order of the CPU path:
 *   u_i = sum_j A(i,j) x_j  (j ascending, separate multiply and add)
 *   r_i = (u_i + (gamma*u_i)*u_i) - b_i ;   dr_i/dx_j = (1 + 2 gamma u_i) A(i,j)
 * Layout: A [nprob][n][m] (each problem column-major m-by-n), b/fvec [nprob][m],
 * x [nprob][n].  Problems are independent; status[k] receives 0 or an NL_* code.
 * ======================================================================== */
int nlh_dq_lm_solve_batch(nlh_handle *h, const nlh_options *opts, int32_t nprob,
                          int32_t m, int32_t n, const double *dA, const double *db,
                          double gamma, double *dx, double *dfvec,
                          nlh_iteration_behavior *ib /* host, [nprob] */,
                          int32_t *status /* host, [nprob] */);

int nlh_dq_newton_solve_batch(nlh_handle *h, const nlh_options *opts, int32_t nprob,
                              int32_t n, const double *dA, const double *db, double gamma,
                              int32_t analytic_jacobian, double *dx, double *dfvec,
                              nlh_iteration_behavior *ib, int32_t *status);

int nlh_dq_quasi_newton_solve_batch(nlh_handle *h, const nlh_options *opts, int32_t jdelta,
                                    int32_t nprob, int32_t n, const double *dA, const double *db,
                                    double gamma, int32_t analytic, double *dx, double *dfvec,
                                    nlh_iteration_behavior *ib /* host, [nprob] */,
                                    int32_t *status /* host, [nprob] */);

/* xl / xu: [n] host arrays shared by every problem, or NULL. */
int nlh_dq_cls_solve_batch(nlh_handle *h, const nlh_options *opts, double delta0, double stepscale0,
                           const double *xl, const double *xu, int32_t nprob, int32_t m, int32_t n,
                           const double *dA, const double *db, double gamma, double *dx,
                           double *dfvec, nlh_iteration_behavior *ib /* host, [nprob] */,
                           int32_t *status /* host, [nprob] */);

/* bfgs on f(x) = 0.5 * sum_i r_i(x)^2 of the device model, forward-difference gradient on the device.
 * hfout: [nprob] host (may be NULL). */
int nlh_dq_bfgs_solve_batch(nlh_handle *h, const nlh_options *opts, int32_t nprob, int32_t m,
                            int32_t n, const double *dA, const double *db, double gamma, double *dx,
                            double *hfout, nlh_iteration_behavior *ib /* host, [nprob] */,
                            int32_t *status /* host, [nprob] */);

/* ---- device residual models behind HOST arrays (no reference counterpart: the extension of vecfcn_helper that lets
 * `solver%solve` reach the batched device path -- nonlin_amd/fortran: vecfcn_helper%set_device_model, device_model_batch,
 * least_squares_solver%solve_batch).  A model owns device copies of nprob problems of the dense-quadratic family
 * r = (u + gamma u u) - b, u = A x: A [nprob][n][m] (each problem column-major m x n), b [nprob][m].
 * x [nprob][n] in/out and fvec [nprob][m] out are host arrays; status[p] = 0 or the NL_* code the reference would stop
 * with for problem p (no process abort). ---- */
typedef struct nlh_dq_model nlh_dq_model;
int  nlh_dq_model_create(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *A, const double *b,
                         double gamma, nlh_dq_model **model);

/* ---- several GPUs behind the boundary (SURVEY.md 8(b) `nlx_init(device, comm)`, 8(e); replaces nothing in the
 * reference, whose solvers are single-threaded -- this is how `solver%solve_batch` reaches every GPU of the node from ONE
 * process).  A device set owns one handle (own stream, own workspaces) per entry of its device list.  A model created
 * ON a set is dealt over the entries block-cyclically (problem k -> entry k mod ndev: iteration counts differ per
 * problem), and every nlh_dq_model_* call on it runs one host thread per entry: independent problems, no collective,
 * the same bits as on one device.  devices == NULL or ndev <= 0: every visible device; an id may repeat (two shares
 * on one GPU).  The handle argument of nlh_dq_model_eval / _lm_solve / _newton_solve is ignored (may be NULL) for a
 * model created on a set.  (One process per GPU over RCCL is the other way to use several GPUs: nonlin_amd/sharding.py,
 * bench.py --gpus N.) ---- */
typedef struct nlh_device_set nlh_device_set;
int  nlh_device_set_create(nlh_device_set **set, const int32_t *devices, int32_t ndev);
void nlh_device_set_destroy(nlh_device_set *set);
int32_t nlh_device_set_size(const nlh_device_set *set);
nlh_handle *nlh_device_set_handle(nlh_device_set *set, int32_t i);        /* entry i's handle (owned by the set) */
const char *nlh_device_set_last_error(const nlh_device_set *set);
int  nlh_dq_model_create_on(nlh_device_set *set, int32_t nprob, int32_t m, int32_t n, const double *A, const double *b,
                            double gamma, nlh_dq_model **model);
int32_t nlh_dq_model_device_count(const nlh_dq_model *model);             /* shares the model is dealt into */
void nlh_dq_model_destroy(nlh_dq_model *model);
void nlh_dq_model_shape(const nlh_dq_model *model, int32_t *nprob, int32_t *m, int32_t *n);
/* vecfcn (src/nonlin_multi_eqn_mult_var.f90:14-25) of the model, every problem */
int  nlh_dq_model_eval(nlh_handle *h, const nlh_dq_model *model, const double *x, double *f);
/* lss_solve (src/nonlin_least_squares.f90:118-391) / ns_solve (src/nonlin_solve.f90:452-638) on every problem */
int  nlh_dq_model_lm_solve(nlh_handle *h, const nlh_options *opts, const nlh_dq_model *model, double *x, double *fvec,
                           nlh_iteration_behavior *ib, int32_t *status);
int  nlh_dq_model_newton_solve(nlh_handle *h, const nlh_options *opts, const nlh_dq_model *model, int32_t analytic,
                               double *x, double *fvec, nlh_iteration_behavior *ib, int32_t *status);
/* The same for quasi_newton_solver%solve (src/nonlin_solve.f90:156-427; jdelta: iterations between fresh Jacobians),
 * constrained_least_squares_solver%solve (src/nonlin_least_squares.f90:938-1176; xl / xu: n entries or NULL, one box for
 * every problem) and bfgs%solve on 0.5 ||F(x)||^2 (src/nonlin_optimize.f90:557-770; fout [nprob]: the objective at the
 * solution, fvec: F there).  Each runs the lock-step device state machine of its solver on every share of the model. */
int  nlh_dq_model_quasi_newton_solve(nlh_handle *h, const nlh_options *opts, const nlh_dq_model *model, int32_t jdelta,
                                     int32_t analytic, double *x, double *fvec, nlh_iteration_behavior *ib, int32_t *status);
int  nlh_dq_model_cls_solve(nlh_handle *h, const nlh_options *opts, const nlh_dq_model *model, double delta0,
                            double stepscale0, const double *xl, const double *xu, double *x, double *fvec,
                            nlh_iteration_behavior *ib, int32_t *status);
int  nlh_dq_model_bfgs_solve(nlh_handle *h, const nlh_options *opts, const nlh_dq_model *model, double *x, double *fvec,
                             double *fout, nlh_iteration_behavior *ib, int32_t *status);


/* ===========================================================================
 * User-supplied DEVICE residuals: vecfcn / jacobianfcn as LAUNCHERS.
 *
 * The reference's plugin layer is "the user hands in a residual": vecfcn (src/nonlin_multi_eqn_mult_var.f90:14-25),
 * set_fcn (:126-140), and the solvers call it at x, at the n perturbed points of the forward-difference Jacobian
 * (:267-273) and at trial points.  A host procedure cannot run on the GPU, so the device form of the plugin is a
 * launcher: a HOST function that ENQUEUES, on the HIP stream it is handed, device work which evaluates F at `npoints`
 * points, and returns at once (0, or non-zero to abort the solve with NLH_ERR_HIP).  It must not synchronise and may be
 * called from several host threads on different streams (sub-batches, device sets).
 *   point q (0 <= q < npoints) belongs to problem dprob[q] (a DEVICE array: the index the problem has in the caller's
 *   batch -- what the user's kernel selects its data with); its variables are dX[q*n .. q*n + n), its residuals go to
 *   dF[q*m .. q*m + m).  For the Jacobian launcher point q's m-by-n Jacobian goes to dJ + q*m*n, column-major (ld = m).
 * The library builds the points itself, on the device, in the reference's order: for a forward-difference Jacobian of
 * problem p, points p*n + j = x with x(j) replaced by x(j) + h_j (:268-271); k_fd_jacobian then forms
 * jac(:,j) = (F(point j) - F(x)) / h_j with a true division (:274).  Everything downstream is the state machine the
 * dense-quadratic entry points run.  A problem's result does not depend on the batch it is solved in.
 * ======================================================================== */
typedef int (*nlh_device_vecfcn)(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n,
                                 const double *dX, int32_t m, double *dF);
typedef int (*nlh_device_jacfcn)(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n,
                                 const double *dX, int32_t m, double *dJ);

/* vecfcn_helper%jacobian (vfh_jac_fcn, :198-277) of every problem: jacfcn != NULL forwards to it (:241-243), otherwise
 * forward differences.  dx [nprob][n], dfv [nprob][m] = F(x) or NULL (then evaluated first, :257-259), dJ [nprob][n][m]
 * (each problem column-major m x n); DEVICE pointers. */
int nlh_fd_jacobian_device(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, nlh_device_vecfcn fcn,
                           nlh_device_jacfcn jacfcn, void *ctx, const double *dx, const double *dfv, double *dJ);
/* least_squares_solver%solve (lss_solve, src/nonlin_least_squares.f90:118-391) on nprob problems of the user's family.
 * dx [nprob][n] in/out, dfvec [nprob][m] out: DEVICE pointers; ib / status: host, [nprob] (NULL allowed). */
int nlh_lm_solve_batch_device(nlh_handle *h, const nlh_options *opts, int32_t nprob, int32_t m, int32_t n,
                              nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx, double *dx, double *dfvec,
                              nlh_iteration_behavior *ib, int32_t *status);
/* newton_solver%solve (ns_solve, src/nonlin_solve.f90:452-638) / quasi_newton_solver%solve (qns_solve, :156-427) on
 * nprob square problems of the user's family. */
int nlh_newton_solve_batch_device(nlh_handle *h, const nlh_options *opts, int32_t nprob, int32_t n,
                                  nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx, double *dx, double *dfvec,
                                  nlh_iteration_behavior *ib, int32_t *status);
int nlh_quasi_newton_solve_batch_device(nlh_handle *h, const nlh_options *opts, int32_t jdelta, int32_t nprob, int32_t n,
                                        nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx, double *dx,
                                        double *dfvec, nlh_iteration_behavior *ib, int32_t *status);
/* constrained_least_squares_solver%solve (cls_solve, src/nonlin_least_squares.f90:938-1176) on the user's family; xl / xu:
 * [n] host arrays shared by every problem, or NULL (as nlh_dq_cls_solve_batch). */
int nlh_cls_solve_batch_device(nlh_handle *h, const nlh_options *opts, double delta0, double stepscale0, const double *xl,
                               const double *xu, int32_t nprob, int32_t m, int32_t n, nlh_device_vecfcn fcn,
                               nlh_device_jacfcn jacfcn, void *ctx, double *dx, double *dfvec, nlh_iteration_behavior *ib,
                               int32_t *status);
int nlh_cls_solve_batch_device_h(nlh_handle *h, const nlh_options *opts, double delta0, double stepscale0, const double *xl,
                                 const double *xu, int32_t nprob, int32_t m, int32_t n, nlh_device_vecfcn fcn,
                                 nlh_device_jacfcn jacfcn, void *ctx, double *x, double *fvec, nlh_iteration_behavior *ib,
                                 int32_t *status);
/* bfgs%solve (src/nonlin_optimize.f90:557-770) on a batch of problems whose objective is the USER'S device fcnnvar
 * (reference plugin: fcnnvar_helper, src/nonlin_multi_var.f90:17-44, 93-104 set_fcn, 182-246 gradient): the launcher is an
 * nlh_device_vecfcn called with m = 1 -- dF[npoints] receives f at each of the npoints points --, gradfcn (NULL: forward
 * differences, n more points per gradient, built on the device in the reference's order :231-243) an nlh_device_jacfcn
 * called with m = 1 -- dJ[npoints][n] receives the gradients (set_gradient_fcn, :126-138).  dx [nprob][n] device, in/out;
 * fout [nprob] host (NULL allowed): f at the solution (:762).  Counts, flags and errors per problem as nlh_bfgs_solve. */
int nlh_bfgs_solve_batch_device(nlh_handle *h, const nlh_options *opts, int32_t nprob, int32_t n, nlh_device_vecfcn fcn,
                                nlh_device_jacfcn gradfcn, void *ctx, double *dx, double *fout, nlh_iteration_behavior *ib,
                                int32_t *status);
/* ... with x [nprob][n] a HOST array. */
int nlh_bfgs_solve_batch_device_h(nlh_handle *h, const nlh_options *opts, int32_t nprob, int32_t n, nlh_device_vecfcn fcn,
                                  nlh_device_jacfcn gradfcn, void *ctx, double *x, double *fout, nlh_iteration_behavior *ib,
                                  int32_t *status);
/* The same three behind HOST arrays x [nprob][n] in/out, fvec [nprob][m] out (what the Fortran shim's
 * vecfcn_helper%set_device_fcn + solver%solve / solve_batch call): staged through the handle's buffers. */
int nlh_lm_solve_batch_device_h(nlh_handle *h, const nlh_options *opts, int32_t nprob, int32_t m, int32_t n,
                                nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx, double *x, double *fvec,
                                nlh_iteration_behavior *ib, int32_t *status);
int nlh_newton_solve_batch_device_h(nlh_handle *h, const nlh_options *opts, int32_t nprob, int32_t n,
                                    nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx, double *x, double *fvec,
                                    nlh_iteration_behavior *ib, int32_t *status);
int nlh_quasi_newton_solve_batch_device_h(nlh_handle *h, const nlh_options *opts, int32_t jdelta, int32_t nprob, int32_t n,
                                          nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx, double *x,
                                          double *fvec, nlh_iteration_behavior *ib, int32_t *status);
/* The built-in dense-quadratic family expressed as such launchers (ctx = nlh_dq_device_ctx): the same residual bits as
 * the nlh_dq_* entry points, through the open path. */
typedef struct nlh_dq_device_ctx {
    const double *dA;     /* [nprob][n][m], device */
    const double *db;     /* [nprob][m], device */
    double gamma;
} nlh_dq_device_ctx;
int nlh_dq_device_fcn(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n, const double *dX,
                      int32_t m, double *dF);
int nlh_dq_device_jac(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n, const double *dX,
                      int32_t m, double *dJ);

/* A user's device residual as a MODEL object (what the Fortran shim's vecfcn_helper%set_device_fcn and
 * device_model_batch%create_from_device_fcn hold): nprob problems of m equations in n unknowns each, evaluated by the
 * launchers; nlh_dq_model_eval / _lm_solve / _newton_solve / _quasi_newton_solve accept it (host arrays, the caller's
 * handle; `analytic` selects the jacobianfcn launcher) and so does nlh_dq_model_cls_solve; the bfgs form takes a model of
 * ONE function (m = 1: the launcher is the user's fcnnvar, the jacobianfcn launcher its gradient -- nlh_bfgs_solve_batch_device)
 * and returns NLH_INVALID_OPERATION_ERROR for m > 1 (bfgs minimises a scalar fcnnvar, not a vecfcn).  Lives on the handle's device (not dealt over a device set: the user's data is wherever the
 * user put it).  Freed with nlh_dq_model_destroy; ctx stays the caller's. */
int nlh_device_fcn_model_create(int32_t nprob, int32_t m, int32_t n, nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx,
                                nlh_dq_model **model);

/* Synthetic problem generator of SURVEY.md 8(d) (bench/test inputs, not part of the
 * reference): counter-based splitmix64, U_k = mix(seed + (k+1)*0x9E3779B97F4A7C15),
 * draw order A (column-major), x_true, noise, x0; problem p uses seed0 + p*seed_stride.
 * A = (2U-1)/sqrt(n) (+2I when square_shift), b = model(x_true) + sigma(2U-1),
 * x0 = x_true + spread(2U-1).  All pointers are DEVICE pointers. */
int nlh_dq_generate(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, uint64_t seed0,
                    uint64_t seed_stride, double gamma, double sigma, double spread, int32_t square_shift,
                    double *dA, double *db, double *dxtrue, double *dx0);

/* ---- stage-level entry points (each is one kernel family of the path; used by
 * the parity tests and the roofline measurement).  DEVICE pointers. ---------- */

/* vecfcn for the dense-quadratic model: f = F(x) for every problem. */
int nlh_dq_residual(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dA,
                    const double *db, double gamma, const double *dx, double *df);
/* The n perturbed evaluations of vfh_jac_fcn (:267-273): P(:,j) = F(x + h_j e_j). */
int nlh_dq_fd_panel(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dA,
                    const double *db, double gamma, const double *dx, double *dP);
/* The forward-difference column write (:274): J(:,j) = (P(:,j) - f0)/h_j,
 * h_j = sqrt(eps)*|x_j| (sqrt(eps) if zero).  HBM-bound streaming kernel. */
int nlh_fd_jacobian_panel(nlh_handle *h, int32_t nprob, int32_t m, int32_t n,
                          const double *dP, const double *df0, const double *dx, double *dJ);
/* Analytic jacobianfcn of the dense-quadratic model. */
int nlh_dq_jacobian(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dA,
                    double gamma, const double *dx, double *dJ);
/* J^T J (fp64 MFMA, deterministic split-K) and J^T f.  dG [nprob][n][n], dg [nprob][n]. */
int nlh_gram(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dJ,
             const double *df, double *dG, double *dg);
/* lmfactor replacement on the Gram matrix: pivoted Cholesky P^T G P = R^T R with
 * MINPACK's pivot rule, acnorm = sqrt(diag G), qtf = R^-T P^T g.  dG is overwritten
 * by R (upper triangle).  ipvt is 0-based.  info[k] != 0 => ill-conditioned/rank-deficient. */
int nlh_chol_factor(nlh_handle *h, int32_t nprob, int32_t n, double *dG, const double *dg,
                    int32_t *dipvt, double *dacnorm, double *dqtf, int32_t *dinfo);
/* lmfactor itself (pivoted Householder QR, src/nonlin_least_squares.f90:569-667) plus
 * Q^T f (:241-253).  dJ is overwritten as in the reference (R strict upper, reflectors
 * below, diagonal restored to rdiag after Q^T f); dqtf [nprob][n]; dwa4 [nprob][m]. */
int nlh_qr_factor(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, double *dJ,
                  const double *df, int32_t *dipvt, double *drdiag, double *dacnorm,
                  double *dqtf, double *dwa4);
/* The same factorisation in the reference's OPERATION ORDER (what NLH_FACTOR_EXACT runs inside the LM solve:
 * streaming lock-step Householder steps, nlh_qrx.hip): bit-identical to the CPU path.  dJ [nprob][n][m] is not
 * modified; dR [nprob][n][n] column-major receives R (strict upper triangle + diagonal = rdiag); requires m >= n. */
int nlh_lmfactor_exact(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dJ,
                       const double *df, double *dR, int32_t *dipvt, double *drdiag, double *dacnorm,
                       double *dqtf, double *dwa4);
/* lmpar (:394-566, including its two deviations from MINPACK) on an n-by-n R
 * (leading dimension ldr) for every problem.  dtailsq[k] = sum of squares of the
 * caller's wa4(n+1:m).  Outputs: dpar (in/out), dxstep [nprob][n], dsdiag [nprob][n]. */
int nlh_lmpar(nlh_handle *h, int32_t nprob, int32_t n, double *dR, int32_t ldr,
              const int32_t *dipvt, const double *ddiag, const double *dqtf,
              const double *ddelta, const double *dtailsq, double *dpar,
              double *dxstep, double *dsdiag);
/* lu_factor / solve_lu stand-ins (call sites src/nonlin_solve.f90:570,577):
 * partial-pivoting LU of [nprob][n][n] in place, 0-based ipvt, then one RHS each. */
int nlh_lu_factor(nlh_handle *h, int32_t nprob, int32_t n, double *dA, int32_t *dipvt,
                  int32_t *dinfo);
int nlh_lu_solve(nlh_handle *h, int32_t nprob, int32_t n, const double *dLU,
                 const int32_t *dipvt, double *db);
/* qr_factor(b, q = q, r = r), qr_rank1_update(q, r, u, v) and solve_triangular_system stand-ins
 * (call sites src/nonlin_solve.f90:289, 307, 327; third-party linalg in the reference).
 * dB, dQ: [nprob][n][n] column-major.  dRt: R stored ROW-major.  Q1 R1 = Q R + u v^T. */
int nlh_qr_factor_full(nlh_handle *h, int32_t nprob, int32_t n, const double *dB, double *dQ,
                       double *dRt);
int nlh_qr_rank1_update(nlh_handle *h, int32_t nprob, int32_t n, double *dQ, double *dRt,
                        const double *du, const double *dv);
int nlh_solve_upper(nlh_handle *h, int32_t nprob, int32_t n, const double *dRt, double *dx);
/* cholesky_rank1_update (downdate = 0) / cholesky_rank1_downdate (1) stand-ins (call sites
 * src/nonlin_optimize.f90:721-722): R1^T R1 = R^T R +- u u^T in place on the ROW-major upper factor dRt;
 * du is consumed; *hinfo = 1 if the downdate would lose positive definiteness. */
int nlh_chol_rank1(nlh_handle *h, int32_t n, int32_t downdate, double *dRt, double *du, int32_t *hinfo);

/* polynomial%fit / polynomial%fit_thru_zero (src/nonlin_polynomials.f90:146-238): least-squares polynomial of
 * the given order through npts points; coef = c0 .. c_order (c0 = 0 for thru_zero).  Returns 4 where the
 * reference stops with 4 (order >= npts or order < 1).  solve_least_squares (third-party linalg) is the
 * Householder QR + back substitution of nlh_cls_solve.  The batch form takes device arrays
 * dx, dy [nprob][npts], dcoef [nprob][order + 1]. */
int nlh_poly_fit(nlh_handle *h, int32_t npts, int32_t order, int32_t thru_zero, const double *x,
                 const double *y, double *coef);
int nlh_poly_fit_batch(nlh_handle *h, int32_t nprob, int32_t npts, int32_t order, int32_t thru_zero,
                       const double *dx, const double *dy, double *dcoef);

/* ---- per-kernel timing (HIP events on the handle's stream) ------------------ */
#define NLH_K_DQ_RESIDUAL   0
#define NLH_K_DQ_PANEL      1
#define NLH_K_FD_JACOBIAN   2
#define NLH_K_GRAM          3
#define NLH_K_GRAM_REDUCE   4
#define NLH_K_JTF           5
#define NLH_K_CHOL          6
#define NLH_K_LMPAR         7
#define NLH_K_QR            8
#define NLH_K_UPDATE        9
#define NLH_K_LU           10
#define NLH_K_DQ_JACOBIAN  11
#define NLH_K_QRX_PASS     12   /* exact lmfactor: trailing pass of a Householder step (nlh_qrx.hip) */
#define NLH_K_QRX_PIVOT    13   /* exact lmfactor: pivot + reflector of a step */
#define NLH_K_COUNT        14
/* on: 0 = off, 1 = every kernel group, otherwise a mask with bit (k + 1) set for each group NLH_K_<k> to time
   (two HIP event records per timed launch on the handle's stream). */
void nlh_timing_enable(nlh_handle *h, int32_t on);
void nlh_timing_reset(nlh_handle *h);
/* Synchronises the stream, then returns total milliseconds and launch count. */
int  nlh_timing_get(nlh_handle *h, int32_t kernel_id, double *total_ms, int64_t *launches);
/* Per-launch durations (ms, launch order) of ONE kernel group since the last nlh_timing_reset.  The first call with a
 * new kernel_id selects that group and returns 0; later calls copy up to cap samples and return how many there are. */
int64_t nlh_timing_samples(nlh_handle *h, int32_t kernel_id, float *out_ms, int64_t cap);
const char *nlh_kernel_name(int32_t kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* NONLIN_HIP_H */
