// nlh_bfgs.hip -- bfgs (bfgs_solve, src/nonlin_optimize.f90:557-770) with ls_search_miso
// (src/nonlin_linesearch.f90:329-492) and fcnnvar_helper%gradient (src/nonlin_multi_var.f90:182-246): host loop for one
// problem, lock-step device state machine for batches; the Cholesky rank-one update / downdate as an entry point.
#include "nlh_internal.h"
#include "nlh_kernels_broyden.h"
#include "nlh_kernels_newton.h"
#include "nlh_kernels_bfgs.h"
#include "nlh_kernels_bfgs_batch.h"

// The blocked Cholesky of the Hessian approximation: as many thread groups per column as 1024 threads allow.
static void launch_bf_chol_blocked(hipStream_t s, int nprob, int n, const double *dB, double *dR, int *dinfo, const LmState *st, int want)
{
    const int CT = ((n + 63) / 64) * 64;
    if (CT * 4 <= 1024) hipLaunchKernelGGL(k_bf_chol_blocked<4>, dim3(nprob), dim3(CT * 4), bf_chol_lds(n), s, n, dB, dR, dinfo, st, want);
    else if (CT * 2 <= 1024) hipLaunchKernelGGL(k_bf_chol_blocked<2>, dim3(nprob), dim3(CT * 2), bf_chol_lds(n), s, n, dB, dR, dinfo, st, want);
    else hipLaunchKernelGGL(k_bf_chol_blocked<1>, dim3(nprob), dim3(CT), bf_chol_lds(n), s, n, dB, dR, dinfo, st, want);
}

void nlh_bfgs_init_device(int lds_max)
{
    hipFuncSetAttribute((const void *)k_bf_solve_upper_t, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_chol_blocked<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_chol_blocked<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_chol_blocked<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_chol_update<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_chol_update<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_chol_update<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_downdate_apply, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    broyden_kernel_attrs(lds_max);
}



// ===========================================================================
// BFGS: bfgs_solve (src/nonlin_optimize.f90:557-770) as a host loop; the Hessian factor R (row-major) and
// B = R^T R live on the device.  Host: ls_search_miso, the convergence tests and the O(n) vector algebra in the
// reference's order.
// ===========================================================================
struct BfgsEval {
    std::function<int(const double *x, double *f)> fcn;                 // objective at host x
    std::function<int(double *x, double fv, double *g)> grad;           // gradient at host x (fv = f(x)) -> host g
};

// ls_search_miso, src/nonlin_linesearch.f90:329-492
static int line_search_scalar(const nlh_options *o, BfgsEval &ev, int n, const double *xold, const double *grad,
                              const double *dir, double *x, double fold, double *fx, int *fcn_count)
{
    const double tolx = 2.0 * DBL_EPSILON, alpha = o->ls_alpha, lambdamin = o->ls_factor;
    const int maxeval = o->ls_max_evals;
    int neval = 0, niter = 0, flag = 0, rc = 0;
    double alam, alam1 = 0.0, alamin, f1 = 0.0, slope, test, tmplam = 0.0, f = 0.0;
    *fcn_count = 0;
    slope = h_dot(n, grad, dir);
    if (slope >= 0.0) return NLH_DIVERGENT_BEHAVIOR_ERROR;
    test = 0.0;
    for (int i = 0; i < n; ++i) {
        const double t = fabs(dir[i]) / fmax(fabs(xold[i]), 1.0);
        if (t > test) test = t;
    }
    alamin = tolx / test;
    alam = 1.0;
    for (;;) {
        for (int i = 0; i < n; ++i) x[i] = xold[i] + alam * dir[i];
        if ((rc = ev.fcn(x, &f))) return rc;
        neval += 1;
        niter += 1;
        if (alam < alamin) {
            double sq = 0.0;
            for (int i = 0; i < n; ++i) { const double d = x[i] - xold[i]; sq = sq + d * d; }
            if (sqrt(sq) == 0.0) { rc = NLH_CONVERGENCE_ERROR; break; }
            for (int i = 0; i < n; ++i) x[i] = xold[i];
            break;
        } else if (f <= fold + alpha * alam * slope) {
            break;
        } else {
            tmplam = min_backtrack_search(niter, fold, f, f1, alam, alam1, slope);
        }
        alam1 = alam;
        f1 = f;
        alam = fmax(tmplam, lambdamin * alam);
        if (neval >= maxeval) { flag = 1; break; }
    }
    *fx = f;
    *fcn_count = neval;
    if (rc) return rc;
    return flag ? NLH_CONVERGENCE_ERROR : 0;
}

static int bfgs_core(nlh_handle *h, const nlh_options *o, int n, BfgsEval &ev, double *x, double *fout,
                     nlh_iteration_behavior *ib)
{
    int rc;
    if (n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    const size_t nn = (size_t)n * n;
    if ((rc = ensure(h, h->bfB, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->bfR, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->bfV, sizeof(double) * ((size_t)6 * n + 8)))) return rc;
    double *dB = (double *)h->bfB.p, *dR = (double *)h->bfR.p, *dv = (double *)h->bfV.p;
    double *dvec = dv, *dout = dv + n, *du = dv + 2 * n, *dc = dv + 3 * n;
    int *dinfo = (int *)(dv + 4 * n);
    hipStream_t s = h->stream;
    std::vector<double> g(n), dx(n), u(n), v(n), y(n), bdx(n), gold(n), xnew(n);
    int xcnvrg = 0, gcnvrg = 0, neval = 0, ngrad = 0, flag = 0, iter = 0, hinfo = 0;
    double fp, stpmax = 0.0, fret = 0.0, xtest = 0.0, gtest, temp, ydx;
    const int bs1 = std::min(1024, ((n + 63) / 64) * 64);
    rc = 0;

    if ((rc = ev.fcn(x, &fp))) return rc;                       // :633-636
    if ((rc = ev.grad(x, fp, g.data()))) return rc;
    neval = 1;
    ngrad = 1;
    gtest = h_norm2(n, g.data());                               // :639-642
    if (gtest < o->gtol) gcnvrg = 1;

    if (!gcnvrg) {
        for (;;) {                                              // :647-748
            iter += 1;
            if (iter == 1) {                                    // :653-656
                for (int i = 0; i < n; ++i) dx[i] = -g[i];
                stpmax = 100.0 * fmax(h_norm2(n, x), (double)n);
            }
            if (o->use_line_search) {                           // :659-669
                const double mag = h_norm2(n, dx.data());       // limit_search_vector
                if (mag != 0.0 && mag > stpmax) {
                    const double sc = stpmax / mag;
                    for (int i = 0; i < n; ++i) dx[i] = sc * dx[i];
                }
                int lcount = 0;
                rc = line_search_scalar(o, ev, n, x, g.data(), dx.data(), xnew.data(), fp, &fret, &lcount);
                neval += lcount;
                if (rc) break;
                fp = fret;
            } else {
                for (int i = 0; i < n; ++i) xnew[i] = x[i] + dx[i];
                if ((rc = ev.fcn(xnew.data(), &fp))) break;
                neval += 1;
            }
            for (int i = 0; i < n; ++i) {                       // :672-678
                dx[i] = xnew[i] - x[i];
                x[i] = xnew[i];
                gold[i] = g[i];
            }
            if ((rc = ev.grad(x, fp, g.data()))) break;
            ngrad += 1;

            xtest = 0.0;                                        // :681-689
            for (int i = 0; i < n; ++i) {
                temp = fabs(dx[i]) / fmax(fabs(x[i]), 1.0);
                xtest = fmax(temp, xtest);
            }
            if (xtest < o->xtol) { xcnvrg = 1; break; }
            gtest = h_norm2(n, g.data());                       // :692-696
            if (gtest < o->gtol) { gcnvrg = 1; break; }

            for (int i = 0; i < n; ++i) y[i] = g[i] - gold[i];  // :699-700
            ydx = h_dot(n, y.data(), dx.data());
            if (iter == 1) {                                    // :703-706
                temp = sqrt(h_dot(n, y.data(), y.data()) / ydx);
                hipLaunchKernelGGL(k_bf_scaled_identity, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, s, n, temp, dR, (const double *)nullptr, (size_t)0, (const int32_t *)nullptr, (size_t)0, (const LmState *)nullptr, -1);
            }
            // B = R^T R (:709), bdx = B dx (:712)
            hipLaunchKernelGGL(k_bf_rtr, dim3((n + 255) / 256, n), dim3(256), 0, s, n, dR, dB, (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(dvec, dx.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_matvec_cm, dim3((n + 255) / 256, 1), dim3(256), sizeof(double) * n, s, n, n, dB, dvec, dout, (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(bdx.data(), dout, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            if (ydx > 1.0e-10 && iter > 1) {                    // :715-724
                const double s1 = sqrt(ydx), s2 = sqrt(h_dot(n, dx.data(), bdx.data()));
                for (int i = 0; i < n; ++i) u[i] = y[i] / s1;
                for (int i = 0; i < n; ++i) v[i] = bdx[i] / s2;
                HIPCHK(h, hipMemcpyAsync(du, u.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
                if (n <= 1024) hipLaunchKernelGGL(k_bf_chol_update<1>, dim3(1), dim3(bs1), sizeof(double) * 2 * n, s, n, dR, du, (const LmState *)nullptr, -1);
                else if (qn_nc8(n)) hipLaunchKernelGGL(k_bf_chol_update<8>, dim3(1), dim3(1024), sizeof(double) * 2 * n, s, n, dR, du, (const LmState *)nullptr, -1);
                else hipLaunchKernelGGL(k_bf_chol_update<4>, dim3(1), dim3(1024), sizeof(double) * 2 * n, s, n, dR, du, (const LmState *)nullptr, -1);
                HIPCHK(h, hipMemcpyAsync(du, v.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_bf_solve_upper_t, dim3(1), dim3(bs1), sizeof(double) * n, s, n, dR, du, (const LmState *)nullptr, -1);
                hipLaunchKernelGGL(k_bf_downdate_rot, dim3(1), dim3(64), 0, s, n, du, dc, dinfo, (const LmState *)nullptr, -1);
                hipLaunchKernelGGL(k_bf_downdate_apply, dim3((n + 255) / 256), dim3(256), sizeof(double) * 2 * n, s, n, dR, dc, du, dinfo, (const LmState *)nullptr, -1);
            } else {
                if (n <= 1024) launch_bf_chol_blocked(s, 1, n, dB, dR, dinfo, nullptr, -1);
                else if (qn_nc8(n)) hipLaunchKernelGGL(k_bf_chol_factor<8>, dim3(1), dim3(1024), sizeof(double) * n, s, n, dB, dR, dinfo, (const LmState *)nullptr, -1);
                else hipLaunchKernelGGL(k_bf_chol_factor<4>, dim3(1), dim3(1024), sizeof(double) * n, s, n, dB, dR, dinfo, (const LmState *)nullptr, -1);
            }
            // dx = -(R^T R)^-1 g (:727)
            for (int i = 0; i < n; ++i) u[i] = -g[i];
            HIPCHK(h, hipMemcpyAsync(dout, u.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_bf_solve_upper_t, dim3(1), dim3(bs1), sizeof(double) * n, s, n, dR, dout, (const LmState *)nullptr, -1);
            hipLaunchKernelGGL(k_qn_solve_upper, dim3(1), dim3(bs1), sizeof(double) * n, s, n, dR, dout, (size_t)n * n, (size_t)n, (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(dx.data(), dout, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipMemcpyAsync(&hinfo, dinfo, sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            if (hinfo) { rc = NLH_INVALID_OPERATION_ERROR; break; }     // linalg: matrix not positive definite

            if (o->print_status) {                              // :730-737
                printf(" \n");
                printf("Iteration: %d\n", iter);
                printf("Function Evaluations: %d\n", neval);
                char e1[16], e2[16], e3[16];
                format_e10_3(fp, e1); format_e10_3(xtest, e2); format_e10_3(gtest, e3);
                printf("Function Value: %s\nChange in Variable: %s\nGradient: %s\n", e1, e2, e3);
            }
            if (neval >= o->max_evals) { flag = 1; break; }     // :740-743
        }
    }
    if (ib) {                                                   // :751-759
        ib->iter_count = iter; ib->fcn_count = neval; ib->jacobian_count = 0; ib->gradient_count = ngrad;
        ib->converge_on_fcn = 0; ib->converge_on_chng = xcnvrg; ib->converge_on_zero_diff = gcnvrg;
    }
    if (fout) *fout = fp;                                       // :762
    if (rc) return rc;
    return flag ? NLH_CONVERGENCE_ERROR : 0;                    // :765-767
}

// fcnnvar_helper%gradient -- fnh_grad_fcn, src/nonlin_multi_var.f90:182-246: the user's gradient routine when there is
// one, otherwise forward differences with h_j = sqrt(eps) |x_j| (sqrt(eps) at x_j = 0), one evaluation per variable in
// ascending order on the calling thread, true division.  The work is n + 1 calls of a host function: nothing here for
// the device; it lives behind the C ABI so that the Fortran shim and nlh_bfgs_solve share one implementation.
int nlh_fd_gradient(int32_t n, nlh_fcnnvar fcn, nlh_gradfcn gradfcn, void *ctx, double *x, const double *fv, double *g)
{
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;
    if (n < 1 || !x || !g) return NLH_INVALID_INPUT_ERROR;
    if (gradfcn) { gradfcn(ctx, n, x, g); return 0; }
    const double f0 = fv ? *fv : fcn(ctx, n, x);
    for (int j = 0; j < n; ++j) {
        const double xj = x[j];
        double step = NLH_SQRT_EPS * fabs(xj);
        if (step == 0.0) step = NLH_SQRT_EPS;
        x[j] = xj + step;
        const double fj = fcn(ctx, n, x);
        x[j] = xj;
        g[j] = (fj - f0) / step;
    }
    return 0;
}

// bfgs%solve -- bfgs_solve, src/nonlin_optimize.f90:557-770; fcnnvar / gradientfcn callbacks flattened to C
int nlh_bfgs_solve(nlh_handle *h, const nlh_options *o, int32_t n, nlh_fcnnvar fcn, nlh_gradfcn gradfcn, void *ctx,
                   double *x, double *fout, nlh_iteration_behavior *ib)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib) memset(ib, 0, sizeof *ib);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :614
    if (!o || n < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    BfgsEval ev;
    ev.fcn = [&](const double *xx, double *f) -> int { *f = fcn(ctx, n, xx); return 0; };
    ev.grad = [&](double *xx, double fv, double *g) -> int { return nlh_fd_gradient(n, fcn, gradfcn, ctx, xx, &fv, g); };
    int rc = bfgs_core(h, o, n, ev, x, fout, ib);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return NLH_ERR_HIP; }
    return rc;
}

// Device model: minimise f(x) = 0.5 * sum_i r_i(x)^2 of the dense-quadratic residual with bfgs; the
// forward-difference gradient (fnh_grad_fcn) is the residual panel kernel + k_bf_fd_gradient.
// bfgs%solve for a batch of device-model problems (objective 0.5 ||F(x)||^2, forward-difference gradient): the lock-step
// state machine of nlh_kernels_bfgs_batch.h.  A round takes every problem that wants a gradient through the residual
// panel and the differences, the tests and the secant pair, the update of the Cholesky factor (rank-one update +
// downdate, or a refactorisation) and the two triangular solves for the direction to the first trial point of its line
// search; every problem with a trial point gets F evaluated there and one turn of the search.  One 8-byte read-back
// per round.
// rs: the dense-quadratic family (objective 0.5 ||F(x)||^2), or -- scalar -- a USER'S fcnnvar handed in as a launcher with
// m = 1 (include/nonlin_hip.h: nlh_bfgs_solve_batch_device): F(x) is the objective itself, the forward-difference gradient
// is the 1 x n Jacobian of residual_jacobian (the same differences, src/nonlin_multi_var.f90:182-246), rs.jac the user's gradient.
static int bfgs_lockstep(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n, const ResidualSource &rs, bool scalar,
                         double *dx, double *hfout, nlh_iteration_behavior *ib, int32_t *status)
{
    int rc;
    if (n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    if (scalar != rs.user() || (scalar && m != 1)) return NLH_INVALID_INPUT_ERROR;
    const size_t mn = (size_t)m * n, nn = (size_t)n * n, np = (size_t)nprob;
    if (!scalar && (rc = ensure(h, h->P, sizeof(double) * mn * np))) return rc;
    if ((rc = ensure(h, h->bfB, sizeof(double) * nn * np))) return rc;
    if ((rc = ensure(h, h->bfR, sizeof(double) * nn * np))) return rc;
    if ((rc = ensure(h, h->bfV, sizeof(double) * ((size_t)m + 10 * (size_t)n + 1) * np + sizeof(int32_t) * np + 64))) return rc;
    if ((rc = ensure(h, h->state, sizeof(LmState) * np))) return rc;
    if ((rc = ensure(h, h->misc, sizeof(BfState) * np + 64))) return rc;
    if ((rc = ensure_pinned(h, sizeof(BfState) * np + 64))) return rc;
    double *dP = (double *)h->P.p, *dB = (double *)h->bfB.p, *dR = (double *)h->bfR.p, *q = (double *)h->bfV.p;
    double *dfv = q; q += (size_t)m * np;                        // F at the point evaluated last
    double *dg = q; q += (size_t)n * np;
    double *dgold = q; q += (size_t)n * np;
    double *ddx = q; q += (size_t)n * np;                        // the direction, then the step taken
    double *dy = q; q += (size_t)n * np;
    double *dbdx = q; q += (size_t)n * np;
    double *du = q; q += (size_t)n * np;
    double *dv = q; q += (size_t)n * np;
    double *dc = q; q += (size_t)n * np;
    double *dw = q; q += (size_t)n * np;
    double *dxnew = q; q += (size_t)n * np;
    double *df0 = q; q += np;                                    // (scalar: the objective at x, contiguous)
    int32_t *dinfo = (int32_t *)q;
    LmState *st = (LmState *)h->state.p;
    int32_t *dcounts = (int32_t *)h->misc.p;
    BfState *bs = (BfState *)((char *)h->misc.p + 64);
    int32_t *hcounts = (int32_t *)h->pinned;
    BfState *hbs = (BfState *)((char *)h->pinned + 64);
    hipStream_t s = h->stream;
    BfOpts bo;
    bo.xtol = o->xtol; bo.gtol = o->gtol; bo.ls_alpha = o->ls_alpha; bo.ls_factor = o->ls_factor;
    bo.max_evals = o->max_evals; bo.ls_max_evals = o->ls_max_evals; bo.use_line_search = o->use_line_search ? 1 : 0;
    bo.rc_divergent = NLH_DIVERGENT_BEHAVIOR_ERROR; bo.rc_convergence = NLH_CONVERGENCE_ERROR; bo.rc_invalid_op = NLH_INVALID_OPERATION_ERROR;
    bo.pad0 = bo.pad1 = 0;
    const int pb = (nprob + 255) / 256;
    const bool echo = o->print_status && nprob == 1;
    const int bs1 = std::min(1024, ((n + 63) / 64) * 64);
    const size_t bstride = sizeof(BfState) / sizeof(double), istride = sizeof(BfState) / sizeof(int32_t);
    static_assert(sizeof(BfState) % sizeof(double) == 0, "BfState is read through strided double / int pointers");
    const double *fp_all = &bs[0].fp, *temp_all = &bs[0].temp;
    const int32_t *iter_all = &bs[0].iter;
    const LmState *cst = st;

    hipLaunchKernelGGL(k_bfl_reset, dim3(pb), dim3(256), 0, s, nprob, st, bs, dinfo);
    if ((rc = residual_eval(h, rs, nprob, m, n, dx, dfv, nullptr, st, BF_START))) return rc;      // :633
    hipLaunchKernelGGL(k_bfl_start, dim3(nprob), dim3(256), 0, s, m, (const double *)dfv, st, bs, scalar ? 1 : 0);
    int need_grad = nprob;                                       // upper bound until the first read-back
    // a round costs every live problem an evaluation at least (a trial point, or an iteration's first one)
    const long max_rounds = (long)o->max_evals + (long)o->ls_max_evals + 16;
    for (long round = 0; round < max_rounds; ++round) {
        if (need_grad > 0) {
            // fnh_grad_fcn: n perturbed evaluations, (f_j - f) / h_j (src/nonlin_multi_var.f90:182-246)
            if (scalar) {
                hipLaunchKernelGGL(k_bfl_gather_fp, dim3(pb), dim3(256), 0, s, nprob, (const BfState *)bs, df0);
                if ((rc = residual_jacobian(h, rs, nprob, 1, n, dx, df0, dg, nullptr, st, BF_GRAD, false, false, true))) return rc;
            } else {
                launch_dq_panel(h, nprob, m, n, rs.dA, rs.db, rs.gamma, dx, dP, st, BF_GRAD);
                hipLaunchKernelGGL(k_bf_fd_gradient, dim3(n, nprob), dim3(64), 0, s, m, n, (const double *)dP, (const double *)dx, 0.0, dg,
                                   fp_all, bstride, cst, (int)BF_GRAD);
            }
            hipLaunchKernelGGL(k_bfl_after_grad, dim3(nprob), dim3(256), 0, s, n, bo, (const double *)dx, (const double *)dg, (const double *)dgold,
                               ddx, dy, dxnew, st, bs);
            // :703-712: R = temp I in the first iteration, B = R^T R, B dx
            hipLaunchKernelGGL(k_bf_scaled_identity, dim3((unsigned)((nn + 255) / 256), nprob), dim3(256), 0, s, n, 0.0, dR, temp_all, bstride,
                               iter_all, istride, cst, (int)BF_UPD_A);
            hipLaunchKernelGGL(k_bf_rtr, dim3((n + 255) / 256, n, nprob), dim3(256), 0, s, n, (const double *)dR, dB, cst, (int)BF_UPD_A);
            hipLaunchKernelGGL(k_matvec_cm, dim3((n + 255) / 256, nprob), dim3(256), sizeof(double) * n, s, n, n, (const double *)dB,
                               (const double *)ddx, dbdx, cst, (int)BF_UPD_A);
            hipLaunchKernelGGL(k_bfl_split, dim3(nprob), dim3(256), 0, s, n, (const double *)ddx, (const double *)dbdx, (const double *)dy, du, dv, st, bs);
            // :716-722: R^T R += u u^T, then -= v v^T
            if (n <= 1024) hipLaunchKernelGGL(k_bf_chol_update<1>, dim3(nprob), dim3(bs1), sizeof(double) * 2 * n, s, n, dR, (const double *)du, cst, (int)BF_UPD_RANK);
            else if (qn_nc8(n)) hipLaunchKernelGGL(k_bf_chol_update<8>, dim3(nprob), dim3(1024), sizeof(double) * 2 * n, s, n, dR, (const double *)du, cst, (int)BF_UPD_RANK);
                else hipLaunchKernelGGL(k_bf_chol_update<4>, dim3(nprob), dim3(1024), sizeof(double) * 2 * n, s, n, dR, (const double *)du, cst, (int)BF_UPD_RANK);
            hipLaunchKernelGGL(k_bf_solve_upper_t, dim3(nprob), dim3(bs1), sizeof(double) * n, s, n, (const double *)dR, dv, cst, (int)BF_UPD_RANK);
            hipLaunchKernelGGL(k_bf_downdate_rot, dim3(nprob), dim3(64), 0, s, n, dv, dc, dinfo, cst, (int)BF_UPD_RANK);
            hipLaunchKernelGGL(k_bf_downdate_apply, dim3((n + 255) / 256, nprob), dim3(256), sizeof(double) * 2 * n, s, n, dR, (const double *)dc,
                               (const double *)dv, (const int *)dinfo, cst, (int)BF_UPD_RANK);
            hipLaunchKernelGGL(k_nt_advance, dim3(pb), dim3(256), 0, s, nprob, st, (int)BF_UPD_RANK, (int)BF_DIR);
            // :724: R = chol(B)
            if (n <= 1024) launch_bf_chol_blocked(s, nprob, n, (const double *)dB, dR, dinfo, cst, (int)BF_UPD_FACTOR);
            else if (qn_nc8(n)) hipLaunchKernelGGL(k_bf_chol_factor<8>, dim3(nprob), dim3(1024), sizeof(double) * n, s, n, (const double *)dB, dR, dinfo, cst, (int)BF_UPD_FACTOR);
                else hipLaunchKernelGGL(k_bf_chol_factor<4>, dim3(nprob), dim3(1024), sizeof(double) * n, s, n, (const double *)dB, dR, dinfo, cst, (int)BF_UPD_FACTOR);
            hipLaunchKernelGGL(k_nt_advance, dim3(pb), dim3(256), 0, s, nprob, st, (int)BF_UPD_FACTOR, (int)BF_DIR);
            // :727: dx = -(R^T R)^-1 g
            hipLaunchKernelGGL(k_bfl_neg, dim3((n + 255) / 256, nprob), dim3(256), 0, s, n, (const double *)dg, dw, cst);
            hipLaunchKernelGGL(k_bf_solve_upper_t, dim3(nprob), dim3(bs1), sizeof(double) * n, s, n, (const double *)dR, dw, cst, (int)BF_DIR);
            hipLaunchKernelGGL(k_qn_solve_upper, dim3(nprob), dim3(bs1), sizeof(double) * n, s, n, (const double *)dR, dw, nn, (size_t)n, cst, (int)BF_DIR);
            hipLaunchKernelGGL(k_bfl_dir_done, dim3(nprob), dim3(256), 0, s, n, bo, (const double *)dx, (const double *)dg, ddx, (const double *)dw,
                               dxnew, (const int32_t *)dinfo, st, bs);
        }
        if ((rc = residual_eval(h, rs, nprob, m, n, dxnew, dfv, nullptr, st, BF_TRIAL))) return rc;
        hipLaunchKernelGGL(k_bfl_trial, dim3(nprob), dim3(256), 0, s, m, n, bo, dx, dxnew, ddx, (const double *)dg, dgold, (const double *)dfv, st, bs,
                           scalar ? 1 : 0);
        hipLaunchKernelGGL(k_bfl_count, dim3(1), dim3(256), 0, s, nprob, cst, dcounts);
        HIPCHK(h, hipMemcpyAsync(hcounts, dcounts, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (echo) HIPCHK(h, hipMemcpyAsync(hbs, bs, sizeof(BfState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (echo && need_grad > 0 && hbs[0].print_due) {         // :730-737
            printf(" \n");
            printf("Iteration: %d\n", hbs[0].pr_iter);
            printf("Function Evaluations: %d\n", hbs[0].pr_neval);
            char e1[16], e2[16], e3[16];
            format_e10_3(hbs[0].pr_fp, e1); format_e10_3(hbs[0].pr_xtest, e2); format_e10_3(hbs[0].pr_gtest, e3);
            printf("Function Value: %s\nChange in Variable: %s\nGradient: %s\n", e1, e2, e3);
        }
        need_grad = hcounts[0];
        if (need_grad == 0 && hcounts[1] == 0) break;
    }
    HIPCHK(h, hipMemcpyAsync(hbs, bs, sizeof(BfState) * np, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    for (int p = 0; p < nprob; ++p) {
        const BfState &c = hbs[p];
        if (ib) {                                                // :751-759
            ib[p].iter_count = c.iter; ib[p].fcn_count = c.neval; ib[p].jacobian_count = 0; ib[p].gradient_count = c.ngrad;
            ib[p].converge_on_fcn = 0; ib[p].converge_on_chng = c.xcnvrg; ib[p].converge_on_zero_diff = c.gcnvrg;
        }
        const bool finished = c.rc || c.flag || c.xcnvrg || c.gcnvrg;
        if (status) status[p] = c.rc ? c.rc : ((c.flag || !finished) ? NLH_CONVERGENCE_ERROR : 0);   // :765-767
        if (hfout) hfout[p] = c.fp;                              // :762
    }
    return 0;
}

int nlh_dq_bfgs_solve_batch(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n, const double *dA,
                            const double *db, double gamma, double *dx, double *hfout, nlh_iteration_behavior *ib,
                            int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!o || n < 1 || m < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    static const int bfgs_host = [] { const char *e = getenv("NLH_BFGS_HOSTLOOP"); return e ? atoi(e) : 0; }();
    if (!bfgs_host) {
        ResidualSource rs;
        rs.dA = dA; rs.db = db; rs.gamma = gamma;
        return lockstep_slices(nprob, [&](int32_t p0, int32_t cnt) {
            return bfgs_lockstep(h, o, cnt, m, n, rs.shifted(p0, m, n), false, dx + (size_t)p0 * n,
                                 hfout ? hfout + p0 : nullptr, ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
        });
    }
    // one problem per call; run_problems deals the problems to worker threads with private handles
    auto solve_one = [&](nlh_handle *h, int p) -> int {
        int rc;
        const size_t mn = (size_t)m * n;
        if ((rc = ensure(h, h->P, sizeof(double) * mn))) return rc;
        if ((rc = ensure(h, h->xdev, sizeof(double) * 2 * n))) return rc;
        if ((rc = ensure(h, h->wa4, sizeof(double) * m))) return rc;
        hipStream_t s = h->stream;
        std::vector<double> x(n), f(m);
        const double *A = dA + (size_t)p * mn, *b = db + (size_t)p * m;
        double *dxp = dx + (size_t)p * n;
        double *dxs = (double *)h->xdev.p, *dgs = dxs + n, *dfs = (double *)h->wa4.p;
        HIPCHK(h, hipMemcpyAsync(x.data(), dxp, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        BfgsEval ev;
        ev.fcn = [&](const double *xx, double *fv) -> int {
            HIPCHK(h, hipMemcpyAsync(dxs, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
            launch_dq_residual(h, 1, m, n, A, b, gamma, dxs, dfs, nullptr, nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(f.data(), dfs, sizeof(double) * m, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            *fv = 0.5 * h_dot(m, f.data(), f.data());
            return 0;
        };
        ev.grad = [&](double *xx, double fv, double *g) -> int {
            HIPCHK(h, hipMemcpyAsync(dxs, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
            launch_dq_panel(h, 1, m, n, A, b, gamma, dxs, (double *)h->P.p, nullptr, -1);
            hipLaunchKernelGGL(k_bf_fd_gradient, dim3(n), dim3(64), 0, s, m, n, (const double *)h->P.p, dxs, fv, dgs, (const double *)nullptr, (size_t)0, (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(g, dgs, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            return 0;
        };
        nlh_iteration_behavior lib;
        memset(&lib, 0, sizeof lib);
        double fo = 0.0;
        rc = bfgs_core(h, o, n, ev, x.data(), &fo, &lib);
        if (rc < 0) return rc;
        if (ib) ib[p] = lib;
        if (status) status[p] = rc;
        if (hfout) hfout[p] = fo;
        HIPCHK(h, hipMemcpyAsync(dxp, x.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    };
    const int rcb = run_problems(h, nprob, solve_one);
    if (rcb) return rcb;
    HIPCHK(h, hipGetLastError());
    return 0;
}

// bfgs%solve on a batch of problems whose objective is the USER'S device fcnnvar: a launcher of the nlh_device_vecfcn type
// with m = 1 (dF[npoints] = f at each point); gradfcn (optional, nlh_device_jacfcn with m = 1: dJ[npoints][n] = the
// gradients) replaces the forward differences as fcnnvar_helper%gradient does (src/nonlin_multi_var.f90:213-217).
int nlh_bfgs_solve_batch_device(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t n, nlh_device_vecfcn fcn,
                                nlh_device_jacfcn gradfcn, void *ctx, double *dx, double *hfout, nlh_iteration_behavior *ib,
                                int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib && nprob > 0) memset(ib, 0, sizeof(*ib) * (size_t)nprob);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // src/nonlin_optimize.f90:611-615
    if (!o || n < 1 || (nprob > 0 && !dx)) return NLH_INVALID_INPUT_ERROR;
    if (nprob <= 0) return 0;
    HIPCHK(h, hipSetDevice(h->device));
    nlh_options oq = *o;
    if (nprob > 1) oq.print_status = 0;
    ResidualSource rs;
    rs.fcn = fcn; rs.jac = gradfcn; rs.ctx = ctx;
    // (a launcher is asked for nprob * n points at once)
    const int32_t slice = (int32_t)std::max<int64_t>(1, std::min<int64_t>(NLH_MAX_LOCKSTEP, ((int64_t)1 << 30) / n));
    for (int32_t p0 = 0; p0 < nprob; p0 += slice) {
        const int32_t cnt = std::min<int32_t>(slice, nprob - p0);
        const int rc = bfgs_lockstep(h, &oq, cnt, 1, n, rs.shifted(p0, 1, n), true, dx + (size_t)p0 * n, hfout ? hfout + p0 : nullptr,
                                     ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
        if (rc) return rc;
    }
    return 0;
}

// The same behind a host array.
int nlh_bfgs_solve_batch_device_h(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t n, nlh_device_vecfcn fcn,
                                  nlh_device_jacfcn gradfcn, void *ctx, double *x, double *fout, nlh_iteration_behavior *ib,
                                  int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (nprob <= 0) return 0;
    if (!x || !o || n < 1) return NLH_INVALID_INPUT_ERROR;
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;
    int rc;
    HIPCHK(h, hipSetDevice(h->device));
    if ((rc = ensure(h, h->xdev, sizeof(double) * (size_t)nprob * n))) return rc;
    double *dx = (double *)h->xdev.p;
    HIPCHK(h, hipMemcpyAsync(dx, x, sizeof(double) * (size_t)nprob * n, hipMemcpyHostToDevice, h->stream));
    if ((rc = nlh_bfgs_solve_batch_device(h, o, nprob, n, fcn, gradfcn, ctx, dx, fout, ib, status))) return rc;
    HIPCHK(h, hipMemcpyAsync(x, dx, sizeof(double) * (size_t)nprob * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// cholesky_rank1_update / cholesky_rank1_downdate stand-ins (call sites src/nonlin_optimize.f90:721-722): in place on the
// row-major upper factor dRt (n x n); du is consumed.  *hinfo = 1 if the downdate would lose positive definiteness.
int nlh_chol_rank1(nlh_handle *h, int32_t n, int32_t downdate, double *dRt, double *du, int32_t *hinfo)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (n < 1) return NLH_INVALID_INPUT_ERROR;
    if (n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->bfV, sizeof(double) * ((size_t)6 * n + 8)))) return rc;
    double *dc = (double *)h->bfV.p;
    int *dinfo = (int *)(dc + n);
    hipStream_t s = h->stream;
    const int bs1 = std::min(1024, ((n + 63) / 64) * 64);
    int info = 0;
    if (!downdate) {
        if (n <= 1024) hipLaunchKernelGGL(k_bf_chol_update<1>, dim3(1), dim3(bs1), sizeof(double) * 2 * n, s, n, dRt, du, (const LmState *)nullptr, -1);
        else if (qn_nc8(n)) hipLaunchKernelGGL(k_bf_chol_update<8>, dim3(1), dim3(1024), sizeof(double) * 2 * n, s, n, dRt, du, (const LmState *)nullptr, -1);
                else hipLaunchKernelGGL(k_bf_chol_update<4>, dim3(1), dim3(1024), sizeof(double) * 2 * n, s, n, dRt, du, (const LmState *)nullptr, -1);
    } else {
        hipLaunchKernelGGL(k_bf_solve_upper_t, dim3(1), dim3(bs1), sizeof(double) * n, s, n, dRt, du, (const LmState *)nullptr, -1);
        hipLaunchKernelGGL(k_bf_downdate_rot, dim3(1), dim3(64), 0, s, n, du, dc, dinfo, (const LmState *)nullptr, -1);
        hipLaunchKernelGGL(k_bf_downdate_apply, dim3((n + 255) / 256), dim3(256), sizeof(double) * 2 * n, s, n, dRt, dc, du, dinfo, (const LmState *)nullptr, -1);
        HIPCHK(h, hipMemcpyAsync(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(h, hipStreamSynchronize(s));
    if (hinfo) *hinfo = info;
    HIPCHK(h, hipGetLastError());
    return 0;
}
