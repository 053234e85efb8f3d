// nlh_cls.hip -- constrained_least_squares_solver: cls_solve (src/nonlin_least_squares.f90:938-1176: bounded trust
// region, Coleman-Li scaling, dog-leg, projected backtracking) for one problem with host callbacks and as a lock-step
// device state machine for batches.
#include "nlh_internal.h"
#include "nlh_kernels_broyden.h"
#include "nlh_kernels_cls.h"
#include "nlh_kernels_exact.h"

void nlh_cls_init_device(int lds_max) { broyden_kernel_attrs(lds_max); }



// ===========================================================================
// Constrained least squares: cls_solve (src/nonlin_least_squares.f90:938-1176) as a host loop.
// Device: FD / analytic Jacobian, Householder QR of J with the reflectors applied to f
// (qr_factor :1047 + solve_qr :1334), the triangular solve, J^T f and the two J v products of dogleg
// (:1331, :1341, :1398).  Host, in the reference's order: Coleman-Li scaling, the dog-leg logic,
// alpha_box, the reduction ratio, the trust-region update and the Armijo fallback.
// ===========================================================================
struct ClsEval {
    std::function<int(const double *x, double *f)> fcn;                          // F at host x -> host f
    std::function<int(double *x, const double *f0_host, double *dJ)> jac;        // Jacobian at host x -> device J (m x n)
};

static double cls_scaled_norm(int n, const double *x, const double *s, double *tmp)   // :1263-1273
{
    for (int i = 0; i < n; ++i) tmp[i] = x[i] * s[i];
    return h_norm2(n, tmp);
}

static bool cls_is_finite(int n, const double *x)               // :1276-1298
{
    for (int i = 0; i < n; ++i) {
        if (!(x[i] == x[i])) return false;
        if (fabs(x[i]) == DBL_MAX) return false;
    }
    return true;
}

static void cls_apply_limits(int n, const double *xl, const double *xu, double *x)   // :858-883
{
    for (int i = 0; i < n; ++i) if (x[i] < xl[i]) x[i] = xl[i];
    for (int i = 0; i < n; ++i) if (x[i] > xu[i]) x[i] = xu[i];
}

static double cls_alpha_box(int n, const double *x, const double *p, const double *xl, const double *xu)   // :1181-1219
{
    double rst = DBL_MAX;
    for (int i = 0; i < n; ++i) {
        if (p[i] > 0.0) {
            if (xu[i] < x[i]) return 0.0;
            const double a = (xu[i] - x[i]) / p[i];
            if (a < rst) rst = a;
        } else if (p[i] < 0.0) {
            if (xl[i] > x[i]) return 0.0;
            const double a = (xl[i] - x[i]) / p[i];
            if (a < rst) rst = a;
        }
    }
    if (rst < 0.0) rst = 0.0;
    return rst;
}

static int cls_core(nlh_handle *h, const nlh_options *o, double delta0, double stepscale0, const double *xl_in,
                    const double *xu_in, int m, int n, ClsEval &ev, double *x, double *fvec, nlh_iteration_behavior *ib)
{
    int rc;
    const size_t mn = (size_t)m * n;
    if ((rc = ensure(h, h->J, sizeof(double) * mn))) return rc;
    if ((rc = ensure(h, h->W2, sizeof(double) * mn))) return rc;
    if ((rc = ensure(h, h->qnV, sizeof(double) * ((size_t)6 * m + 6 * n + 16)))) return rc;
    double *dJ = (double *)h->J.p, *dW = (double *)h->W2.p, *dv = (double *)h->qnV.p;
    double *dE = dv, *dfv = dv + m, *dJv = dv + 2 * m, *vbuf = dv + 3 * m /* 2m */, *dgv = dv + 5 * m, *dvec = dgv + n,
           *dstep = dvec + n, *wbuf = dstep + n /* 2 (n + 1) */, *st = wbuf + 2 * (n + 1);
    hipStream_t s = h->stream;
    std::vector<double> xl(n), xu(n), sc(n), g(n), p(n), xnew(n), tmp(n), pgn(n), psd(n), u(n), v(n), Jg(m), Jp(m), fnew(m);
    int converged = 0, xcnvrg = 0, fcnvrg = 0, gcnvrg = 0, neval = 0, iter = 0, njac = 0;
    double xnorm, fnorm, gnorm, fnewnorm, actred = 0, prered = 0, rho = 0, delta, stepscale, dderiv;
    for (int i = 0; i < n; ++i) {                               // :999-1009
        xl[i] = xl_in ? xl_in[i] : -DBL_MAX;
        xu[i] = xu_in ? xu_in[i] : DBL_MAX;
    }
    cls_apply_limits(n, xl.data(), xu.data(), x);               // :1023-1031
    if ((rc = ev.fcn(x, fvec))) return rc;
    neval = 1;
    fnorm = h_norm2(m, fvec);
    xnorm = h_norm2(n, x);
    if (!cls_is_finite(n, x) || !cls_is_finite(m, fvec)) {
        if (ib) { ib->iter_count = 0; ib->fcn_count = 0; ib->jacobian_count = 0; }
        return 0;                                               // the reference returns silently here
    }
    auto matvec = [&](const double *vec, double *out) -> int {  // out = J vec (dgemv 'N')
        HIPCHK(h, hipMemcpyAsync(dvec, vec, sizeof(double) * n, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_matvec_cm, dim3((m + 255) / 256, 1), dim3(256), sizeof(double) * n, s, m, n, dJ, dvec, dJv, (const LmState *)nullptr, -1);
        HIPCHK(h, hipMemcpyAsync(out, dJv, sizeof(double) * m, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    };

    delta = delta0;                                             // :1034
    iter = 1;
    rc = 0;
    for (;;) {                                                  // :1036-1160
        if ((rc = ev.jac(x, fvec, dJ))) break;                  // :1038
        njac += 1;
        if (o->print_status) print_status(iter, neval, njac, xnorm, fnorm);

        // QR of J with the reflectors applied to f (:1047, :1334), g = J^T f (:1331), u = R^-1 (Q^T f)(1:n)
        HIPCHK(h, hipMemcpyAsync(dfv, fvec, sizeof(double) * m, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(dE, dfv, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
        {
            dim3 grid((m + 31) / 32, (n + 31) / 32, 1);
            hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, s, m, n, dJ, dW, n, (const LmState *)nullptr, -1);
        }
        hipLaunchKernelGGL(k_qn_col0, dim3((m + 255) / 256, 1), dim3(256), 0, s, m, n, dW, vbuf);
        launch_house_steps(h, 1, m, n, 1, dW, dE, vbuf, wbuf, st);
        HIPCHK(h, hipMemcpyAsync(dstep, dE, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_qn_solve_upper, dim3(1), dim3(std::min(1024, ((n + 63) / 64) * 64)), sizeof(double) * n, s, n, dW, dstep, (size_t)m * n, (size_t)n, (const LmState *)nullptr, -1);
        hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, 1), dim3(256), 0, s, m, n, dJ, dfv, dgv, 1.0, (const LmState *)nullptr, -1);
        HIPCHK(h, hipMemcpyAsync(u.data(), dstep, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(g.data(), dgv, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));

        // coleman_li_scaling, :1222-1260
        for (int i = 0; i < n; ++i) {
            const double big = DBL_MAX;
            double di;
            if (xl[i] > -big && xu[i] < big) di = fmin(x[i] - xl[i], xu[i] - x[i]);
            else if (xl[i] > -big) di = x[i] - xl[i];
            else if (xu[i] < big) di = xu[i] - x[i];
            else di = 1.0;
            di = fmax(di, 1.0e-8);
            sc[i] = 1.0 / di;
            if (sc[i] > 1.0e8) sc[i] = 1.0e8;
        }
        // dogleg, :1301-1403
        for (int i = 0; i < n; ++i) pgn[i] = -u[i];
        const double pgnnorm = cls_scaled_norm(n, pgn.data(), sc.data(), tmp.data());
        if (pgnnorm > delta) {
            if ((rc = matvec(g.data(), Jg.data()))) break;
            const double c1 = h_dot(n, g.data(), g.data()), c2 = h_dot(m, Jg.data(), Jg.data());
            const double alpha = (c2 > 0.0 && c1 > 0.0) ? c1 / c2 : 0.0;
            for (int i = 0; i < n; ++i) psd[i] = -alpha * g[i];
            const double psdnorm = cls_scaled_norm(n, psd.data(), sc.data(), tmp.data());
            if (psdnorm >= delta && psdnorm > 0.0) {
                const double f1 = delta / psdnorm;
                for (int i = 0; i < n; ++i) p[i] = f1 * psd[i];
            } else {
                for (int i = 0; i < n; ++i) u[i] = pgn[i] - psd[i];
                for (int i = 0; i < n; ++i) u[i] = sc[i] * u[i];
                for (int i = 0; i < n; ++i) v[i] = sc[i] * psd[i];
                const double a = h_dot(n, u.data(), u.data());
                const double b = 2.0 * h_dot(n, u.data(), v.data());
                const double c = h_dot(n, v.data(), v.data()) - delta * delta;
                if (a <= 0.0) {
                    for (int i = 0; i < n; ++i) p[i] = psd[i];
                } else {
                    const double arg = fmax(0.0, b * b - 4.0 * a * c);
                    double t;
                    if (arg == 0.0) {
                        t = -b / (2.0 * a);
                    } else {
                        t = (-b + sqrt(arg)) / (2.0 * a);
                        if (t < 0.0 || t > 1.0) t = (-b - sqrt(arg)) / (2.0 * a);
                    }
                    t = fmax(0.0, fmin(1.0, t));
                    for (int i = 0; i < n; ++i) p[i] = psd[i] + t * u[i];
                }
            }
        } else {
            for (int i = 0; i < n; ++i) p[i] = pgn[i];
        }
        {
            const double alpha = cls_alpha_box(n, x, p.data(), xl.data(), xu.data());   // :1392-1395
            if (alpha < 1.0)
                for (int i = 0; i < n; ++i) p[i] = alpha * p[i];
        }
        if ((rc = matvec(p.data(), Jp.data()))) break;          // :1398
        prered = -h_dot(n, g.data(), p.data()) - 0.5 * h_dot(m, Jp.data(), Jp.data());

        xnorm = cls_scaled_norm(n, p.data(), sc.data(), tmp.data());   // :1055-1057
        gnorm = h_norm2(n, g.data());
        for (int i = 0; i < n; ++i) xnew[i] = x[i] + p[i];
        if ((rc = ev.fcn(xnew.data(), fnew.data()))) break;     // :1060-1062
        fnewnorm = h_norm2(m, fnew.data());
        neval += 1;

        actred = 0.5 * (fnorm * fnorm - fnewnorm * fnewnorm);   // :1065-1070
        rho = (prered > 0.0 && actred >= 0.0) ? actred / prered : 0.0;
        if (rho < 0.25) delta = fmax(0.25, 1.0e-12);            // :1073-1077 (constant 0.25: as in the reference)
        else if (rho > 0.75 && fabs(xnorm - delta) < 1.0e-12 * delta) delta = fmin(2.0 * delta, 1.0e3);

        if (rho > 0.1 && fnewnorm <= fnorm) {                   // :1080-1086
            memcpy(x, xnew.data(), sizeof(double) * n);
            cls_apply_limits(n, xl.data(), xu.data(), x);
            memcpy(fvec, fnew.data(), sizeof(double) * m);
            fnorm = fnewnorm;
            iter += 1;
        } else {                                                // :1088-1123
            dderiv = h_dot(n, g.data(), p.data());
            if (dderiv >= 0.0) {
                delta = fmax(0.5 * delta, 1.0e-12);
            } else {
                stepscale = stepscale0;
                int k;
                for (k = 1; k <= 10; ++k) {
                    for (int i = 0; i < n; ++i) xnew[i] = x[i] + stepscale * p[i];
                    cls_apply_limits(n, xl.data(), xu.data(), xnew.data());
                    if ((rc = ev.fcn(xnew.data(), fnew.data()))) break;
                    neval += 1;
                    fnewnorm = h_norm2(m, fnew.data());
                    if (fnewnorm <= fnorm + 1.0e-4 * stepscale * dderiv) {
                        memcpy(x, xnew.data(), sizeof(double) * n);
                        memcpy(fvec, fnew.data(), sizeof(double) * m);
                        fnorm = fnewnorm;
                        iter += 1;
                        delta = fmax(stepscale * xnorm, 1.0e-12);
                        break;
                    }
                    stepscale = stepscale * 0.5;
                }
                if (rc) break;
                if (k > 10) delta = fmax(0.5 * delta, 1.0e-12);
            }
        }
        if (!cls_is_finite(n, x) || !cls_is_finite(m, fvec)) break;     // :1125-1127
        if (xnorm <= o->xtol) { converged = 1; xcnvrg = 1; break; }     // :1130-1149
        if (fabs(actred) <= o->ftol && fabs(prered) <= o->ftol && 0.5 * rho <= 1.0) { converged = 1; fcnvrg = 1; break; }
        if (gnorm <= o->gtol) { converged = 1; gcnvrg = 1; break; }
        if (neval >= o->max_evals) break;
    }
    if (ib) {                                                   // :1163-1170
        ib->iter_count = iter; ib->fcn_count = neval; ib->jacobian_count = njac; ib->gradient_count = 0;
        ib->converge_on_fcn = fcnvrg; ib->converge_on_chng = xcnvrg; ib->converge_on_zero_diff = gcnvrg;
    }
    if (rc) return rc;
    return converged ? 0 : NLH_CONVERGENCE_ERROR;               // :1173-1175
}

// constrained_least_squares_solver%solve -- cls_solve, src/nonlin_least_squares.f90:938-1176
int nlh_cls_solve(nlh_handle *h, const nlh_options *o, double delta0, double stepscale0, const double *xl,
                  const double *xu, int32_t m, int32_t n, nlh_vecfcn fcn, nlh_jacfcn jacfcn, void *ctx, double *x,
                  double *fvec, nlh_iteration_behavior *ib)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib) memset(ib, 0, sizeof *ib);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :988
    if (!o || n < 1 || m < 1) return NLH_INVALID_INPUT_ERROR;
    if (n > m) return NLH_UNDERDEFINED_PROBLEM_ERROR;           // :989
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    const size_t mn = (size_t)m * n;
    if ((rc = ensure(h, h->P, sizeof(double) * mn))) return rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
    if ((rc = ensure(h, h->wa4, sizeof(double) * m))) return rc;
    if ((rc = ensure_pinned(h, sizeof(double) * (mn + m)))) return rc;
    double *hP = (double *)h->pinned;
    hipStream_t s = h->stream;
    ClsEval ev;
    ev.fcn = [&](const double *xx, double *ff) -> int { fcn(ctx, n, xx, m, ff); return 0; };
    ev.jac = [&](double *xx, const double *f0, double *dJ) -> int {
        if (jacfcn) {
            jacfcn(ctx, n, xx, m, hP);
            HIPCHK(h, hipMemcpyAsync(dJ, hP, sizeof(double) * mn, hipMemcpyHostToDevice, s));
            HIPCHK(h, hipStreamSynchronize(s));
            return 0;
        }
        for (int j = 0; j < n; ++j) {                           // vfh_jac_fcn :267-273
            const double temp = xx[j];
            double hh = NLH_SQRT_EPS * fabs(temp);
            if (hh == 0.0) hh = NLH_SQRT_EPS;
            xx[j] = temp + hh;
            fcn(ctx, n, xx, m, hP + (size_t)j * m);
            xx[j] = temp;
        }
        HIPCHK(h, hipMemcpyAsync(h->P.p, hP, sizeof(double) * mn, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->wa4.p, f0, sizeof(double) * m, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->xdev.p, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
        launch_fd(h, 1, m, n, (const double *)h->P.p, (const double *)h->wa4.p, (const double *)h->xdev.p, dJ, nullptr, -1);
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    };
    rc = cls_core(h, o, delta0, stepscale0, xl, xu, m, n, ev, x, fvec, ib);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return NLH_ERR_HIP; }
    return rc;
}

// constrained_least_squares_solver%solve for a batch of device-model problems: the lock-step state machine of
// nlh_kernels_cls.h.  A round takes every problem that wants a Jacobian through J (forward differences, fused into the
// panel kernel), the QR of J with the reflectors applied to f, the Gauss-Newton step, the gradient and the dog-leg (with
// J g and J p where the step needs them) to its trial point, evaluates F there and runs the ratio test; problems inside
// the projected backtracking get one more trial point evaluated.  One 8-byte read-back per round.
static int cls_lockstep(nlh_handle *h, const nlh_options *o, double delta0, double stepscale0, const double *xl_in,
                        const double *xu_in, int32_t nprob, int32_t m, int32_t n, const ResidualSource &rs,
                        double *dx, double *dfvec, nlh_iteration_behavior *ib, int32_t *status)
{   // rs: the built-in dense-quadratic family, or a user's device vecfcn / jacobianfcn launchers (nlh_devfcn.hip)
    int rc;
    const size_t mn = (size_t)m * n, np = (size_t)nprob;
    if ((rc = ensure(h, h->J, sizeof(double) * mn * np))) return rc;
    if ((rc = ensure(h, h->W2, sizeof(double) * mn * np))) return rc;
    const size_t per = 5 * (size_t)m + 7 * (size_t)n + 2 * ((size_t)n + 1) + 8;
    if ((rc = ensure(h, h->qnV, sizeof(double) * (per * np + 2 * (size_t)n)))) return rc;
    if ((rc = ensure(h, h->state, sizeof(LmState) * np))) return rc;
    if ((rc = ensure(h, h->misc, sizeof(ClState) * np + 64))) return rc;
    if ((rc = ensure_pinned(h, sizeof(ClState) * np + 64 + sizeof(double) * 2 * (size_t)n))) return rc;
    double *dJ = (double *)h->J.p, *dW = (double *)h->W2.p, *q = (double *)h->qnV.p;
    double *dE = q; q += (size_t)m * np;                         // the QR's extra column, then u in its head
    double *dJv = q; q += (size_t)m * np;                        // J g, later J p
    double *dfnew = q; q += (size_t)m * np;
    double *vbuf = q; q += 2 * (size_t)m * np;
    double *dg = q; q += (size_t)n * np;
    double *dsc = q; q += (size_t)n * np;
    double *dpgn = q; q += (size_t)n * np;
    double *dpsd = q; q += (size_t)n * np;
    double *du = q; q += (size_t)n * np;
    double *dp = q; q += (size_t)n * np;
    double *dxnew = q; q += (size_t)n * np;
    double *wbuf = q; q += 2 * ((size_t)n + 1) * np;
    double *st2 = q; q += 8 * np;
    double *dxl = q, *dxu = q + n;
    LmState *st = (LmState *)h->state.p;
    int32_t *dcounts = (int32_t *)h->misc.p;
    ClState *cs = (ClState *)((char *)h->misc.p + 64);
    int32_t *hcounts = (int32_t *)h->pinned;
    ClState *hcs = (ClState *)((char *)h->pinned + 64);
    double *hb = (double *)((char *)h->pinned + 64 + sizeof(ClState) * np);
    hipStream_t s = h->stream;
    for (int i = 0; i < n; ++i) {                                // :999-1009
        hb[i] = xl_in ? xl_in[i] : -DBL_MAX;
        hb[n + i] = xu_in ? xu_in[i] : DBL_MAX;
    }
    HIPCHK(h, hipMemcpyAsync(dxl, hb, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    ClOpts co;
    co.ftol = o->ftol; co.xtol = o->xtol; co.gtol = o->gtol; co.delta0 = delta0; co.stepscale0 = stepscale0;
    co.max_evals = o->max_evals; co.pad = 0;
    const int pb = (nprob + 255) / 256;
    const bool echo = o->print_status && nprob == 1;
    const int bsn = std::min(1024, ((n + 63) / 64) * 64);

    hipLaunchKernelGGL(k_cls_reset, dim3(pb), dim3(256), 0, s, nprob, delta0, st, cs);
    hipLaunchKernelGGL(k_cls_limits, dim3((n + 255) / 256, nprob), dim3(256), 0, s, n, (const double *)dxl, (const double *)dxu, dx);   // :1023
    if ((rc = residual_eval(h, rs, nprob, m, n, dx, dfvec, nullptr, st, CL_START))) return rc;
    hipLaunchKernelGGL(k_cls_start, dim3(nprob), dim3(256), 0, s, m, n, (const double *)dx, (const double *)dfvec, st, cs);
    int need_jac = nprob;                                        // upper bound until the first read-back
    // a round is an iteration or one backtracking trial: every one of them costs its problem an evaluation
    const long max_rounds = (long)o->max_evals + 16;
    for (long round = 0; round < max_rounds; ++round) {
        if (need_jac > 0) {
            // :1038 fcn%jacobian: forward differences (the built-in family: fused into the panel kernel) or the user's jacobianfcn
            if ((rc = residual_jacobian(h, rs, nprob, m, n, dx, dfvec, dJ, nullptr, st, CL_NEED_JAC, false, true, true))) return rc;
            hipLaunchKernelGGL(k_cls_qr_prep, dim3((m + 255) / 256, nprob), dim3(256), 0, s, m, (const double *)dfvec, dE, (const LmState *)st);
            {
                dim3 grid((m + 31) / 32, (n + 31) / 32, nprob);
                hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, s, m, n, (const double *)dJ, dW, n, (const LmState *)st, (int)CL_NEED_JAC);
            }
            hipLaunchKernelGGL(k_qn_col0, dim3((m + 255) / 256, nprob), dim3(256), 0, s, m, n, (const double *)dW, vbuf);
            launch_house_steps(h, nprob, m, n, 1, dW, dE, vbuf, wbuf, st2, st, CL_NEED_JAC);
            hipLaunchKernelGGL(k_qn_solve_upper, dim3(nprob), dim3(bsn), sizeof(double) * n, s, n, (const double *)dW, dE, mn, (size_t)m,
                               (const LmState *)st, (int)CL_NEED_JAC);
            hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, nprob), dim3(256), 0, s, m, n, (const double *)dJ, (const double *)dfvec, dg, 1.0,
                               (const LmState *)st, (int)CL_NEED_JAC);
            hipLaunchKernelGGL(k_cls_dog1, dim3(nprob), dim3(256), 0, s, m, n, (const double *)dx, (const double *)dxl, (const double *)dxu,
                               (const double *)dE, dsc, dpgn, dp, st, cs);
            hipLaunchKernelGGL(k_matvec_cm, dim3((m + 255) / 256, nprob), dim3(256), sizeof(double) * n, s, m, n, (const double *)dJ,
                               (const double *)dg, dJv, (const LmState *)st, (int)CL_DOG_SD);
            hipLaunchKernelGGL(k_cls_dog2, dim3(nprob), dim3(256), 0, s, m, n, (const double *)dx, (const double *)dxl, (const double *)dxu,
                               (const double *)dg, (const double *)dJv, (const double *)dsc, (const double *)dpgn, dpsd, du, dp, st, cs);
            hipLaunchKernelGGL(k_matvec_cm, dim3((m + 255) / 256, nprob), dim3(256), sizeof(double) * n, s, m, n, (const double *)dJ,
                               (const double *)dp, dJv, (const LmState *)st, (int)CL_PRED);
            hipLaunchKernelGGL(k_cls_pred, dim3(nprob), dim3(256), 0, s, m, n, (const double *)dx, (const double *)dg, (const double *)dp,
                               (const double *)dJv, (const double *)dsc, dxnew, st, cs);
            if ((rc = residual_eval(h, rs, nprob, m, n, dxnew, dfnew, nullptr, st, CL_TRIAL))) return rc;
            hipLaunchKernelGGL(k_cls_judge, dim3(nprob), dim3(256), 0, s, m, n, co, dx, dxnew, dfvec, (const double *)dfnew, (const double *)dg,
                               (const double *)dp, (const double *)dxl, (const double *)dxu, st, cs);
        }
        // (a problem that has just entered the backtracking gets its first point evaluated in the same round)
        if ((rc = residual_eval(h, rs, nprob, m, n, dxnew, dfnew, nullptr, st, CL_BT))) return rc;
        hipLaunchKernelGGL(k_cls_bt, dim3(nprob), dim3(256), 0, s, m, n, co, dx, dxnew, dfvec, (const double *)dfnew, (const double *)dp,
                           (const double *)dxl, (const double *)dxu, st, cs);
        hipLaunchKernelGGL(k_cls_count, dim3(1), dim3(256), 0, s, nprob, (const LmState *)st, dcounts);
        HIPCHK(h, hipMemcpyAsync(hcounts, dcounts, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (echo) HIPCHK(h, hipMemcpyAsync(hcs, cs, sizeof(ClState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (echo && need_jac > 0 && hcs[0].print_due) print_status(hcs[0].pr_iter, hcs[0].pr_neval, hcs[0].pr_njac, hcs[0].pr_xnorm, hcs[0].pr_fnorm);
        need_jac = hcounts[0];
        if (need_jac == 0 && hcounts[1] == 0) break;
    }
    HIPCHK(h, hipMemcpyAsync(hcs, cs, sizeof(ClState) * np, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    for (int p = 0; p < nprob; ++p) {
        const ClState &c = hcs[p];
        if (ib) {                                                // :1163-1170 (a non-finite start leaves them zero)
            memset(&ib[p], 0, sizeof ib[p]);
            if (!c.silent) {
                ib[p].iter_count = c.iter; ib[p].fcn_count = c.neval; ib[p].jacobian_count = c.njac;
                ib[p].converge_on_fcn = c.fcnvrg; ib[p].converge_on_chng = c.xcnvrg; ib[p].converge_on_zero_diff = c.gcnvrg;
            }
        }
        if (status) status[p] = (c.silent || c.converged) ? 0 : NLH_CONVERGENCE_ERROR;      // :1173-1175
    }
    return 0;
}

int nlh_dq_cls_solve_batch(nlh_handle *h, const nlh_options *o, double delta0, double stepscale0, const double *xl,
                           const double *xu, int32_t nprob, int32_t m, int32_t n, const double *dA, const double *db,
                           double gamma, double *dx, double *dfvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!o || n < 1 || m < 1) return NLH_INVALID_INPUT_ERROR;
    if (n > m) return NLH_UNDERDEFINED_PROBLEM_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    static const int cls_host = [] { const char *e = getenv("NLH_CLS_HOSTLOOP"); return e ? atoi(e) : 0; }();
    if (!cls_host) {
        ResidualSource rs;
        rs.dA = dA; rs.db = db; rs.gamma = gamma;
        return lockstep_slices(nprob, [&](int32_t p0, int32_t cnt) {
            return cls_lockstep(h, o, delta0, stepscale0, xl, xu, cnt, m, n, rs.shifted(p0, m, n),
                                dx + (size_t)p0 * n, dfvec + (size_t)p0 * m, ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
        });
    }
    // one problem per call; run_problems deals the problems to worker threads with private handles
    auto solve_one = [&](nlh_handle *h, int p) -> int {
        int rc;
        const size_t mn = (size_t)m * n;
        if ((rc = ensure(h, h->P, sizeof(double) * mn))) return rc;
        if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
        if ((rc = ensure(h, h->wa4, sizeof(double) * m))) return rc;
        hipStream_t s = h->stream;
        std::vector<double> x(n), f(m);
        const double *A = dA + (size_t)p * mn, *b = db + (size_t)p * m;
        double *dxp = dx + (size_t)p * n, *dfp = dfvec + (size_t)p * m;
        double *dxs = (double *)h->xdev.p, *dfs = (double *)h->wa4.p;
        HIPCHK(h, hipMemcpyAsync(x.data(), dxp, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        ClsEval ev;
        ev.fcn = [&](const double *xx, double *ff) -> int {
            HIPCHK(h, hipMemcpyAsync(dxs, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
            launch_dq_residual(h, 1, m, n, A, b, gamma, dxs, dfs, nullptr, nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(ff, dfs, sizeof(double) * m, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            return 0;
        };
        ev.jac = [&](double *xx, const double *f0, double *dJ) -> int {
            HIPCHK(h, hipMemcpyAsync(dxs, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
            HIPCHK(h, hipMemcpyAsync(dfs, f0, sizeof(double) * m, hipMemcpyHostToDevice, s));
            launch_dq_panel(h, 1, m, n, A, b, gamma, dxs, dJ, nullptr, -1, dfs);      // fused FD column write
            return 0;
        };
        nlh_iteration_behavior lib;
        memset(&lib, 0, sizeof lib);
        rc = cls_core(h, o, delta0, stepscale0, xl, xu, m, n, ev, x.data(), f.data(), &lib);
        if (rc < 0) return rc;
        if (ib) ib[p] = lib;
        if (status) status[p] = rc;
        HIPCHK(h, hipMemcpyAsync(dxp, x.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(dfp, f.data(), sizeof(double) * m, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    };
    const int rcb = run_problems(h, nprob, solve_one);
    if (rcb) return rcb;
    HIPCHK(h, hipGetLastError());
    return 0;
}


// constrained_least_squares_solver%solve (cls_solve, src/nonlin_least_squares.f90:938-1176) on a batch of problems whose
// residual is the USER'S device function (launchers, include/nonlin_hip.h): the lock-step state machine above.
int nlh_cls_solve_batch_device(nlh_handle *h, const nlh_options *o, double delta0, double stepscale0, const double *xl, const double *xu,
                               int32_t nprob, int32_t m, int32_t n, nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx, double *dx,
                               double *dfvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib && nprob > 0) memset(ib, 0, sizeof(*ib) * (size_t)nprob);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :988
    if (nprob <= 0) return 0;
    if (!o || n < 1 || m < 1 || !dx || !dfvec) return NLH_INVALID_INPUT_ERROR;
    if (n > m) return NLH_UNDERDEFINED_PROBLEM_ERROR;           // :989
    HIPCHK(h, hipSetDevice(h->device));
    ResidualSource rs;
    rs.fcn = fcn; rs.jac = jacfcn; rs.ctx = ctx;
    nlh_options oq = *o;
    if (nprob > 1) oq.print_status = 0;
    const int32_t slice = (int32_t)std::max<int64_t>(1, std::min<int64_t>(NLH_MAX_LOCKSTEP, ((int64_t)1 << 30) / n));
    for (int32_t p0 = 0; p0 < nprob; p0 += slice) {
        const int32_t cnt = std::min<int32_t>(slice, nprob - p0);
        const int rc = cls_lockstep(h, &oq, delta0, stepscale0, xl, xu, cnt, m, n, rs.shifted(p0, m, n), dx + (size_t)p0 * n,
                                    dfvec + (size_t)p0 * m, ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
        if (rc) return rc;
    }
    return 0;
}

int nlh_cls_solve_batch_device_h(nlh_handle *h, const nlh_options *o, double delta0, double stepscale0, const double *xl, const double *xu,
                                 int32_t nprob, int32_t m, int32_t n, nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx, double *x,
                                 double *fvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (nprob <= 0) return 0;
    if (!o || !x || !fvec || n < 1 || m < 1) return NLH_INVALID_INPUT_ERROR;
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * (size_t)nprob * n))) return rc;
    if ((rc = ensure(h, h->fdev, sizeof(double) * (size_t)nprob * m))) return rc;
    double *dx = (double *)h->xdev.p, *df = (double *)h->fdev.p;
    HIPCHK(h, hipMemcpyAsync(dx, x, sizeof(double) * (size_t)nprob * n, hipMemcpyHostToDevice, h->stream));
    if ((rc = nlh_cls_solve_batch_device(h, o, delta0, stepscale0, xl, xu, nprob, m, n, fcn, jacfcn, ctx, dx, df, ib, status))) return rc;
    HIPCHK(h, hipMemcpyAsync(x, dx, sizeof(double) * (size_t)nprob * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(fvec, df, sizeof(double) * (size_t)nprob * m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}
