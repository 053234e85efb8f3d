/*
 * nonlin_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See nonlin_oracle.h for scope, pinning status and arithmetic conventions.
 * Every routine cites the reference lines it restates (paths relative to
 * /root/reference).  Indices are 0-based here; the reference is 1-based.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include "nonlin_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define A_(a, lda, i, j) ((a)[(size_t)(j) * (size_t)(lda) + (size_t)(i)])

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

void nlo_default_options(nlo_options *o)
{
    o->max_evals = 100;        /* src/nonlin_multi_eqn_mult_var.f90:69 */
    o->ftol = 1.0e-8;          /* :71 */
    o->xtol = 1.0e-12;         /* :73 */
    o->gtol = 1.0e-12;         /* :75 */
    o->print_status = 0;       /* :77 */
    o->factor = 100.0;         /* src/nonlin_least_squares.f90:25 */
    o->use_line_search = 1;    /* src/nonlin_solve.f90:30 */
    o->ls_max_evals = 100;     /* src/nonlin_linesearch.f90:35 */
    o->ls_alpha = 1.0e-4;      /* :38 */
    o->ls_factor = 0.1;        /* :46 */
}

/* Fortran intrinsics as frozen for this project (header, "Arithmetic conventions").
 * NORM2 is processor-dependent in Fortran.  This is the algorithm of the flang
 * runtime shipped with ROCm 7.2 (amdflang 22, _FortranANorm2_8): a running
 * maximum with a rescaled sum of squared ratios, result max*sqrt(1+sum).  It
 * was checked bit-for-bit against amdflang's NORM2 on 300 random vectors
 * (tests/golden/make_norm2_vectors.f90 -> tests/golden/norm2_flang.json). */
static int g_norm2_mode = 0;    /* 0 = flang runtime algorithm (default), 1 = sqrt(sequential sum of squares) */
void nlo_set_norm2_mode(int mode) { g_norm2_mode = mode; }

double nlo_norm2(int32_t n, const double *x)
{
    if (g_norm2_mode == 1) {
        double q = 0.0;
        for (int32_t i = 0; i < n; ++i) q = q + x[i] * x[i];
        return sqrt(q);
    }
    double mx = 0.0, s = 0.0;
    for (int32_t i = 0; i < n; ++i) {
        double a = fabs(x[i]);
        if (mx == 0.0) {
            mx = a;
        } else if (a > mx) {
            double t = mx / a, tsq = t * t;
            s = s * tsq;
            s = s + tsq;
            mx = a;
        } else if (a != 0.0) {
            double t = a / mx;
            s = s + t * t;
        }
    }
    return mx * sqrt(1.0 + s);
}

double nlo_dot(int32_t n, const double *x, const double *y)
{
    double s = 0.0;
    for (int32_t i = 0; i < n; ++i) s = s + x[i] * y[i];
    return s;
}

/* print_status: src/nonlin_helper.f90:17-33 (formats A,I0 and A,E10.3). */
static void fmt_e10_3(char *buf, size_t nbuf, double v)
{
    /* Fortran E10.3: 0.ddddE+ee */
    if (v == 0.0) { snprintf(buf, nbuf, " 0.000E+00"); return; }
    int e = (int)floor(log10(fabs(v))) + 1;
    double mant = v / pow(10.0, e);
    if (fabs(mant) >= 0.9995) { mant /= 10.0; e += 1; }
    snprintf(buf, nbuf, "%s0.%03dE%c%02d", mant < 0 ? "-" : " ",
             (int)floor(fabs(mant) * 1000.0 + 0.5), e < 0 ? '-' : '+', abs(e));
}

static void print_status(int iter, int nfeval, int njaceval, double xnorm, double fnorm)
{
    char b1[32], b2[32];
    fmt_e10_3(b1, sizeof b1, xnorm);
    fmt_e10_3(b2, sizeof b2, fnorm);
    printf(" \n");
    printf("Iteration: %d\n", iter);
    printf("Function Evaluations: %d\n", nfeval);
    if (njaceval > 0) printf("Jacobian Evaluations: %d\n", njaceval);
    printf("Change in Variable: %s\n", b1);
    printf("Residual: %s\n", b2);
}

/* ---------------------------------------------------------------------------
 * vfh_jac_fcn: src/nonlin_multi_eqn_mult_var.f90:198-277
 * ------------------------------------------------------------------------- */
int nlo_fd_jacobian(nlo_vecfcn fcn, nlo_jacfcn jac_or_null, void *ctx,
                    int32_t m, int32_t n, double *x, const double *fv, double *jac)
{
    if (!fcn) return NLO_UNDEFINED_FUNCTION_ERROR;           /* :240 */
    if (jac_or_null) {                                       /* :241-243 */
        jac_or_null(ctx, n, x, m, jac);
        return 0;
    }
    int lwork = fv ? m : 2 * m;                              /* :246-250 */
    double *wrk = (double *)malloc(sizeof(double) * (size_t)(lwork > 0 ? lwork : 1)); /* :253 */
    double *f1 = wrk;                                        /* :254 */
    const double *f0;
    if (fv) {
        f0 = fv;                                             /* :256 */
    } else {
        fcn(ctx, n, x, m, wrk + m);                          /* :258-259 */
        f0 = wrk + m;
    }
    const double eps = sqrt(DBL_EPSILON);                    /* :263-264 */
    for (int32_t j = 0; j < n; ++j) {                        /* :267-275 */
        double temp = x[j];
        double h = eps * fabs(temp);
        if (h == 0.0) h = eps;
        x[j] = temp + h;
        fcn(ctx, n, x, m, f1);
        x[j] = temp;
        for (int32_t i = 0; i < m; ++i)
            A_(jac, m, i, j) = (f1[i] - f0[i]) / h;          /* :274, true division */
    }
    free(wrk);
    return 0;
}

/* ---------------------------------------------------------------------------
 * lmfactor (MINPACK qrfac): src/nonlin_least_squares.f90:569-667
 * ipvt is 0-based here.
 * ------------------------------------------------------------------------- */
void nlo_lmfactor(int32_t m, int32_t n, double *a, int32_t lda, int32_t pivot,
                  int32_t *ipvt, double *rdiag, double *acnorm, double *wa)
{
    const double p05 = 5.0e-2, epsmch = DBL_EPSILON;
    const int32_t minmn = m < n ? m : n;

    for (int32_t j = 0; j < n; ++j) {                        /* :611-616 */
        acnorm[j] = nlo_norm2(m, &A_(a, lda, 0, j));
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
        if (pivot) ipvt[j] = j;
    }

    for (int32_t j = 0; j < minmn; ++j) {                    /* :619-666 */
        if (pivot) {
            int32_t kmax = j;                                /* :622-625 */
            for (int32_t k = j; k < n; ++k)
                if (rdiag[k] > rdiag[kmax]) kmax = k;
            if (kmax != j) {                                 /* :626-637 */
                for (int32_t i = 0; i < m; ++i) {
                    double t = A_(a, lda, i, j);
                    A_(a, lda, i, j) = A_(a, lda, i, kmax);
                    A_(a, lda, i, kmax) = t;
                }
                rdiag[kmax] = rdiag[j];
                wa[kmax] = wa[j];
                int32_t k = ipvt[j];
                ipvt[j] = ipvt[kmax];
                ipvt[kmax] = k;
            }
        }

        double ajnorm = nlo_norm2(m - j, &A_(a, lda, j, j));  /* :642 */
        if (ajnorm != 0.0) {
            if (A_(a, lda, j, j) < 0.0) ajnorm = -ajnorm;    /* :644 */
            for (int32_t i = j; i < m; ++i)                  /* :645 */
                A_(a, lda, i, j) = A_(a, lda, i, j) / ajnorm;
            A_(a, lda, j, j) = A_(a, lda, j, j) + 1.0;       /* :646 */

            for (int32_t k = j + 1; k < n; ++k) {            /* :652-662 */
                double sm = 0.0;
                for (int32_t i = j; i < m; ++i)
                    sm = sm + A_(a, lda, i, j) * A_(a, lda, i, k);
                double temp = sm / A_(a, lda, j, j);
                for (int32_t i = j; i < m; ++i)
                    A_(a, lda, i, k) = A_(a, lda, i, k) - temp * A_(a, lda, i, j);
                if (!pivot || rdiag[k] == 0.0) continue;
                temp = A_(a, lda, j, k) / rdiag[k];
                rdiag[k] = rdiag[k] * sqrt(dmax(0.0, 1.0 - temp * temp));
                double q = rdiag[k] / wa[k];
                if (p05 * (q * q) > epsmch) continue;
                rdiag[k] = nlo_norm2(m - (j + 1), &A_(a, lda, j + 1, k));
                wa[k] = rdiag[k];
            }
        }
        rdiag[j] = -ajnorm;                                  /* :665 */
    }
}

/* ---------------------------------------------------------------------------
 * lmsolve (MINPACK qrsolv): src/nonlin_least_squares.f90:670-791
 * r is the leading n-by-n block of the Jacobian workspace (leading dim ldr).
 * ------------------------------------------------------------------------- */
void nlo_lmsolve(int32_t n, double *r, int32_t ldr, const int32_t *ipvt,
                 const double *diag, const double *qtb, double *x, double *sdiag,
                 double *wa)
{
    const double qtr = 0.25, half = 0.5;

    for (int32_t j = 0; j < n; ++j) {                        /* :710-714 */
        for (int32_t i = j; i < n; ++i) A_(r, ldr, i, j) = A_(r, ldr, j, i);
        x[j] = A_(r, ldr, j, j);
        wa[j] = qtb[j];
    }

    for (int32_t j = 0; j < n; ++j) {                        /* :717-765 */
        int32_t l = ipvt[j];
        if (diag[l] != 0.0) {
            for (int32_t k = j; k < n; ++k) sdiag[k] = 0.0;
            sdiag[j] = diag[l];
            double qtbpj = 0.0;
            for (int32_t k = j; k < n; ++k) {
                double cs, sn;
                if (sdiag[k] == 0.0) continue;
                if (fabs(A_(r, ldr, k, k)) < fabs(sdiag[k])) {   /* :733-741 */
                    double ctan = A_(r, ldr, k, k) / sdiag[k];
                    sn = half / sqrt(qtr + qtr * (ctan * ctan));
                    cs = sn * ctan;
                } else {
                    double tn = sdiag[k] / A_(r, ldr, k, k);
                    cs = half / sqrt(qtr + qtr * (tn * tn));
                    sn = cs * tn;
                }
                A_(r, ldr, k, k) = cs * A_(r, ldr, k, k) + sn * sdiag[k];   /* :745 */
                double temp = cs * wa[k] + sn * qtbpj;
                qtbpj = -sn * wa[k] + cs * qtbpj;
                wa[k] = temp;
                for (int32_t i = k + 1; i < n; ++i) {        /* :753-757 */
                    temp = cs * A_(r, ldr, i, k) + sn * sdiag[i];
                    sdiag[i] = -sn * A_(r, ldr, i, k) + cs * sdiag[i];
                    A_(r, ldr, i, k) = temp;
                }
            }
        }
        sdiag[j] = A_(r, ldr, j, j);                         /* :763-764 */
        A_(r, ldr, j, j) = x[j];
    }

    int32_t nsing = n;                                       /* :769-773 */
    for (int32_t j = 0; j < n; ++j) {
        if (sdiag[j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0.0;
    }
    for (int32_t k = 1; k <= nsing; ++k) {                   /* :774-784 */
        int32_t j = nsing - k;
        double sm = 0.0;
        for (int32_t i = j + 1; i < nsing; ++i) sm = sm + A_(r, ldr, i, j) * wa[i];
        wa[j] = (wa[j] - sm) / sdiag[j];
    }
    for (int32_t j = 0; j < n; ++j) x[ipvt[j]] = wa[j];      /* :787-790 */
}

/* ---------------------------------------------------------------------------
 * lmpar (MINPACK lmpar + two live deviations): src/nonlin_least_squares.f90:394-566
 *   deviation A (:531): norm2 over all m entries of wa2 (caller's wa4)
 *   deviation B (:552): the whole wa1 vector is updated, not only rows j+1..n
 * ------------------------------------------------------------------------- */
/* Test-only switch (tests/test_oracle.py, tests/golden/make_minpack_vectors.py): 1 restores MINPACK's own two lines --
 * the norm over n entries at :531 and the update of rows j+1..n at :552 -- so that everything else in lmpar / lmsolve /
 * the reject path of lss_solve can be checked against MINPACK itself (scipy's lmder).  0 (default) = the reference.
 * g_lmpar_loops counts how often the iteration :522-563 was entered (the fixtures must reach it). */
static int g_lmpar_minpack = 0;
static long g_lmpar_loops = 0;
void nlo_set_lmpar_minpack(int on) { g_lmpar_minpack = on; }
long nlo_lmpar_loop_entries(int reset) { long v = g_lmpar_loops; if (reset) g_lmpar_loops = 0; return v; }

void nlo_lmpar(int32_t m, int32_t n, double *r, int32_t ldr, const int32_t *ipvt,
               const double *diag, const double *qtb, double delta, double *par,
               double *x, double *sdiag, double *wa1, double *wa2)
{
    const double p001 = 1.0e-3, p1 = 0.1;
    const double dwarf = DBL_MIN;                            /* tiny(dwarf), :442 */
    int32_t nsing = n;
    double dxnorm, fp, gnorm, parc, parl, paru, sm, temp;

    for (int32_t j = 0; j < n; ++j) {                        /* :447-451 */
        wa1[j] = qtb[j];
        if (A_(r, ldr, j, j) == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0.0;
    }
    for (int32_t k = 1; k <= nsing; ++k) {                   /* :453-463 */
        int32_t j = nsing - k;
        wa1[j] = wa1[j] / A_(r, ldr, j, j);
        temp = wa1[j];
        for (int32_t i = 0; i < j; ++i) wa1[i] = wa1[i] - A_(r, ldr, i, j) * temp;
    }
    for (int32_t j = 0; j < n; ++j) x[ipvt[j]] = wa1[j];     /* :466-469 */

    int32_t iter = 0;                                        /* :473-481 */
    for (int32_t i = 0; i < n; ++i) wa2[i] = diag[i] * x[i];
    dxnorm = nlo_norm2(n, wa2);
    fp = dxnorm - delta;
    if (fp <= p1 * delta) {
        if (iter == 0) *par = 0.0;
        return;
    }

    parl = 0.0;                                              /* :486-503 */
    if (nsing == n) {
        for (int32_t j = 0; j < n; ++j) {
            int32_t l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int32_t j = 0; j < n; ++j) {
            sm = 0.0;
            if (j >= 1) sm = nlo_dot(j, &A_(r, ldr, 0, j), wa1);
            wa1[j] = (wa1[j] - sm) / A_(r, ldr, j, j);
        }
        temp = nlo_norm2(n, wa1);
        parl = ((fp / delta) / temp) / temp;
    }

    for (int32_t j = 0; j < n; ++j) {                        /* :506-513 */
        sm = nlo_dot(j + 1, &A_(r, ldr, 0, j), qtb);
        wa1[j] = sm / diag[ipvt[j]];
    }
    gnorm = nlo_norm2(n, wa1);
    paru = gnorm / delta;
    if (paru == 0.0) paru = dwarf / dmin(delta, p1);

    *par = dmax(*par, parl);                                 /* :517-519 */
    *par = dmin(*par, paru);
    if (*par == 0.0) *par = gnorm / dxnorm;

    g_lmpar_loops += 1;
    for (;;) {                                               /* :522-563 */
        iter = iter + 1;
        if (*par == 0.0) *par = dmax(dwarf, p001 * paru);
        temp = sqrt(*par);
        for (int32_t i = 0; i < n; ++i) wa1[i] = temp * diag[i];
        nlo_lmsolve(n, r, ldr, ipvt, wa1, qtb, x, sdiag, wa2);
        for (int32_t i = 0; i < n; ++i) wa2[i] = diag[i] * x[i];
        dxnorm = nlo_norm2(g_lmpar_minpack ? n : m, wa2);    /* :531 deviation A (MINPACK: n) */
        temp = fp;
        fp = dxnorm - delta;
#ifdef NLO_TRACE
        fprintf(stderr, "[lmpar] iter=%d par=%.17g parl=%.6g paru=%.6g dxnorm=%.17g fp=%.6g delta=%.6g\n",
                iter, *par, parl, paru, dxnorm, fp, delta);
#endif

        if (fabs(fp) <= p1 * delta ||
            (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;  /* :538-540 */

        for (int32_t j = 0; j < n; ++j) {                    /* :543-546 */
            int32_t l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int32_t j = 0; j < n; ++j) {                    /* :547-553 */
            wa1[j] = wa1[j] / sdiag[j];
            temp = wa1[j];
            if (n < j + 2) continue;
            for (int32_t i = g_lmpar_minpack ? j + 1 : 0; i < n; ++i)   /* :552 deviation B (MINPACK: rows j+1..n) */
                wa1[i] = wa1[i] - A_(r, ldr, i, j) * temp;
        }
        temp = nlo_norm2(n, wa1);
        parc = ((fp / delta) / temp) / temp;

        if (fp > 0.0) parl = dmax(parl, *par);               /* :558-559 */
        if (fp < 0.0) paru = dmin(paru, *par);
        *par = dmax(parl, *par + parc);                      /* :562 */
    }
    /* :564 `if (iter == zero) par = zero` is dead: iter >= 1 here. */
}

/* ---------------------------------------------------------------------------
 * lss_solve (MINPACK lmdif, mode-1 scaling): src/nonlin_least_squares.f90:118-391
 * ------------------------------------------------------------------------- */
/* TEST-ONLY switch (tests/golden/make_fast_policy_study.py): replaces lmfactor (:225) and the Q^T f sweep (:241-253) of
 * lss_solve by a caller-supplied factorisation, so that the study "can ANY factorisation other than the reference's own
 * operation order meet 1e-10 on these problem families?" runs the rest of the reference's iteration unchanged.  The hook
 * receives J (column-major, lda = m) and f and must leave: the n x n upper triangle of jac = R (diagonal included), jpvt
 * (0-based), rdiag = diag(R), acnorm = the column norms of J, qtf (n), and wa4 (m) with wa4(0:n) = qtf and a tail whose
 * norm is ||(Q^T f)(n+1:m)|| (deviation A reads it).  Never set by anything the product or the parity tests run. */
static nlo_factor_hook g_factor_hook = NULL;
static void *g_factor_hook_ctx = NULL;
void nlo_set_factor_hook(nlo_factor_hook hook, void *hctx) { g_factor_hook = hook; g_factor_hook_ctx = hctx; }

int nlo_lm_solve(const nlo_options *opt, nlo_vecfcn fcn, nlo_jacfcn jac_or_null,
                 void *ctx, int32_t m, int32_t n, double *x, double *fvec,
                 nlo_iteration_behavior *ib)
{
    const double p0001 = 1.0e-4, p1 = 0.1, qtr = 0.25, half = 0.5, p75 = 0.75, one = 1.0;
    int xcnvrg = 0, fcnvrg = 0, gcnvrg = 0;
    int32_t neval = 0, iter = 0, njac = 0, flag = 0;
    double fac = opt->factor, ftol = opt->ftol, xtol = opt->xtol, gtol = opt->gtol;
    int32_t maxeval = opt->max_evals;
    const double eps = DBL_EPSILON;
    double fnorm, par, xnorm = 0.0, delta = 0.0, sm, temp = 0.0, gnorm = 0.0, pnorm, fnorm1,
           actred, temp1, temp2, prered, dirder, ratio;

    if (ib) {                                                /* :177-185 */
        memset(ib, 0, sizeof *ib);
    }
    if (!fcn) return NLO_UNDEFINED_FUNCTION_ERROR;           /* :188 */
    if (n > m) return NLO_UNDERDEFINED_PROBLEM_ERROR;        /* :189 */

    int32_t *jpvt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));   /* :199-208 */
    double *jac = (double *)malloc(sizeof(double) * (size_t)m * (size_t)(n > 0 ? n : 1));
    double *diag = (double *)calloc((size_t)(6 * n + m + 1), sizeof(double));
    double *qtf = diag + n, *wa1 = qtf + n, *wa2 = wa1 + n, *wa3 = wa2 + n, *wa4 = wa3 + n;

    fcn(ctx, n, x, m, fvec);                                 /* :211-213 */
    neval = 1;
    fnorm = nlo_norm2(m, fvec);

    par = 0.0;                                               /* :216-218 */
    iter = 1;
    flag = 0;
    for (;;) {                                               /* outer loop :219-375 */
        nlo_fd_jacobian(fcn, jac_or_null, ctx, m, n, x, fvec, jac);   /* :221 */
        njac = njac + 1;

        if (g_factor_hook) g_factor_hook(g_factor_hook_ctx, m, n, jac, m, fvec, jpvt, wa1, wa2, qtf, wa4);   /* test-only, see above */
        else
        nlo_lmfactor(m, n, jac, m, 1, jpvt, wa1, wa2, wa3);  /* :225 */

        if (iter == 1) {                                     /* :229-238 */
            for (int32_t j = 0; j < n; ++j) {
                diag[j] = wa2[j];
                if (wa2[j] == 0.0) diag[j] = one;
            }
            for (int32_t j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
            xnorm = nlo_norm2(n, wa3);
            delta = fac * xnorm;
            if (delta == 0.0) delta = fac;
        }

        if (!g_factor_hook) {
        memcpy(wa4, fvec, sizeof(double) * (size_t)m);       /* :241-253 */
        for (int32_t j = 0; j < n; ++j) {
            if (A_(jac, m, j, j) != 0.0) {
                sm = 0.0;
                for (int32_t i = j; i < m; ++i) sm = sm + A_(jac, m, i, j) * wa4[i];
                temp = -sm / A_(jac, m, j, j);
                for (int32_t i = j; i < m; ++i) wa4[i] = wa4[i] + A_(jac, m, i, j) * temp;
            }
            A_(jac, m, j, j) = wa1[j];
            qtf[j] = wa4[j];
        }
        }

        gnorm = 0.0;                                         /* :256-267 */
        if (fnorm != 0.0) {
            for (int32_t j = 0; j < n; ++j) {
                int32_t l = jpvt[j];
                if (wa2[l] == 0.0) continue;
                sm = 0.0;
                for (int32_t i = 0; i <= j; ++i) sm = sm + A_(jac, m, i, j) * (qtf[i] / fnorm);
                gnorm = dmax(gnorm, fabs(sm / wa2[l]));
            }
        }

        if (gnorm <= gtol) {                                 /* :270-273 */
            gcnvrg = 1;
            break;
        }

        for (int32_t j = 0; j < n; ++j) diag[j] = dmax(diag[j], wa2[j]);   /* :276-278 */

        for (;;) {                                           /* inner loop :281-366 */
            nlo_lmpar(m, n, jac, m, jpvt, diag, qtf, delta, &par, wa1, wa2, wa3, wa4);  /* :283 */

            for (int32_t j = 0; j < n; ++j) {                /* :286-291 */
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            pnorm = nlo_norm2(n, wa3);

            if (iter == 1) delta = dmin(delta, pnorm);       /* :294 */

            fcn(ctx, n, wa2, m, wa4);                        /* :297-299 */
            neval = neval + 1;
            fnorm1 = nlo_norm2(m, wa4);

            actred = -one;                                   /* :302-303 */
            if (p1 * fnorm1 < fnorm) {
                double q = fnorm1 / fnorm;
                actred = one - q * q;
            }

            for (int32_t j = 0; j < n; ++j) {                /* :307-312 */
                wa3[j] = 0.0;
                int32_t l = jpvt[j];
                temp = wa1[l];
                for (int32_t i = 0; i <= j; ++i) wa3[i] = wa3[i] + A_(jac, m, i, j) * temp;
            }
            temp1 = nlo_norm2(n, wa3) / fnorm;               /* :313-316 */
            temp2 = (sqrt(par) * pnorm) / fnorm;
            prered = temp1 * temp1 + (temp2 * temp2) / half;
            dirder = -(temp1 * temp1 + temp2 * temp2);

            ratio = 0.0;                                     /* :319-320 */
            if (prered != 0.0) ratio = actred / prered;
#ifdef NLO_TRACE                                              /* build-time tracing: nothing in the timed loops */
            fprintf(stderr, "[nlo] iter=%d neval=%d fnorm=%.17g fnorm1=%.17g par=%.6g delta=%.6g pnorm=%.6g actred=%.6g prered=%.6g ratio=%.6g gnorm=%.3g\n",
                    iter, neval, fnorm, fnorm1, par, delta, pnorm, actred, prered, ratio, gnorm);
#endif

            if (ratio <= qtr) {                              /* :323-337 */
                if (actred >= 0.0) temp = half;
                if (actred < 0.0) temp = half * dirder / (dirder + half * actred);
                if (p1 * fnorm1 >= fnorm || temp < p1) temp = p1;
                delta = temp * dmin(delta, pnorm / p1);
                par = par / temp;
            } else {
                if (par != 0.0 && ratio < p75) {
                    /* no action */
                } else {
                    delta = pnorm / half;
                    par = half * par;
                }
            }

            if (ratio >= p0001) {                            /* :340-349 */
                for (int32_t j = 0; j < n; ++j) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                }
                memcpy(fvec, wa4, sizeof(double) * (size_t)m);
                xnorm = nlo_norm2(n, wa2);
                fnorm = fnorm1;
                iter = iter + 1;
            }

            if (fabs(actred) <= ftol && prered <= ftol && half * ratio <= one) fcnvrg = 1;  /* :352-355 */
            if (delta <= xtol * xnorm) xcnvrg = 1;
            if (fcnvrg || xcnvrg) break;

            if (neval >= maxeval) flag = NLO_CONVERGENCE_ERROR;    /* :358-363 */
            if (fabs(actred) <= eps && prered <= eps && half * ratio <= one)
                flag = NLO_TOLERANCE_TOO_SMALL_ERROR;
            if (delta <= eps * xnorm) flag = NLO_TOLERANCE_TOO_SMALL_ERROR;
            if (gnorm <= eps) flag = NLO_TOLERANCE_TOO_SMALL_ERROR;
            if (flag != 0) break;

            if (ratio >= p0001) break;                       /* :365 */
        }

        if (fcnvrg || xcnvrg || gcnvrg || flag != 0) break;  /* :369 */

        if (opt->print_status) print_status(iter, neval, njac, xnorm, fnorm);   /* :372-374 */
    }

    if (ib) {                                                /* :378-385 */
        ib->iter_count = iter;
        ib->fcn_count = neval;
        ib->jacobian_count = njac;
        ib->converge_on_fcn = fcnvrg;
        ib->converge_on_chng = xcnvrg;
        ib->converge_on_zero_diff = gcnvrg;
    }
    free(jpvt);
    free(jac);
    free(diag);
    return flag != 0 ? NLO_CONVERGENCE_ERROR : 0;            /* :388-390 */
}

/* ---------------------------------------------------------------------------
 * lu_factor / solve_lu stand-ins (call sites src/nonlin_solve.f90:570, 577).
 * The implementation is in the un-vendored jchristopherson/linalg (unpinned;
 * fpm.toml:15) which forwards to LAPACK dgetrf/dgetrs.  Restated here as the
 * published unblocked right-looking algorithm (LAPACK dgetf2 + dgetrs):
 * first-maximum partial pivoting, reciprocal scaling of the column, rank-1
 * update; then row interchanges, unit-lower forward and upper back solves.
 * PARITY-UNPINNED at the bit level.  ipvt is 0-based.
 * ------------------------------------------------------------------------- */
int nlo_lu_factor(int32_t n, double *a, int32_t lda, int32_t *ipvt)
{
    int info = 0;
    for (int32_t j = 0; j < n; ++j) {
        int32_t p = j;
        double best = fabs(A_(a, lda, j, j));
        for (int32_t i = j + 1; i < n; ++i) {
            double v = fabs(A_(a, lda, i, j));
            if (v > best) { best = v; p = i; }
        }
        ipvt[j] = p;
        if (A_(a, lda, p, j) != 0.0) {
            if (p != j)
                for (int32_t k = 0; k < n; ++k) {
                    double t = A_(a, lda, j, k);
                    A_(a, lda, j, k) = A_(a, lda, p, k);
                    A_(a, lda, p, k) = t;
                }
            double rcp = 1.0 / A_(a, lda, j, j);
            for (int32_t i = j + 1; i < n; ++i) A_(a, lda, i, j) = A_(a, lda, i, j) * rcp;
        } else if (info == 0) {
            info = j + 1;
        }
        for (int32_t k = j + 1; k < n; ++k) {
            double ujk = A_(a, lda, j, k);
            for (int32_t i = j + 1; i < n; ++i)
                A_(a, lda, i, k) = A_(a, lda, i, k) - A_(a, lda, i, j) * ujk;
        }
    }
    return info;
}

void nlo_lu_solve(int32_t n, const double *lu, int32_t lda, const int32_t *ipvt, double *b)
{
    for (int32_t j = 0; j < n; ++j) {
        int32_t p = ipvt[j];
        if (p != j) { double t = b[j]; b[j] = b[p]; b[p] = t; }
    }
    for (int32_t j = 0; j < n; ++j) {                 /* L y = Pb, unit diagonal */
        double bj = b[j];
        if (bj != 0.0)
            for (int32_t i = j + 1; i < n; ++i) b[i] = b[i] - bj * A_(lu, lda, i, j);
    }
    for (int32_t j = n - 1; j >= 0; --j) {            /* U x = y */
        if (b[j] != 0.0) {
            b[j] = b[j] / A_(lu, lda, j, j);
            double bj = b[j];
            for (int32_t i = 0; i < j; ++i) b[i] = b[i] - bj * A_(lu, lda, i, j);
        }
    }
}

/* ---------------------------------------------------------------------------
 * min_backtrack_search: src/nonlin_linesearch.f90:495-551
 * ------------------------------------------------------------------------- */
double nlo_min_backtrack_search(int32_t mode, double f0, double f, double f1,
                                double alam, double alam1, double slope)
{
    const double p5 = 0.5, two = 2.0, three = 3.0;
    double lam;
    if (mode == 1) {
        lam = -slope / (two * (f - f0 - slope));             /* :529 */
    } else {
        double rhs1 = f - f0 - alam * slope;                 /* :532-536 */
        double rhs2 = f1 - f0 - alam1 * slope;
        double a = (rhs1 / (alam * alam) - rhs2 / (alam1 * alam1)) / (alam - alam1);
        double b = (-alam1 * rhs1 / (alam * alam) + alam * rhs2 / (alam1 * alam1)) /
                   (alam - alam1);
        if (a == 0.0) {
            lam = -slope / (two * b);
        } else {
            double disc = b * b - three * a * slope;
            if (disc < 0.0)
                lam = p5 * alam;
            else if (b <= 0.0)
                lam = (-b + sqrt(disc)) / (three * a);
            else
                lam = -slope / (b + sqrt(disc));
        }
        if (lam > p5 * alam) lam = p5 * alam;                /* :549 */
    }
    return lam;
}

/* limit_search_vector: src/nonlin_linesearch.f90:554-572 */
void nlo_limit_search_vector(int32_t n, double *x, double lim)
{
    double mag = nlo_norm2(n, x);
    if (mag == 0.0) return;
    if (mag > lim) {
        double s = lim / mag;
        for (int32_t i = 0; i < n; ++i) x[i] = s * x[i];
    }
}

/* ---------------------------------------------------------------------------
 * ls_search_mimo: src/nonlin_linesearch.f90:152-326 (fold always supplied by
 * ns_solve, so the "evaluate at xold" branch :241-246 is not needed here).
 * ------------------------------------------------------------------------- */
int nlo_line_search(const nlo_options *opt, nlo_vecfcn fcn, void *ctx, int32_t m,
                    int32_t n, const double *xold, const double *grad,
                    const double *dir, double *x, double *fvec, double fold,
                    double *fx, nlo_iteration_behavior *ib)
{
    const double p5 = 0.5, one = 1.0, two = 2.0;
    int xcnvrg = 0, fcnvrg = 0;
    int32_t neval = 0, niter = 0, flag = 0;
    const double tolx = two * DBL_EPSILON;                   /* :209 */
    const double alpha = opt->ls_alpha, lambdamin = opt->ls_factor;
    const int32_t maxeval = opt->ls_max_evals;
    double alam, alam1 = 0.0, alamin, f1 = 0.0, slope, temp, test, tmplam, f = 0.0, fo;
    int rc = 0;

    if (fx) *fx = 0.0;
    if (ib) memset(ib, 0, sizeof *ib);
    if (!fcn) return NLO_UNDEFINED_FUNCTION_ERROR;

    fo = fold;                                               /* :239-240 */

    slope = nlo_dot(n, grad, dir);                           /* :249-253 */
    if (slope >= 0.0) return NLO_DIVERGENT_BEHAVIOR_ERROR;

    test = 0.0;                                              /* :256-262 */
    for (int32_t i = 0; i < n; ++i) {
        temp = fabs(dir[i]) / dmax(fabs(xold[i]), one);
        if (temp > test) test = temp;
    }
    alamin = tolx / test;
    alam = one;

    for (;;) {                                               /* :266-310 */
        for (int32_t i = 0; i < n; ++i) x[i] = xold[i] + alam * dir[i];
        fcn(ctx, n, x, m, fvec);
        f = p5 * nlo_dot(m, fvec, fvec);
        neval = neval + 1;
        niter = niter + 1;

        if (alam < alamin) {                                 /* :275-287 */
            double s = 0.0;
            for (int32_t i = 0; i < n; ++i) { double d = x[i] - xold[i]; s = s + d * d; }
            if (sqrt(s) == 0.0) { rc = NLO_CONVERGENCE_ERROR; break; }
            for (int32_t i = 0; i < n; ++i) x[i] = xold[i];
            xcnvrg = 1;
            break;
        } else if (f <= fo + alpha * alam * slope) {         /* :288-291 */
            fcnvrg = 1;
            break;
        } else {                                             /* :292-296 */
            tmplam = nlo_min_backtrack_search(niter, fo, f, f1, alam, alam1, slope);
        }

        alam1 = alam;                                        /* :300-302 */
        f1 = f;
        alam = dmax(tmplam, lambdamin * alam);

        if (neval >= maxeval) { flag = 1; break; }           /* :305-309 */
    }
    if (fx) *fx = f;                                         /* :311 */

    if (ib) {                                                /* :314-320 */
        ib->iter_count = niter;
        ib->fcn_count = neval;
        ib->converge_on_fcn = fcnvrg;
        ib->converge_on_chng = xcnvrg;
        ib->converge_on_zero_diff = 0;
    }
    if (rc) return rc;
    return flag != 0 ? NLO_CONVERGENCE_ERROR : 0;            /* :323-325 */
}

/* ---------------------------------------------------------------------------
 * test_convergence: src/nonlin_helper.f90:36-124
 * ------------------------------------------------------------------------- */
void nlo_test_convergence(int32_t n, int32_t m, const double *x, const double *xo,
                          const double *f, const double *g, int32_t lg, double xtol,
                          double ftol, double gtol, int32_t *c, int32_t *cx,
                          int32_t *cf, int32_t *cg, double *xnorm, double *fnorm)
{
    const double one = 1.0, half = 0.5;
    *cx = 0; *cf = 0; *cg = 0; *c = 0;
    double fc = half * nlo_dot(m, f, f);
    *fnorm = 0.0;
    *xnorm = 0.0;

    for (int32_t i = 0; i < m; ++i) *fnorm = dmax(fabs(f[i]), *fnorm);   /* :87-94 */
    if (*fnorm < ftol) { *cf = 1; *c = 1; return; }

    for (int32_t i = 0; i < n; ++i) {                        /* :97-105 */
        double test = fabs(x[i] - xo[i]) / dmax(fabs(x[i]), one);
        *xnorm = dmax(test, *xnorm);
    }
    if (*xnorm < xtol) { *cx = 1; *c = 1; return; }

    if (lg) {                                                /* :108-118 */
        double test = 0.0;
        double den = dmax(fc, half * (double)n);
        for (int32_t i = 0; i < n; ++i) {
            double dxmax = fabs(g[i]) * dmax(fabs(x[i]), one) / den;
            test = dmax(test, dxmax);
        }
        if (test < gtol) *cg = 1;
    }
}

/* ---------------------------------------------------------------------------
 * ns_solve: src/nonlin_solve.f90:452-638
 * ------------------------------------------------------------------------- */
int nlo_newton_solve(const nlo_options *opt, nlo_vecfcn fcn, nlo_jacfcn jac_or_null,
                     void *ctx, int32_t n, double *x, double *fvec,
                     nlo_iteration_behavior *ib)
{
    const double half = 0.5, factor = 1.0e2;
    int32_t xcnvrg = 0, fcnvrg = 0, gcnvrg = 0, check = 0;
    int32_t neval = 0, iter = 0, njac = 0, flag = 0;
    const double ftol = opt->ftol, xtol = opt->xtol, gtol = opt->gtol;
    const int32_t maxeval = opt->max_evals;
    double f, fold, stpmax, xnorm, fnorm, temp, test;
    int rc = 0;
    nlo_iteration_behavior lib;

    if (ib) memset(ib, 0, sizeof *ib);                       /* :502-510 */
    if (!fcn) return NLO_UNDEFINED_FUNCTION_ERROR;           /* :518 */

    double *dir = (double *)calloc((size_t)(4 * n + 1), sizeof(double));   /* :529-534 */
    double *grad = dir + n, *xold = grad + n, *rhs = xold + n;
    double *jac = (double *)malloc(sizeof(double) * (size_t)n * (size_t)(n > 0 ? n : 1));
    int32_t *ipvt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));

    /* :535 -- Jacobian requested before fvec exists; result discarded, not
     * counted (Appendix A item 11).  Kept so callbacks fire in the same order. */
    nlo_fd_jacobian(fcn, jac_or_null, ctx, n, n, x, fvec, jac);

    fcn(ctx, n, x, n, fvec);                                 /* :538-547 */
    f = half * nlo_dot(n, fvec, fvec);
    neval = neval + 1;
    test = 0.0;
    for (int32_t i = 0; i < n; ++i) test = dmax(fabs(fvec[i]), test);
    if (test < ftol) fcnvrg = 1;

    flag = 0;
    if (!fcnvrg) {
        stpmax = factor * dmax(nlo_norm2(n, x), (double)n);  /* :553 */

        for (;;) {                                           /* :556-620 */
            iter = iter + 1;

            nlo_fd_jacobian(fcn, jac_or_null, ctx, n, n, x, fvec, jac);   /* :561-562 */
            njac = njac + 1;

            for (int32_t i = 0; i < n; ++i)                  /* :565-567 */
                grad[i] = nlo_dot(n, &A_(jac, n, 0, i), fvec);

            nlo_lu_factor(n, jac, n, ipvt);                  /* :570 */

            memcpy(xold, x, sizeof(double) * (size_t)n);     /* :573-574 */
            fold = f;

            for (int32_t i = 0; i < n; ++i) rhs[i] = -fvec[i];   /* :577 */
            nlo_lu_solve(n, jac, n, ipvt, rhs);
            memcpy(dir, rhs, sizeof(double) * (size_t)n);

            if (opt->use_line_search) {                      /* :580-589 */
                temp = nlo_dot(n, dir, dir);
                if (temp > stpmax) {                         /* squared length vs stpmax: kept */
                    double s = stpmax / temp;
                    for (int32_t i = 0; i < n; ++i) dir[i] = dir[i] * s;
                }
                nlo_limit_search_vector(n, dir, stpmax);
                rc = nlo_line_search(opt, fcn, ctx, n, n, xold, grad, dir, x, fvec, fold, &f, &lib);
                neval = neval + lib.fcn_count;
                if (rc) break;                               /* error stop inside the search */
            } else {                                         /* :591-595 */
                for (int32_t i = 0; i < n; ++i) x[i] = x[i] + dir[i];
                fcn(ctx, n, x, n, fvec);
                f = half * nlo_dot(n, fvec, fvec);
                neval = neval + 1;
            }

            nlo_test_convergence(n, n, x, xold, fvec, grad, 1, xtol, ftol, gtol, &check,
                                 &xcnvrg, &fcnvrg, &gcnvrg, &xnorm, &fnorm);   /* :599 */
            if (check) {
                break;
            } else if (gcnvrg) {                             /* :604-608 */
                rc = NLO_SPURIOUS_CONVERGENCE_ERROR;
                break;
            }

            if (opt->print_status) print_status(iter, neval, njac, xnorm, fnorm);   /* :611-613 */

            if (neval >= maxeval) { flag = 1; break; }       /* :616-619 */
        }
    }

    if (ib) {                                                /* :624-632 */
        ib->iter_count = iter;
        ib->fcn_count = neval;
        ib->jacobian_count = njac;
        ib->gradient_count = 0;
        ib->converge_on_fcn = fcnvrg;
        ib->converge_on_chng = xcnvrg;
        ib->converge_on_zero_diff = gcnvrg;
    }
    free(dir);
    free(jac);
    free(ipvt);
    if (rc) return rc;
    return flag != 0 ? NLO_CONVERGENCE_ERROR : 0;            /* :635-637 */
}

/* ---------------------------------------------------------------------------
 * Dense kernels behind qns_solve.  The reference takes these from the third-party
 * `linalg` library (jchristopherson/linalg, unpinned: fpm.toml:15, src/CMakeLists.txt:26),
 * which forwards to LAPACK / qrupdate: qr_factor = DGEQRF + DORGQR, qr_rank1_update =
 * DQR1UP, solve_triangular_system = DTRSV, mtx_mult = DGEMV, rank1_update = DGER,
 * recip_mult_array = DRSCL.  None of it is under /root/reference, so the published
 * unblocked algorithms are restated here with every sum in ascending index order
 * (parity unpinned at this boundary; see header).  Call sites: src/nonlin_solve.f90:286-336.
 * ------------------------------------------------------------------------- */

/* Plane rotation of LAPACK 3.10 DLARTG: c >= 0, r carries the sign of f. */
void nlo_givens(double f, double g, double *c, double *s, double *r)
{
    if (g == 0.0) { *c = 1.0; *s = 0.0; *r = f; return; }
    if (f == 0.0) { *c = 0.0; *s = 1.0; *r = g; return; }
    const double d = sqrt(f * f + g * g);
    *c = fabs(f) / d;
    *r = copysign(d, f);
    *s = g / *r;
}

/* Householder QR of the n-by-n column-major a (DGEQR2), with Q^T accumulated by applying
 * every reflector to an identity alongside (instead of DORG2R's backward pass).  On exit
 * r = R (upper, zeros below), q = Q, both column-major with leading dimension n. */
void nlo_qr_factor_full(int32_t n, const double *a_in, double *q, double *r)
{
    double *a = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
    double *e = (double *)calloc((size_t)n * (size_t)n, sizeof(double));     /* becomes Q^T */
    double *v = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    memcpy(a, a_in, sizeof(double) * (size_t)n * (size_t)n);
    for (int32_t i = 0; i < n; ++i) A_(e, n, i, i) = 1.0;
    for (int32_t j = 0; j < n; ++j) {
        const double alpha = A_(a, n, j, j);
        double ssq = 0.0;
        for (int32_t i = j + 1; i < n; ++i) ssq = ssq + A_(a, n, i, j) * A_(a, n, i, j);
        if (ssq == 0.0) continue;                                /* H = I (DLARFG with xnorm == 0) */
        const double beta = -copysign(sqrt(alpha * alpha + ssq), alpha);
        const double tau = (beta - alpha) / beta;
        const double scal = 1.0 / (alpha - beta);
        for (int32_t i = j + 1; i < n; ++i) v[i] = A_(a, n, i, j) * scal;
        A_(a, n, j, j) = beta;
        for (int32_t i = j + 1; i < n; ++i) A_(a, n, i, j) = 0.0;
        for (int32_t pass = 0; pass < 2; ++pass) {
            double *t = pass == 0 ? a : e;
            for (int32_t k = (pass == 0 ? j + 1 : 0); k < n; ++k) {
                double w = A_(t, n, j, k);
                for (int32_t i = j + 1; i < n; ++i) w = w + v[i] * A_(t, n, i, k);
                w = tau * w;
                A_(t, n, j, k) = A_(t, n, j, k) - w;
                for (int32_t i = j + 1; i < n; ++i) A_(t, n, i, k) = A_(t, n, i, k) - v[i] * w;
            }
        }
    }
    memcpy(r, a, sizeof(double) * (size_t)n * (size_t)n);
    for (int32_t i = 0; i < n; ++i)
        for (int32_t k = 0; k < n; ++k) A_(q, n, k, i) = A_(e, n, i, k);
    free(a); free(e); free(v);
}

/* Q1 R1 = Q R + u v^T for square Q, R (qrupdate DQR1UP: DQRTV1, DQROT 'B', DQRQH, the
 * first-row update, DQHQR, DQROT 'F').  u is overwritten. */
void nlo_qr_rank1_update(int32_t n, double *q, double *r, double *u, const double *v)
{
    if (n < 1) return;
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    double *c = (double *)malloc(sizeof(double) * (size_t)n);
    double *s = (double *)malloc(sizeof(double) * (size_t)n);
    for (int32_t k = 0; k < n; ++k) {                            /* w = Q^T u */
        double t = 0.0;
        for (int32_t i = 0; i < n; ++i) t = t + A_(q, n, i, k) * u[i];
        w[k] = t;
    }
    /* DQRTV1: rotations that fold w into w(0), generated from the bottom */
    double rr = w[n - 1];
    for (int32_t i = n - 2; i >= 0; --i) {
        double t;
        nlo_givens(w[i], rr, &c[i], &s[i], &t);
        rr = t;
    }
    w[0] = rr;
    /* DQROT 'B': the same rotations on the columns of Q, last pair first */
    for (int32_t i = n - 2; i >= 0; --i)
        for (int32_t row = 0; row < n; ++row) {
            const double x = A_(q, n, row, i), y = A_(q, n, row, i + 1);
            A_(q, n, row, i) = c[i] * x + s[i] * y;
            A_(q, n, row, i + 1) = c[i] * y - s[i] * x;
        }
    /* DQRQH: R -> upper Hessenberg, column by column from the bottom */
    for (int32_t col = 0; col < n; ++col) {
        int32_t ii = col < n - 2 ? col : n - 2;
        if (ii < 0) continue;
        double t = A_(r, n, ii + 1, col);
        for (int32_t j = ii; j >= 0; --j) {
            A_(r, n, j + 1, col) = c[j] * t - s[j] * A_(r, n, j, col);
            t = c[j] * A_(r, n, j, col) + s[j] * t;
        }
        A_(r, n, 0, col) = t;
    }
    for (int32_t col = 0; col < n; ++col) A_(r, n, 0, col) = A_(r, n, 0, col) + w[0] * v[col];
    /* DQHQR: back to upper triangular, rotations generated column by column */
    for (int32_t col = 0; col < n; ++col) {
        double t = A_(r, n, 0, col);
        const int32_t ii = col < n - 1 ? col : n - 1;
        for (int32_t j = 0; j < ii; ++j) {
            A_(r, n, j, col) = c[j] * t + s[j] * A_(r, n, j + 1, col);
            t = c[j] * A_(r, n, j + 1, col) - s[j] * t;
        }
        if (col < n - 1) {
            double rd;
            nlo_givens(t, A_(r, n, col + 1, col), &c[col], &s[col], &rd);
            A_(r, n, col, col) = rd;
            A_(r, n, col + 1, col) = 0.0;
        } else {
            A_(r, n, ii, col) = t;
        }
    }
    /* DQROT 'F' */
    for (int32_t i = 0; i < n - 1; ++i)
        for (int32_t row = 0; row < n; ++row) {
            const double x = A_(q, n, row, i), y = A_(q, n, row, i + 1);
            A_(q, n, row, i) = c[i] * x + s[i] * y;
            A_(q, n, row, i + 1) = c[i] * y - s[i] * x;
        }
    (void)u;
    free(w); free(c); free(s);
}

/* x <- R^-1 x, R upper triangular, column oriented (DTRSV 'U','N','N'). */
void nlo_solve_upper(int32_t n, const double *r, double *x)
{
    for (int32_t j = n - 1; j >= 0; --j) {
        if (x[j] != 0.0) {
            x[j] = x[j] / A_(r, n, j, j);
            const double t = x[j];
            for (int32_t i = j - 1; i >= 0; --i) x[i] = x[i] - t * A_(r, n, i, j);
        }
    }
}

/* ---------------------------------------------------------------------------
 * qns_solve: src/nonlin_solve.f90:156-427 (Broyden's method with QR rank-1 updates)
 * ------------------------------------------------------------------------- */
int nlo_quasi_newton_solve(const nlo_options *opt, int32_t jdelta, nlo_vecfcn fcn,
                           nlo_jacfcn jac_or_null, void *ctx, int32_t n, double *x,
                           double *fvec, nlo_iteration_behavior *ib)
{
    const double half = 0.5, factor = 1.0e2;
    int32_t restart = 1, xcnvrg = 0, fcnvrg = 0, gcnvrg = 0, check = 0;
    int32_t neval = 0, iter = 0, njac = 0, flag = 0, jcount = 0;
    const double ftol = opt->ftol, xtol = opt->xtol, gtol = opt->gtol;
    const int32_t maxeval = opt->max_evals;
    double f, fold, stpmax, xnorm = 0.0, fnorm = 0.0, temp, test, x2;
    int rc = 0;
    nlo_iteration_behavior lib;
    memset(&lib, 0, sizeof lib);              /* :205 `lib` is undefined before the first search; 0 here */

    if (ib) memset(ib, 0, sizeof *ib);                       /* :224-232 */
    if (!fcn) return NLO_UNDEFINED_FUNCTION_ERROR;           /* :240 */

    const size_t nn = (size_t)n * (size_t)(n > 0 ? n : 1);
    double *b = (double *)malloc(sizeof(double) * nn);       /* :251-258 */
    double *q = (double *)malloc(sizeof(double) * nn);
    double *r = (double *)malloc(sizeof(double) * nn);
    double *df = (double *)calloc((size_t)(5 * n + 1), sizeof(double));
    double *fvold = df + n, *xold = fvold + n, *dx = xold + n, *s = dx + n;

    fcn(ctx, n, x, n, fvec);                                 /* :261-270 */
    f = half * nlo_dot(n, fvec, fvec);
    neval = neval + 1;
    test = 0.0;
    for (int32_t i = 0; i < n; ++i) test = dmax(fabs(fvec[i]), test);
    if (test < ftol) fcnvrg = 1;

    if (!fcnvrg) {
        stpmax = factor * dmax(nlo_norm2(n, x), (double)n);  /* :276 */
        for (;;) {                                           /* :279-411 */
            iter = iter + 1;
            if (restart) {                                   /* :284-292 */
                nlo_fd_jacobian(fcn, jac_or_null, ctx, n, n, x, fvec, b);
                njac = njac + 1;
                nlo_qr_factor_full(n, b, q, r);
                jcount = 0;
            } else {                                         /* :294-310 */
                for (int32_t i = 0; i < n; ++i) df[i] = fvec[i] - fvold[i];
                for (int32_t i = 0; i < n; ++i) dx[i] = x[i] - xold[i];
                x2 = nlo_dot(n, dx, dx);
                for (int32_t i = 0; i < n; ++i) {            /* s = (df - matmul(b, dx)) / x2 */
                    double t = 0.0;
                    for (int32_t j = 0; j < n; ++j) t = t + A_(b, n, i, j) * dx[j];
                    s[i] = (df[i] - t) / x2;
                }
                for (int32_t j = 0; j < n; ++j)              /* rank1_update: b += s dx^T */
                    for (int32_t i = 0; i < n; ++i) A_(b, n, i, j) = A_(b, n, i, j) + s[i] * dx[j];
                nlo_qr_rank1_update(n, q, r, s, dx);
                jcount = jcount + 1;
            }

            for (int32_t j = 0; j < n; ++j)                  /* :313 grad = b^T f, kept in dx */
                dx[j] = nlo_dot(n, &A_(b, n, 0, j), fvec);

            memcpy(xold, x, sizeof(double) * (size_t)n);     /* :316-318 */
            memcpy(fvold, fvec, sizeof(double) * (size_t)n);
            fold = f;

            for (int32_t k = 0; k < n; ++k)                  /* :322 df = -q^T f */
                df[k] = -nlo_dot(n, &A_(q, n, 0, k), fvec);
            nlo_solve_upper(n, r, df);                       /* :327-328 */

            temp = nlo_dot(n, dx, df);                       /* :332-339 */
            if (temp >= 0.0) {
                restart = 1;
                if (opt->print_status) print_status(iter, neval, njac, xnorm, fnorm);
                if (iter > 10 * maxeval + 100) { flag = 1; break; }   /* the reference would spin; bounded here */
                continue;
            }

            if (opt->use_line_search) {                      /* :342-351 */
                temp = nlo_dot(n, df, df);
                if (temp > stpmax) {                         /* squared length vs stpmax: kept */
                    const double sc = stpmax / temp;
                    for (int32_t i = 0; i < n; ++i) df[i] = df[i] * sc;
                }
                nlo_limit_search_vector(n, df, stpmax);
                rc = nlo_line_search(opt, fcn, ctx, n, n, xold, dx, df, x, fvec, fold, &f, &lib);
                neval = neval + lib.fcn_count;
                if (rc) break;
            } else {                                         /* :353-357 */
                for (int32_t i = 0; i < n; ++i) x[i] = x[i] + df[i];
                fcn(ctx, n, x, n, fvec);
                f = half * nlo_dot(n, fvec, fvec);
                neval = neval + 1;
            }

            nlo_test_convergence(n, n, x, xold, fvec, dx,    /* :360-367 */
                                 (lib.converge_on_zero_diff && opt->use_line_search) ? 1 : 0,
                                 xtol, ftol, gtol, &check, &xcnvrg, &fcnvrg, &gcnvrg, &xnorm, &fnorm);
            if (!check) {                                    /* :368-391 */
                if (gcnvrg) {
                    if (restart) { rc = NLO_SPURIOUS_CONVERGENCE_ERROR; break; }
                    restart = 1;
                } else {
                    restart = jcount >= jdelta ? 1 : 0;
                }
            } else {
                break;
            }

            if (opt->print_status) print_status(iter, neval, njac, xnorm, fnorm);   /* :398-400 */
            if (neval >= maxeval) { flag = 1; break; }       /* :403-406 */
        }
    }

    if (ib) {                                                /* :414-422 */
        ib->iter_count = iter;
        ib->fcn_count = neval;
        ib->jacobian_count = njac;
        ib->gradient_count = 0;
        ib->converge_on_fcn = fcnvrg;
        ib->converge_on_chng = xcnvrg;
        ib->converge_on_zero_diff = gcnvrg;
    }
    free(b); free(q); free(r); free(df);
    if (rc) return rc;
    return flag != 0 ? NLO_CONVERGENCE_ERROR : 0;            /* :425-427 */
}

/* ---------------------------------------------------------------------------
 * constrained_least_squares_solver: src/nonlin_least_squares.f90:33-74, 793-1403
 * (bounded trust-region dog-leg with Coleman-Li scaling and an Armijo fallback).
 * Third-party pieces restated as for qns_solve: qr_factor(jac, tau, qr) = DGEQR2,
 * solve_qr = reflectors applied to f (DORM2R) + DTRSM on the top n rows, DGEMV.
 * ------------------------------------------------------------------------- */

/* Householder QR of the m-by-n column-major a (m >= n) with the same reflectors applied to the
 * right-hand side f (length m): on exit a holds R in its upper triangle (zeros below), f = Q^T f. */
void nlo_qr_factor_rhs(int32_t m, int32_t n, double *a, double *f)
{
    double *v = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    const int32_t steps = (n < m - 1) ? n : m - 1;
    for (int32_t j = 0; j < steps; ++j) {
        const double alpha = A_(a, m, j, j);
        double ssq = 0.0;
        for (int32_t i = j + 1; i < m; ++i) ssq = ssq + A_(a, m, i, j) * A_(a, m, i, j);
        if (ssq == 0.0) continue;
        const double beta = -copysign(sqrt(alpha * alpha + ssq), alpha);
        const double tau = (beta - alpha) / beta;
        const double scal = 1.0 / (alpha - beta);
        for (int32_t i = j + 1; i < m; ++i) v[i] = A_(a, m, i, j) * scal;
        A_(a, m, j, j) = beta;
        for (int32_t i = j + 1; i < m; ++i) A_(a, m, i, j) = 0.0;
        for (int32_t k = j + 1; k <= n; ++k) {                    /* k == n: the right-hand side */
            double *t = (k < n) ? &A_(a, m, 0, k) : f;
            double w = t[j];
            for (int32_t i = j + 1; i < m; ++i) w = w + v[i] * t[i];
            w = tau * w;
            t[j] = t[j] - w;
            for (int32_t i = j + 1; i < m; ++i) t[i] = t[i] - v[i] * w;
        }
    }
    free(v);
}

static double cls_scaled_norm(int32_t n, const double *x, const double *s, double *tmp)   /* :1263-1273 */
{
    for (int32_t i = 0; i < n; ++i) tmp[i] = x[i] * s[i];
    return nlo_norm2(n, tmp);
}

static int cls_is_finite(int32_t n, const double *x)         /* :1276-1298 */
{
    for (int32_t i = 0; i < n; ++i) {
        if (!(x[i] == x[i])) return 0;
        if (fabs(x[i]) == DBL_MAX) return 0;
    }
    return 1;
}

static void cls_apply_limits(int32_t n, const double *xl, const double *xu, double *x)   /* :858-883 */
{
    for (int32_t i = 0; i < n; ++i) if (x[i] < xl[i]) x[i] = xl[i];
    for (int32_t i = 0; i < n; ++i) if (x[i] > xu[i]) x[i] = xu[i];
}

double nlo_alpha_box(int32_t n, const double *x, const double *p, const double *xl, const double *xu)   /* :1181-1219 */
{
    double rst = DBL_MAX;
    for (int32_t i = 0; i < n; ++i) {
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

void nlo_coleman_li_scaling(int32_t n, const double *x, const double *xl, const double *xu, double *s)   /* :1222-1260 */
{
    const double min_scale = 1.0e-8, max_scale = 1.0e8, big = DBL_MAX;
    for (int32_t i = 0; i < n; ++i) {
        double di;
        if (xl[i] > -big && xu[i] < big) di = dmin(x[i] - xl[i], xu[i] - x[i]);
        else if (xl[i] > -big) di = x[i] - xl[i];
        else if (xu[i] < big) di = xu[i] - x[i];
        else di = 1.0;
        di = dmax(di, min_scale);
        s[i] = 1.0 / di;
        if (s[i] > max_scale) s[i] = max_scale;
    }
}

/* dogleg, :1301-1403.  jac: the Jacobian (m x n); r: R of its QR in the top n rows of an m x n array;
 * qtf: Q^T f.  Outputs p, g, Jp, prered.  work: 4n + m doubles. */
void nlo_dogleg(int32_t m, int32_t n, double delta, const double *x, const double *f, const double *jac,
                const double *r, const double *qtf, const double *s, const double *xl, const double *xu,
                double *p, double *g, double *Jp, double *prered, double *work)
{
    double *pgn = work, *psd = work + n, *u = work + 2 * n, *v = work + 3 * n, *Jg = work + 4 * n;
    double alpha, pgnnorm, psdnorm, t, c1, c2, a, b, c, arg;
    for (int32_t j = 0; j < n; ++j) g[j] = nlo_dot(m, &A_(jac, m, 0, j), f);      /* :1331 dgemv 'T' */
    for (int32_t i = 0; i < n; ++i) u[i] = qtf[i];                                /* :1334 solve_qr */
    {
        /* back substitution on the leading n x n block of r (leading dimension m) */
        for (int32_t j = n - 1; j >= 0; --j) {
            if (u[j] != 0.0) {
                u[j] = u[j] / A_(r, m, j, j);
                const double tj = u[j];
                for (int32_t i = j - 1; i >= 0; --i) u[i] = u[i] - tj * A_(r, m, i, j);
            }
        }
    }
    for (int32_t i = 0; i < n; ++i) pgn[i] = -u[i];
    pgnnorm = cls_scaled_norm(n, pgn, s, v);
    if (pgnnorm > delta) {                                                         /* :1339 */
        for (int32_t i = 0; i < m; ++i) Jg[i] = 0.0;                               /* dgemv 'N' */
        for (int32_t j = 0; j < n; ++j) {
            const double tj = g[j];
            for (int32_t i = 0; i < m; ++i) Jg[i] = Jg[i] + tj * A_(jac, m, i, j);
        }
        c1 = nlo_dot(n, g, g);
        c2 = nlo_dot(m, Jg, Jg);
        alpha = (c2 > 0.0 && c1 > 0.0) ? c1 / c2 : 0.0;
        for (int32_t i = 0; i < n; ++i) psd[i] = -alpha * g[i];
        psdnorm = cls_scaled_norm(n, psd, s, v);
        if (psdnorm >= delta && psdnorm > 0.0) {
            const double sc = delta / psdnorm;
            for (int32_t i = 0; i < n; ++i) p[i] = sc * psd[i];
        } else {
            for (int32_t i = 0; i < n; ++i) u[i] = pgn[i] - psd[i];
            for (int32_t i = 0; i < n; ++i) u[i] = s[i] * u[i];
            for (int32_t i = 0; i < n; ++i) v[i] = s[i] * psd[i];
            a = nlo_dot(n, u, u);
            b = 2.0 * nlo_dot(n, u, v);
            c = nlo_dot(n, v, v) - delta * delta;
            if (a <= 0.0) {
                for (int32_t i = 0; i < n; ++i) p[i] = psd[i];
            } else {
                arg = dmax(0.0, b * b - 4.0 * a * c);
                if (arg == 0.0) {
                    t = -b / (2.0 * a);
                } else {
                    t = (-b + sqrt(arg)) / (2.0 * a);
                    if (t < 0.0 || t > 1.0) t = (-b - sqrt(arg)) / (2.0 * a);
                }
                t = dmax(0.0, dmin(1.0, t));
                for (int32_t i = 0; i < n; ++i) p[i] = psd[i] + t * u[i];          /* u is the SCALED difference: kept */
            }
        }
    } else {
        for (int32_t i = 0; i < n; ++i) p[i] = pgn[i];
    }
    alpha = nlo_alpha_box(n, x, p, xl, xu);                                        /* :1392-1395 */
    if (alpha < 1.0)
        for (int32_t i = 0; i < n; ++i) p[i] = alpha * p[i];
    for (int32_t i = 0; i < m; ++i) Jp[i] = 0.0;                                   /* :1398 */
    for (int32_t j = 0; j < n; ++j) {
        const double tj = p[j];
        for (int32_t i = 0; i < m; ++i) Jp[i] = Jp[i] + tj * A_(jac, m, i, j);
    }
    c1 = nlo_dot(n, g, p);
    c2 = 0.5 * nlo_dot(m, Jp, Jp);
    *prered = -c1 - c2;
}

/* cls_solve, :938-1176.  xl / xu may be NULL (unbounded).  delta0 = get_trust_region_radius() (1),
 * stepscale0 = get_step_scaling_factor() (1). */
int nlo_cls_solve(const nlo_options *opt, double delta0, double stepscale0, const double *xl_in,
                  const double *xu_in, nlo_vecfcn fcn, nlo_jacfcn jac_or_null, void *ctx, int32_t m,
                  int32_t n, double *x, double *fvec, nlo_iteration_behavior *ib)
{
    const double delta_max = 1.0e3, eta = 1.0e-1, ls_cl = 1.0e-4, ls_beta = 0.5;
    const int32_t ls_max_iter = 10;
    int32_t converged = 0, xcnvrg = 0, fcnvrg = 0, gcnvrg = 0;
    int32_t neval = 0, iter = 0, njac = 0, k;
    const double ftol = opt->ftol, xtol = opt->xtol, gtol = opt->gtol;
    const int32_t maxeval = opt->max_evals;
    double xnorm, fnorm, gnorm, fnewnorm, actred, prered, rho, delta, stepscale, dderiv;

    if (ib) memset(ib, 0, sizeof *ib);                       /* :977-985 */
    if (!fcn) return NLO_UNDEFINED_FUNCTION_ERROR;           /* :988 */
    if (n > m) return NLO_UNDERDEFINED_PROBLEM_ERROR;        /* :989 */

    const size_t mn = (size_t)m * (size_t)n;
    double *jac = (double *)malloc(sizeof(double) * (mn ? mn : 1));
    double *qr = (double *)malloc(sizeof(double) * (mn ? mn : 1));
    double *w = (double *)calloc((size_t)(11 * n + 4 * m + 1), sizeof(double));
    double *xl = w, *xu = xl + n, *s = xu + n, *g = s + n, *p = g + n, *xnew = p + n, *tmp = xnew + n,
           *dwork = tmp + n /* 4n + m */, *Jp = dwork + 4 * n + m, *fnew = Jp + m, *qtf = fnew + m;
    for (int32_t i = 0; i < n; ++i) {                        /* :999-1009 */
        xl[i] = xl_in ? xl_in[i] : -DBL_MAX;
        xu[i] = xu_in ? xu_in[i] : DBL_MAX;
    }

    cls_apply_limits(n, xl, xu, x);                          /* :1023-1031 */
    fcn(ctx, n, x, m, fvec);
    neval = 1;
    fnorm = nlo_norm2(m, fvec);
    xnorm = nlo_norm2(n, x);
    if (!cls_is_finite(n, x) || !cls_is_finite(m, fvec)) {   /* silent return, :1029-1031 */
        free(jac); free(qr); free(w);
        return 0;
    }

    delta = delta0;                                          /* :1034 */
    iter = 1;
    for (;;) {                                               /* :1036-1160 */
        nlo_fd_jacobian(fcn, jac_or_null, ctx, m, n, x, fvec, jac);
        njac = njac + 1;
        if (opt->print_status) print_status(iter, neval, njac, xnorm, fnorm);

        memcpy(qr, jac, sizeof(double) * mn);                /* :1047 */
        memcpy(qtf, fvec, sizeof(double) * (size_t)m);
        nlo_qr_factor_rhs(m, n, qr, qtf);
        nlo_coleman_li_scaling(n, x, xl, xu, s);             /* :1050 */
        nlo_dogleg(m, n, delta, x, fvec, jac, qr, qtf, s, xl, xu, p, g, Jp, &prered, dwork);   /* :1053 */
        xnorm = cls_scaled_norm(n, p, s, tmp);
        gnorm = nlo_norm2(n, g);
        for (int32_t i = 0; i < n; ++i) xnew[i] = x[i] + p[i];

        fcn(ctx, n, xnew, m, fnew);                          /* :1060-1062 */
        fnewnorm = nlo_norm2(m, fnew);
        neval = neval + 1;

        actred = 0.5 * (fnorm * fnorm - fnewnorm * fnewnorm);   /* :1065-1070 */
        rho = (prered > 0.0 && actred >= 0.0) ? actred / prered : 0.0;

        if (rho < 0.25) {                                    /* :1073-1077 (constant 0.25: kept) */
            delta = dmax(0.25, 1.0e-12);
        } else if (rho > 0.75 && fabs(xnorm - delta) < 1.0e-12 * delta) {
            delta = dmin(2.0 * delta, delta_max);
        }

        if (rho > eta && fnewnorm <= fnorm) {                /* :1080-1086 */
            memcpy(x, xnew, sizeof(double) * (size_t)n);
            cls_apply_limits(n, xl, xu, x);
            memcpy(fvec, fnew, sizeof(double) * (size_t)m);
            fnorm = fnewnorm;
            iter = iter + 1;
        } else {                                             /* :1088-1123 */
            dderiv = nlo_dot(n, g, p);
            if (dderiv >= 0.0) {
                delta = dmax(0.5 * delta, 1.0e-12);
            } else {
                stepscale = stepscale0;
                for (k = 1; k <= ls_max_iter; ++k) {
                    for (int32_t i = 0; i < n; ++i) xnew[i] = x[i] + stepscale * p[i];
                    cls_apply_limits(n, xl, xu, xnew);
                    fcn(ctx, n, xnew, m, fnew);
                    neval = neval + 1;
                    fnewnorm = nlo_norm2(m, fnew);
                    if (fnewnorm <= fnorm + ls_cl * stepscale * dderiv) {
                        memcpy(x, xnew, sizeof(double) * (size_t)n);
                        memcpy(fvec, fnew, sizeof(double) * (size_t)m);
                        fnorm = fnewnorm;
                        iter = iter + 1;
                        delta = dmax(stepscale * xnorm, 1.0e-12);
                        break;
                    }
                    stepscale = stepscale * ls_beta;
                }
                if (k > ls_max_iter) delta = dmax(0.5 * delta, 1.0e-12);
            }
        }

        if (!cls_is_finite(n, x) || !cls_is_finite(m, fvec)) break;   /* :1125-1127 */

        if (xnorm <= xtol) { converged = 1; xcnvrg = 1; break; }      /* :1130-1149 */
        if (fabs(actred) <= ftol && fabs(prered) <= ftol && 0.5 * rho <= 1.0) { converged = 1; fcnvrg = 1; break; }
        if (gnorm <= gtol) { converged = 1; gcnvrg = 1; break; }
        if (neval >= maxeval) break;
    }

    if (ib) {                                                /* :1163-1170 */
        ib->iter_count = iter;
        ib->fcn_count = neval;
        ib->jacobian_count = njac;
        ib->converge_on_fcn = fcnvrg;
        ib->converge_on_chng = xcnvrg;
        ib->converge_on_zero_diff = gcnvrg;
    }
    free(jac); free(qr); free(w);
    return converged ? 0 : NLO_CONVERGENCE_ERROR;            /* :1173-1175 */
}

/* ---------------------------------------------------------------------------
 * polynomial%fit / fit_thru_zero: src/nonlin_polynomials.f90:146-238.  Vandermonde panel exactly as
 * the reference builds it (column j = column j-1 * x), then solve_least_squares (third-party linalg,
 * DGELS for a full-rank tall system = Householder QR + Q^T y + back substitution), restated with
 * nlo_qr_factor_rhs / nlo_solve_upper.  coef has order + 1 entries (coef[0] = 0 for thru_zero).
 * y is not modified (the reference's y is intent(inout) scratch).
 * ------------------------------------------------------------------------- */
int nlo_poly_fit(int32_t npts, int32_t order, const double *x, const double *y, int32_t thru_zero, double *coef)
{
    if (order >= npts || order < 1) return 4;                /* :163-166 */
    const int32_t ncols = thru_zero ? order : order + 1;
    double *a = (double *)malloc(sizeof(double) * (size_t)npts * (size_t)ncols);
    double *rhs = (double *)malloc(sizeof(double) * (size_t)npts);
    for (int32_t j = 0; j < npts; ++j) {                     /* :177-184 / :222-225 */
        if (thru_zero) {
            A_(a, npts, j, 0) = x[j];
        } else {
            A_(a, npts, j, 0) = 1.0;
            A_(a, npts, j, 1) = x[j];
        }
    }
    for (int32_t c = (thru_zero ? 1 : 2); c < ncols; ++c)
        for (int32_t j = 0; j < npts; ++j) A_(a, npts, j, c) = A_(a, npts, j, c - 1) * x[j];
    memcpy(rhs, y, sizeof(double) * (size_t)npts);
    nlo_qr_factor_rhs(npts, ncols, a, rhs);
    /* back substitution on the leading ncols x ncols block (leading dimension npts) */
    for (int32_t j = ncols - 1; j >= 0; --j) {
        if (rhs[j] != 0.0) {
            rhs[j] = rhs[j] / A_(a, npts, j, j);
            const double t = rhs[j];
            for (int32_t i = j - 1; i >= 0; --i) rhs[i] = rhs[i] - t * A_(a, npts, i, j);
        }
    }
    if (thru_zero) {
        coef[0] = 0.0;                                       /* :228 */
        for (int32_t c = 0; c < ncols; ++c) coef[c + 1] = rhs[c];
    } else {
        for (int32_t c = 0; c < ncols; ++c) coef[c] = rhs[c];
    }
    free(a); free(rhs);
    return 0;
}

/* polynomial%evaluate (real): src/nonlin_polynomials.f90:241-268 (Horner from the top). */
double nlo_poly_eval(int32_t order, const double *c, double x)
{
    const int32_t n = order + 1;
    if (order == -1) return 0.0;
    if (order == 0) return c[0];
    double y = c[n - 1] * x + c[order - 1];
    for (int32_t j = n - 2; j >= 1; --j) y = y * x + c[j - 1];
    return y;
}

/* ---------------------------------------------------------------------------
 * fcnnvar_helper%gradient (src/nonlin_multi_var.f90:182-246) and bfgs%solve
 * (src/nonlin_optimize.f90:557-770) with ls_search_miso (src/nonlin_linesearch.f90:329-492).
 * Third-party pieces (linalg -> BLAS / qrupdate, unpinned), restated with ascending-index sums:
 * tri_mtx_mult(.true., 1, r, 0, b) = B <- R^T R; DSYMV 'U'; cholesky_rank1_update = DCH1UP;
 * cholesky_rank1_downdate = DCH1DN; cholesky_factor(b, upper) = DPOTF2 'U'; solve_cholesky = two DTRSV.
 * ------------------------------------------------------------------------- */
/* fnh_grad_fcn, :182-246.  fv_or_null: the function value at x, if known. */
void nlo_fd_gradient(nlo_fcnnvar fcn, nlo_gradfcn grad_or_null, void *ctx, int32_t n, double *x,
                     const double *fv_or_null, double *g)
{
    if (grad_or_null) { grad_or_null(ctx, n, x, g); return; }
    const double f = fv_or_null ? *fv_or_null : fcn(ctx, n, x);
    const double eps = sqrt(DBL_EPSILON);
    for (int32_t j = 0; j < n; ++j) {
        const double temp = x[j];
        double h = eps * fabs(temp);
        if (h == 0.0) h = eps;
        x[j] = temp + h;
        const double f1 = fcn(ctx, n, x);
        x[j] = temp;
        g[j] = (f1 - f) / h;
    }
}

/* B <- R^T R for upper triangular R (tri_mtx_mult with trans = .true.); b is filled completely. */
void nlo_rtr(int32_t n, const double *r, double *b)
{
    for (int32_t j = 0; j < n; ++j)
        for (int32_t i = 0; i <= j; ++i) {
            double t = 0.0;
            for (int32_t k = 0; k <= i; ++k) t = t + A_(r, n, k, i) * A_(r, n, k, j);
            A_(b, n, i, j) = t;
            A_(b, n, j, i) = t;
        }
}

/* y <- B x, B symmetric (DSYMV 'U' reads the upper triangle; written here as full row sums in ascending j
 * over the symmetric matrix, which nlo_rtr fills on both sides). */
void nlo_symv(int32_t n, const double *b, const double *x, double *y)
{
    for (int32_t i = 0; i < n; ++i) {
        double t = 0.0;
        for (int32_t j = 0; j < n; ++j) t = t + A_(b, n, i, j) * x[j];
        y[i] = t;
    }
}

/* R1^T R1 = R^T R + u u^T (qrupdate DCH1UP).  u is overwritten. */
void nlo_chol_update(int32_t n, double *r, double *u)
{
    double *w = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int32_t i = 0; i < n; ++i) {
        double ui = u[i];
        for (int32_t j = 0; j < i; ++j) {
            const double t = w[j] * A_(r, n, j, i) + u[j] * ui;
            ui = w[j] * ui - u[j] * A_(r, n, j, i);
            A_(r, n, j, i) = t;
        }
        double rr;
        nlo_givens(A_(r, n, i, i), ui, &w[i], &u[i], &rr);
        A_(r, n, i, i) = rr;
    }
    free(w);
}

/* R1^T R1 = R^T R - u u^T (qrupdate DCH1DN).  Returns 1 if the result would not be positive definite
 * (r is then unchanged up to the triangular solve of u).  u is overwritten. */
int nlo_chol_downdate(int32_t n, double *r, double *u)
{
    double *w = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    /* u <- R^-T u (DTRSV 'U','T','N'): forward substitution, dot form */
    for (int32_t j = 0; j < n; ++j) {
        double t = u[j];
        for (int32_t i = 0; i < j; ++i) t = t - A_(r, n, i, j) * u[i];
        u[j] = t / A_(r, n, j, j);
    }
    double rho = nlo_norm2(n, u);
    rho = 1.0 - rho * rho;
    if (rho <= 0.0) { free(w); return 1; }
    rho = sqrt(rho);
    for (int32_t i = n - 1; i >= 0; --i) {
        const double ui = u[i];
        double rr;
        nlo_givens(rho, ui, &w[i], &u[i], &rr);
        rho = rr;
    }
    for (int32_t i = n - 1; i >= 0; --i) {
        double ui = 0.0;
        for (int32_t j = i; j >= 0; --j) {
            const double t = w[j] * ui + u[j] * A_(r, n, j, i);
            A_(r, n, j, i) = w[j] * A_(r, n, j, i) - u[j] * ui;
            ui = t;
        }
    }
    free(w);
    return 0;
}

/* Upper Cholesky factor of the symmetric positive definite b (DPOTF2 'U'), strict lower triangle zeroed.
 * Returns the 1-based index of a non-positive pivot, 0 on success. */
int nlo_chol_factor_upper(int32_t n, const double *b, double *r)
{
    memcpy(r, b, sizeof(double) * (size_t)n * (size_t)n);
    for (int32_t j = 0; j < n; ++j) {
        double ajj = A_(r, n, j, j);
        for (int32_t k = 0; k < j; ++k) ajj = ajj - A_(r, n, k, j) * A_(r, n, k, j);
        if (!(ajj > 0.0)) return j + 1;
        ajj = sqrt(ajj);
        A_(r, n, j, j) = ajj;
        for (int32_t c = j + 1; c < n; ++c) {
            double t = A_(r, n, j, c);
            for (int32_t k = 0; k < j; ++k) t = t - A_(r, n, k, j) * A_(r, n, k, c);
            A_(r, n, j, c) = t / ajj;
        }
    }
    for (int32_t j = 0; j < n; ++j)
        for (int32_t i = j + 1; i < n; ++i) A_(r, n, i, j) = 0.0;
    return 0;
}

/* x <- (R^T R)^-1 x: R^T y = x (forward, dot form), then R x = y (nlo_solve_upper). */
void nlo_solve_cholesky_upper(int32_t n, const double *r, double *x)
{
    for (int32_t j = 0; j < n; ++j) {
        double t = x[j];
        for (int32_t i = 0; i < j; ++i) t = t - A_(r, n, i, j) * x[i];
        x[j] = t / A_(r, n, j, j);
    }
    nlo_solve_upper(n, r, x);
}

/* ls_search_miso, src/nonlin_linesearch.f90:329-492 (fold is always supplied by bfgs). */
static int nlo_line_search_scalar(const nlo_options *opt, nlo_fcnnvar fcn, void *ctx, int32_t n, const double *xold,
                                  const double *grad, const double *dir, double *x, double fold, double *fx,
                                  int32_t *fcn_count)
{
    const double tolx = 2.0 * DBL_EPSILON, alpha = opt->ls_alpha, lambdamin = opt->ls_factor;
    const int32_t maxeval = opt->ls_max_evals;
    int32_t neval = 0, niter = 0, flag = 0;
    double alam, alam1 = 0.0, alamin, f1 = 0.0, slope, test, tmplam = 0.0, f = 0.0;
    int rc = 0;
    *fcn_count = 0;
    slope = nlo_dot(n, grad, dir);
    if (slope >= 0.0) return NLO_DIVERGENT_BEHAVIOR_ERROR;
    test = 0.0;
    for (int32_t i = 0; i < n; ++i) {
        const double t = fabs(dir[i]) / dmax(fabs(xold[i]), 1.0);
        if (t > test) test = t;
    }
    alamin = tolx / test;
    alam = 1.0;
    for (;;) {
        for (int32_t i = 0; i < n; ++i) x[i] = xold[i] + alam * dir[i];
        f = fcn(ctx, n, x);
        neval = neval + 1;
        niter = niter + 1;
        if (alam < alamin) {
            double sq = 0.0;
            for (int32_t i = 0; i < n; ++i) { const double d = x[i] - xold[i]; sq = sq + d * d; }
            if (sqrt(sq) == 0.0) { rc = NLO_CONVERGENCE_ERROR; break; }
            for (int32_t i = 0; i < n; ++i) x[i] = xold[i];
            break;
        } else if (f <= fold + alpha * alam * slope) {
            break;
        } else {
            tmplam = nlo_min_backtrack_search(niter, fold, f, f1, alam, alam1, slope);
        }
        alam1 = alam;
        f1 = f;
        alam = dmax(tmplam, lambdamin * alam);
        if (neval >= maxeval) { flag = 1; break; }
    }
    *fx = f;
    *fcn_count = neval;
    if (rc) return rc;
    return flag ? NLO_CONVERGENCE_ERROR : 0;
}

/* bfgs_solve, src/nonlin_optimize.f90:557-770.  opt->max_evals = get_max_fcn_evals() (500), opt->gtol =
 * get_tolerance() (1e-12), opt->xtol = get_var_tolerance() (1e-12); ib->gradient_count is filled. */
int nlo_bfgs_solve(const nlo_options *opt, nlo_fcnnvar fcn, nlo_gradfcn grad_or_null, void *ctx, int32_t n,
                   double *x, double *fout, nlo_iteration_behavior *ib)
{
    const double factor = 1.0e2, small = 1.0e-10;
    int32_t xcnvrg = 0, gcnvrg = 0, neval = 0, ngrad = 0, flag = 0, iter = 0;
    const int32_t maxeval = opt->max_evals;
    const double gtol = opt->gtol, xtol = opt->xtol;
    double fp, stpmax = 0.0, fret, xtest = 0.0, gtest, temp, ydx;
    int rc = 0;

    if (ib) memset(ib, 0, sizeof *ib);
    if (!fcn) return NLO_UNDEFINED_FUNCTION_ERROR;           /* :614 */
    const size_t nn = (size_t)n * (size_t)(n > 0 ? n : 1);
    double *w = (double *)calloc((size_t)(8 * n + 1), sizeof(double));
    double *g = w, *dx = g + n, *u = dx + n, *v = u + n, *y = v + n, *bdx = y + n, *gold = bdx + n, *xnew = gold + n;
    double *b = (double *)calloc(nn, sizeof(double)), *r = (double *)calloc(nn, sizeof(double));

    fp = fcn(ctx, n, x);                                     /* :633-636 */
    nlo_fd_gradient(fcn, grad_or_null, ctx, n, x, &fp, g);
    neval = 1;
    ngrad = 1;
    gtest = nlo_norm2(n, g);                                 /* :639-642 */
    if (gtest < gtol) gcnvrg = 1;

    if (!gcnvrg) {
        for (;;) {                                           /* :647-748 */
            iter = iter + 1;
            if (iter == 1) {                                 /* :653-656 */
                for (int32_t i = 0; i < n; ++i) dx[i] = -g[i];
                stpmax = factor * dmax(nlo_norm2(n, x), (double)n);
            }
            if (opt->use_line_search) {                      /* :659-669 */
                int32_t lcount = 0;
                nlo_limit_search_vector(n, dx, stpmax);
                rc = nlo_line_search_scalar(opt, fcn, ctx, n, x, g, dx, xnew, fp, &fret, &lcount);
                neval = neval + lcount;
                if (rc) break;
                fp = fret;
            } else {
                for (int32_t i = 0; i < n; ++i) xnew[i] = x[i] + dx[i];
                fp = fcn(ctx, n, xnew);
                neval = neval + 1;
            }
            for (int32_t i = 0; i < n; ++i) {                /* :672-678 */
                dx[i] = xnew[i] - x[i];
                x[i] = xnew[i];
                gold[i] = g[i];
            }
            nlo_fd_gradient(fcn, grad_or_null, ctx, n, x, &fp, g);
            ngrad = ngrad + 1;

            xtest = 0.0;                                     /* :681-689 */
            for (int32_t i = 0; i < n; ++i) {
                temp = fabs(dx[i]) / dmax(fabs(x[i]), 1.0);
                xtest = dmax(temp, xtest);
            }
            if (xtest < xtol) { xcnvrg = 1; break; }
            gtest = nlo_norm2(n, g);                         /* :692-696 */
            if (gtest < gtol) { gcnvrg = 1; break; }

            for (int32_t i = 0; i < n; ++i) y[i] = g[i] - gold[i];   /* :699-700 */
            ydx = nlo_dot(n, y, dx);
            if (iter == 1) {                                 /* :703-706: R = temp * I */
                temp = sqrt(nlo_dot(n, y, y) / ydx);
                for (size_t e = 0; e < nn; ++e) r[e] = 0.0;
                for (int32_t i = 0; i < n; ++i) A_(r, n, i, i) = temp;
            }
            nlo_rtr(n, r, b);                                /* :709 */
            nlo_symv(n, b, dx, bdx);                         /* :712 */
            if (ydx > small && iter > 1) {                   /* :715-724 */
                const double s1 = sqrt(ydx), s2 = sqrt(nlo_dot(n, dx, bdx));
                for (int32_t i = 0; i < n; ++i) u[i] = y[i] / s1;
                for (int32_t i = 0; i < n; ++i) v[i] = bdx[i] / s2;
                nlo_chol_update(n, r, u);
                if (nlo_chol_downdate(n, r, v)) { rc = NLO_INVALID_OPERATION_ERROR; break; }   /* linalg raises LA_MATRIX_FORMAT_ERROR */
            } else {
                if (nlo_chol_factor_upper(n, b, r)) { rc = NLO_INVALID_OPERATION_ERROR; break; }
            }
            for (int32_t i = 0; i < n; ++i) dx[i] = -g[i];   /* :727 dx = solve_cholesky(.true., r, -g) */
            nlo_solve_cholesky_upper(n, r, dx);

            if (opt->print_status) {                         /* :730-737 */
                printf("\n");
                printf("Iteration: %d\n", iter);
                printf("Function Evaluations: %d\n", neval);
                printf("Function Value: %10.3E\n", fp);
                printf("Change in Variable: %10.3E\n", xtest);
                printf("Gradient: %10.3E\n", gtest);
            }
            if (neval >= maxeval) { flag = 1; break; }       /* :740-743 */
        }
    }
    if (ib) {                                                /* :751-759 */
        ib->iter_count = iter;
        ib->fcn_count = neval;
        ib->jacobian_count = 0;
        ib->gradient_count = ngrad;
        ib->converge_on_fcn = 0;
        ib->converge_on_chng = xcnvrg;
        ib->converge_on_zero_diff = gcnvrg;
    }
    if (fout) *fout = fp;                                    /* :762 */
    free(w); free(b); free(r);
    if (rc) return rc;
    return flag ? NLO_CONVERGENCE_ERROR : 0;                 /* :765-767 */
}

/* ---------------------------------------------------------------------------
 * Synthetic dense-quadratic family; not reference code.
 * ------------------------------------------------------------------------- */
static void trace_push(nlo_trace *t, int32_t n, const double *x)
{
    if (!t) return;
    if (t->count < t->capacity && t->xs)
        memcpy(t->xs + (size_t)t->count * (size_t)n, x, sizeof(double) * (size_t)n);
    t->count += 1;
}

void nlo_dq_fcn(void *ctx, int32_t n, const double *x, int32_t m, double *f)
{
    nlo_dq_problem *p = (nlo_dq_problem *)ctx;
    const double *A = p->A, *b = p->b;
    const double g = p->gamma;
    p->ncalls += 1;
    trace_push(p->trace, n, x);
    for (int32_t i = 0; i < m; ++i) f[i] = 0.0;
    /* column sweep: row i still accumulates j ascending, one mul + one add per term */
    for (int32_t j = 0; j < n; ++j) {
        const double xj = x[j];
        const double *col = A + (size_t)j * (size_t)m;
        for (int32_t i = 0; i < m; ++i) f[i] = f[i] + col[i] * xj;
    }
    for (int32_t i = 0; i < m; ++i) {
        double u = f[i];
        f[i] = (u + g * u * u) - b[i];
    }
}

void nlo_dq_jac(void *ctx, int32_t n, const double *x, int32_t m, double *jac)
{
    nlo_dq_problem *p = (nlo_dq_problem *)ctx;
    const double *A = p->A;
    const double g = p->gamma;
    double *u = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
    for (int32_t j = 0; j < n; ++j) {
        const double xj = x[j];
        const double *col = A + (size_t)j * (size_t)m;
        for (int32_t i = 0; i < m; ++i) u[i] = u[i] + col[i] * xj;
    }
    for (int32_t i = 0; i < m; ++i) u[i] = 1.0 + 2.0 * g * u[i];
    for (int32_t j = 0; j < n; ++j)
        for (int32_t i = 0; i < m; ++i)
            A_(jac, m, i, j) = u[i] * A_(A, m, i, j);
    free(u);
}

static inline double sm64_uniform(uint64_t *state)
{
    *state += 0x9E3779B97F4A7C15ULL;
    uint64_t z = *state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1.0p-53;
}

void nlo_dq_generate(uint64_t seed, int32_t m, int32_t n, double gamma, double sigma,
                     double spread, int32_t square_shift, double *A, double *b,
                     double *x_true, double *x0)
{
    uint64_t st = seed;
    const double rs = sqrt((double)n);
    for (int32_t j = 0; j < n; ++j)
        for (int32_t i = 0; i < m; ++i)
            A_(A, m, i, j) = (2.0 * sm64_uniform(&st) - 1.0) / rs;
    if (square_shift)
        for (int32_t j = 0; j < n && j < m; ++j) A_(A, m, j, j) = 2.0 + A_(A, m, j, j);
    for (int32_t j = 0; j < n; ++j) x_true[j] = 2.0 * sm64_uniform(&st) - 1.0;
    double *zero = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
    nlo_dq_problem p = { m, n, A, zero, gamma, 0, NULL };
    nlo_dq_fcn(&p, n, x_true, m, b);          /* b = model(x_true), i.e. the residual with b == 0 */
    free(zero);
    for (int32_t i = 0; i < m; ++i) b[i] = b[i] + sigma * (2.0 * sm64_uniform(&st) - 1.0);
    for (int32_t j = 0; j < n; ++j) x0[j] = x_true[j] + spread * (2.0 * sm64_uniform(&st) - 1.0);
}

int nlo_dq_lm_solve(const nlo_options *opt, const nlo_dq_problem *p, double *x,
                    double *fvec, nlo_iteration_behavior *ib)
{
    nlo_dq_problem q = *p;
    int rc = nlo_lm_solve(opt, nlo_dq_fcn, NULL, &q, q.m, q.n, x, fvec, ib);
    ((nlo_dq_problem *)p)->ncalls = q.ncalls;
    return rc;
}

int nlo_dq_newton_solve(const nlo_options *opt, const nlo_dq_problem *p, int32_t analytic,
                        double *x, double *fvec, nlo_iteration_behavior *ib)
{
    nlo_dq_problem q = *p;
    int rc = nlo_newton_solve(opt, nlo_dq_fcn, analytic ? nlo_dq_jac : NULL, &q, q.n, x, fvec, ib);
    ((nlo_dq_problem *)p)->ncalls = q.ncalls;
    return rc;
}

int nlo_dq_quasi_newton_solve(const nlo_options *opt, int32_t jdelta, const nlo_dq_problem *p,
                              int32_t analytic, double *x, double *fvec, nlo_iteration_behavior *ib)
{
    nlo_dq_problem q = *p;
    int rc = nlo_quasi_newton_solve(opt, jdelta, nlo_dq_fcn, analytic ? nlo_dq_jac : NULL, &q, q.n, x, fvec, ib);
    ((nlo_dq_problem *)p)->ncalls = q.ncalls;
    return rc;
}

int nlo_dq_cls_solve(const nlo_options *opt, double delta0, double stepscale0, const double *xl, const double *xu,
                     const nlo_dq_problem *p, double *x, double *fvec, nlo_iteration_behavior *ib)
{
    nlo_dq_problem q = *p;
    int rc = nlo_cls_solve(opt, delta0, stepscale0, xl, xu, nlo_dq_fcn, NULL, &q, q.m, q.n, x, fvec, ib);
    ((nlo_dq_problem *)p)->ncalls = q.ncalls;
    return rc;
}

/* Scalar objective of the device model for bfgs: f(x) = 0.5 * sum_i r_i(x)^2 (ascending sum). */
static double dq_objective(void *ctx, int32_t n, const double *x)
{
    nlo_dq_problem *p = (nlo_dq_problem *)ctx;
    double *r = (double *)malloc(sizeof(double) * (size_t)p->m);
    nlo_dq_fcn(ctx, n, x, p->m, r);
    const double f = 0.5 * nlo_dot(p->m, r, r);
    free(r);
    return f;
}

int nlo_dq_bfgs_solve(const nlo_options *opt, const nlo_dq_problem *p, double *x, double *fout,
                      nlo_iteration_behavior *ib)
{
    nlo_dq_problem q = *p;
    int rc = nlo_bfgs_solve(opt, dq_objective, NULL, &q, q.n, x, fout, ib);
    ((nlo_dq_problem *)p)->ncalls = q.ncalls;
    return rc;
}
