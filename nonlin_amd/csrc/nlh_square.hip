// nlh_square.hip -- newton_solver (ns_solve, src/nonlin_solve.f90:452-638) and quasi_newton_solver (qns_solve, :156-427)
// with ls_search_mimo (src/nonlin_linesearch.f90:152-326): host loops around device kernels for one problem with host
// callbacks, lock-step device state machines for batches; lu_factor / solve_lu (call sites :570, :577), the Householder
// steps every QR of the library runs (also behind constrained least squares and polynomial fits), and their entry points.
#include "nlh_internal.h"
#include "nlh_kernels_model.h"
#include "nlh_kernels_gram.h"
#include "nlh_kernels_lu.h"
#include "nlh_kernels_newton.h"
#include "nlh_kernels_broyden.h"
#include "nlh_kernels_exact.h"

template <bool FAST> static void lu_panel_reg_attrs(int lds_max);

void nlh_square_init_device(int lds_max)
{
    hipFuncSetAttribute((const void *)k_lu_solve, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lu_panel_lds, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    broyden_kernel_attrs(lds_max);
    lu_panel_reg_attrs<false>(lds_max);
    lu_panel_reg_attrs<true>(lds_max);
}

// The register panel with several rows per thread (k_lu_panel_reg): instance by the number of panel rows.
// Width of the register panel that starts with `rows` rows left (0: beyond the register panel's reach) -- the rule of
// launch_lu_panel_reg below.
static int lu_panel_reg_width(int rows)
{
    if (rows <= 0 || rows > 1024) return 0;
    return std::min(rows <= 512 ? 32 : 16, rows);
}

// mvbuf: where this panel leaves its move list; pnb / pmv: the previous panel, not yet applied to this panel's columns (look-ahead)
template <bool FAST>
static int launch_lu_panel_reg(nlh_handle *h, int nprob, int n, double *dA, int32_t *dipvt, int32_t *dinfo, int jb,
                               const LmState *st, int want, int wide, int32_t *mvbuf, int pnb = 0, const int32_t *pmv = nullptr)
{
    const int rows = n - jb;
    auto go = [&](auto kern, int rpt, int pw) {
        const int nb = std::min(pw, rows);
        const int T = std::max(64, (((rows + rpt - 1) / rpt) + 63) & ~63);
        hipLaunchKernelGGL(kern, dim3(nprob), dim3(T), lu_panel_reg_lds(rpt, pw, T), h->stream, n, dA, dipvt, dinfo, mvbuf,
                           jb, nb, st, want, pnb, pmv);
        return nb;
    };
    if (rows <= 256) return go(k_lu_panel_reg<1, 32, 256, FAST>, 1, 32);
    if (rows <= 512) return go(k_lu_panel_reg<2, 32, 256, FAST>, 2, 32);
    if (rows <= 1024) return wide ? go(k_lu_panel_reg<2, 16, 512, FAST>, 2, 16) : go(k_lu_panel_reg<4, 16, 256, FAST>, 4, 16);
    return 0;
}
template <bool FAST>
static void lu_panel_reg_attrs(int lds_max)
{
    hipFuncSetAttribute((const void *)k_lu_panel_reg<1, 32, 256, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lu_panel_reg<2, 32, 256, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lu_panel_reg<2, 16, 512, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lu_panel_reg<4, 16, 256, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
}

template <bool FAST>
static void lu_blocked(nlh_handle *h, int nprob, int n, double *dA, int32_t *dipvt, int32_t *dinfo, const LmState *st, int want,
                       int panel_mode)
{
    // panels factored in registers while they have at most 1024 rows (several rows per thread, implicit interchanges),
    // 32-column panels in global memory before.
    // LOOK-AHEAD (opt-in, NLH_LU_LOOKAHEAD=1; a handful of problems, n <= 1024): the panel kernel applies the previous panel
    // to ITS OWN columns itself (k_lu_panel_reg, pnb > 0) and starts as soon as that panel is done; the rest of a step's
    // update -- the moves left of the panel, the block row and the trailing product of every other column -- runs on a side
    // stream under the next panel, which is one workgroup per problem and leaves the chip idle (n = 1024: 1.8 of 2.9 ms are
    // panels).  Every element still receives the same operations in the same order: the same bits (tests force it on).
    // OFF by default: measured at n = 1024, the fused update costs the panel kernel 23 us (16 wide) to 37 us (32 wide) where
    // the two update launches it takes off the critical path cost 21 -- 2.93 -> 3.67 ms.  One workgroup of four waves forms
    // its 1024 x 16 x 16 product at a wave per SIMD; the stream-level variant before it (narrow launches of the update
    // kernels for the next panel's columns) bought nothing either, those kernels being latency-bound whatever their width.
    // docs/lab_notebook.md.
    static const int la_env = [] { const char *e = getenv("NLH_LU_LOOKAHEAD"); return e ? atoi(e) : 0; }();
    const bool la_ok = la_env > 0 && panel_mode == 1 && (long)nprob * n <= 8192 && n <= 1024 && n >= 256;
    if (la_ok && !h->lu_side) {
        bool ok = hipStreamCreateWithFlags(&h->lu_side, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&h->lu_panel_done, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&h->lu_bulk_done[0], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&h->lu_bulk_done[1], hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); h->lu_side = nullptr; }
    }
    const bool la = la_ok && h->lu_side;
    int32_t *mvb[2] = {(int32_t *)h->lumv.p, h->lumv.p ? (int32_t *)h->lumv.p + (size_t)LU_MV_STRIDE * nprob : nullptr};
    bool bulk_rec[2] = {false, false};                           // lu_bulk_done[i] has been recorded (by the bulk update of a step of parity i)
    int pnb = 0, step = 0;                                       // pnb > 0: the previous panel is still to be applied to the next panel's columns
    for (int jb = 0; jb < n; ++step) {
        int nb = 0;
        int32_t *mv = mvb[la ? (step & 1) : 0];
        if (panel_mode >= 1) {
            // this panel's columns were last touched by the bulk update two steps back (it also read the move list this panel overwrites)
            if (la && bulk_rec[step & 1]) { hipStreamWaitEvent(h->stream, h->lu_bulk_done[step & 1], 0); bulk_rec[step & 1] = false; }
            nb = launch_lu_panel_reg<FAST>(h, nprob, n, dA, dipvt, dinfo, jb, st, want, panel_mode >= 2, mv, pnb, pnb ? mvb[(step - 1) & 1] : nullptr);
        }
        pnb = 0;
        const bool reg = nb != 0;
        if (nb == 0) {
            const bool lds = panel_mode == 0 && (n - jb) <= LU_PROWS;
            const int pw = lds ? LU_PNB : LU_NB;
            nb = (n - jb < pw) ? (n - jb) : pw;
            if (lds)
                // one thread per panel row: waves without rows would still run the step's instruction stream and barriers
                hipLaunchKernelGGL(k_lu_panel_lds, dim3(nprob), dim3(std::min(1024, (n - jb + 63) & ~63)), 0, h->stream, n, dA, dipvt, dinfo, jb, nb,
                                   st, want);
            else
                hipLaunchKernelGGL(k_lu_panel, dim3(nprob), dim3(1024), 0, h->stream, n, dA, dipvt, dinfo, jb, nb, st, want);
        }
        const int nt = n - jb - nb;
        const int nb2 = (la && reg) ? lu_panel_reg_width(nt) : 0;            // the next panel's width: its columns are left to its own kernel
        hipStream_t us = h->stream;
        if (nb2 > 0) {
            hipEventRecord(h->lu_panel_done, h->stream);
            hipStreamWaitEvent(h->lu_side, h->lu_panel_done, 0);
            us = h->lu_side;
        } else {                                                 // back on the panel's stream: behind whatever the side stream still holds
            for (int i = 0; i < 2; ++i)
                if (bulk_rec[i]) { hipStreamWaitEvent(h->stream, h->lu_bulk_done[i], 0); bulk_rec[i] = false; }
        }
        if (n - nb > 0) {
            if (reg) {
                const int slo = nb2 > 0 ? jb : 0, shi = nb2 > 0 ? jb + nb2 : 0;   // outside-column indices jb .. jb + nb2 = the next panel's columns
                if (nb <= 16)
                    hipLaunchKernelGGL((k_lu_move_trsm<16, FAST>), dim3((n - nb + 255) / 256, nprob), dim3(256), 0, us, n, dA,
                                       (const int32_t *)mv, jb, nb, st, want, slo, shi);
                else
                    hipLaunchKernelGGL((k_lu_move_trsm<32, FAST>), dim3((n - nb + 255) / 256, nprob), dim3(256), 0, us, n, dA,
                                       (const int32_t *)mv, jb, nb, st, want, slo, shi);
            } else {
                hipLaunchKernelGGL(k_lu_swap_trsm, dim3((n - nb + 255) / 256, nprob), dim3(256), 0, us, n, dA,
                                   (const int32_t *)dipvt, jb, nb, st, want);
            }
        }
        if (nt > 0)
            hipLaunchKernelGGL(k_lu_gemm<FAST>, dim3((nt + 63) / 64, (nt + 63) / 64, nprob), dim3(256), 0, us, n, dA, jb, nb, st, want, nb2);
        if (nb2 > 0) {
            hipEventRecord(h->lu_bulk_done[step & 1], h->lu_side);
            bulk_rec[step & 1] = true;
            pnb = nb;
        }
        jb += nb;
    }
    for (int i = 0; i < 2; ++i)
        if (bulk_rec[i]) hipStreamWaitEvent(h->stream, h->lu_bulk_done[i], 0);
}

// lu_factor: unblocked single-workgroup kernel for small n, blocked multi-kernel path otherwise.
void launch_lu_factor(nlh_handle *h, int nprob, int n, double *dA, int32_t *dipvt, int32_t *dinfo,
                             const LmState *st, int want)
{
    static const int panel_env = [] { const char *e = getenv("NLH_LU_PANEL"); return e ? atoi(e) : 1; }();
    int panel_mode = panel_env;
    // the move lists of the register panel: allocated BEFORE the timed bracket and the first launch (growing a workspace
    // frees the old one, a device-wide synchronisation).  Without the memory the global-memory panel serves: say nothing
    if (n >= 128 && panel_mode >= 1 && ensure(h, h->lumv, sizeof(int32_t) * 2 * LU_MV_STRIDE * (size_t)nprob)) { panel_mode = 0; h->err.clear(); }
    Timed t(h, NLH_K_LU);
    if (n < 128) {
        hipLaunchKernelGGL(k_lu_factor, dim3(nprob), dim3(n >= 96 ? 1024 : 256), 0, h->stream, n, dA, dipvt, dinfo, st, want);
        return;
    }
    if (dinfo) hipMemsetAsync(dinfo, 0, sizeof(int32_t) * (size_t)nprob, h->stream);
    // panels factored in registers while they have at most 2048 rows (several rows per thread, implicit interchanges),
    // 32-column panels in global memory before
    // NLH_LU_CONTRACT=1 (measurement only): the same kernels with every multiply-subtract pair contracted
    // into one FMA -- the pivots stay, the bits do not
    static const bool contract = [] { const char *e = getenv("NLH_LU_CONTRACT"); return e && atoi(e) != 0; }();
    if (contract && panel_mode >= 1) lu_blocked<true>(h, nprob, n, dA, dipvt, dinfo, st, want, panel_mode);
    else lu_blocked<false>(h, nprob, n, dA, dipvt, dinfo, st, want, panel_mode);
}


// ===========================================================================
// Newton: ns_solve as a host loop; Jacobian, gradient, LU on the device.
// The O(n) line-search and convergence arithmetic stays on the host in the
// reference's exact order (sequential dot products), so given the same search
// direction every accept/backtrack decision matches the CPU path.
// ===========================================================================
struct NewtonEval {
    // evaluate F at host x -> host f (device copy of f kept in dfvec when keep_dev)
    std::function<int(const double *x, double *f)> fcn;
    // Jacobian at host x (device copy of f0 in dfvec) -> device J (n x n)
    std::function<int(double *x, const double *f0_host, double *dJ)> jac;
};




// ls_search_mimo, src/nonlin_linesearch.f90:152-326
static int line_search(const nlh_options *o, NewtonEval &ev, int n, const double *xold, const double *grad,
                       const double *dir, double *x, double *fvec, double fold, double *fx, int *fcn_count)
{
    const double tolx = 2.0 * DBL_EPSILON, alpha = o->ls_alpha, lambdamin = o->ls_factor;
    const int maxeval = o->ls_max_evals;
    int neval = 0, niter = 0, flag = 0, rc = 0;
    double alam, alam1 = 0.0, alamin, f1 = 0.0, slope, test, tmplam = 0.0, f = 0.0;
    *fcn_count = 0;
    slope = h_dot(n, grad, dir);                                // :249-253
    if (slope >= 0.0) return NLH_DIVERGENT_BEHAVIOR_ERROR;
    test = 0.0;                                                 // :256-262
    for (int i = 0; i < n; ++i) {
        const double t = fabs(dir[i]) / fmax(fabs(xold[i]), 1.0);
        if (t > test) test = t;
    }
    alamin = tolx / test;
    alam = 1.0;
    for (;;) {                                                  // :266-310
        for (int i = 0; i < n; ++i) x[i] = xold[i] + alam * dir[i];
        if ((rc = ev.fcn(x, fvec))) return rc;
        f = 0.5 * h_dot(n, fvec, fvec);
        neval += 1;
        niter += 1;
        if (alam < alamin) {                                    // :275-287
            double sq = 0.0;
            for (int i = 0; i < n; ++i) { const double d = x[i] - xold[i]; sq = sq + d * d; }
            if (sqrt(sq) == 0.0) { rc = NLH_CONVERGENCE_ERROR; break; }
            for (int i = 0; i < n; ++i) x[i] = xold[i];
            break;
        } else if (f <= fold + alpha * alam * slope) {          // :288-291
            break;
        } else {
            tmplam = min_backtrack_search(niter, fold, f, f1, alam, alam1, slope);
        }
        alam1 = alam;                                           // :300-302
        f1 = f;
        alam = fmax(tmplam, lambdamin * alam);
        if (neval >= maxeval) { flag = 1; break; }              // :305-309
    }
    *fx = f;
    *fcn_count = neval;
    if (rc) return rc;
    return flag ? NLH_CONVERGENCE_ERROR : 0;
}

static int newton_core(nlh_handle *h, const nlh_options *o, int n, NewtonEval &ev, double *x, double *fvec,
                       nlh_iteration_behavior *ib)
{
    int rc;
    const size_t nn = (size_t)n * n;
    if ((rc = ensure(h, h->J, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->lu, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->gvec, sizeof(double) * 2 * n))) return rc;
    if ((rc = ensure(h, h->ipvt, sizeof(int32_t) * n))) return rc;
    if ((rc = ensure(h, h->fdev, sizeof(double) * n))) return rc;
    double *dJ = (double *)h->J.p, *dLU = (double *)h->lu.p, *dgrad = (double *)h->gvec.p, *drhs = dgrad + n;
    double *dfvec = (double *)h->fdev.p;
    int32_t *dipvt = (int32_t *)h->ipvt.p;
    hipStream_t s = h->stream;
    std::vector<double> dir(n), grad(n), xold(n), rhs(n);
    int xcnvrg = 0, fcnvrg = 0, gcnvrg = 0, neval = 0, iter = 0, njac = 0, flag = 0;
    double f, fold, stpmax, xnorm = 0, fnorm = 0, test;
    rc = 0;

    // :535  Jacobian requested before fvec is defined; result discarded, not counted.
    if ((rc = ev.jac(x, fvec, dJ))) return rc;

    if ((rc = ev.fcn(x, fvec))) return rc;                      // :538-547
    f = 0.5 * h_dot(n, fvec, fvec);
    neval += 1;
    test = 0.0;
    for (int i = 0; i < n; ++i) test = fmax(fabs(fvec[i]), test);
    if (test < o->ftol) fcnvrg = 1;

    if (!fcnvrg) {
        stpmax = 100.0 * fmax(h_norm2(n, x), (double)n);        // :553
        for (;;) {                                              // :556-620
            iter += 1;
            if ((rc = ev.jac(x, fvec, dJ))) break;              // :561-562
            njac += 1;
            // grad(i) = dot(jac(:,i), fvec)  (:565-567), rows ascending as in the reference
            HIPCHK(h, hipMemcpyAsync(dfvec, fvec, sizeof(double) * n, hipMemcpyHostToDevice, s));
            {
                Timed t(h, NLH_K_JTF);
                hipLaunchKernelGGL(k_jtf_exact, dim3((n + 255) / 256, 1), dim3(256), 0, s, n, n, (const double *)dJ, (const double *)dfvec, dgrad,
                                   (const LmState *)nullptr, -1);
            }
            // LU of a copy (:570) and solve for -fvec (:577)
            HIPCHK(h, hipMemcpyAsync(dLU, dJ, sizeof(double) * nn, hipMemcpyDeviceToDevice, s));
            launch_lu_factor(h, 1, n, dLU, dipvt, (int32_t *)nullptr);
            for (int i = 0; i < n; ++i) rhs[i] = -fvec[i];
            HIPCHK(h, hipMemcpyAsync(drhs, rhs.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(n >= 96 ? 1024 : 256), lu_solve_lds(n), s, n, (const double *)dLU, (const int32_t *)dipvt, drhs,
                               (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(dir.data(), drhs, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipMemcpyAsync(grad.data(), dgrad, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));

            memcpy(xold.data(), x, sizeof(double) * n);         // :573-574
            fold = f;

            if (o->use_line_search) {                           // :580-589
                const double temp = h_dot(n, dir.data(), dir.data());
                if (temp > stpmax) {
                    const double sc = stpmax / temp;
                    for (int i = 0; i < n; ++i) dir[i] = dir[i] * sc;
                }
                const double mag = h_norm2(n, dir.data());      // limit_search_vector, linesearch.f90:554-572
                if (mag != 0.0 && mag > stpmax) {
                    const double sc = stpmax / mag;
                    for (int i = 0; i < n; ++i) dir[i] = sc * dir[i];
                }
                int lcount = 0;
                rc = line_search(o, ev, n, xold.data(), grad.data(), dir.data(), x, fvec, fold, &f, &lcount);
                neval += lcount;
                if (rc) break;
            } else {                                            // :591-595
                for (int i = 0; i < n; ++i) x[i] = x[i] + dir[i];
                if ((rc = ev.fcn(x, fvec))) break;
                f = 0.5 * h_dot(n, fvec, fvec);
                neval += 1;
            }

            // test_convergence, src/nonlin_helper.f90:36-124
            int check = 0;
            xcnvrg = fcnvrg = gcnvrg = 0;
            {
                const double fc = 0.5 * h_dot(n, fvec, fvec);
                fnorm = 0.0; xnorm = 0.0;
                for (int i = 0; i < n; ++i) fnorm = fmax(fabs(fvec[i]), fnorm);
                if (fnorm < o->ftol) { fcnvrg = 1; check = 1; }
                else {
                    for (int i = 0; i < n; ++i) {
                        const double t = fabs(x[i] - xold[i]) / fmax(fabs(x[i]), 1.0);
                        xnorm = fmax(t, xnorm);
                    }
                    if (xnorm < o->xtol) { xcnvrg = 1; check = 1; }
                    else {
                        double tg = 0.0;
                        const double den = fmax(fc, 0.5 * (double)n);
                        for (int i = 0; i < n; ++i) tg = fmax(tg, fabs(grad[i]) * fmax(fabs(x[i]), 1.0) / den);
                        if (tg < o->gtol) gcnvrg = 1;
                    }
                }
            }
            if (check) break;
            if (gcnvrg) { rc = NLH_SPURIOUS_CONVERGENCE_ERROR; break; }     // :604-608
            if (o->print_status) print_status(iter, neval, njac, xnorm, fnorm);   // :611-613
            if (neval >= o->max_evals) { flag = 1; break; }     // :616-619
        }
    }
    if (ib) {                                                   // :624-632
        ib->iter_count = iter; ib->fcn_count = neval; ib->jacobian_count = njac; ib->gradient_count = 0;
        ib->converge_on_fcn = fcnvrg; ib->converge_on_chng = xcnvrg; ib->converge_on_zero_diff = gcnvrg;
    }
    if (rc) return rc;
    return flag ? NLH_CONVERGENCE_ERROR : 0;
}


// ===========================================================================
// Quasi-Newton (Broyden): qns_solve as a host loop; B, Q, R live on the device.
// Same division of labour as Newton: O(n) vector logic on the host in the reference's order,
// every O(n^2)/O(n^3) operation in the kernels of nlh_kernels_broyden.h.
// ===========================================================================

// B (column-major) -> Q, R: Householder QR with Q formed (qr_factor(b, q = q, r = r), :289)
// Householder steps on the row-major work array [A | E] (rows x ncA | rows x ncE); vbuf slot 0 must hold column 0 of A.
void launch_house_steps(nlh_handle *h, int nprob, int rows, int ncA, int ncE, double *dA, double *dE,
                               double *vbuf, double *wbuf, double *st, const LmState *gst, int gwant)
{
    // vbuf: [nprob][2][rows]; wbuf: room for [nprob][2][ncA + ncE]; st: room for [nprob][2][4]
    hipStream_t s = h->stream;
    const int steps = std::min(ncA, rows - 1), nc = ncA + ncE;
    if (steps < 1) return;
    if (rows <= QN_FUSED_MAXROWS) {
        // one pass per step: the update of step j-1 rides along with the sums of step j
        const bool skinny = (long)nc * nprob < 1536;     // few columns in total: 4 per workgroup so that the chip has work
        // more workgroups than CUs: one product tile each, so that two workgroups share a CU (their chains overlap)
        // (the four-column form always has two tiles: its summing waves run a tile behind its producing waves)
        const int dbuf = skinny ? 1 : ((long)((nc + 15) / 16) * nprob > 256 ? 0 : 1);
        const size_t sh3 = qn_fused_lds(rows, dbuf);
        for (int j = 0; j < steps; ++j) {
            if (skinny)
                hipLaunchKernelGGL(k_qn_house_fused<4>, dim3((nc + 3) / 4, nprob), dim3(512), sh3, s,
                                   rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, gst, gwant, dbuf);
            else
                hipLaunchKernelGGL(k_qn_house_fused<16>, dim3((nc + 15) / 16, nprob), dim3(256), sh3, s,
                                   rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, gst, gwant, dbuf);
        }
        const int jl = steps - 1, slot = jl & 1;
        hipLaunchKernelGGL(k_qn_house_apply, dim3((nc + 255) / 256, (rows - jl + QN_RC - 1) / QN_RC, nprob), dim3(256), 0, s,
                           rows, ncA, ncE, jl, dA, dE, vbuf, wbuf + (size_t)slot * nc, st + (size_t)slot * 4, 2 * nc, 8, 1, gst, gwant);
        return;
    }
    const bool wide = rows <= QN_DOT2_MAXROWS;          // workgroup-wide loads (reflector + two product tiles in LDS)
    const size_t sh2 = sizeof(double) * ((size_t)rows + 2 * QN_DOT2_TR * QN_DOT2_CG);
    for (int j = 0; j < steps; ++j) {
        if (wide)
            hipLaunchKernelGGL(k_qn_house_dot2, dim3((nc + QN_DOT2_CG - 1) / QN_DOT2_CG, nprob), dim3(256), sh2, s,
                               rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, gst, gwant);
        else if (rows <= QN_LDS_ROWS)
            hipLaunchKernelGGL(k_qn_house_dot<false>, dim3((nc + QN_DOT_BS - 1) / QN_DOT_BS, nprob), dim3(QN_DOT_BS),
                               sizeof(double) * rows, s, rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, gst, gwant);
        else                                                // the reflector does not fit LDS: it stays in global memory
            hipLaunchKernelGGL(k_qn_house_dot<true>, dim3((nc + QN_DOT_BS - 1) / QN_DOT_BS, nprob), dim3(QN_DOT_BS),
                               0, s, rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, gst, gwant);
        hipLaunchKernelGGL(k_qn_house_apply, dim3((nc + 255) / 256, (rows - j + QN_RC - 1) / QN_RC, nprob), dim3(256), 0, s,
                           rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, nc, 4, 0, gst, gwant);
    }
}

static void launch_qn_qr(nlh_handle *h, int nprob, int n, const double *dB, double *dQ, double *dRt, double *dvb,
                         const LmState *gst = nullptr, int gwant = -1)
{
    // dvb: per problem 2n (reflector column, two slots) + 2 x 2n (w, two slots) + 2 x 4 (tau, scal, beta)
    hipStream_t s = h->stream;
    double *vbuf = dvb, *wbuf = dvb + (size_t)nprob * 2 * n, *st = wbuf + (size_t)nprob * 4 * n;
    {
        dim3 grid((n + 31) / 32, (n + 31) / 32, nprob);
        hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, s, n, n, dB, dRt, n, gst, gwant);
    }
    hipLaunchKernelGGL(k_qn_qr_init, dim3(std::min(1024, (n * n + 255) / 256), nprob), dim3(256), 0, s, n, dRt, dQ, vbuf, gst, gwant);
    launch_house_steps(h, nprob, n, n, n, dRt, dQ, vbuf, wbuf, st, gst, gwant);
}

// Q1 R1 = Q R + u v^T (qr_rank1_update(q, r, s, dx), :307).  dwcs: 3n doubles per problem of scratch.
static void launch_qn_update(nlh_handle *h, int nprob, int n, double *dQ, double *dRt, const double *du,
                             const double *dv, double *dwcs, const LmState *gst = nullptr, int gwant = -1)
{
    hipStream_t s = h->stream;
    double *dw = dwcs, *dc = dwcs + (size_t)nprob * n, *dsn = dc + (size_t)nprob * n;
    hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, nprob), dim3(256), 0, s, n, n, dQ, du, dw, 1.0, gst, gwant);
    hipLaunchKernelGGL(k_qn_fold, dim3(nprob), dim3(64), 0, s, n, dw, dc, dsn, gst, gwant);
    const dim3 g1((n + 255) / 256, nprob);
    hipLaunchKernelGGL(k_qn_rot_q, g1, dim3(256), sizeof(double) * 2 * n, s, n, dQ, dc, dsn, 1, gst, gwant);
    hipLaunchKernelGGL(k_qn_hess_r, g1, dim3(256), sizeof(double) * 2 * n, s, n, dRt, dc, dsn, dw, dv, gst, gwant);
    if (n <= 1024) {
        const int bs = std::min(1024, ((n + 63) / 64) * 64);
        hipLaunchKernelGGL(k_qn_retri<1>, dim3(nprob), dim3(bs), sizeof(double) * 2 * n, s, n, dRt, dc, dsn, gst, gwant);
    } else if (qn_nc8(n)) {
        hipLaunchKernelGGL(k_qn_retri<8>, dim3(nprob), dim3(1024), sizeof(double) * 2 * n, s, n, dRt, dc, dsn, gst, gwant);
    } else {
        hipLaunchKernelGGL(k_qn_retri<4>, dim3(nprob), dim3(1024), sizeof(double) * 2 * n, s, n, dRt, dc, dsn, gst, gwant);
    }
    hipLaunchKernelGGL(k_qn_rot_q, g1, dim3(256), sizeof(double) * 2 * n, s, n, dQ, dc, dsn, 0, gst, gwant);
}

static int quasi_newton_core(nlh_handle *h, const nlh_options *o, int jdelta, int n, NewtonEval &ev, double *x,
                             double *fvec, nlh_iteration_behavior *ib)
{
    int rc;
    const size_t nn = (size_t)n * n;
    if (n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    if ((rc = ensure(h, h->J, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->qnQ, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->qnR, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->qnV, sizeof(double) * ((size_t)16 * n + 8)))) return rc;
    double *dB = (double *)h->J.p, *dQ = (double *)h->qnQ.p, *dRt = (double *)h->qnR.p;
    double *dv = (double *)h->qnV.p;
    double *ddx = dv, *ddf = dv + n, *dsv = dv + 2 * n, *dwcs = dv + 3 * n /* 3n */, *dgrad = dv + 6 * n,
           *dstep = dv + 7 * n, *dfv = dv + 8 * n, *dvb = dv + 9 * n /* 4n + 4 */;
    hipStream_t s = h->stream;
    std::vector<double> dx(n), df(n), fvold(n), xold(n);
    int restart = 1, xcnvrg = 0, fcnvrg = 0, gcnvrg = 0, neval = 0, iter = 0, njac = 0, flag = 0, jcount = 0;
    int ls_zero_diff = 0;                                       // lib%converge_on_zero_diff: .false. after every search (:318 of linesearch)
    double f, fold, stpmax, xnorm = 0, fnorm = 0, test;
    rc = 0;

    if ((rc = ev.fcn(x, fvec))) return rc;                      // :261-270
    f = 0.5 * h_dot(n, fvec, fvec);
    neval += 1;
    test = 0.0;
    for (int i = 0; i < n; ++i) test = fmax(fabs(fvec[i]), test);
    if (test < o->ftol) fcnvrg = 1;

    if (!fcnvrg) {
        stpmax = 100.0 * fmax(h_norm2(n, x), (double)n);        // :276
        for (;;) {                                              // :279-411
            iter += 1;
            if (restart) {                                      // :284-292
                if ((rc = ev.jac(x, fvec, dB))) break;
                njac += 1;
                launch_qn_qr(h, 1, n, dB, dQ, dRt, dvb);
                jcount = 0;
            } else {                                            // :294-310
                for (int i = 0; i < n; ++i) df[i] = fvec[i] - fvold[i];
                for (int i = 0; i < n; ++i) dx[i] = x[i] - xold[i];
                const double x2 = h_dot(n, dx.data(), dx.data());
                HIPCHK(h, hipMemcpyAsync(ddx, dx.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
                HIPCHK(h, hipMemcpyAsync(ddf, df.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_qn_resid, dim3((n + 255) / 256, 1), dim3(256), sizeof(double) * n, s, n, dB, ddx, ddf, x2, (const double *)nullptr, dsv, (const LmState *)nullptr, -1);
                hipLaunchKernelGGL(k_qn_rank1, dim3((n + 255) / 256, n, 1), dim3(256), 0, s, n, dB, dsv, ddx, (const LmState *)nullptr, -1);
                launch_qn_update(h, 1, n, dQ, dRt, dsv, ddx, dwcs);
                jcount += 1;
            }
            // grad = B^T f (:313), step = -R^-1 Q^T f (:322-328)
            HIPCHK(h, hipMemcpyAsync(dfv, fvec, sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, 1), dim3(256), 0, s, n, n, dB, dfv, dgrad, 1.0, (const LmState *)nullptr, -1);
            hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, 1), dim3(256), 0, s, n, n, dQ, dfv, dstep, -1.0, (const LmState *)nullptr, -1);
            hipLaunchKernelGGL(k_qn_solve_upper, dim3(1), dim3(std::min(1024, ((n + 63) / 64) * 64)), sizeof(double) * n, s, n, dRt, dstep, (size_t)n * n, (size_t)n, (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(dx.data(), dgrad, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipMemcpyAsync(df.data(), dstep, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));

            memcpy(xold.data(), x, sizeof(double) * n);         // :316-318
            memcpy(fvold.data(), fvec, sizeof(double) * n);
            fold = f;

            double temp = h_dot(n, dx.data(), df.data());       // :332-339
            if (temp >= 0.0) {
                restart = 1;
                if (o->print_status) print_status(iter, neval, njac, xnorm, fnorm);
                if (iter > 10 * o->max_evals + 100) { flag = 1; break; }    // the reference would spin here
                continue;
            }

            if (o->use_line_search) {                           // :342-351
                temp = h_dot(n, df.data(), df.data());
                if (temp > stpmax) {
                    const double sc = stpmax / temp;
                    for (int i = 0; i < n; ++i) df[i] = df[i] * sc;
                }
                const double mag = h_norm2(n, df.data());       // limit_search_vector
                if (mag != 0.0 && mag > stpmax) {
                    const double sc = stpmax / mag;
                    for (int i = 0; i < n; ++i) df[i] = sc * df[i];
                }
                int lcount = 0;
                rc = line_search(o, ev, n, xold.data(), dx.data(), df.data(), x, fvec, fold, &f, &lcount);
                neval += lcount;
                ls_zero_diff = 0;
                if (rc) break;
            } else {                                            // :353-357
                for (int i = 0; i < n; ++i) x[i] = x[i] + df[i];
                if ((rc = ev.fcn(x, fvec))) break;
                f = 0.5 * h_dot(n, fvec, fvec);
                neval += 1;
            }

            // test_convergence (:360-367); the gradient test runs only if the search reported a zero slope
            int check = 0;
            xcnvrg = fcnvrg = gcnvrg = 0;
            {
                const double fc = 0.5 * h_dot(n, fvec, fvec);
                fnorm = 0.0; xnorm = 0.0;
                for (int i = 0; i < n; ++i) fnorm = fmax(fabs(fvec[i]), fnorm);
                if (fnorm < o->ftol) { fcnvrg = 1; check = 1; }
                else {
                    for (int i = 0; i < n; ++i) {
                        const double t = fabs(x[i] - xold[i]) / fmax(fabs(x[i]), 1.0);
                        xnorm = fmax(t, xnorm);
                    }
                    if (xnorm < o->xtol) { xcnvrg = 1; check = 1; }
                    else if (ls_zero_diff && o->use_line_search) {
                        double tg = 0.0;
                        const double den = fmax(fc, 0.5 * (double)n);
                        for (int i = 0; i < n; ++i) tg = fmax(tg, fabs(dx[i]) * fmax(fabs(x[i]), 1.0) / den);
                        if (tg < o->gtol) gcnvrg = 1;
                    }
                }
            }
            if (!check) {                                       // :368-391
                if (gcnvrg) {
                    if (restart) { rc = NLH_SPURIOUS_CONVERGENCE_ERROR; break; }
                    restart = 1;
                } else {
                    restart = jcount >= jdelta ? 1 : 0;
                }
            } else {
                break;
            }
            if (o->print_status) print_status(iter, neval, njac, xnorm, fnorm);   // :398-400
            if (neval >= o->max_evals) { flag = 1; break; }     // :403-406
        }
    }
    if (ib) {                                                   // :414-422
        ib->iter_count = iter; ib->fcn_count = neval; ib->jacobian_count = njac; ib->gradient_count = 0;
        ib->converge_on_fcn = fcnvrg; ib->converge_on_chng = xcnvrg; ib->converge_on_zero_diff = gcnvrg;
    }
    if (rc) return rc;
    return flag ? NLH_CONVERGENCE_ERROR : 0;
}


int nlh_newton_solve(nlh_handle *h, const nlh_options *o, int32_t n, nlh_vecfcn fcn, nlh_jacfcn jacfcn,
                     void *ctx, double *x, double *fvec, nlh_iteration_behavior *ib)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib) memset(ib, 0, sizeof *ib);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :518
    if (!o || n < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    const size_t nn = (size_t)n * n;
    if ((rc = ensure(h, h->P, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
    if ((rc = ensure(h, h->wa4, sizeof(double) * n))) return rc;
    if ((rc = ensure_pinned(h, sizeof(double) * (nn + n)))) return rc;
    double *hP = (double *)h->pinned;
    hipStream_t s = h->stream;
    NewtonEval ev;
    ev.fcn = [&](const double *xx, double *ff) -> int { fcn(ctx, n, xx, n, ff); return 0; };
    ev.jac = [&](double *xx, const double *f0, double *dJ) -> int {
        if (jacfcn) {
            jacfcn(ctx, n, xx, n, hP);
            HIPCHK(h, hipMemcpyAsync(dJ, hP, sizeof(double) * nn, hipMemcpyHostToDevice, s));
            return 0;
        }
        for (int j = 0; j < n; ++j) {                           // vfh_jac_fcn :267-273
            const double temp = xx[j];
            double hh = NLH_SQRT_EPS * fabs(temp);
            if (hh == 0.0) hh = NLH_SQRT_EPS;
            xx[j] = temp + hh;
            fcn(ctx, n, xx, n, hP + (size_t)j * n);
            xx[j] = temp;
        }
        HIPCHK(h, hipMemcpyAsync(h->P.p, hP, sizeof(double) * nn, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->wa4.p, f0, sizeof(double) * n, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->xdev.p, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
        launch_fd(h, 1, n, n, (const double *)h->P.p, (const double *)h->wa4.p, (const double *)h->xdev.p, dJ, nullptr, -1);
        HIPCHK(h, hipStreamSynchronize(s));   // hP is reused by the next call
        return 0;
    };
    rc = newton_core(h, o, n, ev, x, fvec, ib);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return NLH_ERR_HIP; }
    return rc;
}

// newton_solver%solve / quasi_newton_solver%solve for a batch of device-model problems: the lock-step state machine of
// nlh_kernels_newton.h.  A round serves every problem in whatever stage it is.  Newton: the ones that want a Jacobian
// get J (analytic or forward differences), grad = J^T F in the reference's row order, the LU of J in place (the
// reference factors a copy; J is not read again in the iteration), the direction and the set-up of the line search.
// Quasi-Newton (broyden): an iteration starts either from a fresh Jacobian B and its QR factors with Q formed (:284-292)
// or from Broyden's rank-one update of B, Q and R (:294-310); then grad = B^T F, step = -R^-1 Q^T F (:313-328).  The ones
// with a trial point get F(x) and one turn of the search loop / the convergence test.  One 12-byte read-back per round.
static int square_lockstep(nlh_handle *h, const nlh_options *o, bool broyden, int jdelta, int32_t nprob, int32_t n,
                           const ResidualSource &rs, int32_t analytic, double *dx, double *dfvec,
                           nlh_iteration_behavior *ib, int32_t *status)
{
    const double *dA = rs.dA; const double gamma = rs.gamma;
    if (rs.user()) analytic = rs.jac != nullptr;                 // is_jacobian_defined (src/nonlin_multi_eqn_mult_var.f90:241)
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (broyden && n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    const size_t nn = (size_t)n * n, np = (size_t)nprob;
    if ((rc = ensure(h, h->J, sizeof(double) * nn * np))) return rc;
    if ((!analytic || rs.user()) && (rc = ensure(h, h->P, sizeof(double) * nn * np))) return rc;
    if ((rc = ensure(h, h->vecs, sizeof(double) * (broyden ? 17 : 3) * n * np + sizeof(double) * 16 * np))) return rc;
    if ((rc = ensure(h, h->ipvt, sizeof(int32_t) * n * np))) return rc;
    if ((rc = ensure(h, h->state, sizeof(LmState) * np))) return rc;
    if ((rc = ensure(h, h->misc, sizeof(NtState) * np + 64))) return rc;
    if ((rc = ensure_pinned(h, sizeof(NtState) * np + 64))) return rc;
    if (broyden) {
        if ((rc = ensure(h, h->qnQ, sizeof(double) * nn * np))) return rc;
        if ((rc = ensure(h, h->qnR, sizeof(double) * nn * np))) return rc;
    }
    double *dJ = (double *)h->J.p, *dP = (double *)h->P.p;       // J: the Jacobian / its LU; quasi-Newton: B
    double *dxold = (double *)h->vecs.p, *ddir = dxold + (size_t)n * np, *dgrad = ddir + (size_t)n * np;
    // quasi-Newton only: F(xold), dx, df, s, the update's scratch (3n), the QR's reflector / w slots (6n + 8), x2
    double *dfvold = dgrad + (size_t)n * np, *dddx = dfvold + (size_t)n * np, *dddf = dddx + (size_t)n * np;
    double *dsv = dddf + (size_t)n * np, *dwcs = dsv + (size_t)n * np, *dvb = dwcs + 3 * (size_t)n * np;
    double *dx2 = dvb + (6 * (size_t)n + 8) * np;
    double *dQ = (double *)h->qnQ.p, *dRt = (double *)h->qnR.p;
    int32_t *dipvt = (int32_t *)h->ipvt.p;
    LmState *st = (LmState *)h->state.p;
    int32_t *dcounts = (int32_t *)h->misc.p;
    NtState *ns = (NtState *)((char *)h->misc.p + 64);
    int32_t *hcounts = (int32_t *)h->pinned;
    NtState *hns = (NtState *)((char *)h->pinned + 64);
    hipStream_t s = h->stream;
    NtOpts no;
    no.ftol = o->ftol; no.xtol = o->xtol; no.gtol = o->gtol; no.ls_alpha = o->ls_alpha; no.ls_factor = o->ls_factor;
    no.max_evals = o->max_evals; no.ls_max_evals = o->ls_max_evals; no.use_line_search = o->use_line_search ? 1 : 0;
    no.broyden = broyden ? 1 : 0; no.jdelta = jdelta; no.pad = 0;
    const int pb = (nprob + 255) / 256;
    const bool echo = o->print_status && nprob == 1;             // the status block is a single solve's (:611-613)
    auto jacobian = [&]() -> int {                               // for the problems in stage NT_NEED_JAC
        if (rs.user())                                           // the user's jacobianfcn, or vfh_jac_fcn on the user's vecfcn
            return residual_jacobian(h, rs, nprob, n, n, dx, dfvec, dJ, dP, st, NT_NEED_JAC, false, false, true);
        if (analytic) {
            Timed t(h, NLH_K_DQ_JACOBIAN);
            hipLaunchKernelGGL(k_dq_jacobian<RB>, dim3((n + RB - 1) / RB, nprob), dim3(RB), sizeof(double) * n, s,
                               n, n, dA, gamma, (const double *)dx, dJ, (const LmState *)st, (int)NT_NEED_JAC);
        } else {                                                 // vfh_jac_fcn: n perturbed evaluations, (f1 - f0) / h
            launch_dq_panel(h, nprob, n, n, dA, rs.db, gamma, dx, dP, st, NT_NEED_JAC);
            launch_fd(h, nprob, n, n, dP, dfvec, dx, dJ, st, NT_NEED_JAC);
        }
        return 0;
    };

    // (ns_solve :535 asks for a Jacobian before fvec is defined and discards it: nothing observable for a device model.)
    hipLaunchKernelGGL(k_nt_reset, dim3(pb), dim3(256), 0, s, nprob, st, ns);
    if ((rc = residual_eval(h, rs, nprob, n, n, dx, dfvec, nullptr, st, NT_START))) return rc;   // :538 / :261
    hipLaunchKernelGGL(k_nt_start, dim3(nprob), dim3(256), 0, s, n, no, (const double *)dx, (const double *)dfvec, st, ns);
    int need_jac = nprob, update = 0;                            // upper bounds until the first read-back
    // a round advances every live problem by one evaluation at least (or, quasi-Newton, turns an iteration without a
    // descent direction into a restart: bounded by 10 max_evals + 100 iterations as in the host loop)
    const long max_rounds = broyden ? 11L * o->max_evals + (long)o->ls_max_evals + 128 : (long)o->max_evals + (long)o->ls_max_evals + 8;
    for (long round = 0; round < max_rounds; ++round) {
        if (!broyden && need_jac > 0) {
            if ((rc = jacobian())) return rc;
            {
                Timed t(h, NLH_K_JTF);                           // :565-567
                hipLaunchKernelGGL(k_jtf_exact, dim3((n + 255) / 256, nprob), dim3(256), 0, s, n, n, (const double *)dJ,
                                   (const double *)dfvec, dgrad, (const LmState *)st, (int)NT_NEED_JAC);
            }
            hipLaunchKernelGGL(k_nt_rhs, dim3((n + 255) / 256, nprob), dim3(256), 0, s, n, (const double *)dfvec, ddir,
                               (const LmState *)st);
            launch_lu_factor(h, nprob, n, dJ, dipvt, nullptr, st, NT_NEED_JAC);          // :570
            hipLaunchKernelGGL(k_lu_solve, dim3(nprob), dim3(n >= 96 ? 1024 : 256), lu_solve_lds(n), s, n, (const double *)dJ,
                               (const int32_t *)dipvt, ddir, (const LmState *)st, (int)NT_NEED_JAC);   // :577
            hipLaunchKernelGGL(k_nt_step_begin, dim3(nprob), dim3(256), 0, s, n, no, (int)NT_NEED_JAC, dx, dxold, ddir, (const double *)dgrad,
                               (const double *)dfvec, (double *)nullptr, st, ns);
        }
        if (broyden && (need_jac > 0 || update > 0)) {
            if (need_jac > 0) {                                  // :284-292: B = J(x), QR with Q formed
                if ((rc = jacobian())) return rc;
                launch_qn_qr(h, nprob, n, dJ, dQ, dRt, dvb, st, NT_NEED_JAC);
                hipLaunchKernelGGL(k_nt_advance, dim3(pb), dim3(256), 0, s, nprob, st, (int)NT_NEED_JAC, (int)NT_DIR);
            }
            if (update > 0) {                                    // :294-310: B += s dx^T, Q R <- Q R + s dx^T
                hipLaunchKernelGGL(k_qn_prep, dim3(nprob), dim3(256), 0, s, n, (const double *)dx, (const double *)dxold,
                                   (const double *)dfvec, (const double *)dfvold, dddx, dddf, dx2, (const LmState *)st);
                hipLaunchKernelGGL(k_qn_resid, dim3((n + 255) / 256, nprob), dim3(256), sizeof(double) * n, s, n, (const double *)dJ,
                                   (const double *)dddx, (const double *)dddf, 0.0, (const double *)dx2, dsv, (const LmState *)st, (int)NT_UPDATE);
                hipLaunchKernelGGL(k_qn_rank1, dim3((n + 255) / 256, n, nprob), dim3(256), 0, s, n, dJ, (const double *)dsv,
                                   (const double *)dddx, (const LmState *)st, (int)NT_UPDATE);
                launch_qn_update(h, nprob, n, dQ, dRt, dsv, dddx, dwcs, st, NT_UPDATE);
                hipLaunchKernelGGL(k_nt_advance, dim3(pb), dim3(256), 0, s, nprob, st, (int)NT_UPDATE, (int)NT_DIR);
            }
            // grad = B^T F (:313), step = -R^-1 Q^T F (:322-328)
            hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, nprob), dim3(256), 0, s, n, n, (const double *)dJ, (const double *)dfvec,
                               dgrad, 1.0, (const LmState *)st, (int)NT_DIR);
            hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, nprob), dim3(256), 0, s, n, n, (const double *)dQ, (const double *)dfvec,
                               ddir, -1.0, (const LmState *)st, (int)NT_DIR);
            hipLaunchKernelGGL(k_qn_solve_upper, dim3(nprob), dim3(std::min(1024, ((n + 63) / 64) * 64)), sizeof(double) * n, s, n,
                               (const double *)dRt, ddir, nn, (size_t)n, (const LmState *)st, (int)NT_DIR);
            hipLaunchKernelGGL(k_nt_step_begin, dim3(nprob), dim3(256), 0, s, n, no, (int)NT_DIR, dx, dxold, ddir, (const double *)dgrad,
                               (const double *)dfvec, dfvold, st, ns);
        }
        if ((rc = residual_eval(h, rs, nprob, n, n, dx, dfvec, nullptr, st, NT_TRIAL))) return rc;
        hipLaunchKernelGGL(k_nt_trial, dim3(nprob), dim3(256), 0, s, n, no, dx, (const double *)dxold, (const double *)ddir,
                           (const double *)dgrad, (const double *)dfvec, st, ns);
        hipLaunchKernelGGL(k_nt_count, dim3(1), dim3(256), 0, s, nprob, (const LmState *)st, dcounts);
        HIPCHK(h, hipMemcpyAsync(hcounts, dcounts, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (echo) HIPCHK(h, hipMemcpyAsync(hns, ns, sizeof(NtState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (echo && hns[0].print_due) print_status(hns[0].iter, hns[0].neval, hns[0].njac, hns[0].xnorm, hns[0].fnorm);
        need_jac = hcounts[0]; update = hcounts[2];
        if (need_jac == 0 && update == 0 && hcounts[1] == 0) break;
    }
    HIPCHK(h, hipMemcpyAsync(hns, ns, sizeof(NtState) * np, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    for (int p = 0; p < nprob; ++p) {
        const NtState &q = hns[p];
        if (ib) {                                                // :624-632 / :414-422
            ib[p].iter_count = q.iter; ib[p].fcn_count = q.neval; ib[p].jacobian_count = q.njac; ib[p].gradient_count = 0;
            ib[p].converge_on_fcn = q.fcnvrg; ib[p].converge_on_chng = q.xcnvrg; ib[p].converge_on_zero_diff = q.gcnvrg;
        }
        const bool finished = q.rc || q.flag || q.fcnvrg || q.xcnvrg;
        if (status) status[p] = q.rc ? q.rc : ((q.flag || !finished) ? NLH_CONVERGENCE_ERROR : 0);       // :635-637 / :425-427
    }
    return 0;
}

// One square solver over a batch, in slices the lock-step kernels' grid dimensions (and, for a user's launcher, the point
// count of one Jacobian call) hold.
static int square_batch_rs(nlh_handle *h, const nlh_options *o, bool broyden, int jdelta, int32_t nprob, int32_t n, const ResidualSource &rs,
                           int32_t analytic, double *dx, double *dfvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!o || n < 1 || nprob < 1) return NLH_INVALID_INPUT_ERROR;
    const int32_t slice = rs.user() ? (int32_t)std::max<int64_t>(1, std::min<int64_t>(NLH_MAX_LOCKSTEP, ((int64_t)1 << 30) / n)) : NLH_MAX_LOCKSTEP;
    for (int32_t p0 = 0; p0 < nprob; p0 += slice) {
        const int32_t cnt = std::min<int32_t>(slice, nprob - p0);
        const int rc = square_lockstep(h, o, broyden, jdelta, cnt, n, rs.shifted(p0, n, n), analytic, dx + (size_t)p0 * n, dfvec + (size_t)p0 * n,
                                       ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
        if (rc) return rc;
    }
    return 0;
}

int nlh_dq_newton_solve_batch(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t n, const double *dA,
                              const double *db, double gamma, int32_t analytic, double *dx, double *dfvec,
                              nlh_iteration_behavior *ib, int32_t *status)
{
    ResidualSource rs;
    rs.dA = dA; rs.db = db; rs.gamma = gamma;
    return square_batch_rs(h, o, false, 0, nprob, n, rs, analytic, dx, dfvec, ib, status);
}

// newton_solver%solve / quasi_newton_solver%solve on a batch of square problems whose residual (and, optionally, Jacobian)
// is the USER'S device function (launchers, include/nonlin_hip.h).
static int square_device(nlh_handle *h, const nlh_options *o, bool broyden, int jdelta, int32_t nprob, int32_t n, nlh_device_vecfcn fcn,
                         nlh_device_jacfcn jacfcn, void *ctx, double *dx, double *dfvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib && nprob > 0) memset(ib, 0, sizeof(*ib) * (size_t)nprob);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // src/nonlin_solve.f90:516 / :238
    if (nprob <= 0) return 0;
    if (!o || !dx || !dfvec) return NLH_INVALID_INPUT_ERROR;
    ResidualSource rs;
    rs.fcn = fcn; rs.jac = jacfcn; rs.ctx = ctx;
    nlh_options oq = *o;
    if (nprob > 1) oq.print_status = 0;                         // the status block is a single solve's (:611-613)
    return square_batch_rs(h, &oq, broyden, jdelta, nprob, n, rs, 0, dx, dfvec, ib, status);
}

static int square_device_h(nlh_handle *h, const nlh_options *o, bool broyden, int jdelta, int32_t nprob, int32_t n, nlh_device_vecfcn fcn,
                           nlh_device_jacfcn jacfcn, void *ctx, double *x, double *fvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (nprob <= 0) return 0;
    if (!o || !x || !fvec || n < 1) return NLH_INVALID_INPUT_ERROR;
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    const size_t cnt = (size_t)nprob * n;
    if ((rc = ensure(h, h->xdev, sizeof(double) * cnt))) return rc;
    if ((rc = ensure(h, h->fdev, sizeof(double) * cnt))) return rc;
    double *dx = (double *)h->xdev.p, *df = (double *)h->fdev.p;
    HIPCHK(h, hipMemcpyAsync(dx, x, sizeof(double) * cnt, hipMemcpyHostToDevice, h->stream));
    if ((rc = square_device(h, o, broyden, jdelta, nprob, n, fcn, jacfcn, ctx, dx, df, ib, status))) return rc;
    HIPCHK(h, hipMemcpyAsync(x, dx, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(fvec, df, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int nlh_newton_solve_batch_device(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t n, nlh_device_vecfcn fcn,
                                  nlh_device_jacfcn jacfcn, void *ctx, double *dx, double *dfvec, nlh_iteration_behavior *ib,
                                  int32_t *status)
{
    return square_device(h, o, false, 0, nprob, n, fcn, jacfcn, ctx, dx, dfvec, ib, status);
}

int nlh_quasi_newton_solve_batch_device(nlh_handle *h, const nlh_options *o, int32_t jdelta, int32_t nprob, int32_t n,
                                        nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx, double *dx, double *dfvec,
                                        nlh_iteration_behavior *ib, int32_t *status)
{
    return square_device(h, o, true, jdelta, nprob, n, fcn, jacfcn, ctx, dx, dfvec, ib, status);
}

int nlh_newton_solve_batch_device_h(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t n, nlh_device_vecfcn fcn,
                                    nlh_device_jacfcn jacfcn, void *ctx, double *x, double *fvec, nlh_iteration_behavior *ib,
                                    int32_t *status)
{
    return square_device_h(h, o, false, 0, nprob, n, fcn, jacfcn, ctx, x, fvec, ib, status);
}

int nlh_quasi_newton_solve_batch_device_h(nlh_handle *h, const nlh_options *o, int32_t jdelta, int32_t nprob, int32_t n,
                                          nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx, double *x, double *fvec,
                                          nlh_iteration_behavior *ib, int32_t *status)
{
    return square_device_h(h, o, true, jdelta, nprob, n, fcn, jacfcn, ctx, x, fvec, ib, status);
}

// quasi_newton_solver%solve -- qns_solve, src/nonlin_solve.f90:156-427
int nlh_quasi_newton_solve(nlh_handle *h, const nlh_options *o, int32_t jdelta, int32_t n, nlh_vecfcn fcn,
                           nlh_jacfcn jacfcn, void *ctx, double *x, double *fvec, nlh_iteration_behavior *ib)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib) memset(ib, 0, sizeof *ib);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :240
    if (!o || n < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    const size_t nn = (size_t)n * n;
    if ((rc = ensure(h, h->P, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
    if ((rc = ensure(h, h->wa4, sizeof(double) * n))) return rc;
    if ((rc = ensure_pinned(h, sizeof(double) * (nn + n)))) return rc;
    double *hP = (double *)h->pinned;
    hipStream_t s = h->stream;
    NewtonEval ev;
    ev.fcn = [&](const double *xx, double *ff) -> int { fcn(ctx, n, xx, n, ff); return 0; };
    ev.jac = [&](double *xx, const double *f0, double *dJ) -> int {
        if (jacfcn) {
            jacfcn(ctx, n, xx, n, hP);
            HIPCHK(h, hipMemcpyAsync(dJ, hP, sizeof(double) * nn, hipMemcpyHostToDevice, s));
            HIPCHK(h, hipStreamSynchronize(s));
            return 0;
        }
        for (int j = 0; j < n; ++j) {                           // vfh_jac_fcn :267-273
            const double temp = xx[j];
            double hh = NLH_SQRT_EPS * fabs(temp);
            if (hh == 0.0) hh = NLH_SQRT_EPS;
            xx[j] = temp + hh;
            fcn(ctx, n, xx, n, hP + (size_t)j * n);
            xx[j] = temp;
        }
        HIPCHK(h, hipMemcpyAsync(h->P.p, hP, sizeof(double) * nn, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->wa4.p, f0, sizeof(double) * n, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->xdev.p, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
        launch_fd(h, 1, n, n, (const double *)h->P.p, (const double *)h->wa4.p, (const double *)h->xdev.p, dJ, nullptr, -1);
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    };
    rc = quasi_newton_core(h, o, jdelta, n, ev, x, fvec, ib);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return NLH_ERR_HIP; }
    return rc;
}

int nlh_dq_quasi_newton_solve_batch(nlh_handle *h, const nlh_options *o, int32_t jdelta, int32_t nprob, int32_t n,
                                    const double *dA, const double *db, double gamma, int32_t analytic, double *dx,
                                    double *dfvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!o || n < 1 || nprob < 1) return NLH_INVALID_INPUT_ERROR;
    ResidualSource rs;
    rs.dA = dA; rs.db = db; rs.gamma = gamma;
    return square_batch_rs(h, o, true, jdelta, nprob, n, rs, analytic, dx, dfvec, ib, status);   // the same state machine
}

int nlh_lu_factor(nlh_handle *h, int32_t nprob, int32_t n, double *dA, int32_t *dipvt, int32_t *dinfo)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    launch_lu_factor(h, nprob, n, dA, dipvt, dinfo);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_lu_solve(nlh_handle *h, int32_t nprob, int32_t n, const double *dLU, const int32_t *dipvt, double *db)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(k_lu_solve, dim3(nprob), dim3(n >= 96 ? 1024 : 256), lu_solve_lds(n), h->stream, n, dLU, dipvt, db,
                       (const LmState *)nullptr, -1);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// qr_factor(b, q = q, r = r) / qr_rank1_update / solve_triangular_system stand-ins (call sites :289, :307, :327).
// dB, dQ column-major [nprob][n][n]; dRt is R ROW-major.
int nlh_qr_factor_full(nlh_handle *h, int32_t nprob, int32_t n, const double *dB, double *dQ, double *dRt)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (n < 1 || nprob < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->qnV, sizeof(double) * ((size_t)nprob * (16 * (size_t)n + 8))))) return rc;
    launch_qn_qr(h, nprob, n, dB, dQ, dRt, (double *)h->qnV.p);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_qr_rank1_update(nlh_handle *h, int32_t nprob, int32_t n, double *dQ, double *dRt, const double *du,
                        const double *dv)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (n < 1 || nprob < 1) return NLH_INVALID_INPUT_ERROR;
    if (n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->qnV, sizeof(double) * ((size_t)nprob * (16 * (size_t)n + 8))))) return rc;
    launch_qn_update(h, nprob, n, dQ, dRt, du, dv, (double *)h->qnV.p);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_solve_upper(nlh_handle *h, int32_t nprob, int32_t n, const double *dRt, double *dx)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (n < 1 || nprob < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(k_qn_solve_upper, dim3(nprob), dim3(std::min(1024, ((n + 63) / 64) * 64)), sizeof(double) * n,
                       h->stream, n, dRt, dx, (size_t)n * n, (size_t)n, (const LmState *)nullptr, -1);
    HIPCHK(h, hipGetLastError());
    return 0;
}
