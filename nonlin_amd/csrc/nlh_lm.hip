// nlh_lm.hip -- least_squares_solver: lss_solve (src/nonlin_least_squares.f90:118-391) as a batched, device-resident
// lock-step state machine (nlh_dq_lm_solve_batch), with host callbacks (nlh_lm_solve, nlh_fd_jacobian = vfh_jac_fcn,
// src/nonlin_multi_eqn_mult_var.f90:198-277), and its stages as entry points of their own (nlh_gram, nlh_chol_factor,
// nlh_qr_factor, nlh_lmfactor_exact, nlh_lmpar).
#include "nlh_internal.h"
#include <chrono>
#include "nlh_kernels_gram.h"
#include "nlh_kernels_factor.h"
#include "nlh_kernels_lm.h"
#include "nlh_kernels_exact.h"
#include "nlh_qrx.h"


static __global__ void k_lmpar_standalone(int n, double *Rall, int ldr, const int32_t *ipvt_all, const double *diag_all,
                                   const double *qtf_all, const double *delta_all, const double *tailsq_all,
                                   double *par_all, double *x_all, double *sdiag_all, double *Wall, int ringcap);



void nlh_lm_init_device(int lds_max)
{
    hipFuncSetAttribute((const void *)k_gram_tri<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_gram_tri<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_gram_512, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_chol_factor, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_chol_nopiv<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_chol_mc_step<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qr_factor, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lmpar<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lmpar<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lmpar<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lmpar_standalone, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
}


// Brackets for the launches of nlh_qrx.hip (another translation unit): which = 0 pivot kernel, 1 trailing pass, 2 rest.
static void qrx_time_begin(nlh_handle *h, int which, hipStream_t s)
{
    const int kid = which == 1 ? NLH_K_QRX_PASS : which == 0 ? NLH_K_QRX_PIVOT : NLH_K_QR;
    h->qrx_open_on = (h->timing >> kid) & 1u;
    if (h->qrx_open_on) { h->qrx_a = ev_get(h); h->qrx_b = ev_get(h); hipEventRecord(h->qrx_a, s); }
    h->qrx_kid = kid;
}
static void qrx_time_end(nlh_handle *h, int, hipStream_t s)
{
    if (!h->qrx_open_on) return;
    hipEventRecord(h->qrx_b, s);
    h->pending.push_back({h->qrx_a, h->qrx_b, h->qrx_kid});
    if (h->pending.size() > 65536) timing_flush(h);
}
// ---------------------------------------------------------------------------
// small helper kernels of the drivers
// ---------------------------------------------------------------------------
static __global__ void k_stage_advance(int nprob, LmState *st, int from, int to, int njac_inc)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprob) return;
    if (st[p].stage == from) { st[p].stage = to; st[p].njac += njac_inc; }
}

// :211-218: fnorm of the starting residual, counters.
static __global__ void k_lm_init(int nprob, int nblk, const double *part, LmState *st, int first_stage)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprob) return;
    double sq = 0.0;
    for (int k = 0; k < nblk; ++k) sq = sq + part[((size_t)p * nblk + k) * 2];
    LmState s;
    memset(&s, 0, sizeof s);
    s.fnorm = sqrt(sq);
    s.neval = 1;
    s.iter = 1;
    s.par = 0.0;
    s.stage = first_stage;
    st[p] = s;
}

static __global__ void k_count_active(int nprob, const LmState *st, int *out)
{
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    __shared__ int cntj;
    if (threadIdx.x == 0) cntj = 0;
    __syncthreads();
    int c = 0, cj = 0;
    for (int p = threadIdx.x; p < nprob; p += blockDim.x) { c += (st[p].stage != ST_DONE); cj += (st[p].stage == ST_NEED_JAC); }
    atomicAdd(&cnt, c);
    atomicAdd(&cntj, cj);
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = cnt; out[1] = cntj; }      // still iterating; of those, due for a Jacobian + factorisation
}

// partial sums of squares of a device vector: the block structure AND the in-block order of k_dq_residual (PAIR false) /
// k_dq_residual2 (PAIR true: two rows per thread, their squares added first), so that a residual that arrives from a
// launcher or a host callback gives the normal-equations policies the bits the built-in family's fused sums give
template <int BS, bool PAIR>
__global__ void k_sumsq_part(int m, int n, const double *__restrict__ f, double *__restrict__ part)
{
    __shared__ double red[16];
    const int p = blockIdx.y;
    double sq, tq;
    if (PAIR) {
        const int i = (blockIdx.x * BS + threadIdx.x) * 2;      // m even: i + 1 < m as well
        const double v0 = (i < m) ? f[(size_t)p * m + i] : 0.0, v1 = (i < m) ? f[(size_t)p * m + i + 1] : 0.0;
        const double q0 = v0 * v0, q1 = v1 * v1;
        sq = q0 + q1;
        tq = ((i < m && i >= n) ? q0 : 0.0) + ((i < m && i + 1 >= n) ? q1 : 0.0);
    } else {
        const int i = blockIdx.x * BS + threadIdx.x;
        const double v = (i < m) ? f[(size_t)p * m + i] : 0.0;
        sq = v * v;
        tq = (i >= n) ? sq : 0.0;
    }
    const double s = block_reduce_sum(sq, red);
    const double t = block_reduce_sum(tq, red);
    if (threadIdx.x == 0) {
        part[((size_t)p * gridDim.x + blockIdx.x) * 2 + 0] = s;
        part[((size_t)p * gridDim.x + blockIdx.x) * 2 + 1] = t;
    }
}

void launch_sumsq_part(nlh_handle *h, int nprob, int m, int n, const double *f, double *part)
{
    const dim3 grid((m + RB - 1) / RB, nprob);
    if (m % 2 == 0) hipLaunchKernelGGL((k_sumsq_part<RB / 2, true>), grid, dim3(RB / 2), 0, h->stream, m, n, f, part);
    else hipLaunchKernelGGL((k_sumsq_part<RB, false>), grid, dim3(RB), 0, h->stream, m, n, f, part);
}

// K-splits of the Gram contraction.  The split count and the kernel are functions of the problem SHAPE only, never of how
// many problems share the launch: G = sum over splits (in split order) of a row-ascending accumulation, so a problem's
// bits do not depend on its batch, the round it is active in, the sub-batch or the rank it was dealt to.
static int gram_splits(int m)
{
    const int s = (m + 1023) / 1024;                  // 1024 rows per split (4096 rows per split was measured: no gain for
                                                      // 512 x 4096x256, and one problem alone 3.3 -> 7.0 ms per solve)
    return s < 1 ? 1 : s;
}

static bool gram512_on() { const char *e = getenv("NLH_GRAM512"); return !e || atoi(e) != 0; }   // (0: k_gram_mfma for 256 < n <= 512, for comparison)

static int launch_gram(nlh_handle *h, int nprob, int m, int n, const double *J, const double *f,
                       double *G, double *g, const LmState *st, int want)
{
    const int nb = (n + GRAM_BT - 1) / GRAM_BT;
    const int nblk = nb * (nb + 1) / 2;
    const int ns = gram_splits(m);
    int rps = (m + ns - 1) / ns;
    rps = ((rps + GRAM_KT - 1) / GRAM_KT) * GRAM_KT;
    int rc = ensure(h, h->Gpart, sizeof(double) * ((size_t)nprob * ns * n * n + (size_t)nprob * ns * n));
    if (rc) return rc;
    double *Gp = (double *)h->Gpart.p;
    double *gp = Gp + (size_t)nprob * ns * n * n;
    {
        Timed t(h, NLH_K_GRAM);
        const long items = (long)ns * nprob;
        const bool tri16 = n > 224 && n <= 256;
        const bool tri8 = n > 96 && n <= 128;
        if (tri16 || tri8) {
            // whole lower triangle per workgroup, J staged once
            const int nt = tri16 ? 16 : 8;
            const size_t sh = sizeof(double) * (size_t)(16 * nt * GRAM_LD + GRAM_KT + 64 * nt);
            const bool direct = ns == 1;
            if (tri16)
                hipLaunchKernelGGL(k_gram_tri<16>, dim3((unsigned)items), dim3(512), sh, h->stream, m, n, rps, J, Gp,
                                   g ? f : (const double *)nullptr, gp, st, want, ns, direct ? G : (double *)nullptr,
                                   direct ? g : (double *)nullptr);
            else
                hipLaunchKernelGGL(k_gram_tri<8>, dim3((unsigned)items), dim3(256), sh, h->stream, m, n, rps, J, Gp,
                                   g ? f : (const double *)nullptr, gp, st, want, ns, direct ? G : (double *)nullptr,
                                   direct ? g : (double *)nullptr);
            if (direct) return 0;          // one split: G and g are final, nothing to reduce
        } else if (n > 256 && n <= 512 && gram512_on()) {
            // four workgroups per item: the two diagonal 256-column blocks and the two halves of the square between them
            const long groups = (items + 7) / 8;
            const size_t sh = sizeof(double) * (size_t)(2 * 384 * GRAM_LD2 + 2 * GRAM_KT2 + 1024);   // two 16-row tile buffers
            hipLaunchKernelGGL(k_gram_512, dim3((unsigned)(groups * 32)), dim3(512), sh, h->stream, m, n, rps, J, Gp,
                               g ? f : (const double *)nullptr, gp, st, want, ns, nprob);
        } else {
            const long groups = (items + 7) / 8;
            hipLaunchKernelGGL(k_gram_mfma, dim3((unsigned)(groups * 8 * nblk)), dim3(256), 0, h->stream, m, n, rps, J, Gp,
                               g ? f : (const double *)nullptr, gp, st, want, nblk, ns, nprob);
        }
    }
    {
        Timed t(h, NLH_K_GRAM_REDUCE);
        dim3 grid((unsigned)(((size_t)n * n + 255) / 256), nprob);
        hipLaunchKernelGGL(k_gram_reduce, grid, dim3(256), 0, h->stream, n, ns, (const double *)Gp, G,
                           (const double *)gp, g, st, want);
    }
    return 0;
}

// up to here k_lmpar's six n-vectors (+ the exact reductions' scratch) fit 158 KB of LDS (NLH_LM_LDS_MAX_N: a smaller bound,
// so that tests reach the global-memory form at small sizes)
static const int LM_LDS_MAX_N = [] { const char *e = getenv("NLH_LM_LDS_MAX_N"); const int v = e ? atoi(e) : 3000; return v < 3000 ? v : 3000; }();

// LDS of lmsolve's on-chip sweep for a k_lmpar launch (0 / *cap = 0: the global-memory wavefront).  NLH_LMSOLVE_GLOBAL=1
// (read per call: tests compare the two forms bit for bit) forces the latter at any n.
static size_t lm_ring_bytes(int n, int threads, int *cap)
{
    const char *e = getenv("NLH_LMSOLVE_GLOBAL");
    if (e && atoi(e)) { *cap = 0; return 0; }
    return lmsolve_ring_bytes(n, threads, cap);
}

struct LmWs {
    double *J, *P, *wa4, *scratch, *G, *g, *part, *W2, *R;
    LmVecs v;
    LmState *st;
    int32_t *info;
    int nblk;
};

static int lm_workspace(nlh_handle *h, int nprob, int m, int n, LmWs &w, bool need_panel, bool need_J = true)
{
    int rc;
    const size_t mn = (size_t)nprob * m * n, pm = (size_t)nprob * m, pn = (size_t)nprob * n;
    // need_J = false: the exact policy with the fused FD epilogue writes the Jacobian straight into the factorisation's
    // working matrix (the panel buffer) and nothing reads a column-major J: 17 GB less at 2048 x 4096x256
    if (need_J && (rc = ensure(h, h->J, sizeof(double) * mn))) return rc;
    // the panel doubles as the exact factorisation's row-major working matrix (nlh_qrx.hip) and as lmsolve's scratch
    if (need_panel && (rc = ensure(h, h->P, sizeof(double) * std::max(mn + pm + (size_t)512 * (n + 1),
                                                                      qrx_matrix_doubles(nprob, m, n))))) return rc;
    if ((rc = ensure(h, h->wa4, sizeof(double) * pm))) return rc;
    if ((rc = ensure(h, h->scratch, sizeof(double) * pm))) return rc;
    if ((rc = ensure(h, h->G, sizeof(double) * (size_t)nprob * n * n))) return rc;
    if ((rc = ensure(h, h->W2, sizeof(double) * (size_t)nprob * n * n))) return rc;
    if ((rc = ensure(h, h->R, sizeof(double) * (size_t)nprob * n * n))) return rc;
    if ((rc = ensure(h, h->vecs, sizeof(double) * pn * 10))) return rc;
    if ((rc = ensure(h, h->ipvt, sizeof(int32_t) * pn))) return rc;
    if ((rc = ensure(h, h->gvec, sizeof(double) * pn))) return rc;
    w.nblk = (m + RB - 1) / RB;
    if ((rc = ensure(h, h->part, sizeof(double) * (size_t)nprob * w.nblk * 2))) return rc;
    if ((rc = ensure(h, h->state, sizeof(LmState) * (size_t)nprob))) return rc;
    if ((rc = ensure(h, h->info, sizeof(int32_t) * (size_t)(nprob + 16)))) return rc;
    w.J = (double *)h->J.p; w.P = (double *)h->P.p; w.wa4 = (double *)h->wa4.p;
    w.scratch = (double *)h->scratch.p; w.G = (double *)h->G.p; w.g = (double *)h->gvec.p;
    w.part = (double *)h->part.p; w.st = (LmState *)h->state.p; w.info = (int32_t *)h->info.p;
    w.W2 = (double *)h->W2.p;
    w.R = (double *)h->R.p;
    double *vb = (double *)h->vecs.p;
    w.v.diag = vb; w.v.diag_prev = vb + pn; w.v.qtf = vb + 2 * pn; w.v.acnorm = vb + 3 * pn;
    w.v.rdiag = vb + 4 * pn; w.v.g = vb + 5 * pn; w.v.wa1 = vb + 6 * pn; w.v.wa2 = vb + 7 * pn;
    w.v.wa3 = vb + 8 * pn; w.v.sdiag = vb + 9 * pn;
    w.v.ipvt = (int32_t *)h->ipvt.p;
    return 0;
}

// One pass over the factorisation + lmpar stages for every problem whose Jacobian is in
// w.J (stage ST_HAVE_JAC or ST_NEED_QR) or whose factors are ready (inner-loop repeat).
static int lm_factor_and_step(nlh_handle *h, const nlh_options *o, int nprob, int m, int n, LmWs &w,
                              double *dx, const double *dfvec, int nact = -1, bool jac_in_qrx_layout = false, bool any_fresh = true)
{   // any_fresh = false: no problem has a fresh Jacobian this round (the exact policy then skips the factorisation's launches)
    const int ft = factor_threads(n), lt_ = lmpar_threads(n);
    const size_t shl = sizeof(double) * (size_t)(6 * n + 72);
    if (o->factor_policy == NLH_FACTOR_EXACT) {
        // reference operation order: exact lmfactor + Q^T f (streaming form, the batch advances through the
        // Householder steps in lock step: nlh_qrx.hip), exact lmpar
        if (any_fresh) {
            int rc;
            if ((rc = ensure(h, h->qxV, qrx_workspace_bytes(nprob, m, n)))) return rc;
            QrxTimer tm{h, [](void *c, int which, hipStream_t s) { qrx_time_begin((nlh_handle *)c, which, s); },
                        [](void *c, int which, hipStream_t s) { qrx_time_end((nlh_handle *)c, which, s); }};
            qrx_factor(h->stream, nprob, m, n, jac_in_qrx_layout ? (const double *)nullptr : w.J, w.P, dfvec, w.R, w.v, w.wa4,
                       w.scratch, dx, w.st, o->factor, o->gtol, h->qxV.p, &tm, nact);
        }
        {
            // (Measured and dropped in round 5: lmpar's ITERATION -- the few problems whose Gauss-Newton step is not accepted; ten
            // lmsolve sweeps, milliseconds of one workgroup -- on a side stream, the problems joining the next round at their
            // trial point.  It happens only in a problem's LAST outer iteration on these families (deviation A sends par to
            // +Inf), the round it is pushed into exists anyway, and every extra round costs the batch its launches: 3,603 ->
            // 3,585 LM it/s for one lock-step batch of 2048 x 4096x256, 3,681 -> 3,511 with sub-batches; 47 x 4096x256 153.5 ->
            // 155.7 ms.  docs/lab_notebook.md.)
            Timed t(h, NLH_K_LMPAR);
            if (n <= LM_LDS_MAX_N) {
                int cap;
                const size_t rb = lm_ring_bytes(n, lt_, &cap);
                hipLaunchKernelGGL(k_lmpar<true>, dim3(nprob), dim3(lt_), shl + sizeof(double) * lmpar_scratch_doubles(lt_) + rb,
                                   h->stream, m, n, w.R, w.v, dx, w.wa4, w.P, w.J, w.W2, w.st, (int)ST_QR_READY, (double *)nullptr, cap);
            }
            else {                                              // lmpar's n-vectors in global memory (the misc buffer)
                int rc2;
                if ((rc2 = ensure(h, h->misc, sizeof(double) * (size_t)nprob * (6 * (size_t)n + 8)))) return rc2;
                hipLaunchKernelGGL((k_lmpar<true, true>), dim3(nprob), dim3(ft), sizeof(double) * (size_t)(64 + lmpar_scratch_doubles(ft)),
                                   h->stream, m, n, w.R, w.v, dx, w.wa4, w.P, w.J, w.W2, w.st, (int)ST_QR_READY, (double *)h->misc.p);
            }
        }
        return 0;
    }
    int rc = launch_gram(h, nprob, m, n, w.J, dfvec, w.G, w.g, w.st, ST_HAVE_JAC);
    if (rc) return rc;
    int ringcap;                                                // lmsolve's on-chip sweep (the launches that can reach lmpar's iteration)
    const size_t ringb = lm_ring_bytes(n, lt_, &ringcap);
    constexpr int NB = 16;
    {   // fast path: blocked Cholesky in natural order, G -> R
        Timed t(h, NLH_K_CHOL);
        size_t sh = sizeof(double) * ((size_t)NB * n + NB * NB + n + NB + 64);
        // a handful of problems (BASELINE config 5's one 65536 x 512 problem): a launch per panel step over many CUs
        // instead of one workgroup per problem; same bits (nlh_kernels_factor.h)
        const char *mc_e = getenv("NLH_CHOL_MC");                   // (read per call: tests switch between the two forms)
        const int mc_max = mc_e ? atoi(mc_e) : 8;
        const int na = nact < 0 ? nprob : nact;
        // (its launches cover every problem of the batch, active or not: only for batches that are small themselves)
        if (sh <= 150 * 1024 && n >= 192 && na <= mc_max && nprob <= 4 * std::max(mc_max, 1)) {
            int rc2;
            if ((rc2 = ensure(h, h->cholmc, sizeof(double) * (size_t)nprob * 2 * (NB * n + NB * NB + NB) + sizeof(int32_t) * (size_t)nprob + 64))) return rc2;
            double *side = (double *)h->cholmc.p;
            double *fact = side + (size_t)nprob * 2 * NB * n;
            int32_t *bad = (int32_t *)(fact + (size_t)nprob * 2 * (NB * NB + NB));
            hipLaunchKernelGGL(k_chol_mc_begin<NB>, dim3(64, nprob), dim3(256), 0, h->stream, n, (const double *)w.G, (const double *)w.g, w.R, w.v,
                               fact, bad, (const LmState *)w.st, o->ne_pivot_tol);
            const size_t shm = sizeof(double) * ((size_t)NB * n + 2 * (NB * NB + 2 * NB));
            for (int jb = 0; jb < n; jb += NB)
                hipLaunchKernelGGL(k_chol_mc_step<NB>, dim3(CHOLMC_NWG + 2, nprob), dim3(512), shm, h->stream, n, jb, w.R, w.v, side, fact, bad,
                                   (const LmState *)w.st, o->ne_pivot_tol);
            hipLaunchKernelGGL(k_chol_mc_end<NB>, dim3(nprob), dim3(ft), sizeof(double) * (size_t)n, h->stream, n, w.R, w.v, (const double *)side, (const int32_t *)bad, dx,
                               w.st, o->factor, o->gtol);
        } else if (sh <= 150 * 1024) {
            // more problems than CUs: 512-thread workgroups, two of which fit a CU (128 VGPRs each), so that the
            // latency-bound phases of one factorisation overlap the MFMA phase of the other
            const int ct = (ft == 1024 && (nact < 0 ? nprob : nact) > 256) ? 512 : ft;
            hipLaunchKernelGGL(k_chol_nopiv<NB>, dim3(nprob), dim3(ct), sh, h->stream, n, w.G, w.g, w.R, w.v, dx, w.st,
                               o->factor, o->gtol, o->ne_pivot_tol);
        } else {    // panel does not fit LDS: go straight to the pivoted (unblocked) factorisation
            size_t sh2 = sizeof(double) * (size_t)(3 * n + 64);
            hipLaunchKernelGGL(k_chol_factor, dim3(nprob), dim3(ft), sh2, h->stream, n, w.R, w.G, w.g, w.v, dx, w.st,
                               (int32_t *)nullptr, o->factor, o->gtol, o->ne_pivot_tol, 0, (int)ST_HAVE_JAC);
        }
    }
    {
        Timed t(h, NLH_K_LMPAR);
        const int lt = (ft == 1024 && n <= 512 && (nact < 0 ? nprob : nact) > 256) ? 512 : ft;   // two workgroups per CU, as for the Cholesky
        hipLaunchKernelGGL(k_lmpar<false>, dim3(nprob), dim3(lt), shl, h->stream, m, n, w.R, w.v, dx, w.wa4, w.P,
                           w.J, w.W2, w.st, (int)ST_NE_READY);
    }
    {   // problems whose lmpar iteration needs lmfactor's pivot order (or with a weak pivot)
        Timed t(h, NLH_K_CHOL);
        size_t sh = sizeof(double) * (size_t)(3 * n + 64);
        hipLaunchKernelGGL(k_chol_factor, dim3(nprob), dim3(ft), sh, h->stream, n, w.R, w.G, w.g, w.v, dx, w.st,
                           (int32_t *)nullptr, o->factor, o->gtol, o->ne_pivot_tol, 0, (int)ST_NEED_PCHOL);
    }
    {
        Timed t(h, NLH_K_LMPAR);
        hipLaunchKernelGGL(k_lmpar<false>, dim3(nprob), dim3(lt_), shl + ringb, h->stream, m, n, w.R, w.v, dx, w.wa4, w.P,
                           w.J, w.W2, w.st, (int)ST_NE_READY, (double *)nullptr, ringcap);
    }
    {
        Timed t(h, NLH_K_QR);
        size_t sh = sizeof(double) * (size_t)(3 * n + 64);
        hipLaunchKernelGGL(k_qr_factor, dim3(nprob), dim3(1024), sh, h->stream, m, n, w.J, dfvec, w.R, w.v,
                           w.wa4, w.scratch, dx, w.st, o->factor, o->gtol, 0);
    }
    {
        Timed t(h, NLH_K_LMPAR);
        hipLaunchKernelGGL(k_lmpar<false>, dim3(nprob), dim3(lt_), shl + ringb, h->stream, m, n, w.R, w.v, dx, w.wa4, w.P,
                           w.J, w.W2, w.st, (int)ST_QR_READY, (double *)nullptr, ringcap);
    }
    return 0;
}

static void lm_update(nlh_handle *h, const nlh_options *o, int nprob, int m, int n, LmWs &w, double *dx, double *dfvec)
{
    Timed t(h, NLH_K_UPDATE);
    if (o->factor_policy == NLH_FACTOR_EXACT)
        hipLaunchKernelGGL(k_lm_update<true>, dim3(nprob), dim3(256), 0, h->stream, m, n, w.nblk, w.part, w.v, dx,
                           dfvec, w.wa4, w.st, o->ftol, o->xtol, (int)o->max_evals);
    else
        hipLaunchKernelGGL(k_lm_update<false>, dim3(nprob), dim3(256), 0, h->stream, m, n, w.nblk, w.part, w.v, dx,
                           dfvec, w.wa4, w.st, o->ftol, o->xtol, (int)o->max_evals);
}

static void fill_ib(const LmState &s, nlh_iteration_behavior *ib)
{
    ib->iter_count = s.iter;
    ib->fcn_count = s.neval;
    ib->jacobian_count = s.njac;
    ib->gradient_count = 0;
    ib->converge_on_fcn = s.fcnvrg;
    ib->converge_on_chng = s.xcnvrg;
    ib->converge_on_zero_diff = s.gcnvrg;
}

static int check_opts_lm(const nlh_options *o, int m, int n)
{
    if (!o) return NLH_INVALID_INPUT_ERROR;
    if (n > m) return NLH_UNDERDEFINED_PROBLEM_ERROR;          // :189
    if (n < 1 || m < 1) return NLH_INVALID_INPUT_ERROR;
    // the normal-equations and tree-reduced QR policies keep n-vectors in LDS; the exact policy (the default) moves
    // lmpar's to global memory beyond that and takes any n
    if (n > LM_LDS_MAX_N && o->factor_policy != NLH_FACTOR_EXACT) return NLH_ARRAY_SIZE_ERROR;
    return 0;
}


// ===========================================================================
// Device-model LM, batched: lss_solve as a lock-step state machine.
// ===========================================================================
static int lm_solve_range(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n,
                          const ResidualSource &rs, double *dx, double *dfvec,
                          nlh_iteration_behavior *ib, int32_t *status)
{
    int rc;
    HIPCHK(h, hipSetDevice(h->device));
    LmWs w;
    const bool exact = o->factor_policy == NLH_FACTOR_EXACT;
    // the dense-quadratic family's fused epilogue writes the Jacobian straight into the exact factorisation's working
    // matrix; a user's residual leaves a panel (in the J buffer) that k_fd_jacobian_qrx turns into the same matrix
    const bool fuse = o->fuse_fd && !rs.user();
    const bool to_qrx = exact && (fuse || rs.user());
    if ((rc = lm_workspace(h, nprob, m, n, w, true, !to_qrx))) return rc;   // (to_qrx: nothing reads a column-major Jacobian)
    if ((rc = ensure_pinned(h, sizeof(LmState) * (size_t)nprob + 64))) return rc;
    int *d_active = (int *)(w.info + nprob);
    int *h_active = (int *)h->pinned;
    LmState *h_state = (LmState *)((char *)h->pinned + 64);
    const int pb = (nprob + 255) / 256;
    const int first_stage = ST_NEED_JAC;

    // :211-213  f(x0), fnorm
    if ((rc = residual_eval(h, rs, nprob, m, n, dx, dfvec, exact && rs.user() ? (double *)nullptr : w.part, nullptr, -1))) return rc;
    if (o->factor_policy == NLH_FACTOR_EXACT)
        hipLaunchKernelGGL(k_lm_init_exact, dim3(nprob), dim3(256), 0, h->stream, m, dfvec, w.st, first_stage);
    else
        hipLaunchKernelGGL(k_lm_init, dim3(pb), dim3(256), 0, h->stream, nprob, w.nblk, w.part, w.st, first_stage);

    const bool echo = o->print_status && nprob == 1;
    const int max_rounds = o->max_evals + 8;
    int last_printed_iter = -1;
    int nact = nprob, njac_due = nprob;                         // problems still iterating / due for a Jacobian (from the previous round)
    static const bool dbg_rounds = getenv("NLH_DEBUG_ROUNDS") != nullptr;   // per lock-step round: active problems, Jacobians due, wall ms
    auto tround = std::chrono::steady_clock::now();
    for (int round = 0; round < max_rounds; ++round) {
        // outer-loop head for problems that need a Jacobian (:221): n perturbed evaluations + FD
        // (the exact factorisation is the Jacobian's only reader: to_qrx writes it in that working layout, no re-layout
        // pass; the panel of the unfused forms lives in whichever of the two big buffers the Jacobian does not)
        // A round in which no problem is due for a Jacobian (rejected trial points only) skips the Jacobian and the factorisation's ~3 n launches; lmpar for the repeats still runs.
        const bool any_jac = njac_due > 0;
        if (any_jac) {
            // (njac_due is exact: the first round has every problem at ST_NEED_JAC, later ones have the previous read-back)
            if ((rc = residual_jacobian(h, rs, nprob, m, n, dx, dfvec, to_qrx ? w.P : w.J, to_qrx ? w.J : w.P, w.st, ST_NEED_JAC, to_qrx,
                                        fuse, true, njac_due))) return rc;
            hipLaunchKernelGGL(k_stage_advance, dim3(pb), dim3(256), 0, h->stream, nprob, w.st, (int)ST_NEED_JAC,
                               o->factor_policy != NLH_FACTOR_AUTO ? (int)ST_NEED_QR : (int)ST_HAVE_JAC, 1);
        }
        if ((rc = lm_factor_and_step(h, o, nprob, m, n, w, dx, dfvec, std::min(nact, std::max(njac_due, 1)), to_qrx, any_jac))) return rc;
        // trial residual (:297-299)
        if ((rc = residual_eval(h, rs, nprob, m, n, w.v.wa2, w.wa4, exact && rs.user() ? (double *)nullptr : w.part, w.st,
                                ST_TRIAL_READY))) return rc;
        hipLaunchKernelGGL(k_stage_advance, dim3(pb), dim3(256), 0, h->stream, nprob, w.st, (int)ST_TRIAL_READY,
                           (int)ST_TRIAL_DONE, 0);
        lm_update(h, o, nprob, m, n, w, dx, dfvec);
        hipLaunchKernelGGL(k_count_active, dim3(1), dim3(256), 0, h->stream, nprob, w.st, d_active);
        HIPCHK(h, hipMemcpyAsync(h_active, d_active, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (echo) HIPCHK(h, hipMemcpyAsync(h_state, w.st, sizeof(LmState), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        // a single solve with print_status set: the reference's status block at the end of every outer iteration that
        // goes on (:372-374), printed from the state that came back with the count
        if (echo && h_state[0].stage == ST_NEED_JAC && h_state[0].iter != last_printed_iter) {
            print_status(h_state[0].iter, h_state[0].neval, h_state[0].njac, h_state[0].xnorm, h_state[0].fnorm);
            last_printed_iter = h_state[0].iter;
        }
        if (dbg_rounds) {
            const auto tn = std::chrono::steady_clock::now();
            fprintf(stderr, "round %d base %d nprob %d active_before %d jac_due_before %d -> active %d ms %.3f\n", round, rs.pbase, nprob, nact, njac_due,
                    h_active[0], std::chrono::duration<double, std::milli>(tn - tround).count());
            tround = tn;
        }
        if (*h_active == 0) break;
        nact = h_active[0];
        njac_due = h_active[1];
    }
    HIPCHK(h, hipMemcpyAsync(h_state, w.st, sizeof(LmState) * (size_t)nprob, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (getenv("NLH_DEBUG_LAG"))
        for (int p = 0; p < nprob; ++p)
            fprintf(stderr, "lag %d njac %d slow_lmpar %d rejects %d first %d\n", p + rs.pbase, h_state[p].njac, h_state[p].slow_lmpar,
                    h_state[p].rejects, h_state[p].first_slow);
    for (int p = 0; p < nprob; ++p) {
        if (ib) fill_ib(h_state[p], &ib[p]);
        if (status) status[p] = (h_state[p].flag != 0 || h_state[p].stage != ST_DONE) ? NLH_CONVERGENCE_ERROR : 0;  // :388-390
    }
    return 0;
}

// Several sub-batches in flight.  A batch is a lock-step state machine whose rounds contain latency-bound stages (pivot /
// NORM2 chains of the exact lmfactor, Cholesky, lmpar's iteration for the few problems that need it, straggler rounds,
// the status read-back): with the batch dealt to S host threads, each driving its own stream and workspace, those stages
// of one sub-batch run under the streaming kernels of the others.  Problems are independent and a problem's arithmetic
// does not depend on its neighbours, so x, fvec and all counts are the same bits for any S.
static int lm_sub_batches(const nlh_options *o, int nprob, int m, int n)
{
    int S = o->sub_batches;
    if (S <= 0)                                     // the environment only fills in for "automatic", never overrides a caller
        if (const char *e = getenv("NLH_SUB_BATCHES")) S = atoi(e);
    if (S <= 0) {                                   // auto: >= 128 problems per sub-batch, at most 3 in flight (measured
        S = nprob / 128;                            // on 512 x 4096x256 exact: 1 / 2 / 3 / 4 -> 1032 / 959 / 928 / 1036 ms)
        if (S > 3) S = 3;
        // Round 6 re-measured the cap.  On the bench family (four to six lock-step rounds) three sub-batches against one batch
        // are a wash since k_lmpar halved: 3,822 / 3,772, 3,694 / 3,772, 3,700 / 3,770, 3,800 / 3,750, 3,640 / 3,755 LM it/s on five
        // boxes (a kernel trace has a pass resident 94 % of the wall time, profiles/r06_overlap_defaults.json -- but two or three
        // HBM-bound passes side by side deliver 12 % less than one alone).  On a family with a long straggler tail they are
        // not: Lorentzian peak fits through a user launcher (tests/device_model, 416 lock-step rounds), 2048 x 4096x96 /
        // 4096 x 2048x24, seconds per solve: one batch 20.6 / 1.86, THREE 15.0 / 1.51, four 23.8 / 2.22, six 23.0 / 2.59,
        // eight 27.7 / 3.39, twelve 33.1 / 3.26 (profiles/r06_sub_batch_levers.txt).  Three stays.
        // 32 to 255 problems: two halves.  A half of such a batch takes the wide pass form (at most 256 (problem, window)
        // pairs), whose passes are bound by one adder wave per window rather than by HBM, so two of them side by side
        // cost little more than one, and the half that holds a straggler runs its rounds at the smaller batch's pace
        // (ms per solve, one batch / two halves: 4096x256: 16 problems 63.3 / 63.1, 32: 71.4 / 71.1, 44: 87.7 / 76.9,
        // 48: 132.7 / 120.7, 64: 151.8 / 137.1, 96: 200 / 168, 128: 223 / 207, 192: 316 / 279; 2048x128: 48: 22.7 / 22.3,
        // 96: 30.5 / 28.5, 128: 34.0 / 32.2, 192: 54.9 / 48.3; 4096x512: 32: 210 / 179; 8192x256: 64: 279 / 236; three
        // pieces: 16 x 4096x256 98 -- pieces of five problems fall to the column sweep)
        // -- measured from 2048x128 up; tiny problems (m n < 65536) have no latency-bound pass to hide and would only pay for
        // a second host thread, workspace and stream (and have a user's launcher called from two threads): one batch
        if (S < 2 && nprob >= 32 && (size_t)m * n >= 65536) S = 2;
        if (o->factor_policy != NLH_FACTOR_EXACT) S = 1;   // the normal-equations pipeline has no long latency-bound
    }                                                      // stage to hide (1 / 2 / 4 -> 40.0 / 40.6 / 41.7 ms)
    if (S > nprob) S = nprob;
    return S < 1 ? 1 : S;
}

static int lm_solve_batch_rs(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n, const ResidualSource &rs,
                             double *dx, double *dfvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (nprob <= 0) return 0;
    int rc = check_opts_lm(o, m, n);
    if (rc) return rc;
    // a user's launcher is asked for nprob * n points at once and its panel is addressed with 31-bit point counts
    const int32_t slice = rs.user() ? (int32_t)std::max<int64_t>(1, std::min<int64_t>(NLH_MAX_LOCKSTEP, ((int64_t)1 << 30) / n)) : NLH_MAX_LOCKSTEP;
    if (nprob > slice) {
        for (int32_t p0 = 0; p0 < nprob; p0 += slice) {
            const int32_t cnt = std::min<int32_t>(slice, nprob - p0);
            if ((rc = lm_solve_batch_rs(h, o, cnt, m, n, rs.shifted(p0, m, n), dx + (size_t)p0 * n, dfvec + (size_t)p0 * m, ib ? ib + p0 : nullptr,
                                        status ? status + p0 : nullptr))) return rc;
        }
        return 0;
    }
    const int S = lm_sub_batches(o, nprob, m, n);
    if (S == 1) return lm_solve_range(h, o, nprob, m, n, rs, dx, dfvec, ib, status);
    if ((rc = ensure_workers(h, S))) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));                 // inputs written on the caller's stream are complete
    std::vector<int> rcs(S, 0);
    std::vector<std::thread> pool;
    for (int t = 0; t < S; ++t) {
        // (equal shares: shares skewed by +-15 / 30 / 50 % so that the sub-batches' phases drift apart were measured --
        // 3,626 / 3,579 / 3,486 against 3,640 LM it/s for three equal sub-batches of the 2048 batch -- and dropped)
        const int p0 = (int)((long)nprob * t / S), p1 = (int)((long)nprob * (t + 1) / S);
        pool.emplace_back([&, t, p0, p1]() {
            nlh_handle *wk = h->workers[t];
            wk->timing = h->timing;
            rcs[t] = lm_solve_range(wk, o, p1 - p0, m, n, rs.shifted(p0, m, n), dx + (size_t)p0 * n, dfvec + (size_t)p0 * m,
                                    ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
        });
    }
    for (auto &th : pool) th.join();
    for (int t = 0; t < S; ++t) {
        nlh_handle *wk = h->workers[t];
        if (h->timing) {                                        // fold the workers' kernel timers into the caller's
            timing_flush(wk);
            for (int k = 0; k < NLH_K_COUNT; ++k) { h->ms[k] += wk->ms[k]; h->launches[k] += wk->launches[k]; wk->ms[k] = 0; wk->launches[k] = 0; }
        }
        if (rcs[t]) { h->err = wk->err; return rcs[t]; }
    }
    return 0;
}

int nlh_dq_lm_solve_batch(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n,
                          const double *dA, const double *db, double gamma, double *dx, double *dfvec,
                          nlh_iteration_behavior *ib, int32_t *status)
{
    ResidualSource rs;
    rs.dA = dA; rs.db = db; rs.gamma = gamma;
    return lm_solve_batch_rs(h, o, nprob, m, n, rs, dx, dfvec, ib, status);
}

// least_squares_solver%solve on a batch of problems whose residual is the USER'S device function (launchers,
// include/nonlin_hip.h): the same lock-step state machine, the same kernels downstream of the Jacobian.
int nlh_lm_solve_batch_device(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n, nlh_device_vecfcn fcn,
                              nlh_device_jacfcn jacfcn, void *ctx, double *dx, double *dfvec, nlh_iteration_behavior *ib,
                              int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib && nprob > 0) memset(ib, 0, sizeof(*ib) * (size_t)nprob);          // :177-185
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :188
    if (!o || (nprob > 0 && (!dx || !dfvec))) return NLH_INVALID_INPUT_ERROR;
    ResidualSource rs;
    rs.fcn = fcn; rs.jac = jacfcn; rs.ctx = ctx;
    nlh_options oq = *o;
    if (nprob > 1) oq.print_status = 0;                         // the status block is a single solve's (:372-374)
    return lm_solve_batch_rs(h, &oq, nprob, m, n, rs, dx, dfvec, ib, status);
}

// The same behind host arrays (the Fortran shim's set_device_fcn + solve / solve_batch).
int nlh_lm_solve_batch_device_h(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n, nlh_device_vecfcn fcn,
                                nlh_device_jacfcn jacfcn, void *ctx, double *x, double *fvec, nlh_iteration_behavior *ib,
                                int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (nprob <= 0) return 0;
    if (!x || !fvec || !o) return NLH_INVALID_INPUT_ERROR;
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;
    int rc = check_opts_lm(o, m, n);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    if ((rc = ensure(h, h->xdev, sizeof(double) * (size_t)nprob * n))) return rc;
    if ((rc = ensure(h, h->fdev, sizeof(double) * (size_t)nprob * m))) return rc;
    double *dx = (double *)h->xdev.p, *df = (double *)h->fdev.p;
    HIPCHK(h, hipMemcpyAsync(dx, x, sizeof(double) * (size_t)nprob * n, hipMemcpyHostToDevice, h->stream));
    rc = nlh_lm_solve_batch_device(h, o, nprob, m, n, fcn, jacfcn, ctx, dx, df, ib, status);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(x, dx, sizeof(double) * (size_t)nprob * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(fvec, df, sizeof(double) * (size_t)nprob * m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ===========================================================================
// Host-callback LM: the same kernels with nprob = 1; residuals come from fcn.
// ===========================================================================
int nlh_lm_solve(nlh_handle *h, const nlh_options *o, int32_t m, int32_t n, nlh_vecfcn fcn,
                 nlh_jacfcn jacfcn, void *ctx, double *x, double *fvec, nlh_iteration_behavior *ib)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib) memset(ib, 0, sizeof *ib);                          // :177-185
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :188
    int rc = check_opts_lm(o, m, n);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    LmWs w;
    if ((rc = lm_workspace(h, 1, m, n, w, true))) return rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
    if ((rc = ensure(h, h->fdev, sizeof(double) * m))) return rc;
    const size_t pin_bytes = 256 + sizeof(double) * ((size_t)m * n + 2 * (size_t)m + 2 * (size_t)n);
    if ((rc = ensure_pinned(h, pin_bytes))) return rc;
    LmState *hs = (LmState *)h->pinned;
    double *hP = (double *)((char *)h->pinned + 256);   // m*n panel / Jacobian staging
    double *hf = hP + (size_t)m * n;                     // m
    double *hx = hf + m;                                 // n
    double *dx = (double *)h->xdev.p, *dfvec = (double *)h->fdev.p;
    hipStream_t s = h->stream;

    fcn(ctx, n, x, m, fvec);                                    // :211
    HIPCHK(h, hipMemcpyAsync(dx, x, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(dfvec, fvec, sizeof(double) * m, hipMemcpyHostToDevice, s));
    launch_sumsq_part(h, 1, m, n, dfvec, w.part);
    if (o->factor_policy == NLH_FACTOR_EXACT)
        hipLaunchKernelGGL(k_lm_init_exact, dim3(1), dim3(256), 0, s, m, dfvec, w.st, (int)ST_NEED_JAC);
    else
        hipLaunchKernelGGL(k_lm_init, dim3(1), dim3(64), 0, s, 1, w.nblk, w.part, w.st, (int)ST_NEED_JAC);
    HIPCHK(h, hipStreamSynchronize(s));

    const int max_rounds = o->max_evals + 8;
    int last_printed_iter = -1;
    for (int round = 0; round < max_rounds; ++round) {
        HIPCHK(h, hipMemcpyAsync(hs, w.st, sizeof(LmState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (hs->stage == ST_DONE) break;
        if (hs->stage == ST_NEED_JAC) {
            if (round > 0 && o->print_status && hs->iter != last_printed_iter) {   // :372-374
                print_status(hs->iter, hs->neval, hs->njac, hs->xnorm, hs->fnorm);
                last_printed_iter = hs->iter;
            }
            // vfh_jac_fcn (:221).  x and fvec on the host are kept equal to the device copies.
            if (jacfcn) {
                jacfcn(ctx, n, x, m, hP);
                HIPCHK(h, hipMemcpyAsync(w.J, hP, sizeof(double) * (size_t)m * n, hipMemcpyHostToDevice, s));
            } else {
                for (int j = 0; j < n; ++j) {                   // src/nonlin_multi_eqn_mult_var.f90:267-273
                    const double temp = x[j];
                    double hh = NLH_SQRT_EPS * fabs(temp);
                    if (hh == 0.0) hh = NLH_SQRT_EPS;
                    x[j] = temp + hh;
                    fcn(ctx, n, x, m, hP + (size_t)j * m);
                    x[j] = temp;
                }
                HIPCHK(h, hipMemcpyAsync(w.P, hP, sizeof(double) * (size_t)m * n, hipMemcpyHostToDevice, s));
                launch_fd(h, 1, m, n, w.P, dfvec, dx, w.J, nullptr, -1);      // :274
            }
            hipLaunchKernelGGL(k_stage_advance, dim3(1), dim3(64), 0, s, 1, w.st, (int)ST_NEED_JAC,
                               o->factor_policy != NLH_FACTOR_AUTO ? (int)ST_NEED_QR : (int)ST_HAVE_JAC, 1);
        }
        if ((rc = lm_factor_and_step(h, o, 1, m, n, w, dx, dfvec))) return rc;
        HIPCHK(h, hipMemcpyAsync(hs, w.st, sizeof(LmState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(hx, w.v.wa2, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (hs->stage == ST_DONE) break;                       // gradient convergence (:270-273)
        if (hs->stage != ST_TRIAL_READY) { h->err = "lm: unexpected stage"; return NLH_ERR_HIP; }
        fcn(ctx, n, hx, m, hf);                                 // :297
        HIPCHK(h, hipMemcpyAsync(w.wa4, hf, sizeof(double) * m, hipMemcpyHostToDevice, s));
        launch_sumsq_part(h, 1, m, n, w.wa4, w.part);
        hipLaunchKernelGGL(k_stage_advance, dim3(1), dim3(64), 0, s, 1, w.st, (int)ST_TRIAL_READY, (int)ST_TRIAL_DONE, 0);
        const int iter_before = hs->iter;
        lm_update(h, o, 1, m, n, w, dx, dfvec);
        HIPCHK(h, hipMemcpyAsync(hs, w.st, sizeof(LmState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (hs->iter != iter_before) {                          // accepted: mirror x, fvec on the host (:341-345)
            memcpy(x, hx, sizeof(double) * n);
            memcpy(fvec, hf, sizeof(double) * m);
        }
    }
    HIPCHK(h, hipMemcpyAsync(hs, w.st, sizeof(LmState), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    if (ib) fill_ib(*hs, ib);
    return (hs->flag != 0 || hs->stage != ST_DONE) ? NLH_CONVERGENCE_ERROR : 0;
}

// ===========================================================================
// vecfcn_helper%jacobian for host callbacks.
// ===========================================================================
int nlh_fd_jacobian(nlh_handle *h, int32_t m, int32_t n, nlh_vecfcn fcn, nlh_jacfcn jacfcn, void *ctx,
                    double *x, const double *fv, double *jac)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :240
    if (jacfcn) { jacfcn(ctx, n, x, m, jac); return 0; }       // :241-243
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    const size_t mn = (size_t)m * n;
    if ((rc = ensure(h, h->J, sizeof(double) * mn))) return rc;
    if ((rc = ensure(h, h->P, sizeof(double) * mn))) return rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
    if ((rc = ensure(h, h->fdev, sizeof(double) * m))) return rc;
    if ((rc = ensure_pinned(h, sizeof(double) * (mn + m)))) return rc;
    double *hP = (double *)h->pinned, *hf0 = hP + mn;
    if (fv) memcpy(hf0, fv, sizeof(double) * m);
    else fcn(ctx, n, x, m, hf0);                                // :257-259
    for (int j = 0; j < n; ++j) {                               // :267-273
        const double temp = x[j];
        double hh = NLH_SQRT_EPS * fabs(temp);
        if (hh == 0.0) hh = NLH_SQRT_EPS;
        x[j] = temp + hh;
        fcn(ctx, n, x, m, hP + (size_t)j * m);
        x[j] = temp;
    }
    hipStream_t s = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->P.p, hP, sizeof(double) * mn, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->fdev.p, hf0, sizeof(double) * m, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->xdev.p, x, sizeof(double) * n, hipMemcpyHostToDevice, s));
    launch_fd(h, 1, m, n, (const double *)h->P.p, (const double *)h->fdev.p, (const double *)h->xdev.p,
              (double *)h->J.p, nullptr, -1);
    HIPCHK(h, hipMemcpyAsync(jac, h->J.p, sizeof(double) * mn, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_gram(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dJ, const double *df, double *dG,
             double *dg)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = launch_gram(h, nprob, m, n, dJ, df, dG, df ? dg : nullptr, nullptr, -1);
    if (rc) return rc;
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_chol_factor(nlh_handle *h, int32_t nprob, int32_t n, double *dG, const double *dg, int32_t *dipvt,
                    double *dacnorm, double *dqtf, int32_t *dinfo)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    LmVecs v;
    memset(&v, 0, sizeof v);
    v.ipvt = dipvt; v.acnorm = dacnorm; v.qtf = dqtf;
    nlh_options o;
    nlh_default_options(&o);
    {
        Timed t(h, NLH_K_CHOL);
        size_t sh = sizeof(double) * (size_t)(3 * n + 64);
        hipLaunchKernelGGL(k_chol_factor, dim3(nprob), dim3(factor_threads(n)), sh, h->stream, n, dG,
                           (const double *)nullptr, dg, v, (const double *)nullptr, (LmState *)nullptr, dinfo,
                           o.factor, o.gtol, 0.0, 1, -1);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_qr_factor(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, double *dJ, const double *df,
                  int32_t *dipvt, double *drdiag, double *dacnorm, double *dqtf, double *dwa4)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->G, sizeof(double) * (size_t)nprob * n * n))) return rc;
    LmVecs v;
    memset(&v, 0, sizeof v);
    v.ipvt = dipvt; v.acnorm = dacnorm; v.qtf = dqtf; v.rdiag = drdiag;
    {
        Timed t(h, NLH_K_QR);
        size_t sh = sizeof(double) * (size_t)(3 * n + 64);
        hipLaunchKernelGGL(k_qr_factor, dim3(nprob), dim3(1024), sh, h->stream, m, n, dJ, df, (double *)h->G.p, v,
                           dwa4, dwa4, (const double *)nullptr, (LmState *)nullptr, 100.0, 0.0, 1);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

// lmfactor + Q^T f in the reference's operation order (the factorisation the exact LM policy runs): nlh_qrx.hip on
// caller-supplied matrices, every problem factored.  dJ: [nprob][n][m] column-major, not modified.
int nlh_lmfactor_exact(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dJ, const double *df,
                       double *dR, int32_t *dipvt, double *drdiag, double *dacnorm, double *dqtf, double *dwa4)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (nprob <= 0) return 0;
    if (m < n || n < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->P, sizeof(double) * qrx_matrix_doubles(nprob, m, n)))) return rc;
    if ((rc = ensure(h, h->qxV, qrx_workspace_bytes(nprob, m, n)))) return rc;
    LmVecs v;
    memset(&v, 0, sizeof v);
    v.ipvt = dipvt; v.acnorm = dacnorm; v.qtf = dqtf; v.rdiag = drdiag;
    qrx_factor(h->stream, nprob, m, n, dJ, (double *)h->P.p, df, dR, v, dwa4, dwa4, (const double *)nullptr, (LmState *)nullptr,
               100.0, 0.0, h->qxV.p, (const QrxTimer *)nullptr, nprob);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// lmpar on caller-supplied factors (parity tests): wraps lmpar_dev.

static __global__ void __launch_bounds__(1024)
k_lmpar_standalone(int n, double *Rall, int ldr, const int32_t *ipvt_all, const double *diag_all,
                   const double *qtf_all, const double *delta_all, const double *tailsq_all, double *par_all,
                   double *x_all, double *sdiag_all, double *Wall, int ringcap)
{
    extern __shared__ double smem[];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    double *xs = smem, *sdiag = smem + n, *wa1 = smem + 2 * n, *wa2n = smem + 3 * n, *z = smem + 4 * n;
    double *red = smem + 5 * n;
    double *rot = red + 64;
    double par = par_all[p];
    lmpar_dev<false>(n, n, Rall + (size_t)p * ldr * n, ldr, ipvt_all + (size_t)p * n, diag_all + (size_t)p * n,
                     qtf_all + (size_t)p * n, delta_all[p], &par, tailsq_all[p], nullptr, xs, sdiag, wa1, wa2n, z,
                     red, nullptr, Wall + (size_t)p * n * n, rot, 0, ringcap > 0 ? smem + 6 * n + 72 : nullptr, ringcap);
    __syncthreads();
    for (int j = tid; j < n; j += BS) {
        x_all[(size_t)p * n + j] = xs[j];
        sdiag_all[(size_t)p * n + j] = sdiag[j];
    }
    if (tid == 0) par_all[p] = par;
}


int nlh_lmpar(nlh_handle *h, int32_t nprob, int32_t n, double *dR, int32_t ldr, const int32_t *dipvt,
              const double *ddiag, const double *dqtf, const double *ddelta, const double *dtailsq,
              double *dpar, double *dxstep, double *dsdiag)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure(h, h->misc, sizeof(double) * (size_t)nprob * n * n);
    if (rc) return rc;
    {
        Timed t(h, NLH_K_LMPAR);
        int cap;
        const size_t rb = lm_ring_bytes(n, lmpar_threads(n), &cap);
        size_t sh = sizeof(double) * (size_t)(6 * n + 72) + rb;
        hipLaunchKernelGGL(k_lmpar_standalone, dim3(nprob), dim3(lmpar_threads(n)), sh, h->stream, n, dR, ldr, dipvt,
                           ddiag, dqtf, ddelta, dtailsq, dpar, dxstep, dsdiag, (double *)h->misc.p, cap);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}
