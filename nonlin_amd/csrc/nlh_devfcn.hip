// nlh_devfcn.hip -- the open device-residual path: a user's vecfcn / jacobianfcn (src/nonlin_multi_eqn_mult_var.f90:14-38)
// handed in as LAUNCHERS (include/nonlin_hip.h: nlh_device_vecfcn, nlh_device_jacfcn) and everything the lock-step
// drivers need around them:
//   * residual_eval      F at one point per problem (the starting point :211, a trial point :297, the line search's trials);
//   * residual_jacobian  vecfcn_helper%jacobian (vfh_jac_fcn, :198-277): the n perturbed points of every problem that is
//                        due, built on the device in the reference's order (:267-273), ONE call of the user's launcher for
//                        all of them, and the forward-difference column write (:274) -- k_fd_jacobian (column-major J) or
//                        k_fd_jacobian_qrx (straight into the exact factorisation's row-blocked working matrix);
//   * the dense-quadratic family as such launchers (nlh_dq_device_fcn / nlh_dq_device_jac) -- the same bits as the
//     fused kernels of nlh_kernels_model.h, through the open path;
//   * nlh_fd_jacobian_device.
// A lock-step round serves the problems that are in a given stage; the user's launcher is only ever asked for those: the
// problems are compacted on the device (ascending), the count comes back to the host (one read-back), and the launcher
// sees npoints contiguous points with the problem index of each in dprob.
#include "nlh_internal.h"
#include "nlh_kernels_model.h"
#include "nlh_qrx.h"

// ---------------------------------------------------------------------------
// compaction of the problems at one stage: list[k] = k-th such problem (ascending), *cnt = how many
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(1024)
k_dv_compact(int nprob, const LmState *__restrict__ st, int want, int32_t *__restrict__ list, int32_t *__restrict__ cnt)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int p0 = 0; p0 < nprob; p0 += 1024) {
        const int p = p0 + tid;
        const bool on = p < nprob && (!st || st[p].stage == want);
        const unsigned long long bal = __ballot(on);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wid] = __popcll(bal);
        __syncthreads();
        int off = base_s, tot = 0;
        for (int w = 0; w < 16; ++w) { const int c = wsum[w]; if (w < wid) off += c; tot += c; }
        if (on) list[off + before] = p;
        __syncthreads();
        if (tid == 0) base_s += tot;
        __syncthreads();
    }
    if (tid == 0) *cnt = base_s;
}

// dprob for one point per listed problem (list == nullptr: the identity), and the gather of their x
static __global__ void k_dv_gather_x(int cnt, int n, const int32_t *__restrict__ list, int pbase, const double *__restrict__ x,
                                     double *__restrict__ X, int32_t *__restrict__ dprob)
{
    const int k = blockIdx.x;
    const int p = list ? list[k] : k;
    if (threadIdx.x == 0) dprob[k] = pbase + p;
    if (X)
        for (int c = threadIdx.x; c < n; c += blockDim.x) X[(size_t)k * n + c] = x[(size_t)p * n + c];
}

// compact residuals back to the problems' rows
static __global__ void k_dv_scatter_f(int m, const int32_t *__restrict__ list, const double *__restrict__ F, double *__restrict__ f)
{
    const int k = blockIdx.y, p = list[k];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) f[(size_t)p * m + i] = F[(size_t)k * m + i];
}

// The n perturbed points of vfh_jac_fcn for the listed problems (:267-273): point k*n + j = x with x(j) = x(j) + h_j.
static __global__ void k_dv_fd_points(int n, const int32_t *__restrict__ list, int pbase, const double *__restrict__ x,
                                      double *__restrict__ X, int32_t *__restrict__ dprob)
{
    const int k = blockIdx.y, j = blockIdx.x;
    const int p = list ? list[k] : k;
    const double *xp = x + (size_t)p * n;
    double *Xq = X + ((size_t)k * n + j) * n;
    if (threadIdx.x == 0) dprob[(size_t)k * n + j] = pbase + p;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const double v = xp[c];
        Xq[c] = (c == j) ? v + fd_step(v) : v;                  // x(j) = temp + h, :271
    }
}

// ---------------------------------------------------------------------------
// The forward-difference column write (:274) into the exact factorisation's working matrix (nlh_qrx.hip: element (i, c)
// at ((i / 8) * ld + c) * 8 + i % 8, Jacobian column j at c = coff + j).  The panel is column-major (a lane per row pair:
// a wave reads 1 KB contiguous of one column); the working matrix keeps eight rows of a column per 64-byte sector and all
// columns of a row block contiguous, so a tile of 128 rows x 32 columns is turned in LDS and leaves as sixteen runs of
// 2 KB.  Streaming: 8 bytes read + 8 bytes written per element, f0 and h_j from cache.
// Panel slot = blockIdx.z (compact), problem = list[blockIdx.z].
// ---------------------------------------------------------------------------
#define FDQ_ROWS 128
#define FDQ_COLS 32
#define FDQ_LDB (FDQ_COLS * 8 + 8)        // doubles between two row blocks of the tile in LDS (padded: 64 bytes)
template <bool VEC2, bool NT>
__global__ void __launch_bounds__(256)
k_fd_jacobian_qrx(int m, int n, const double *__restrict__ P, const double *__restrict__ f0, const double *__restrict__ x,
                  double *__restrict__ T, const int32_t *__restrict__ list, const LmState *__restrict__ st, int want,
                  int ld, int coff, size_t tst, int nfull)
{   // n: columns of this launch's panel per problem (a column group of ONE problem when n < nfull: x, T arrive shifted)
    __shared__ __attribute__((aligned(16))) double tile[(FDQ_ROWS / 8) * FDQ_LDB];
    const int k = blockIdx.z;
    const int p = list ? list[k] : k;
    if (st && st[p].stage != want) return;
    const double *Pp = P + (size_t)k * m * n;
    const double *fp = f0 + (size_t)p * m;
    const double *xp = x + (size_t)p * nfull;
    double *Tp = T + (size_t)p * tst;
    const int i0 = blockIdx.x * FDQ_ROWS, j0 = blockIdx.y * FDQ_COLS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    typedef double v2d __attribute__((ext_vector_type(2)));
    {
        const int i = i0 + 2 * lane;
        double fa = 0.0, fb = 0.0;
        if (i < m) fa = fp[i];
        if (i + 1 < m) fb = fp[i + 1];
        double va[8], vb[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int j = j0 + w * 8 + c;
            va[c] = 0.0; vb[c] = 0.0;
            if (j < n) {
                if (VEC2) {                                     // m even: i + 1 < m whenever i < m, 16-byte aligned
                    if (i < m) {
                        const v2d *src = reinterpret_cast<const v2d *>(Pp + (size_t)j * m + i);
                        const v2d v = NT ? __builtin_nontemporal_load(src) : *src;
                        va[c] = v.x; vb[c] = v.y;
                    }
                } else {
                    if (i < m) va[c] = Pp[(size_t)j * m + i];
                    if (i + 1 < m) vb[c] = Pp[(size_t)j * m + i + 1];
                }
            }
        }
        double *tl = tile + (lane >> 2) * FDQ_LDB + (w * 8) * 8 + ((2 * lane) & 7);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int j = j0 + w * 8 + c;
            const double h = fd_step(j < n ? xp[j] : 1.0);
            v2d o;
            o.x = (i < m) ? (va[c] - fa) / h : 0.0;             // rows beyond m: the zero padding of the last row block
            o.y = (i + 1 < m) ? (vb[c] - fb) / h : 0.0;
            *reinterpret_cast<v2d *>(tl + c * 8) = o;
        }
    }
    __syncthreads();
    const int mblk = (m + 7) >> 3;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int q = it * 256 + tid;                           // pair index inside the tile: 16 row blocks x 128 pairs
        const int rb = q >> 7, wi = q & 127;                    // pair wi of row block rb: column wi / 4, rows 2 (wi % 4), + 1
        const int c = wi >> 2;
        if (i0 / 8 + rb < mblk && j0 + c < n) {
            const v2d o = *reinterpret_cast<const v2d *>(tile + rb * FDQ_LDB + wi * 2);
            __builtin_nontemporal_store(o, reinterpret_cast<v2d *>(Tp + ((size_t)(i0 / 8 + rb) * ld + coff + j0) * 8 + wi * 2));
        }
    }
}

// A user's analytic Jacobian (column-major, compact slots) into the working matrix / the problems' column-major slots.
template <bool TOQRX>
static __global__ void __launch_bounds__(256)
k_dv_place_jac(int m, int n, const double *__restrict__ Jc, double *__restrict__ out, const int32_t *__restrict__ list, int ld, int coff,
               size_t tst)
{
    const int k = blockIdx.z, p = list ? list[k] : k;
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int mp = (m + 7) & ~7;
    if (i >= mp) return;
    const double v = i < m ? Jc[((size_t)k * n + j) * m + i] : 0.0;
    if (TOQRX) out[(size_t)p * tst + ((size_t)(i >> 3) * ld + coff + j) * 8 + (i & 7)] = v;
    else if (i < m) out[((size_t)p * n + j) * m + i] = v;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline int32_t *dv_list(nlh_handle *h) { return (int32_t *)h->dvIdx.p + 16; }

// Compacts the problems at `want` into the handle's list and reads their number back (*cnt); returns 0 or a library error.
// The count lands in the pinned block's header, behind the words the solvers keep their own round read-backs in.
static const int DV_PINNED_SLOT = 12;                            // int32 index inside the 64-byte header of h->pinned
// known >= 0: the caller already holds the count (the previous round's read-back says how many problems are due for a
// Jacobian): the list is built on the device and the host does not wait for it -- one stream synchronisation less per round.
static int dv_select(nlh_handle *h, int nprob, const LmState *st, int want, size_t extra_ints, int *cnt, int known = -1)
{
    int rc;
    *cnt = 0;
    if ((rc = ensure(h, h->dvIdx, sizeof(int32_t) * ((size_t)nprob + 16 + extra_ints)))) return rc;
    if ((rc = ensure_pinned(h, 64))) return rc;
    int32_t *dcnt = (int32_t *)h->dvIdx.p;
    hipLaunchKernelGGL(k_dv_compact, dim3(1), dim3(1024), 0, h->stream, nprob, st, want, dv_list(h), dcnt);
    if (known >= 0) { *cnt = known; return 0; }
    int32_t *hcnt = (int32_t *)h->pinned + DV_PINNED_SLOT;
    HIPCHK(h, hipMemcpyAsync(hcnt, dcnt, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *cnt = *hcnt;
    return 0;
}

static int dv_fail(nlh_handle *h, int rc, const char *what)
{
    h->err = std::string(what) + ": the user's launcher returned " + std::to_string(rc);
    return NLH_ERR_HIP;
}

int residual_eval(nlh_handle *h, const ResidualSource &rs, int nprob, int m, int n, const double *x, double *f, double *part,
                  const LmState *st, int want)
{
    if (!rs.user()) {
        launch_dq_residual(h, nprob, m, n, rs.dA, rs.db, rs.gamma, x, f, part, st, want);
        return 0;
    }
    int rc;
    int cnt = nprob;
    const bool all = st == nullptr;
    if (!all) {
        if ((rc = dv_select(h, nprob, st, want, (size_t)nprob, &cnt))) return rc;
        if (cnt == 0) return 0;
    } else if ((rc = ensure(h, h->dvIdx, sizeof(int32_t) * ((size_t)2 * nprob + 16)))) return rc;
    int32_t *list = dv_list(h), *dprob = list + nprob;
    const bool direct = all || cnt == nprob;                    // every problem: no gather / scatter, the caller's arrays
    const double *X = x;
    double *F = f;
    if (!direct) {
        if ((rc = ensure(h, h->dvX, sizeof(double) * (size_t)cnt * n))) return rc;
        if ((rc = ensure(h, h->dvF, sizeof(double) * (size_t)cnt * m))) return rc;
        X = (const double *)h->dvX.p; F = (double *)h->dvF.p;
    }
    hipLaunchKernelGGL(k_dv_gather_x, dim3(cnt), dim3(64), 0, h->stream, cnt, n, direct ? (const int32_t *)nullptr : (const int32_t *)list,
                       (int)rs.pbase, x, direct ? (double *)nullptr : (double *)h->dvX.p, dprob);
    {
        Timed t(h, NLH_K_DQ_RESIDUAL);
        const int urc = rs.fcn(rs.ctx, (void *)h->stream, cnt, dprob, n, X, m, F);
        if (urc) return dv_fail(h, urc, "vecfcn");
    }
    if (!direct)
        hipLaunchKernelGGL(k_dv_scatter_f, dim3((m + 255) / 256, cnt), dim3(256), 0, h->stream, m, (const int32_t *)list, (const double *)F, f);
    if (part) launch_sumsq_part(h, nprob, m, n, f, part);       // (every problem's row: only the ones at `want` are read)
    return 0;
}

// Bytes of panel the user's launcher fills before k_fd_jacobian turns them into Jacobian columns: a cap on the panel
// buffer (16 GiB), not a tuning knob -- measured at 512 x 4096x256 inside a solve (profiles/scripts/devfcn_fd.sh), chunks small
// enough for the Infinity Cache to serve the panel read do NOT pay: 16 / 32 / 64 / 128 / 256 MiB chunks ran the
// forward-difference kernel at 0.42 / 0.56 / 0.67 / 0.74 / 0.77 of 8 TB/s, one 8.6 GB launch at 0.756 (6.0 TB/s of
// algorithmic traffic; without the non-temporal hint on the panel loads 0.726).  NLH_FD_CHUNK_MB overrides (0: no cap).
static size_t fd_chunk_bytes()
{
    static const long mb = [] { const char *e = getenv("NLH_FD_CHUNK_MB"); return e ? atol(e) : 16384L; }();
    return mb <= 0 ? ~(size_t)0 : (size_t)mb << 20;
}

int residual_jacobian(nlh_handle *h, const ResidualSource &rs, int nprob, int m, int n, const double *x, const double *f0,
                      double *out, double *panel, const LmState *st, int want, bool to_qrx, bool fuse, bool use_jac, int known_cnt)
{
    if (!rs.user()) {
        if (fuse) launch_dq_panel(h, nprob, m, n, rs.dA, rs.db, rs.gamma, x, out, st, want, f0, to_qrx);
        else {
            launch_dq_panel(h, nprob, m, n, rs.dA, rs.db, rs.gamma, x, panel, st, want);
            launch_fd(h, nprob, m, n, panel, f0, x, out, st, want);      // (to_qrx is the fused form's)
        }
        return 0;
    }
    int rc;
    int cnt = nprob;
    const bool all = st == nullptr;
    const size_t npts_max = (size_t)nprob * n;
    if (!all) {
        if ((rc = dv_select(h, nprob, st, want, (size_t)nprob + npts_max, &cnt, known_cnt))) return rc;
        if (cnt == 0) return 0;
    } else if ((rc = ensure(h, h->dvIdx, sizeof(int32_t) * ((size_t)2 * nprob + 16 + npts_max)))) return rc;
    int32_t *list = dv_list(h), *dprob = list + 2 * (size_t)nprob;
    const int32_t *lp = (all || cnt == nprob) ? nullptr : list;
    const int ld = qrx_ld(n), coff = ld - (n + 1);
    const size_t tst = qrx_matrix_stride(m, n);
    const bool analytic = use_jac && rs.jac;                    // :241-243: the user's jacobianfcn, one point per problem
    // chunks: whole problems while one problem's panel fits the chunk, otherwise one problem in groups of 32 columns
    const size_t per = sizeof(double) * (size_t)m * n, cb = fd_chunk_bytes();
    int pc = (int)std::min<size_t>((size_t)cnt, std::max<size_t>(1, cb / per));       // problems per chunk
    int jc = n;                                                                         // columns per chunk
    if (per > cb && !analytic) { pc = 1; jc = (int)std::min<size_t>((size_t)n, std::max<size_t>(32, (cb / (sizeof(double) * m)) / 32 * 32)); }
    if ((rc = ensure(h, h->dvP, sizeof(double) * (size_t)pc * jc * m))) return rc;
    double *Pc = (double *)h->dvP.p;
    if ((rc = ensure(h, h->dvX, sizeof(double) * (size_t)cnt * n * (analytic ? 1 : n)))) return rc;
    double *X = (double *)h->dvX.p;
    if (analytic) hipLaunchKernelGGL(k_dv_gather_x, dim3(cnt), dim3(64), 0, h->stream, cnt, n, lp, (int)rs.pbase, x, X, dprob);
    else hipLaunchKernelGGL(k_dv_fd_points, dim3(n, cnt), dim3(64), 0, h->stream, n, lp, (int)rs.pbase, x, X, dprob);
    for (int k0 = 0; k0 < cnt; k0 += pc) {
        const int kc = std::min(pc, cnt - k0);
        for (int j0 = 0; j0 < n; j0 += jc) {
            const int jn = std::min(jc, n - j0);
            // slots k0 .. k0 + kc of the compact order, columns j0 .. j0 + jn of each (jn == n unless kc == 1)
            const int32_t *lq = lp ? lp + k0 : nullptr;
            const size_t pshift = lp ? 0 : (size_t)k0;         // identity order: the problems' own arrays start at slot k0
            const double *xq = x + pshift * n + j0, *fq = f0 ? f0 + pshift * m : nullptr;
            if (analytic) {
                {
                    Timed t(h, NLH_K_DQ_JACOBIAN);
                    const int urc = rs.jac(rs.ctx, (void *)h->stream, kc, dprob + k0, n, X + (size_t)k0 * n, m, Pc);
                    if (urc) return dv_fail(h, urc, "jacobianfcn");
                }
                const dim3 grid((((m + 7) & ~7) + 255) / 256, n, kc);
                if (to_qrx) hipLaunchKernelGGL(k_dv_place_jac<true>, grid, dim3(256), 0, h->stream, m, n, (const double *)Pc, out + pshift * tst, lq, ld, coff, tst);
                else hipLaunchKernelGGL(k_dv_place_jac<false>, grid, dim3(256), 0, h->stream, m, n, (const double *)Pc, out + pshift * m * n, lq, ld, coff, tst);
                continue;
            }
            {
                Timed t(h, NLH_K_DQ_PANEL);                     // (a share of) the n perturbed evaluations of the problems that are due
                const size_t q0 = (size_t)k0 * n + j0;
                const int urc = rs.fcn(rs.ctx, (void *)h->stream, kc * jn, dprob + q0, n, (const double *)X + q0 * n, m, Pc);
                if (urc) return dv_fail(h, urc, "vecfcn");
            }
            Timed t(h, NLH_K_FD_JACOBIAN);                      // :274
            // (a column group of one problem: the kernels see a problem of jn columns whose arrays start at column j0)
            if (to_qrx) {
                const dim3 grid((m + FDQ_ROWS - 1) / FDQ_ROWS, (jn + FDQ_COLS - 1) / FDQ_COLS, kc);
                double *Tq = out + pshift * tst;
                const bool vec2 = (m % 2 == 0) && (((uintptr_t)Pc & 15) == 0);
                static const int nt_env = [] { const char *e = getenv("NLH_FDQ_NT"); return e ? atoi(e) : 1; }();
                if (vec2 && nt_env) hipLaunchKernelGGL((k_fd_jacobian_qrx<true, true>), grid, dim3(256), 0, h->stream, m, jn, (const double *)Pc, fq, xq, Tq, lq,
                                                       (const LmState *)nullptr, -1, ld, coff + j0, tst, n);
                else if (vec2) hipLaunchKernelGGL((k_fd_jacobian_qrx<true, false>), grid, dim3(256), 0, h->stream, m, jn, (const double *)Pc, fq, xq, Tq, lq,
                                                  (const LmState *)nullptr, -1, ld, coff + j0, tst, n);
                else hipLaunchKernelGGL((k_fd_jacobian_qrx<false, false>), grid, dim3(256), 0, h->stream, m, jn, (const double *)Pc, fq, xq, Tq, lq,
                                        (const LmState *)nullptr, -1, ld, coff + j0, tst, n);
            } else {
                constexpr int CJ = 8;
                double *Jq = out + pshift * m * n + (size_t)j0 * m;
                const bool vec2 = (m % 2 == 0) && ((((uintptr_t)Pc | (uintptr_t)Jq | (uintptr_t)fq) & 15) == 0);
                if (vec2)
                    hipLaunchKernelGGL((k_fd_jacobian<RB, CJ, true>), dim3((m / 2 + RB - 1) / RB, (jn + CJ - 1) / CJ, kc), dim3(RB), 0, h->stream, m, jn,
                                       (const double *)Pc, fq, xq, Jq, (const LmState *)nullptr, -1, lq, n);
                else
                    hipLaunchKernelGGL((k_fd_jacobian<RB, CJ, false>), dim3((m + RB - 1) / RB, (jn + CJ - 1) / CJ, kc), dim3(RB), 0, h->stream, m, jn,
                                       (const double *)Pc, fq, xq, Jq, (const LmState *)nullptr, -1, lq, n);
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------
// the dense-quadratic family as launchers
// ---------------------------------------------------------------------------
// One residual row per thread, a point per blockIdx: the row sum in ascending j, one multiply and one add per term --
// the bits of k_dq_residual and of every column of k_dq_panel.
static __global__ void __launch_bounds__(256)
k_dqv_fcn(int m, int n, int nblk, const double *__restrict__ A, const double *__restrict__ b, double gamma, const int32_t *__restrict__ dprob,
          const double *__restrict__ X, double *__restrict__ F)
{
    extern __shared__ double xs[];
    const int q = blockIdx.x / nblk, rb = blockIdx.x - q * nblk;
    const int p = dprob[q];
    for (int c = threadIdx.x; c < n; c += 256) xs[c] = X[(size_t)q * n + c];
    __syncthreads();
    const int i = rb * 256 + threadIdx.x;
    if (i >= m) return;
    const double *a = A + (size_t)p * m * n + i;
    double u = 0.0;
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        double av[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) av[t] = a[(size_t)(k + t) * m];
#pragma unroll
        for (int t = 0; t < 8; ++t) u = u + av[t] * xs[k + t];
    }
    for (; k < n; ++k) u = u + a[(size_t)k * m] * xs[k];
    F[(size_t)q * m + i] = (u + (gamma * u) * u) - b[(size_t)p * m + i];
}

// The same for DQV_PT consecutive points at a time.  A forward-difference Jacobian asks for n points of the same problem
// in a row: a workgroup takes a block of 256 rows and a tile of DQV_PT points, reads its rows of A ONCE for the tile (the
// per-point form reads them once per point: n times per Jacobian, 240 of the 875 ms of a 512 x 4096x256 solve) and keeps
// one accumulator per point -- each still the plain sum over j ascending, one multiply and one add per term, of that
// point's own x: the same bits.  A tile whose points belong to more than one problem (the seam between two problems when n
// is not a multiple of the tile, mixed calls) is taken point by point by the same workgroup.
#define DQV_PT 16
static __global__ void __launch_bounds__(256)
k_dqv_fcn_tile(int m, int n, int nblk, int npoints, const double *__restrict__ A, const double *__restrict__ b, double gamma,
               const int32_t *__restrict__ dprob, const double *__restrict__ X, double *__restrict__ F)
{
    extern __shared__ __attribute__((aligned(16))) double xs[];  // [n][DQV_PT]
    const int tile = blockIdx.x / nblk, rb = blockIdx.x - tile * nblk;
    const int q0 = tile * DQV_PT, nq = min(DQV_PT, npoints - q0);
    const int p = dprob[q0];
    bool same = true;
    for (int t = 1; t < nq; ++t) same = same && dprob[q0 + t] == p;          // (uniform: scalar loads)
    const int i = rb * 256 + threadIdx.x;
    if (!same) {                                                 // a seam between two problems: this tile point by point
        for (int t = 0; t < nq; ++t) {
            const int pt = dprob[q0 + t];
            __syncthreads();
            for (int c = threadIdx.x; c < n; c += 256) xs[c] = X[(size_t)(q0 + t) * n + c];
            __syncthreads();
            if (i < m) {
                const double *a = A + (size_t)pt * m * n + i;
                double u = 0.0;
                for (int k = 0; k < n; ++k) u = u + a[(size_t)k * m] * xs[k];
                F[(size_t)(q0 + t) * m + i] = (u + (gamma * u) * u) - b[(size_t)pt * m + i];
            }
        }
        return;
    }
    // x of the tile's points, index-major ([k][point]: the sixteen values a step needs are 128 contiguous bytes)
    for (int e = threadIdx.x; e < DQV_PT * n; e += 256) {
        const int t = e / n, k = e - t * n;
        xs[k * DQV_PT + t] = t < nq ? X[(size_t)q0 * n + e] : 0.0;
    }
    __syncthreads();
    if (i >= m) return;
    const double *a = A + (size_t)p * m * n + i;
    double u[DQV_PT];
#pragma unroll
    for (int t = 0; t < DQV_PT; ++t) u[t] = 0.0;
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        double av[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) av[c] = a[(size_t)(k + c) * m];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < DQV_PT; ++t) u[t] = u[t] + av[c] * xs[(k + c) * DQV_PT + t];
    }
    for (; k < n; ++k) {
        const double av = a[(size_t)k * m];
#pragma unroll
        for (int t = 0; t < DQV_PT; ++t) u[t] = u[t] + av * xs[k * DQV_PT + t];
    }
    const double bi = b[(size_t)p * m + i];
#pragma unroll
    for (int t = 0; t < DQV_PT; ++t)
        if (t < nq) F[(size_t)(q0 + t) * m + i] = (u[t] + (gamma * u[t]) * u[t]) - bi;
}

static __global__ void __launch_bounds__(256)
k_dqv_jac(int m, int n, int nblk, const double *__restrict__ A, double gamma, const int32_t *__restrict__ dprob, const double *__restrict__ X,
          double *__restrict__ J)
{
    extern __shared__ double xs[];
    const int q = blockIdx.x / nblk, rb = blockIdx.x - q * nblk;
    const int p = dprob[q];
    for (int c = threadIdx.x; c < n; c += 256) xs[c] = X[(size_t)q * n + c];
    __syncthreads();
    const int i = rb * 256 + threadIdx.x;
    if (i >= m) return;
    const double *a = A + (size_t)p * m * n + i;
    double u = 0.0;
    for (int k = 0; k < n; ++k) u = u + a[(size_t)k * m] * xs[k];
    const double s = 1.0 + 2.0 * gamma * u;                     // k_dq_jacobian's expression
    double *Jq = J + (size_t)q * m * n + i;
    for (int k = 0; k < n; ++k) Jq[(size_t)k * m] = s * a[(size_t)k * m];
}

void nlh_devfcn_init_device(int lds_max)
{   // x of a point lives in dynamic LDS: beyond 8192 columns that exceeds the 64 KB a kernel gets without asking
    (void)hipFuncSetAttribute((const void *)k_dqv_fcn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    (void)hipFuncSetAttribute((const void *)k_dqv_fcn_tile, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    (void)hipFuncSetAttribute((const void *)k_dqv_jac, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
}

int nlh_dq_device_fcn(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n, const double *dX, int32_t m,
                      double *dF)
{
    const nlh_dq_device_ctx *c = (const nlh_dq_device_ctx *)ctx;
    if (!c || npoints <= 0) return c ? 0 : 1;
    if (n > NLH_DQ_MAX_N) return NLH_ARRAY_SIZE_ERROR;          // a launcher's failure code is what the solver returns
    const int nblk = (m + 255) / 256;
    hipStream_t s = (hipStream_t)hip_stream;
    // a handful of points (a trial point per problem), or x vectors that do not fit LDS sixteen at a time: point by point
    if (npoints < 4 * DQV_PT || sizeof(double) * (size_t)DQV_PT * n > 64 * 1024) {
        hipLaunchKernelGGL(k_dqv_fcn, dim3((unsigned)((size_t)npoints * nblk)), dim3(256), sizeof(double) * (size_t)n, s, m, n, nblk,
                           c->dA, c->db, c->gamma, dprob, dX, dF);
        return 0;
    }
    const int ntile = (npoints + DQV_PT - 1) / DQV_PT;
    hipLaunchKernelGGL(k_dqv_fcn_tile, dim3((unsigned)((size_t)ntile * nblk)), dim3(256), sizeof(double) * (size_t)DQV_PT * n, s, m, n, nblk, npoints,
                       c->dA, c->db, c->gamma, dprob, dX, dF);
    return 0;
}

int nlh_dq_device_jac(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n, const double *dX, int32_t m,
                      double *dJ)
{
    const nlh_dq_device_ctx *c = (const nlh_dq_device_ctx *)ctx;
    if (!c || npoints <= 0) return c ? 0 : 1;
    if (n > NLH_DQ_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    const int nblk = (m + 255) / 256;
    hipLaunchKernelGGL(k_dqv_jac, dim3((unsigned)((size_t)npoints * nblk)), dim3(256), sizeof(double) * (size_t)n, (hipStream_t)hip_stream, m, n, nblk,
                       c->dA, c->gamma, dprob, dX, dJ);
    return 0;
}

// ---------------------------------------------------------------------------
// vecfcn_helper%jacobian for a device residual, every problem
// ---------------------------------------------------------------------------
int nlh_fd_jacobian_device(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn,
                           void *ctx, const double *dx, const double *dfv, double *dJ)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :240
    if (nprob <= 0) return 0;
    if (m < 1 || n < 1 || !dx || !dJ) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    ResidualSource rs;
    rs.fcn = fcn; rs.jac = jacfcn; rs.ctx = ctx;
    // slices keep the point count of one launcher call (nprob * n) and the panel inside 31 bits / a bounded workspace
    // ... and the problem count inside what the kernels carry in gridDim.y / .z (NLH_MAX_LOCKSTEP, as every other *_device entry point)
    const int32_t per = (int32_t)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)nprob, (size_t)NLH_MAX_LOCKSTEP),
                                                                      ((size_t)1 << 30) / ((size_t)n * std::max(m, n))));
    for (int32_t p0 = 0; p0 < nprob; p0 += per) {
        const int32_t cnt = std::min(per, nprob - p0);
        int rc;
        const double *f0 = dfv ? dfv + (size_t)p0 * m : nullptr;
        ResidualSource r = rs.shifted(p0, m, n);
        if (!f0 && !(jacfcn)) {                                 // :257-259
            if ((rc = ensure(h, h->fdev, sizeof(double) * (size_t)cnt * m))) return rc;
            if ((rc = residual_eval(h, r, cnt, m, n, dx + (size_t)p0 * n, (double *)h->fdev.p, nullptr, nullptr, -1))) return rc;
            f0 = (const double *)h->fdev.p;
        }
        if ((rc = residual_jacobian(h, r, cnt, m, n, dx + (size_t)p0 * n, f0, dJ + (size_t)p0 * m * n, nullptr, nullptr, -1, false,
                                    false, true))) return rc;
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}
