// nlh_common.h -- shared device helpers for the gfx950 kernels (wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define NLH_WAVE 64

// Per-problem scalar state of the batched Levenberg-Marquardt driver
// (the locals of lss_solve, src/nonlin_least_squares.f90:152-160).
enum NlhStage : int32_t {
    ST_NEED_JAC = 0,   // outer-loop head: Jacobian + factorisation due
    ST_HAVE_JAC = 1,   // J formed, factorisation due
    ST_NE_READY = 2,   // Gram/Cholesky factors valid, lmpar due
    ST_NEED_QR = 3,    // Householder QR due (policy, ill-conditioning, or GN step rejected)
    ST_QR_READY = 4,   // QR factors valid, lmpar due
    ST_TRIAL_READY = 5,// trial point in wa2, residual evaluation due
    ST_TRIAL_DONE = 6, // residual at trial point in wa4, update due
    ST_DONE = 7,
    ST_NEED_PCHOL = 8  // Gram factors with lmfactor's pivot order due (lmpar iteration or weak pivot)
};

struct LmState {
    double fnorm, fnorm1, par, delta, xnorm, gnorm, pnorm;
    double temp1n;   // || R P^T p ||  (src/nonlin_least_squares.f90:307-313)
    double tailsq;   // sum of squares of wa4(n+1:m) (lmpar deviation A, :531)
    int32_t iter, neval, njac;
    int32_t stage;
    int32_t factor_kind;   // 0 = normal equations, 1 = Householder QR (this outer iteration)
    int32_t inner_pass;    // number of lmpar calls already made in this outer iteration
    int32_t fcnvrg, xcnvrg, gcnvrg;
    int32_t flag;          // NL_* failure flag (:358-363)
    int32_t qr_count;      // diagnostics: how many QR fallbacks happened
    int32_t head_done;     // the outer-loop head already ran in this outer iteration
    int32_t signs_done;    // normal-equations factors already carry lmfactor's row signs
    int32_t pivoted;       // normal-equations factors use lmfactor's pivot order (else natural order)
    int32_t slow_lmpar;    // how often lmpar's iteration ran (the Gauss-Newton step was not accepted at :477-481)
    int32_t rejects;       // trial points rejected (:340 not taken)
    int32_t first_slow;    // outer iteration of the first of either (0: none yet)
};

__device__ __forceinline__ double wave_reduce_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_down(v, off, 64);
    return v;   // valid in lane 0
}

__device__ __forceinline__ double wave_reduce_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// Deterministic block-wide sum; result broadcast to all threads.
// sh must hold at least blockDim.x/64 + 1 doubles.  Two barriers.
// Value of v in lane l (l uniform; cheapest with a compile-time l): two v_readlane_b32.
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// sqrt(x) and 1/sqrt(x) together from the hardware estimate plus two coupled Goldschmidt steps
// (a short dependent chain; the IEEE sqrt and divide sequences are several times longer).
// Only used on the non-exact factor policies.  x must be a positive normal number.
__device__ __forceinline__ void sqrt_rsqrt(double x, double &s, double &r)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    s = g; r = 2.0 * h;
}

__device__ __forceinline__ double block_reduce_sum(double v, double *sh)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    v = wave_reduce_sum(v);
    __syncthreads();            // protect sh from the previous use
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < nw; ++w) r = r + sh[w];   // fixed order, every thread
    return r;
}

__device__ __forceinline__ double block_reduce_max(double v, double *sh)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    v = wave_reduce_max(v);
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    double r = sh[0];
    for (int w = 1; w < nw; ++w) r = fmax(r, sh[w]);
    return r;
}

// First index of the maximum of v over the block (strict '>' => first max wins,
// as in lmfactor's pivot search, src/nonlin_least_squares.f90:622-625).
// Each thread passes its best (value, index) with the smallest index among ties;
// threads with no candidate pass idx = INT_MAX and any value.
__device__ __forceinline__ int block_argmax_first(double v, int idx, double *shv, int *shi)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_down(v, off, 64);
        int oi = __shfl_down(idx, off, 64);
        bool take = (oi != 0x7fffffff) && (idx == 0x7fffffff || ov > v || (ov == v && oi < idx));
        if (take) { v = ov; idx = oi; }
    }
    __syncthreads();
    if (lane == 0) { shv[wid] = v; shi[wid] = idx; }
    __syncthreads();
    double bv = shv[0];
    int bi = shi[0];
    for (int w = 1; w < nw; ++w) {
        double ov = shv[w];
        int oi = shi[w];
        bool take = (oi != 0x7fffffff) && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi));
        if (take) { bv = ov; bi = oi; }
    }
    return bi;
}

// The same search when thread t holds candidate index base + t (indices ascend with the thread id) and the values are
// norms (>= +0.0, or NaN): the maximum of the BIT PATTERNS by 32-bit DPP steps (high words, then the low words of the lanes
// that hold the high maximum) instead of twelve shuffles of (value, index) pairs, one barrier instead of two.  Ties go to
// the lowest thread; a NaN is never greater than anything and nothing is greater than a NaN, so a NaN wins if and only
// if it is the very first candidate (thread 0), exactly as in the reference's scan from the left.  sh: 3 * (blockDim / 64)
// words of LDS used for nothing else.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t nlh_dpp_umax_step(uint32_t v)
{
    // bound_ctrl: lanes without a source (and rows outside the mask) read 0 -- neutral for an unsigned maximum
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
    return o > v ? o : v;
}
__device__ __forceinline__ uint32_t nlh_wave_umax(uint32_t v)
{
    v = nlh_dpp_umax_step<0x111, 0xf>(v);
    v = nlh_dpp_umax_step<0x112, 0xf>(v);
    v = nlh_dpp_umax_step<0x114, 0xf>(v);
    v = nlh_dpp_umax_step<0x118, 0xf>(v);
    v = nlh_dpp_umax_step<0x142, 0xa>(v);
    v = nlh_dpp_umax_step<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int block_argmax_first_norms(double v, bool valid, uint32_t *sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = (blockDim.x + 63) >> 6;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    unsigned long long key = (bits >> 63) ? 1ull : bits + 1ull;               // (-0.0 counts as +0.0; norms are never negative)
    if (v != v) key = (tid == 0) ? 0x7ff0000000000002ull : 1ull;
    if (!valid) key = 0ull;
    const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
    const uint32_t mh = nlh_wave_umax(hi);
    const uint32_t ml = nlh_wave_umax(hi == mh ? lo : 0u);
    const unsigned long long hit = __ballot(hi == mh && lo == ml);
    if (lane == 0) { sh[3 * wid] = mh; sh[3 * wid + 1] = ml; sh[3 * wid + 2] = (uint32_t)(wid * 64 + __ffsll((long long)hit) - 1); }
    __syncthreads();
    uint32_t bh = sh[0], bl = sh[1], bp = sh[2];
    for (int w = 1; w < nw; ++w) {
        const uint32_t oh = sh[3 * w], ol = sh[3 * w + 1];
        if (oh > bh || (oh == bh && ol > bl)) { bh = oh; bl = ol; bp = sh[3 * w + 2]; }
    }
    return (int)bp;
}

#define NLH_SQRT_EPS 1.4901161193847656e-08   /* sqrt(epsilon(1d0)), multi_eqn_mult_var.f90:263-264 */
#define NLH_EPS      2.220446049250313e-16
#define NLH_DWARF    2.2250738585072014e-308  /* tiny(1d0), least_squares.f90:442 */

// Forward-difference step of vfh_jac_fcn (src/nonlin_multi_eqn_mult_var.f90:268-270).
__device__ __forceinline__ double fd_step(double xj)
{
    double h = NLH_SQRT_EPS * fabs(xj);
    if (h == 0.0) h = NLH_SQRT_EPS;
    return h;
}

// ---------------------------------------------------------------------------
// Reference-order ("exact") reductions.  With EXACT = true every sum is formed in the
// CPU path's order, so results are bit-identical to it; elementwise work stays parallel.
//
// NORM2 follows the flang runtime (the reference's compiler here; the intrinsic is
// processor-dependent): a running maximum mx and s = sum (x/mx)^2 rescaled whenever the
// maximum grows; result mx*sqrt(1+s).  The divisions and squares of that algorithm are
// independent once the prefix maxima are known, so a chunk of elements is prepared in
// parallel (prefix-max scan in LDS, then c_i, d_i with s <- s*c_i + d_i) and only the
// two-flop recurrence is folded serially by one thread.
// ---------------------------------------------------------------------------
#define NLH_NCH 512     // chunk length of the exact NORM2; scratch = 3*NLH_NCH + 8 doubles

struct Norm2State { double mx, s; };

// get(i) returns element i (any sign).  scratch: LDS, 3*NLH_NCH + 8 doubles.  Whole block.
template <typename Get>
__device__ double norm2_flang_block(Get get, int len, double *scratch)
{
    const int tid = threadIdx.x, BS = blockDim.x;
    double *av = scratch, *pm = scratch + NLH_NCH, *dd = scratch + 2 * NLH_NCH;
    double *carry = scratch + 3 * NLH_NCH;    // [0] = mx, [1] = s
    __syncthreads();
    if (tid == 0) { carry[0] = 0.0; carry[1] = 0.0; }
    __syncthreads();
    for (int base = 0; base < len; base += NLH_NCH) {
        const int cl = min(NLH_NCH, len - base);
        for (int i = tid; i < cl; i += BS) { const double a = fabs(get(base + i)); av[i] = a; pm[i] = a; }
        __syncthreads();
        // inclusive prefix maximum (exact: max is associative); ping-pong between pm and dd
        double *src = pm, *dst = dd;
        for (int off = 1; off < cl; off <<= 1) {
            for (int i = tid; i < cl; i += BS) dst[i] = (i >= off) ? fmax(src[i], src[i - off]) : src[i];
            __syncthreads();
            double *t = src; src = dst; dst = t;
        }
        if (src != pm) {
            for (int i = tid; i < cl; i += BS) pm[i] = src[i];
            __syncthreads();
        }
        const double mx_in = carry[0];
        // per-element recurrence coefficients: s <- s*c + d  (c stored in av, d in dd)
        for (int i = tid; i < cl; i += BS) {
            const double a = av[i];
            const double prev = (i == 0) ? mx_in : fmax(mx_in, pm[i - 1]);   // running max before element i
            double c = 1.0, d = 0.0;
            if (prev == 0.0) {
                // mx was zero: element becomes the maximum, s untouched
            } else if (a > prev) {
                const double t = prev / a, tsq = t * t;
                c = tsq; d = tsq;
            } else if (a != 0.0) {
                const double t = a / prev;
                d = t * t;
            }
            av[i] = c;
            dd[i] = d;
        }
        __syncthreads();
        if (tid == 0) {
            // the recurrence itself is serial; read eight (c, d) pairs at a time so that the LDS latency
            // is paid once per batch, and take the multiply out of the chain when no new maximum occurs
            double s = carry[1];
            int i = 0;
            for (; i + 8 <= cl; i += 8) {
                double c8[8], d8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { c8[u] = av[i + u]; d8[u] = dd[i + u]; }
                bool plain = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) plain = plain && (c8[u] == 1.0);
                if (plain) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) s = s + d8[u];
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (c8[u] != 1.0) s = s * c8[u];
                        s = s + d8[u];
                    }
                }
            }
            for (; i < cl; ++i) {
                const double c = av[i];
                if (c != 1.0) s = s * c;
                s = s + dd[i];
            }
            carry[1] = s;
            carry[0] = fmax(mx_in, pm[cl - 1]);
        }
        __syncthreads();
    }
    const double r = carry[0] * sqrt(1.0 + carry[1]);
    __syncthreads();
    return r;
}

// The same NORM2 with a wide chunk: cd holds 2*cap + 4*EMAX doubles (cap <= EMAX*blockDim), aux 40 + blockDim/2 + 4
// doubles (16-byte aligned; the tails are read-ahead padding of the serial phase).  The running maximum before every element comes from a wave scan (shuffles) plus one LDS
// exchange, i.e. three barriers per chunk instead of log2(chunk) of them.  The serial recurrence is
// unchanged, but the one thread that runs it executes as few instructions as possible: every thread
// flags whether its EMAX consecutive elements are free of a new maximum (all c == 1), and a flagged
// run is a plain chain of adds over terms fetched one run ahead.  Bit-identical to norm2_flang_block.
template <int EMAX, typename Get>
__device__ double norm2_flang_block_wide(Get get, int len, double *cd, int cap, double *aux)
{
    static_assert(EMAX % 2 == 0, "runs are read as pairs");
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = BS >> 6;
    double *cs = cd, *dsv = cd + cap, *wmax = aux, *carry = aux + 32;
    int *flags = reinterpret_cast<int *>(aux + 40);
    __syncthreads();
    if (tid == 0) { carry[0] = 0.0; carry[1] = 0.0; }
    for (int base = 0; base < len; base += cap) {
        const int cl = min(cap, len - base);
        const int E = (cl + BS - 1) / BS, i0 = tid * E;              // E <= EMAX consecutive elements per thread
        double a[EMAX], lm = 0.0;
#pragma unroll
        for (int u = 0; u < EMAX; ++u) {
            a[u] = (u < E && i0 + u < cl) ? fabs(get(base + i0 + u)) : 0.0;
            lm = fmax(lm, a[u]);
        }
        double sc = lm;                                              // inclusive prefix maximum over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(sc, off, 64);
            if (lane >= off) sc = fmax(sc, o);
        }
        double ex = __shfl_up(sc, 1, 64);
        if (lane == 0) ex = 0.0;
        __syncthreads();                                             // carry of the previous chunk is final
        if (lane == 63) wmax[wid] = sc;
        __syncthreads();
        const double mx_in = carry[0];
        double prev = fmax(mx_in, ex);
        for (int w = 0; w < wid; ++w) prev = fmax(prev, wmax[w]);
        bool plain = (E == EMAX) && (i0 + EMAX <= cl);
#pragma unroll
        for (int u = 0; u < EMAX; ++u) {
            if (u < E && i0 + u < cl) {
                double c = 1.0, d = 0.0;
                if (prev == 0.0) {
                    // mx was zero: element becomes the maximum, s untouched
                } else if (a[u] > prev) {
                    const double t = prev / a[u], tsq = t * t;
                    c = tsq; d = tsq;
                } else if (a[u] != 0.0) {
                    const double t = a[u] / prev;
                    d = t * t;
                }
                plain = plain && (c == 1.0);
                cs[i0 + u] = c;
                dsv[i0 + u] = d;
                prev = fmax(prev, a[u]);
            }
        }
        flags[tid] = plain ? 1 : 0;
        __syncthreads();
        if (tid == 0) {
            double s = carry[1];
            const int nrun = (cl + E - 1) / E;                       // runs of E elements, one per contributing thread
            auto general = [&](int i, int cnt) {
                for (int u = 0; u < cnt; ++u) {
                    const double c = cs[i + u];
                    if (c != 1.0) s = s * c;
                    s = s + dsv[i + u];
                }
            };
            if (E == EMAX) {
                // Four runs per iteration, four register buffers: the LDS reads of a run are issued four runs (32 adds)
                // before its terms enter the chain, in straight-line code, so the waits are counted ones and the chain
                // itself is the only latency left (one lgkmcnt(0) per run cost 2/3 of this function's time).  dsv and
                // flags carry 4 runs / 8 entries of padding behind them for the read-ahead.
                const int nfast = nrun & ~3;
                double d0[EMAX], d1[EMAX], d2[EMAX], d3[EMAX];
                auto loadrun = [&](double (&d)[EMAX], int r) {
                    const double2 *src = reinterpret_cast<const double2 *>(dsv + r * EMAX);
#pragma unroll
                    for (int u = 0; u < EMAX / 2; ++u) { const double2 v2 = src[u]; d[2 * u] = v2.x; d[2 * u + 1] = v2.y; }
                };
                auto adds = [&](const double (&d)[EMAX]) {
#pragma unroll
                    for (int u = 0; u < EMAX; ++u) s = s + d[u];
                };
                int r = 0;
                int4 f = make_int4(0, 0, 0, 0);
                if (nfast > 0) {
                    loadrun(d0, 0); loadrun(d1, 1); loadrun(d2, 2); loadrun(d3, 3);
                    f = *reinterpret_cast<const int4 *>(flags);
                }
                for (; r < nfast; r += 4) {
                    const int4 fn = *reinterpret_cast<const int4 *>(flags + r + 4);
                    if (f.x & f.y & f.z & f.w) {
                        adds(d0); loadrun(d0, r + 4); __builtin_amdgcn_sched_barrier(0);   // keep each reload right behind
                        adds(d1); loadrun(d1, r + 5); __builtin_amdgcn_sched_barrier(0);   // the adds that free its registers
                        adds(d2); loadrun(d2, r + 6); __builtin_amdgcn_sched_barrier(0);
                        adds(d3); loadrun(d3, r + 7); __builtin_amdgcn_sched_barrier(0);
                    } else {
                        if (f.x) adds(d0); else general(r * E, E);
                        if (f.y) adds(d1); else general((r + 1) * E, E);
                        if (f.z) adds(d2); else general((r + 2) * E, E);
                        if (f.w) adds(d3); else general((r + 3) * E, min(E, cl - (r + 3) * E));
                        loadrun(d0, r + 4); loadrun(d1, r + 5); loadrun(d2, r + 6); loadrun(d3, r + 7);
                    }
                    f = fn;
                }
                for (; r < nrun; ++r) general(r * E, min(E, cl - r * E));
            } else {
                general(0, cl);
            }
            double mx = mx_in;
            for (int w = 0; w < nw; ++w) mx = fmax(mx, wmax[w]);
            carry[1] = s;
            carry[0] = mx;
        }
    }
    __syncthreads();
    const double r = carry[0] * sqrt(1.0 + carry[1]);
    __syncthreads();
    return r;
}

// The same NORM2 with the serial recurrence spread over the LANES of one wave.  A chunk is 64 runs of EL consecutive
// elements; run l belongs to lane l of wave 0, which holds its EL terms d_i in registers.  The running sum travels
// down the lanes: at step l every lane takes its lower neighbour's value (one DPP wave shift) and adds its own EL
// terms, a straight-line chain of dependent adds with register operands -- lane l's result is the true partial sum
// after run l, the other lanes' results at that step are never used.  Measured on gfx950 a dependent fp64 add with
// register operands issues every ~2 ns; a taken branch costs ~13 ns and an LDS round trip ~100 ns, which is what the
// one-thread forms above spend most of their time on (7.6 ns per element).  A run that contains a new maximum
// (c_i != 1, flagged by the threads that prepared it) takes the general s <- s*c + d form for that step only.
// Bit-identical to norm2_flang_block.  BSZ = blockDim.x (64 .. 1024, a power of two), EL a multiple of BSZ / 64 (and of 2);
// cd: 64 * EL + 128 doubles of LDS, 16-byte aligned; aux: 40 doubles + BSZ ints.
__device__ __forceinline__ double nlh_wave_shr1(double t)
{
    int lo = __double2loint(t), hi = __double2hiint(t);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);      // wave_shr:1 (lane 0 keeps its value)
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Which wave of a workgroup runs a serial chain: spread over the wave indices by workgroup id.  Measured since
// (profiles/ubench/chain_waves.hip): the waves of a workgroup do NOT sit on the SIMDs in index order (HW_ID: wave 0 of
// workgroups 0 and 256, which share a CU, ran on SIMDs 0 and 2), and two chains on one CU cost nothing either way
// (2.32 ns per add with wave 0 in both, 2.30 spread, 2.23 alone) -- the choice is neutral; a dependent fp64 add issues
// every ~1.9 ns whatever the EXEC mask (no skipping of idle 16-lane passes), so 2.2 ns per term is the floor of every
// ordered sum in this library.
__device__ __forceinline__ int nlh_chain_wave(int nwaves)
{
    const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    return (int)((b + (b >> 3) + (b >> 8)) & (unsigned)(nwaves - 1));    // differs for ids 1, 8 and 256 apart
}

// REGGEN: the general runs (those with a new maximum) multiply every term by a per-lane selected operand -- more
// registers, faster when the workgroup has its CU to itself (a handful of problems); without it the flags of the one
// lane that matters are read into a scalar and tested a quarter of a run at a time (see below): +1.4 % on the
// 2048-problem headline over the form it replaced (the terms once more from LDS, a branch per element), 1 % slower
// than REGGEN for a problem alone.
template <int EL, int BSZ, bool REGGEN = false, typename Get>
__device__ double norm2_flang_block_lanes(Get get, int len, double *cd, double *aux)
{
    constexpr int CAP = 64 * EL, E = CAP / BSZ, TPR = EL / E;   // E elements per thread, TPR threads per run
    static_assert(CAP % BSZ == 0 && EL % E == 0 && E % 2 == 0 && EL % 16 == 0, "chunk must split evenly");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int nw = BSZ / 64;
    const int cw = nlh_chain_wave(nw);                                   // the wave that runs the serial recurrence
    double *dsv = cd, *wmax = aux, *carry = aux + 32;
    int *tflags = reinterpret_cast<int *>(aux + 40);
    __syncthreads();
    if (tid == 0) { carry[0] = 0.0; carry[1] = 0.0; }
    for (int base = 0; base < len; base += CAP) {
        const int cl = min(CAP, len - base);
        const int i0 = tid * E;
        double a[E], lm = 0.0;
#pragma unroll
        for (int u = 0; u < E; ++u) {
            a[u] = (i0 + u < cl) ? fabs(get(base + i0 + u)) : 0.0;
            lm = fmax(lm, a[u]);
        }
        double sc = lm;                                              // inclusive prefix maximum over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(sc, off, 64);
            if (lane >= off) sc = fmax(sc, o);
        }
        double ex = __shfl_up(sc, 1, 64);
        if (lane == 0) ex = 0.0;
        __syncthreads();                                             // carry of the previous chunk is final
        if (lane == 63) wmax[wid] = sc;
        __syncthreads();
        const double mx_in = carry[0];
        double prev = fmax(mx_in, ex);
        for (int w = 0; w < wid; ++w) prev = fmax(prev, wmax[w]);
        // The scale factor of an element is 1 unless the element is a new maximum, and then it EQUALS its term (both
        // (mx / a)^2): a bit per element says which, the terms alone go to LDS.
        unsigned newmax = 0;
        // element i of the chunk lives at i + 2 * (i / EL): two doubles of padding behind every run, so that the 16-byte
        // reads of a run's owner fall on other banks than its neighbours'
        double2 *ddst = reinterpret_cast<double2 *>(dsv + i0 + 2 * (i0 / EL));
#pragma unroll
        for (int u = 0; u < E; u += 2) {
            double dd[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // branch-free (sixteen independent divisions the compiler can interleave): a new maximum divides the
                // old one by the element, anything else the element by the maximum; mx == 0: the element becomes the
                // maximum and s stays (0/0 is discarded by the select); a zero element gives 0 / mx = +0.0
                const double av = a[u + h];
                const bool gt = av > prev;
                const double t = (gt ? prev : av) / (gt ? av : prev), tsq = t * t;
                const double d = (prev == 0.0) ? 0.0 : tsq;
                if (gt && prev != 0.0 && tsq != 1.0) newmax |= 1u << (u + h);    // s <- s * tsq + tsq
                dd[h] = d;
                prev = fmax(prev, av);
            }
            ddst[u >> 1] = make_double2(dd[0], dd[1]);
        }
        tflags[tid] = (int)newmax;
        __syncthreads();
        if (wid == cw) {
            const int nl = (cl + EL - 1) / EL;                       // runs in use
            double d[EL];
            const double2 *mine = reinterpret_cast<const double2 *>(dsv + lane * (EL + 2));
#pragma unroll
            for (int u = 0; u < EL / 2; ++u) { const double2 v2 = mine[u]; d[2 * u] = v2.x; d[2 * u + 1] = v2.y; }
            unsigned long long nm = 0;                               // this lane's run: which elements are new maxima
#pragma unroll
            for (int k = 0; k < TPR; ++k) nm |= (unsigned long long)(unsigned)tflags[lane * TPR + k] << (k * E);
            const unsigned long long mask = __ballot(nm == 0);       // runs without one
            double t = carry[1];
            // Eight runs per trip, and a trip whose runs are all free of a new maximum is straight-line code: a taken
            // branch costs ~40 cycles on this part and the test-and-branch per run was a quarter of the
            // chain's time.  t is the same in every lane on entry, so the shift before run 0 changes nothing; the terms
            // behind the vector's end are +0.0, so the runs that fill up the last trip hand the sum on unchanged (sums of
            // squares are never -0.0) and the result is read from the last lane of the last trip.
            const int ntrip = (nl + 7) >> 3;
#pragma unroll 1
            for (int g = 0; g < ntrip; ++g) {
                const unsigned m8 = (unsigned)(mask >> (8 * g)) & 0xffu;
                if (m8 == 0xffu) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        t = nlh_wave_shr1(t);
#pragma unroll
                        for (int u = 0; u < EL; ++u) t = t + d[u];
                    }
                } else {
#pragma unroll 1
                    for (int q = 0; q < 8; ++q) {
                        t = nlh_wave_shr1(t);
                        if ((m8 >> q) & 1u) {
#pragma unroll
                            for (int u = 0; u < EL; ++u) t = t + d[u];
                        } else if (REGGEN) {
#pragma unroll
                            for (int u = 0; u < EL; ++u) {           // s <- s * c + d with c = d at a new maximum, 1 elsewhere
                                t = t * (((nm >> u) & 1ull) ? d[u] : 1.0);
                                t = t + d[u];
                            }
                        } else {
                            // A run with new maxima: s <- s * c + d with c = d at a new maximum, 1 elsewhere.  Only lane
                            // l's result of step l is ever used, so the flags of lane l serve every lane: a scalar,
                            // tested sixteen elements at a time -- a quarter without a new maximum is sixteen plain adds,
                            // the others multiply by a selected operand (x * 1.0 == x), no branch per element and no
                            // second copy of the terms.
                            const int l = 8 * g + q;
                            const unsigned long long nml = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(nm >> 32), l) << 32)
                                                           | (unsigned)__builtin_amdgcn_readlane((int)nm, l);
#pragma unroll
                            for (int qd = 0; qd < EL; qd += 16) {
                                if (((nml >> qd) & 0xffffull) == 0) {
#pragma unroll
                                    for (int u = qd; u < qd + 16; ++u) t = t + d[u];
                                } else {
#pragma unroll
                                    for (int u = qd; u < qd + 16; ++u) {
                                        t = t * (((nml >> u) & 1ull) ? d[u] : 1.0);
                                        t = t + d[u];
                                    }
                                }
                            }
                        }
                    }
                }
            }
            const int lo = __builtin_amdgcn_readlane(__double2loint(t), 8 * ntrip - 1);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(t), 8 * ntrip - 1);
            double mx = mx_in;
#pragma unroll
            for (int w = 0; w < nw; ++w) mx = fmax(mx, wmax[w]);
            if (lane == 0) { carry[1] = __hiloint2double(hi, lo); carry[0] = mx; }
        }
    }
    __syncthreads();
    const double r = carry[0] * sqrt(1.0 + carry[1]);
    __syncthreads();
    return r;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the global loads in flight
// (s_waitcnt vmcnt(0)) -- a full memory latency per round of a pipelined loop.
__device__ __forceinline__ void nlh_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The left-to-right sum s0 + d(0) + d(1) + ... of NON-NEGATIVE terms WITHOUT the serial chain, bit for bit (round 4).
// While the running sum s stays in the binade of s0, fl(s + d) = s + RNE(d to a multiple of ulp(s0)): additions of
// multiples of ulp, exact and hence associative -- except where d / ulp ends in exactly one half, where round-half-even
// looks at the parity of s / ulp.  A term is therefore a map "add c0 if the running sum is even (in ulps), c1 if it is
// odd"; such maps compose into maps of the same form, so a lane folds its EL terms into one pair (c0, c1) and the wave
// folds its 64 pairs in lane order with shuffles: 6 fp64 instructions a term instead of a dependent add, no chain.
//   * RNE(d): t = C + d, q = t - C with C = 1.5 * 2^e (ulp(C) = ulp(s0)); the remainder r = d - q is exact, and the term
//     is a tie iff |r| == ulp / 2 (the rounding of C + d resolves ties by C's parity, so ties are redone: floor part
//     d - ulp / 2, plus one ulp where the running sum plus that floor is odd).  Zero and subnormal terms need no case.
//   * parity of a multiple c of ulp: fract(c * (0.5 / ulp)) != 0.
//   * ok = false -- the caller runs the chain instead -- when the sum leaves the binade (the term that carries it over
//     must be rounded once, at the coarser ulp) or a term exceeds 2^e / 4 (C + d must stay in C's binade).
// d[]: this lane's EL consecutive terms (all >= +0, no NaN); all 64 lanes call; s0 wave-uniform, normal, >= 4.
// profiles/ubench/intsum2.py (and intsum.py, the integer formulation) are CPU prototypes: 1500 (3000) chunks of 3072 terms
// incl. exact ties, zeros and sums next to a binade boundary against the plain loop, no mismatch.  Requires -ffp-contract=off.
__device__ __forceinline__ bool nlh_odd_ulps(double c, double inv2ulp)
{
    const double z = c * inv2ulp;
    return z != floor(z);
}
template <int EL>
__device__ __forceinline__ double ordered_possum_wave_int(const double (&d)[EL], double s0, bool &ok)
{
    const long long sb = __double_as_longlong(s0);
    const long long es = sb >> 52;                                   // (s0 > 0: no sign bit)
    const double ulp = __longlong_as_double((es - 52) << 52), hu = 0.5 * ulp, inv2ulp = 0.5 / ulp;
    const double C = __longlong_as_double((es << 52) | (1ll << 51)), lim = 0.25 * __longlong_as_double(es << 52);
    double c0 = 0.0, c1 = 0.0;
    bool fine = true;
#pragma unroll
    for (int u = 0; u < EL; ++u) {
        const double x = d[u];
        fine = fine && (x <= lim);
        const double t = C + x, q = t - C, r = x - q;
        const bool tie = fabs(r) == hu;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tie) != 0ull, 0)) {       // (uniform; one term in 2^12 or so)
            const double f = x - hu;
            const double a0 = tie ? f + (nlh_odd_ulps(c0 + f, inv2ulp) ? ulp : 0.0) : q;
            const double a1 = tie ? f + (nlh_odd_ulps(c1 + f, inv2ulp) ? 0.0 : ulp) : q;
            c0 = c0 + a0;
            c1 = c1 + a1;
        } else {
            c0 = c0 + q;
            c1 = c1 + q;
        }
    }
    // fold the lanes in order: (left then right)(b) = left(b) + right(parity after left)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double r0 = __shfl_down(c0, off, 64), r1 = __shfl_down(c1, off, 64);
        const double n0 = c0 + (nlh_odd_ulps(c0, inv2ulp) ? r1 : r0);         // running sum even on entry
        const double n1 = c1 + (nlh_odd_ulps(c1, inv2ulp) ? r0 : r1);         // odd on entry
        c0 = n0; c1 = n1;                                           // (lanes whose partner is out of range fold garbage: lane 0 does not)
    }
    const double totl = (sb & 1ll) ? c1 : c0;
    const double tot = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(totl)), __builtin_amdgcn_readfirstlane(__double2loint(totl)));
    const double out = s0 + tot;                                    // (exact while the binade holds)
    ok = (__builtin_amdgcn_ballot_w64(!fine) == 0ull) && ((__double_as_longlong(out) >> 52) == es);
    return out;
}

// norm2_flang_block_lanes for vectors of MANY chunks (a 65536-row column is 16), software-pipelined, with a wave that
// does nothing but the chain.  In the plain form a chunk costs a memory latency for its elements, the prefix maximum
// with two barriers, the divisions, an LDS hand-over and then the 64 * EL-add chain -- 21 us at EL = 64, of which the
// chain (the only part that is serial by definition) is 9.2.  The running MAXIMUM does not go through the chain, so
// everything but the chain can run a chunk ahead: the workgroup has BSZ / 64 PREPARING waves plus one CHAIN wave
// (blockDim.x >= BSZ + 64; waves beyond those only pass the barriers); while the chain wave adds chunk c the others form the coefficients of chunk c + 1 (second
// LDS buffer), the maxima of chunk c + 2 and have the elements of chunk c + 3 in flight; ONE LDS-only barrier per chunk.
// (Measured with the chain wave also preparing its share: 4 us of divisions, shuffles and load issue per chunk in front
// of every chain, 16.4 us per chunk instead of 12.)  Same coefficients, same order: bit-identical.
// cd: 2 * (64 * EL + 128) doubles of LDS (two buffers of terms), 16-byte aligned; aux: 8 doubles + 2 * BSZ ints;
// wm: nchunks * (BSZ / 64) doubles (the caller bounds len accordingly).
template <int EL, int BSZ, typename Get>
__device__ double norm2_flang_block_lanes_pipe(Get get, int len, double *cd, double *aux, double *wm)
{
    constexpr int CAP = 64 * EL, PADCAP = CAP + 128, E = CAP / BSZ, TPR = EL / E, nw = BSZ / 64;
    static_assert(CAP % BSZ == 0 && EL % E == 0 && E % 2 == 0, "chunk must split evenly");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool prep = wid < nw, chain = (wid == nw);                 // (wave-uniform; further waves only pass the barriers)
    const int nch = (len + CAP - 1) / CAP, i0 = tid * E;
    int *tflags = reinterpret_cast<int *>(aux + 8);
    auto loadabs = [&](int c, double (&a)[E]) __attribute__((always_inline)) {
        const int base = c * CAP, cl = min(CAP, len - base);
#pragma unroll
        for (int u = 0; u < E; ++u) a[u] = (i0 + u < cl) ? fabs(get(base + i0 + u)) : 0.0;
    };
    // the maximum of everything in this wave's part of chunk c that lies before this thread's elements; the wave's own
    // maximum goes to wm
    auto scanmax = [&](int c, const double (&a)[E]) __attribute__((always_inline)) {
        double lm = 0.0;
#pragma unroll
        for (int u = 0; u < E; ++u) lm = fmax(lm, a[u]);
        double sc = lm;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(sc, off, 64);
            if (lane >= off) sc = fmax(sc, o);
        }
        double ex = __shfl_up(sc, 1, 64);
        if (lane == 0) ex = 0.0;
        if (lane == 63) wm[c * nw + wid] = sc;
        return ex;
    };
    double a0[E], a1[E], ex0 = 0.0;
    __syncthreads();
    if (prep) {
        loadabs(0, a0);
        if (nch > 1) loadabs(1, a1);
        ex0 = scanmax(0, a0);
    }
    __syncthreads();
    double mxrun = 0.0, s = 0.0;
    // (s_setprio 3 for the chain wave: measured slower here, 268 against 253 us per 65536-row pivot step -- the preparing
    // wave that shares its SIMD falls behind; it helps k_qrx_pass_col, whose preparing waves have less to do)
    // (The chain wave one chunk BEHIND the barrier -- chunk c's terms requested from LDS while chunk c - 1 is added out of
    // a second register set -- was measured too: 264 against 237 us per step.  96 more live registers push part of the
    // terms into AGPRs, and every add of the chain then waits for a move.)
    for (int c = 0; c < nch; ++c) {
        const int cl = min(CAP, len - c * CAP);
        double *dsv = cd + (size_t)(c & 1) * PADCAP;
        int *tf = tflags + (c & 1) * BSZ;
        double mxc = mxrun;
#pragma unroll
        for (int w = 0; w < nw; ++w) mxc = fmax(mxc, wm[c * nw + w]);
        if (prep) {
            double prev = fmax(mxrun, ex0);
#pragma unroll
            for (int w = 0; w < nw; ++w)
                if (w < wid) prev = fmax(prev, wm[c * nw + w]);
            unsigned newmax = 0;                                     // see norm2_flang_block_lanes: a bit per new maximum
            // element i of the chunk lives at i + 2 * (i / EL), see norm2_flang_block_lanes
            double2 *ddst = reinterpret_cast<double2 *>(dsv + i0 + 2 * (i0 / EL));
#pragma unroll
            for (int u = 0; u < E; u += 2) {
                double dd[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double d = 0.0;
                    const double av = a0[u + h];
                    if (prev == 0.0) {
                        // mx was zero: element becomes the maximum, s untouched
                    } else if (av > prev) {
                        const double t = prev / av, tsq = t * t;
                        d = tsq;
                        if (tsq != 1.0) newmax |= 1u << (u + h);
                    } else if (av != 0.0) {
                        const double t = av / prev;
                        d = t * t;
                    }
                    dd[h] = d;
                    prev = fmax(prev, av);
                }
                ddst[u >> 1] = make_double2(dd[0], dd[1]);
            }
            tf[tid] = (int)newmax;
            // a chunk ahead: the maxima of chunk c + 1 (its elements arrived long ago), the loads of chunk c + 2
            if (c + 1 < nch) {
#pragma unroll
                for (int u = 0; u < E; ++u) a0[u] = a1[u];
                ex0 = scanmax(c + 1, a0);
            }
            if (c + 2 < nch) loadabs(c + 2, a1);
        }
        mxrun = mxc;
        // Publishes buffer c & 1 and wm[c + 1].  The preparing waves wait here for the chain of chunk c - 1 to end: the
        // buffer they fill next (c + 1) & 1 is the one that chain read.
        nlh_lds_barrier();
        if (chain) {
            const int nl = (cl + EL - 1) / EL;                       // runs in use
            double d[EL];
            const double2 *mine = reinterpret_cast<const double2 *>(dsv + lane * (EL + 2));
#pragma unroll
            for (int u = 0; u < EL / 2; ++u) { const double2 v2 = mine[u]; d[2 * u] = v2.x; d[2 * u + 1] = v2.y; }
            unsigned long long nm = 0;
#pragma unroll
            for (int k = 0; k < TPR; ++k) nm |= (unsigned long long)(unsigned)tf[lane * TPR + k] << (k * E);
            const unsigned long long mask = __ballot(nm == 0);
            double t = s;
            bool viaint = false;
            if (mask == ~0ull && nl == 64 && s >= 4.0 && s < 1.0e300) {
                // the usual chunk once the sum has left the first binades: no chain at all (ordered_possum_wave_int);
                // the chain below only if the sum crosses a binade inside the chunk (a few chunks per column)
                const double r = ordered_possum_wave_int<EL>(d, s, viaint);
                if (viaint) t = r;
            }
            if (viaint) {
                // (t holds the chunk's sum in every lane; the read-back below takes lane nl - 1)
            } else if (mask == ~0ull && nl == 64) {
                // the usual chunk -- full, no new maximum in it: nothing in the loop but the shift and the adds (the mask
                // test of the general loop below costs 25 ns a step, 1.6 us a chunk)
                // -- in straight-line trips of eight runs: a taken branch costs ~40 cycles; lane 0 keeps its value under
                // the shift, so the one before run 0 is idle
#pragma unroll 1
                for (int l = 0; l < 64; l += 8) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        t = nlh_wave_shr1(t);
#pragma unroll
                        for (int u = 0; u < EL; ++u) t = t + d[u];
                    }
                }
            } else
#pragma unroll 1
            for (int l = 0; l < nl; ++l) {
                if (l > 0) t = nlh_wave_shr1(t);
                if ((mask >> l) & 1ull) {
#pragma unroll
                    for (int u = 0; u < EL; ++u) t = t + d[u];
                } else {
#pragma unroll
                    for (int u = 0; u < EL; ++u) {
                        t = t * (((nm >> u) & 1ull) ? d[u] : 1.0);
                        t = t + d[u];
                    }
                }
            }
            const int lo = __builtin_amdgcn_readlane(__double2loint(t), nl - 1);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(t), nl - 1);
            s = __hiloint2double(hi, lo);
        }
    }
    if (chain && lane == 0) aux[0] = s;
    __syncthreads();
    const double r = mxrun * sqrt(1.0 + aux[0]);
    __syncthreads();
    return r;
}

// Same algorithm, one thread, for short vectors or per-column use.
template <typename Get>
__device__ __forceinline__ double norm2_flang_serial(Get get, int len)
{
    double mx = 0.0, s = 0.0;
    for (int i = 0; i < len; ++i) {
        const double a = fabs(get(i));
        if (mx == 0.0) {
            mx = a;
        } else if (a > mx) {
            const double t = mx / a, tsq = t * t;
            s = s * tsq;
            s = s + tsq;
            mx = a;
        } else if (a != 0.0) {
            const double t = a / mx;
            s = s + t * t;
        }
    }
    return mx * sqrt(1.0 + s);
}

// norm2_flang_serial over p[0], p[stride], ...: sixteen loads in flight, the next sixteen issued before the
// recurrence consumes the current ones (a column walk would otherwise pay one memory latency per element).
__device__ __forceinline__ double norm2_flang_serial_strided(const double *p, size_t stride, int len)
{
    double mx = 0.0, s = 0.0;
    auto step = [&](double v) {
        const double a = fabs(v);
        if (mx == 0.0) {
            mx = a;
        } else if (a > mx) {
            const double t = mx / a, tsq = t * t;
            s = s * tsq;
            s = s + tsq;
            mx = a;
        } else if (a != 0.0) {
            const double t = a / mx;
            s = s + t * t;
        }
    };
    double xa[16], xb[16];
    const int nb = len >> 4;
    auto load = [&](double (&x)[16], int g) {
        const double *q = p + (size_t)g * 16 * stride;
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = q[(size_t)u * stride];
    };
    if (nb > 0) load(xa, 0);
    for (int g = 0; g < nb; g += 2) {
        if (g + 1 < nb) load(xb, g + 1);
#pragma unroll
        for (int u = 0; u < 16; ++u) step(xa[u]);
        if (g + 2 < nb) load(xa, g + 2);
        if (g + 1 < nb) {
#pragma unroll
            for (int u = 0; u < 16; ++u) step(xb[u]);
        }
    }
    for (int i = nb << 4; i < len; ++i) step(p[(size_t)i * stride]);
    return mx * sqrt(1.0 + s);
}

// Norm of get(0..len-1): reference order when EXACT, tree sum of squares otherwise.
template <bool EXACT, typename Get>
__device__ __forceinline__ double nrm2_block(Get get, int len, double *red, double *scratch)
{
    if (EXACT) {
        return norm2_flang_block(get, len, scratch);
    } else {
        double sq = 0.0;
        for (int i = threadIdx.x; i < len; i += blockDim.x) { const double v = get(i); sq = sq + v * v; }
        return sqrt(block_reduce_sum(sq, red));
    }
}

// Left-to-right sum of buf[0..len) (LDS, 16-byte aligned) by the calling thread: the reads of a batch are
// issued together and one batch ahead, so only the adds themselves are serial.
__device__ __forceinline__ double ordered_sum_lds(const double *buf, int len, double s0 = 0.0)
{
    double s = s0, a[8], b[8];
    const double2 *src = reinterpret_cast<const double2 *>(buf);
    const int nb = len >> 3;
    auto load = [&](double (&d)[8], int g) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const double2 v2 = src[g * 4 + u]; d[2 * u] = v2.x; d[2 * u + 1] = v2.y; }
    };
    if (nb > 0) load(a, 0);
    for (int g = 0; g < nb; g += 2) {
        if (g + 1 < nb) load(b, g + 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) s = s + a[u];
        if (g + 2 < nb) load(a, g + 2);
        if (g + 1 < nb) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s = s + b[u];
        }
    }
    for (int i = nb << 3; i < len; ++i) s = s + buf[i];
    return s;
}

// Left-to-right sum s0 + t(0) + t(1) + ... + t(len-1) by ONE WAVE, "down the lanes" (see norm2_flang_block_lanes): the
// terms of a chunk of 64 * E are in registers, E consecutive ones per lane; at step l every lane takes its lower
// neighbour's running sum and adds its own E terms -- only lane l's result is the true partial sum at that step, the
// others' are never used.  The chain is register-to-register adds (~2.2 ns each) plus one DPP shift per E terms,
// against 6 - 10 ns per term for a single thread that reads its terms from LDS.  Slots past the end hold -0.0, the
// exact identity of IEEE addition (s + -0.0 == s for every s, signed zeros included).  get(i) must be callable by
// every lane for any i in [0, len); all 64 lanes of the wave must call; the result is returned in every lane.
template <int E, typename Get>
__device__ __forceinline__ double ordered_sum_wave(Get get, int len, double s0)
{
    const int lane = threadIdx.x & 63;
    double t = s0;
    for (int base = 0; base < len; base += 64 * E) {
        double d[E];
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const int i = base + lane * E + u;
            const double v = get(i < len ? i : len - 1);
            d[u] = i < len ? v : -0.0;
        }
        const int rem = len - base, nl = rem >= 64 * E ? 64 : (rem + E - 1) / E;
        // lane 0 starts from the carry (it keeps its value under the shift); the others' start values are overwritten
        // by the shift.  Eight runs and more go in straight-line trips of eight (a taken branch costs ~40 cycles on this
        // part): the lanes behind the last run hold -0.0 and hand the sum on unchanged, so the result is in the
        // last lane of the last trip.
        int last = nl - 1;
        if (nl >= 8) {
            const int ntrip = (nl + 7) >> 3;
            last = 8 * ntrip - 1;
#pragma unroll 1
            for (int g = 0; g < ntrip; ++g) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    t = nlh_wave_shr1(t);
#pragma unroll
                    for (int u = 0; u < E; ++u) t = t + d[u];
                }
            }
        } else {
#pragma unroll 1
            for (int l = 0; l < nl; ++l) {
                t = nlh_wave_shr1(t);
#pragma unroll
                for (int u = 0; u < E; ++u) t = t + d[u];
            }
        }
        t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), last), __builtin_amdgcn_readlane(__double2loint(t), last));
    }
    return t;
}

// The same for at most 64 * E terms that the lanes already hold: lane l has terms E l .. E l + E - 1 in d[] (slots past the
// end must hold -0.0).  All 64 lanes of the wave call; the sum s0 + d(0) + d(1) + ... is returned in every lane.
template <int E>
__device__ __forceinline__ double ordered_sum_wave_regs(const double (&d)[E], int len, double s0)
{
    double t = s0;
    const int nl = (len + E - 1) / E;
    int last = nl - 1;
    if (nl >= 8) {
        const int ntrip = (nl + 7) >> 3;
        last = 8 * ntrip - 1;
#pragma unroll 1
        for (int g = 0; g < ntrip; ++g) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                t = nlh_wave_shr1(t);
#pragma unroll
                for (int u = 0; u < E; ++u) t = t + d[u];
            }
        }
    } else {
#pragma unroll 1
        for (int l = 0; l < nl; ++l) {
            t = nlh_wave_shr1(t);
#pragma unroll
            for (int u = 0; u < E; ++u) t = t + d[u];
        }
    }
    if (nl <= 0) return s0;
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), last), __builtin_amdgcn_readlane(__double2loint(t), last));
}

// Sum of term(0..len-1): left-to-right by one thread when EXACT, tree otherwise.  Broadcast.
template <bool EXACT, typename Term>
__device__ __forceinline__ double sum_block(Term term, int len, double *red)
{
    if (EXACT) {
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int i = 0; i < len; ++i) s = s + term(i);
            red[0] = s;
        }
        __syncthreads();
        const double r = red[0];
        __syncthreads();
        return r;
    } else {
        double s = 0.0;
        for (int i = threadIdx.x; i < len; i += blockDim.x) s = s + term(i);
        return block_reduce_sum(s, red);
    }
}
