// nlh_qrx.h -- host interface of the streaming exact lmfactor (nlh_qrx.hip): lmfactor + Q^T f of
// src/nonlin_least_squares.f90:569-667 / :241-253 in the reference's operation order, for a
// whole batch in lock step.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "nlh_lm_head.h"

// Optional HIP-event brackets around the launches (bench.py's live roofline):
// which = 0 pivot / reflector kernel, 1 trailing pass, 2 everything else.
struct QrxTimer {
    void *ctx;
    void (*begin)(void *ctx, int which, hipStream_t s);
    void (*end)(void *ctx, int which, hipStream_t s);
};

// Row stride of the row-major working matrix (n columns + the residual, padded to 64 bytes).
void qrx_init_device();          // once per device a handle is created on (kernel attributes)
int qrx_ld(int n);
// Doubles between the working matrices of two consecutive problems.
size_t qrx_matrix_stride(int m, int n);
// Doubles the row-major working matrix T needs for nprob problems (incl. read-ahead padding).
size_t qrx_matrix_doubles(int nprob, int m, int n);
// Bytes of private workspace (reflector banks, pending multipliers, column map, norms, step records).
size_t qrx_workspace_bytes(int nprob, int m, int n);

// Factor every problem whose stage is ST_NEED_QR (st == nullptr: all).  J: column-major m x n per problem, or nullptr when
// the caller has already written the Jacobian into T in the working layout (qrx_ld / qrx_matrix_stride; k_dq_panel does);
// T: row-major scratch (qrx_matrix_doubles); outputs: R (n x n column-major, upper + diagonal),
// v.ipvt / acnorm / qtf / rdiag, wa4 = Q^T f, and -- when st != nullptr -- the outer-loop head (lm_head) with
// stage -> ST_QR_READY / ST_DONE.  nact: how many of the nprob problems are expected to be at ST_NEED_QR (<= 0: all) --
// only picks the kernel variant for nearly empty launches, never the result.
void qrx_factor(hipStream_t stream, int nprob, int m, int n, const double *J, double *T, const double *fvec,
                double *R, LmVecs v, double *wa4, double *scratch, const double *x, LmState *st, double factor,
                double gtol, void *ws, const QrxTimer *tm, int nact);
