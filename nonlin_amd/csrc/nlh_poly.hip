// nlh_poly.hip -- polynomial%fit / fit_thru_zero (src/nonlin_polynomials.f90:146-238): Vandermonde panel, Householder QR
// with y as the extra column, back substitution; batched over independent data sets.
#include "nlh_internal.h"
#include "nlh_kernels_broyden.h"

void nlh_poly_init_device(int lds_max) { broyden_kernel_attrs(lds_max); }


// polynomial%fit / fit_thru_zero (src/nonlin_polynomials.f90:146-238) for nprob independent data sets of npts
// points each: Vandermonde panel, Householder QR with y as the extra column, back substitution.
// dx, dy: [nprob][npts] device; dcoef: [nprob][order + 1] device (c0 first; c0 = 0 for thru_zero).
int nlh_poly_fit_batch(nlh_handle *h, int32_t nprob, int32_t npts, int32_t order, int32_t thru_zero, const double *dx,
                       const double *dy, double *dcoef)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (nprob < 1) return 0;
    if (order >= npts || order < 1) return 4;                   // :163-166
    HIPCHK(h, hipSetDevice(h->device));
    const int ncols = thru_zero ? order : order + 1;
    int rc;
    if ((rc = ensure(h, h->W2, sizeof(double) * (size_t)nprob * npts * ncols))) return rc;
    if ((rc = ensure(h, h->qnV, sizeof(double) * ((size_t)nprob * (3 * (size_t)npts + 2 * ncols + 16))))) return rc;
    double *dA = (double *)h->W2.p, *dv = (double *)h->qnV.p;
    double *rhs = dv, *vbuf = dv + (size_t)nprob * npts, *wbuf = vbuf + (size_t)nprob * 2 * npts,
           *st = wbuf + (size_t)nprob * 2 * (ncols + 1);
    hipStream_t s = h->stream;
    hipLaunchKernelGGL(k_vandermonde, dim3((npts + 255) / 256, nprob), dim3(256), 0, s, npts, ncols, thru_zero, dx, dy, dA, rhs);
    hipLaunchKernelGGL(k_qn_col0, dim3((npts + 255) / 256, nprob), dim3(256), 0, s, npts, ncols, dA, vbuf);
    launch_house_steps(h, nprob, npts, ncols, 1, dA, rhs, vbuf, wbuf, st);
    hipLaunchKernelGGL(k_qn_solve_upper, dim3(nprob), dim3(64), sizeof(double) * ncols, s, ncols, dA, rhs,
                       (size_t)npts * ncols, (size_t)npts, (const LmState *)nullptr, -1);
    if (thru_zero) HIPCHK(h, hipMemsetAsync(dcoef, 0, sizeof(double) * (size_t)nprob * (order + 1), s));
    HIPCHK(h, hipMemcpy2DAsync(dcoef + (thru_zero ? 1 : 0), sizeof(double) * (order + 1), rhs, sizeof(double) * npts,
                               sizeof(double) * ncols, nprob, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipGetLastError());
    return 0;
}

// Host-array front end for one data set (what polynomial%fit marshals to).
int nlh_poly_fit(nlh_handle *h, int32_t npts, int32_t order, int32_t thru_zero, const double *x, const double *y, double *coef)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (order >= npts || order < 1) return 4;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * ((size_t)2 * npts + order + 1)))) return rc;
    double *dxv = (double *)h->xdev.p, *dyv = dxv + npts, *dc = dyv + npts;
    hipStream_t s = h->stream;
    HIPCHK(h, hipMemcpyAsync(dxv, x, sizeof(double) * npts, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(dyv, y, sizeof(double) * npts, hipMemcpyHostToDevice, s));
    if ((rc = nlh_poly_fit_batch(h, 1, npts, order, thru_zero, dxv, dyv, dc))) return rc;
    HIPCHK(h, hipMemcpyAsync(coef, dc, sizeof(double) * (order + 1), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return 0;
}
