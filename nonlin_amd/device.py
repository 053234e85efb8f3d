"""Device-model ("mode D") front end: batches of independent dense-quadratic problems
resident in HBM, solved by the batched entry points of include/nonlin_hip.h.

torch is plumbing only (device memory, streams); every computation is a call into
libnonlin_hip.so.  Layout: A [nprob, n, m] (each problem column-major m-by-n, i.e.
A[p, j, i] = A_p(i, j)), b/fvec [nprob, m], x [nprob, n], all float64 on the GPU.
"""
import ctypes as C

import torch

from . import _lib


def _chk(t, shape, name):
    if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and tuple(t.shape) == tuple(shape)):
        raise ValueError(f"{name}: expected contiguous float64 GPU tensor of shape {tuple(shape)}, "
                         f"got {t.dtype} {tuple(t.shape)} cuda={t.is_cuda}")


class DeviceSolver:
    """Owns an nlh handle bound to torch's current stream on `device`."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise _lib.NonlinHipUnavailable("no GPU visible: nonlin_amd has no CPU fallback")
        self.device = torch.device("cuda", device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self.h = _lib.Handle(device, stream)
        self.lib = self.h.lib

    def check(self, rc, what):
        return self.h.check(rc, what)

    def model(self, A, b, gamma):
        """A device residual model behind host arrays on this solver's device (HostModel)."""
        return HostModel(self, A, b, gamma)

    # -- inputs -------------------------------------------------------------
    def generate(self, nprob, m, n, seed0=12345, gamma=0.5, sigma=1e-3, spread=0.3, square_shift=False,
                 seed_stride=1):
        """SURVEY.md 8(d) generator, on the device; problem p uses seed0 + p*seed_stride.
        Returns (A, b, x_true, x0)."""
        dev = self.device
        A = torch.empty((nprob, n, m), dtype=torch.float64, device=dev)
        b = torch.empty((nprob, m), dtype=torch.float64, device=dev)
        xt = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        x0 = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        rc = self.lib.nlh_dq_generate(self.h.ptr, nprob, m, n, seed0, seed_stride, gamma, sigma, spread, int(square_shift),
                                      A.data_ptr(), b.data_ptr(), xt.data_ptr(), x0.data_ptr())
        self.h.check(rc, "nlh_dq_generate")
        return A, b, xt, x0

    # -- solvers ------------------------------------------------------------
    def options(self, **kw):
        o = _lib.default_options()
        for k, v in kw.items():
            if k == "factor":      # lss_set_factor clamp, src/nonlin_least_squares.f90:108-114
                v = min(max(float(v), 0.1), 100.0)
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        return o

    def lm_solve_batch(self, A, b, gamma, x, opts=None):
        """least_squares_solver%solve for every problem.  x is updated in place.
        Returns (fvec, ib_list, status_list)."""
        nprob, n, m = A.shape
        _chk(A, (nprob, n, m), "A"); _chk(b, (nprob, m), "b"); _chk(x, (nprob, n), "x")
        fvec = torch.empty((nprob, m), dtype=torch.float64, device=A.device)
        ib = (_lib.IterationBehavior * nprob)()
        status = (C.c_int32 * nprob)()
        o = opts or self.options()
        rc = self.lib.nlh_dq_lm_solve_batch(self.h.ptr, C.byref(o), nprob, m, n, A.data_ptr(), b.data_ptr(),
                                            float(gamma), x.data_ptr(), fvec.data_ptr(), ib, status)
        self.h.check(rc, "nlh_dq_lm_solve_batch")
        if rc:
            raise RuntimeError(f"nlh_dq_lm_solve_batch returned {rc}")
        return fvec, [ib[k].as_dict() for k in range(nprob)], [int(status[k]) for k in range(nprob)]

    def newton_solve_batch(self, A, b, gamma, x, analytic=True, opts=None):
        """newton_solver%solve for every (square) problem.  x is updated in place."""
        nprob, n, m = A.shape
        assert m == n
        _chk(A, (nprob, n, n), "A"); _chk(b, (nprob, n), "b"); _chk(x, (nprob, n), "x")
        fvec = torch.empty((nprob, n), dtype=torch.float64, device=A.device)
        ib = (_lib.IterationBehavior * nprob)()
        status = (C.c_int32 * nprob)()
        o = opts or self.options()
        rc = self.lib.nlh_dq_newton_solve_batch(self.h.ptr, C.byref(o), nprob, n, A.data_ptr(), b.data_ptr(),
                                                float(gamma), int(analytic), x.data_ptr(), fvec.data_ptr(),
                                                ib, status)
        self.h.check(rc, "nlh_dq_newton_solve_batch")
        if rc:
            raise RuntimeError(f"nlh_dq_newton_solve_batch returned {rc}")
        return fvec, [ib[k].as_dict() for k in range(nprob)], [int(status[k]) for k in range(nprob)]

    def quasi_newton_solve_batch(self, A, b, gamma, x, analytic=True, opts=None, jdelta=5):
        """quasi_newton_solver%solve for every (square) problem.  x is updated in place."""
        nprob, n, m = A.shape
        assert m == n
        _chk(A, (nprob, n, n), "A"); _chk(b, (nprob, n), "b"); _chk(x, (nprob, n), "x")
        fvec = torch.empty((nprob, n), dtype=torch.float64, device=A.device)
        ib = (_lib.IterationBehavior * nprob)()
        status = (C.c_int32 * nprob)()
        o = opts or self.options()
        rc = self.lib.nlh_dq_quasi_newton_solve_batch(self.h.ptr, C.byref(o), int(jdelta), nprob, n, A.data_ptr(),
                                                      b.data_ptr(), float(gamma), int(analytic), x.data_ptr(),
                                                      fvec.data_ptr(), ib, status)
        self.h.check(rc, "nlh_dq_quasi_newton_solve_batch")
        if rc:
            raise RuntimeError(f"nlh_dq_quasi_newton_solve_batch returned {rc}")
        return fvec, [ib[k].as_dict() for k in range(nprob)], [int(status[k]) for k in range(nprob)]

    def cls_solve_batch(self, A, b, gamma, x, opts=None, lower=None, upper=None, delta=1.0, stepscale=1.0):
        """constrained_least_squares_solver%solve for every problem (the same bounds for all).  x in place."""
        import numpy as np
        nprob, n, m = A.shape
        _chk(A, (nprob, n, m), "A"); _chk(b, (nprob, m), "b"); _chk(x, (nprob, n), "x")
        fvec = torch.empty((nprob, m), dtype=torch.float64, device=A.device)
        ib = (_lib.IterationBehavior * nprob)()
        status = (C.c_int32 * nprob)()
        o = opts or self.options()
        lo = None if lower is None else np.ascontiguousarray(lower, dtype=np.float64)
        hi = None if upper is None else np.ascontiguousarray(upper, dtype=np.float64)
        plo = None if lo is None else lo.ctypes.data_as(_lib.c_double_p)
        phi = None if hi is None else hi.ctypes.data_as(_lib.c_double_p)
        rc = self.lib.nlh_dq_cls_solve_batch(self.h.ptr, C.byref(o), float(delta), float(stepscale), plo, phi, nprob, m, n,
                                             A.data_ptr(), b.data_ptr(), float(gamma), x.data_ptr(), fvec.data_ptr(),
                                             ib, status)
        self.h.check(rc, "nlh_dq_cls_solve_batch")
        if rc:
            raise RuntimeError(f"nlh_dq_cls_solve_batch returned {rc}")
        return fvec, [ib[k].as_dict() for k in range(nprob)], [int(status[k]) for k in range(nprob)]

    def bfgs_solve_batch(self, A, b, gamma, x, opts=None):
        """bfgs%solve on f(x) = 0.5 ||r(x)||^2 of every problem (FD gradient on the device).  x in place.
        Returns (fout list, ib list, status list)."""
        nprob, n, m = A.shape
        _chk(A, (nprob, n, m), "A"); _chk(b, (nprob, m), "b"); _chk(x, (nprob, n), "x")
        ib = (_lib.IterationBehavior * nprob)()
        status = (C.c_int32 * nprob)()
        fout = (C.c_double * nprob)()
        o = opts or self.options(max_evals=500)
        rc = self.lib.nlh_dq_bfgs_solve_batch(self.h.ptr, C.byref(o), nprob, m, n, A.data_ptr(), b.data_ptr(), float(gamma),
                                              x.data_ptr(), fout, ib, status)
        self.h.check(rc, "nlh_dq_bfgs_solve_batch")
        if rc:
            raise RuntimeError(f"nlh_dq_bfgs_solve_batch returned {rc}")
        return [float(v) for v in fout], [ib[k].as_dict() for k in range(nprob)], [int(status[k]) for k in range(nprob)]

    # -- user-supplied device residuals (launchers) ------------------------------
    @staticmethod
    def _devfcn(f):
        """A launcher as ctypes sees it: a symbol of a user's shared object, an address, or None."""
        if f is None:
            return C.cast(None, _lib.DEVFCN)
        return f if isinstance(f, _lib.DEVFCN) else C.cast(f, _lib.DEVFCN)

    def dq_launchers(self, A, b, gamma):
        """The built-in dense-quadratic family expressed through the open path: (fcn, jac, ctx) for the *_device entry
        points.  Keep the returned ctx (and A, b) alive while solving."""
        ctx = _lib.DqDeviceCtx(A.data_ptr(), b.data_ptr(), float(gamma))
        return (C.cast(self.lib.nlh_dq_device_fcn, _lib.DEVFCN), C.cast(self.lib.nlh_dq_device_jac, _lib.DEVFCN), ctx)

    def _ctxp(self, ctx):
        return ctx if isinstance(ctx, (int, C.c_void_p)) or ctx is None else C.cast(C.byref(ctx), C.c_void_p)

    def lm_solve_batch_device(self, fcn, ctx, m, x, jac=None, opts=None):
        """least_squares_solver%solve on x.shape[0] problems of a USER'S device residual (launcher fcn, context ctx).
        x [nprob, n] is updated in place.  Returns (fvec, ib_list, status_list)."""
        nprob, n = x.shape
        _chk(x, (nprob, n), "x")
        fvec = torch.empty((nprob, m), dtype=torch.float64, device=x.device)
        ib = (_lib.IterationBehavior * nprob)()
        status = (C.c_int32 * nprob)()
        o = opts or self.options()
        rc = self.lib.nlh_lm_solve_batch_device(self.h.ptr, C.byref(o), nprob, m, n, self._devfcn(fcn), self._devfcn(jac),
                                                self._ctxp(ctx), x.data_ptr(), fvec.data_ptr(), ib, status)
        self.h.check(rc, "nlh_lm_solve_batch_device")
        if rc:
            raise RuntimeError(f"nlh_lm_solve_batch_device returned {rc}")
        return fvec, [ib[k].as_dict() for k in range(nprob)], [int(status[k]) for k in range(nprob)]

    def cls_solve_batch_device(self, fcn, ctx, m, x, jac=None, opts=None, lower=None, upper=None, delta=1.0, stepscale=1.0):
        """constrained_least_squares_solver%solve on x.shape[0] problems of a user's device residual (the same box for all)."""
        import numpy as np
        nprob, n = x.shape
        _chk(x, (nprob, n), "x")
        fvec = torch.empty((nprob, m), dtype=torch.float64, device=x.device)
        ib = (_lib.IterationBehavior * nprob)()
        status = (C.c_int32 * nprob)()
        o = opts or self.options()
        lo = None if lower is None else np.ascontiguousarray(lower, dtype=np.float64)
        hi = None if upper is None else np.ascontiguousarray(upper, dtype=np.float64)
        plo = None if lo is None else lo.ctypes.data_as(_lib.c_double_p)
        phi = None if hi is None else hi.ctypes.data_as(_lib.c_double_p)
        rc = self.lib.nlh_cls_solve_batch_device(self.h.ptr, C.byref(o), float(delta), float(stepscale), plo, phi, nprob, m, n,
                                                 self._devfcn(fcn), self._devfcn(jac), self._ctxp(ctx), x.data_ptr(), fvec.data_ptr(),
                                                 ib, status)
        self.h.check(rc, "nlh_cls_solve_batch_device")
        if rc:
            raise RuntimeError(f"nlh_cls_solve_batch_device returned {rc}")
        return fvec, [ib[k].as_dict() for k in range(nprob)], [int(status[k]) for k in range(nprob)]

    def square_solve_batch_device(self, fcn, ctx, x, jac=None, opts=None, broyden=False, jdelta=5):
        """newton_solver%solve (or quasi_newton_solver%solve) on x.shape[0] square problems of a user's device residual."""
        nprob, n = x.shape
        _chk(x, (nprob, n), "x")
        fvec = torch.empty((nprob, n), dtype=torch.float64, device=x.device)
        ib = (_lib.IterationBehavior * nprob)()
        status = (C.c_int32 * nprob)()
        o = opts or self.options()
        if broyden:
            name = "nlh_quasi_newton_solve_batch_device"
            rc = self.lib.nlh_quasi_newton_solve_batch_device(self.h.ptr, C.byref(o), int(jdelta), nprob, n, self._devfcn(fcn),
                                                              self._devfcn(jac), self._ctxp(ctx), x.data_ptr(), fvec.data_ptr(),
                                                              ib, status)
        else:
            name = "nlh_newton_solve_batch_device"
            rc = self.lib.nlh_newton_solve_batch_device(self.h.ptr, C.byref(o), nprob, n, self._devfcn(fcn), self._devfcn(jac),
                                                        self._ctxp(ctx), x.data_ptr(), fvec.data_ptr(), ib, status)
        self.h.check(rc, name)
        if rc:
            raise RuntimeError(f"{name} returned {rc}")
        return fvec, [ib[k].as_dict() for k in range(nprob)], [int(status[k]) for k in range(nprob)]

    def bfgs_solve_batch_device(self, fcn, ctx, x, grad=None, opts=None):
        """bfgs%solve for every problem with the USER'S device fcnnvar: fcn a launcher called with m = 1 (dF[npoints] = f),
        grad (optional) one that fills dJ[npoints][n] with the gradients.  x [nprob][n] device tensor, in place.
        Returns (fout list, ib list, status list)."""
        nprob, n = x.shape
        _chk(x, (nprob, n), "x")
        ib = (_lib.IterationBehavior * nprob)()
        status = (C.c_int32 * nprob)()
        fout = (C.c_double * nprob)()
        o = opts or self.options(max_evals=500)
        rc = self.lib.nlh_bfgs_solve_batch_device(self.h.ptr, C.byref(o), nprob, n, self._devfcn(fcn), self._devfcn(grad), self._ctxp(ctx),
                                                  x.data_ptr(), fout, ib, status)
        self.h.check(rc, "nlh_bfgs_solve_batch_device")
        if rc:
            raise RuntimeError(f"nlh_bfgs_solve_batch_device returned {rc}")
        return [float(v) for v in fout], [ib[k].as_dict() for k in range(nprob)], [int(status[k]) for k in range(nprob)]

    def fd_jacobian_device(self, fcn, ctx, m, x, fv=None, jac=None):
        """vecfcn_helper%jacobian of every problem of a user's device residual: J [nprob, n, m]."""
        nprob, n = x.shape
        _chk(x, (nprob, n), "x")
        J = torch.empty((nprob, n, m), dtype=torch.float64, device=x.device)
        rc = self.lib.nlh_fd_jacobian_device(self.h.ptr, nprob, m, n, self._devfcn(fcn), self._devfcn(jac), self._ctxp(ctx),
                                             x.data_ptr(), fv.data_ptr() if fv is not None else None, J.data_ptr())
        self.h.check(rc, "nlh_fd_jacobian_device")
        if rc:
            raise RuntimeError(f"nlh_fd_jacobian_device returned {rc}")
        return J

    # -- stage-level kernels (parity tests, roofline) --------------------------
    def residual(self, A, b, gamma, x):
        nprob, n, m = A.shape
        f = torch.empty((nprob, m), dtype=torch.float64, device=A.device)
        self.h.check(self.lib.nlh_dq_residual(self.h.ptr, nprob, m, n, A.data_ptr(), b.data_ptr(), float(gamma),
                                              x.data_ptr(), f.data_ptr()), "nlh_dq_residual")
        return f

    def fd_panel(self, A, b, gamma, x):
        nprob, n, m = A.shape
        P = torch.empty((nprob, n, m), dtype=torch.float64, device=A.device)
        self.h.check(self.lib.nlh_dq_fd_panel(self.h.ptr, nprob, m, n, A.data_ptr(), b.data_ptr(), float(gamma),
                                              x.data_ptr(), P.data_ptr()), "nlh_dq_fd_panel")
        return P

    def fd_jacobian_panel(self, P, f0, x, out=None):
        nprob, n, m = P.shape
        J = out if out is not None else torch.empty_like(P)
        self.h.check(self.lib.nlh_fd_jacobian_panel(self.h.ptr, nprob, m, n, P.data_ptr(), f0.data_ptr(),
                                                    x.data_ptr(), J.data_ptr()), "nlh_fd_jacobian_panel")
        return J

    def jacobian(self, A, gamma, x):
        nprob, n, m = A.shape
        J = torch.empty_like(A)
        self.h.check(self.lib.nlh_dq_jacobian(self.h.ptr, nprob, m, n, A.data_ptr(), float(gamma), x.data_ptr(),
                                              J.data_ptr()), "nlh_dq_jacobian")
        return J

    def gram(self, J, f):
        nprob, n, m = J.shape
        G = torch.empty((nprob, n, n), dtype=torch.float64, device=J.device)
        g = torch.empty((nprob, n), dtype=torch.float64, device=J.device)
        self.h.check(self.lib.nlh_gram(self.h.ptr, nprob, m, n, J.data_ptr(), f.data_ptr(), G.data_ptr(),
                                       g.data_ptr()), "nlh_gram")
        return G, g

    def chol_factor(self, G, g):
        """Overwrites G's upper triangle (G[p, c, r], r <= c) with R.  Returns (ipvt0, acnorm, qtf, info)."""
        nprob, n, _ = G.shape
        dev = G.device
        ipvt = torch.empty((nprob, n), dtype=torch.int32, device=dev)
        acnorm = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        qtf = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        info = torch.empty((nprob,), dtype=torch.int32, device=dev)
        self.h.check(self.lib.nlh_chol_factor(self.h.ptr, nprob, n, G.data_ptr(), g.data_ptr(), ipvt.data_ptr(),
                                              acnorm.data_ptr(), qtf.data_ptr(), info.data_ptr()), "nlh_chol_factor")
        return ipvt, acnorm, qtf, info

    def qr_factor(self, J, f):
        """lmfactor + Q^T f.  J [nprob, n, m] is overwritten.  Returns (ipvt0, rdiag, acnorm, qtf, wa4)."""
        nprob, n, m = J.shape
        dev = J.device
        ipvt = torch.empty((nprob, n), dtype=torch.int32, device=dev)
        rdiag = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        acnorm = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        qtf = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        wa4 = torch.empty((nprob, m), dtype=torch.float64, device=dev)
        self.h.check(self.lib.nlh_qr_factor(self.h.ptr, nprob, m, n, J.data_ptr(), f.data_ptr(), ipvt.data_ptr(),
                                            rdiag.data_ptr(), acnorm.data_ptr(), qtf.data_ptr(), wa4.data_ptr()),
                     "nlh_qr_factor")
        return ipvt, rdiag, acnorm, qtf, wa4

    def lmfactor_exact(self, J, f):
        """lmfactor + Q^T f in the reference's operation order (the exact LM policy's factorisation, bit-identical to
        the CPU path).  J [nprob, n, m] (column-major problems, not modified), f [nprob, m], m >= n.
        Returns (R [nprob, n, n] column-major: R[p].T is R with rdiag on the diagonal, ipvt0, rdiag, acnorm, qtf, wa4)."""
        nprob, n, m = J.shape
        _chk(J, (nprob, n, m), "J"); _chk(f, (nprob, m), "f")
        dev = J.device
        R = torch.zeros((nprob, n, n), dtype=torch.float64, device=dev)
        ipvt = torch.empty((nprob, n), dtype=torch.int32, device=dev)
        rdiag = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        acnorm = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        qtf = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        wa4 = torch.empty((nprob, m), dtype=torch.float64, device=dev)
        self.h.check(self.lib.nlh_lmfactor_exact(self.h.ptr, nprob, m, n, J.data_ptr(), f.data_ptr(), R.data_ptr(),
                                                 ipvt.data_ptr(), rdiag.data_ptr(), acnorm.data_ptr(), qtf.data_ptr(),
                                                 wa4.data_ptr()), "nlh_lmfactor_exact")
        return R, ipvt, rdiag, acnorm, qtf, wa4

    def lmpar(self, R, ipvt, diag, qtf, delta, tailsq, par):
        """R [nprob, n, ldr] column-major n-by-n blocks with leading dimension ldr."""
        nprob, n, ldr = R.shape
        dev = R.device
        for t, nm in ((R, "R"), (diag, "diag"), (qtf, "qtf"), (delta, "delta"), (tailsq, "tailsq"), (par, "par")):
            if t.dtype != torch.float64 or not t.is_cuda:
                raise ValueError(f"{nm} must be a float64 GPU tensor")
        x = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        sdiag = torch.empty((nprob, n), dtype=torch.float64, device=dev)
        par = par.clone()
        self.h.check(self.lib.nlh_lmpar(self.h.ptr, nprob, n, R.data_ptr(), ldr, ipvt.data_ptr(), diag.data_ptr(),
                                        qtf.data_ptr(), delta.data_ptr(), tailsq.data_ptr(), par.data_ptr(),
                                        x.data_ptr(), sdiag.data_ptr()), "nlh_lmpar")
        return par, x, sdiag

    def lu_factor(self, A):
        nprob, n, _ = A.shape
        ipvt = torch.empty((nprob, n), dtype=torch.int32, device=A.device)
        info = torch.empty((nprob,), dtype=torch.int32, device=A.device)
        self.h.check(self.lib.nlh_lu_factor(self.h.ptr, nprob, n, A.data_ptr(), ipvt.data_ptr(), info.data_ptr()),
                     "nlh_lu_factor")
        return ipvt, info

    def qr_factor_full(self, B):
        """Householder QR with Q formed.  B: [nprob, n, n] column-major problems (B[p].T is the matrix).
        Returns (Q column-major like B, Rt = R stored row-major, i.e. Rt[p] IS R as a torch matrix)."""
        nprob, n, _ = B.shape
        _chk(B, (nprob, n, n), "B")
        Q = torch.empty_like(B)
        Rt = torch.empty_like(B)
        self.h.check(self.lib.nlh_qr_factor_full(self.h.ptr, nprob, n, B.data_ptr(), Q.data_ptr(), Rt.data_ptr()),
                     "nlh_qr_factor_full")
        return Q, Rt

    def qr_rank1_update(self, Q, Rt, u, v):
        """In place: Q1 R1 = Q R + u v^T."""
        nprob, n, _ = Q.shape
        _chk(Q, (nprob, n, n), "Q"); _chk(Rt, (nprob, n, n), "Rt"); _chk(u, (nprob, n), "u"); _chk(v, (nprob, n), "v")
        self.h.check(self.lib.nlh_qr_rank1_update(self.h.ptr, nprob, n, Q.data_ptr(), Rt.data_ptr(), u.data_ptr(),
                                                  v.data_ptr()), "nlh_qr_rank1_update")
        return Q, Rt

    def solve_upper(self, Rt, x):
        nprob, n, _ = Rt.shape
        _chk(Rt, (nprob, n, n), "Rt"); _chk(x, (nprob, n), "x")
        self.h.check(self.lib.nlh_solve_upper(self.h.ptr, nprob, n, Rt.data_ptr(), x.data_ptr()), "nlh_solve_upper")
        return x

    def poly_fit_batch(self, x, y, order, thru_zero=False):
        """polynomial%fit for every row of x / y ([nprob, npts]).  Returns coefficients [nprob, order + 1]."""
        nprob, npts = x.shape
        _chk(x, (nprob, npts), "x"); _chk(y, (nprob, npts), "y")
        coef = torch.empty((nprob, order + 1), dtype=torch.float64, device=x.device)
        rc = self.lib.nlh_poly_fit_batch(self.h.ptr, nprob, npts, int(order), int(thru_zero), x.data_ptr(), y.data_ptr(),
                                         coef.data_ptr())
        self.h.check(rc, "nlh_poly_fit_batch")
        if rc:
            raise RuntimeError(f"nlh_poly_fit_batch returned {rc}")
        return coef

    def chol_rank1(self, Rt, u, downdate=False):
        """In place on the row-major upper Cholesky factor Rt (n x n): R1^T R1 = R^T R +- u u^T.  Returns info."""
        n = Rt.shape[0]
        _chk(Rt, (n, n), "Rt"); _chk(u, (n,), "u")
        info = C.c_int32(0)
        self.h.check(self.lib.nlh_chol_rank1(self.h.ptr, n, int(downdate), Rt.data_ptr(), u.data_ptr(), C.byref(info)),
                     "nlh_chol_rank1")
        return int(info.value)

    def lu_solve(self, LU, ipvt, b):
        nprob, n, _ = LU.shape
        self.h.check(self.lib.nlh_lu_solve(self.h.ptr, nprob, n, LU.data_ptr(), ipvt.data_ptr(), b.data_ptr()),
                     "nlh_lu_solve")
        return b


class HostModel:
    """A device residual model behind HOST arrays (nlh_dq_model): the boundary object the Fortran shim's
    `device_model_batch` wraps.  A [nprob, n, m] (each problem column-major m x n), b [nprob, m] are numpy arrays;
    the copies live on one device (`owner` a DeviceSolver) or are dealt over the devices of a DeviceSet."""

    def __init__(self, owner, A, b, gamma):
        import numpy as np
        self.owner = owner
        self.lib = owner.lib
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        self.nprob, self.n, self.m = A.shape
        if b.shape != (self.nprob, self.m):
            raise ValueError("b must be [nprob, m]")
        self._md = C.c_void_p()
        dp = lambda a: a.ctypes.data_as(_lib.c_double_p)
        if isinstance(owner, DeviceSet):
            rc = self.lib.nlh_dq_model_create_on(owner.ptr, self.nprob, self.m, self.n, dp(A), dp(b), float(gamma), C.byref(self._md))
        else:
            rc = self.lib.nlh_dq_model_create(owner.h.ptr, self.nprob, self.m, self.n, dp(A), dp(b), float(gamma), C.byref(self._md))
        owner.check(rc, "nlh_dq_model_create")
        if rc != 0:
            raise RuntimeError(f"nlh_dq_model_create: {rc}")

    @property
    def shares(self):
        return int(self.lib.nlh_dq_model_device_count(self._md))

    def _solve(self, fn, x, opts, *extra):
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        f = np.empty((self.nprob, self.m))
        ib = (_lib.IterationBehavior * self.nprob)()
        st = (C.c_int32 * self.nprob)()
        dp = lambda a: a.ctypes.data_as(_lib.c_double_p)
        hptr = None if isinstance(self.owner, DeviceSet) else self.owner.h.ptr
        rc = fn(hptr, C.byref(opts if opts is not None else _lib.default_options()), self._md, *extra, dp(x), dp(f), ib, st)
        self.owner.check(rc, fn.__name__)
        if rc != 0:
            raise RuntimeError(f"{fn.__name__}: {rc}")
        return x, f, [ib[p].as_dict() for p in range(self.nprob)], [int(st[p]) for p in range(self.nprob)]

    def lm_solve(self, x, opts=None):
        """least_squares_solver%solve on every problem: returns (x, fvec, iteration behaviours, status codes)."""
        return self._solve(self.lib.nlh_dq_model_lm_solve, x, opts)

    def newton_solve(self, x, analytic=True, opts=None):
        return self._solve(self.lib.nlh_dq_model_newton_solve, x, opts, 1 if analytic else 0)

    def quasi_newton_solve(self, x, analytic=True, jdelta=5, opts=None):
        """quasi_newton_solver%solve on every (square) problem; jdelta: iterations between fresh Jacobians."""
        return self._solve(self.lib.nlh_dq_model_quasi_newton_solve, x, opts, int(jdelta), 1 if analytic else 0)

    def cls_solve(self, x, lower=None, upper=None, delta=1.0, stepscale=1.0, opts=None):
        """constrained_least_squares_solver%solve on every problem inside the box [lower, upper] (n entries each, or None)."""
        import numpy as np
        lo = None if lower is None else np.ascontiguousarray(lower, dtype=np.float64)
        hi = None if upper is None else np.ascontiguousarray(upper, dtype=np.float64)
        for v in (lo, hi):
            if v is not None and v.shape != (self.n,):
                raise ValueError("bounds must have n entries")
        dp = lambda a: None if a is None else a.ctypes.data_as(_lib.c_double_p)
        return self._solve(self.lib.nlh_dq_model_cls_solve, x, opts, float(delta), float(stepscale), dp(lo), dp(hi))

    def bfgs_solve(self, x, opts=None):
        """bfgs%solve on 0.5 ||F(x)||^2 of every problem: returns (x, F(x), objective values, behaviours, status codes)."""
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        f = np.empty((self.nprob, self.m))
        fo = np.empty(self.nprob)
        ib = (_lib.IterationBehavior * self.nprob)()
        st = (C.c_int32 * self.nprob)()
        dp = lambda a: a.ctypes.data_as(_lib.c_double_p)
        hptr = None if isinstance(self.owner, DeviceSet) else self.owner.h.ptr
        rc = self.lib.nlh_dq_model_bfgs_solve(hptr, C.byref(opts if opts is not None else _lib.default_options()), self._md,
                                              dp(x), dp(f), dp(fo), ib, st)
        self.owner.check(rc, "nlh_dq_model_bfgs_solve")
        if rc != 0:
            raise RuntimeError(f"nlh_dq_model_bfgs_solve: {rc}")
        return x, f, fo, [ib[p].as_dict() for p in range(self.nprob)], [int(st[p]) for p in range(self.nprob)]

    def close(self):
        if getattr(self, "_md", None) is not None and self._md.value:
            self.lib.nlh_dq_model_destroy(self._md)
            self._md = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceSet:
    """nlh_device_set: one handle per listed device inside ONE process; models created on it are dealt over the devices
    block-cyclically and solved by one host thread per device (no collective).  devices=None: every visible device;
    an id may repeat (two shares on one GPU)."""

    def __init__(self, devices=None):
        self.lib = _lib.load()
        if self.lib.nlh_device_count() <= 0:
            raise _lib.NonlinHipUnavailable("no HIP device visible: the nonlin_amd compute path needs a GPU "
                                          "(there is no CPU fallback)")
        self._s = C.c_void_p()
        if devices is None:
            rc = self.lib.nlh_device_set_create(C.byref(self._s), None, 0)
        else:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            rc = self.lib.nlh_device_set_create(C.byref(self._s), arr, len(devices))
        if rc != 0:
            raise _lib.NonlinHipUnavailable(f"nlh_device_set_create failed with {rc}")

    @property
    def ptr(self):
        return self._s

    def __len__(self):
        return int(self.lib.nlh_device_set_size(self._s))

    def check(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what}: library error {rc}: {self.lib.nlh_device_set_last_error(self._s).decode()}")
        return rc

    def model(self, A, b, gamma):
        return HostModel(self, A, b, gamma)

    def close(self):
        if getattr(self, "_s", None) is not None and self._s.value:
            self.lib.nlh_device_set_destroy(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
