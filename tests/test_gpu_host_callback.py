"""The literal drop-in case of BASELINE config 2: least_squares_solver through nlh_lm_solve (C ABI) with a COMPILED host
callback (tests/host_callback/dq_callback.c) -- the user's function stays on the host and is called n + 1 times per
Jacobian in the reference's order (src/nonlin_multi_eqn_mult_var.f90:198-277), everything else runs on the GPU.  The
CPU oracle drives the very same callback; x, fvec and the counts must be the same bits."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "host_callback", "libdq_callback.so")


class Ctx(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("A", C.POINTER(C.c_double)), ("b", C.POINTER(C.c_double)),
                ("gamma", C.c_double), ("ncalls", C.c_int64), ("u", C.POINTER(C.c_double))]


def _cb():
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-C", os.path.dirname(SO), "-s"])
    return C.CDLL(SO)


def test_compiled_callback_is_the_oracle_residual(oracle):
    """(CPU) the user's C function computes the SURVEY 8(d) residual with the same roundings as the oracle's own."""
    cb = _cb()
    dp = C.POINTER(C.c_double)
    A, b, xt, x0 = oracle.dq_generate(7, 96, 12, gamma=2.0, sigma=0.1, spread=1.0)
    u, f = np.zeros(96), np.zeros(96)
    ctx = Ctx(96, 12, A.ctypes.data_as(dp), b.ctypes.data_as(dp), 2.0, 0, u.ctypes.data_as(dp))
    cb.dq_user_fcn(C.byref(ctx), 12, x0.ctypes.data_as(dp), 96, f.ctypes.data_as(dp))
    assert ctx.ncalls == 1
    assert np.array_equal(f, oracle.dq_residual(A, b, 2.0, x0))


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,kw,factor", [(512, 64, {}, 100.0), (300, 37, dict(gamma=2.0, sigma=0.1, spread=5.0), 0.1),
                                           (2100, 40, {}, 100.0), (4096, 256, {}, 100.0)])   # the last: BASELINE config 2, full size
def test_lm_solve_with_compiled_host_callback_bitwise(ds, oracle, m, n, kw, factor):
    from nonlin_amd import _lib
    cb = _cb()
    dp = C.POINTER(C.c_double)
    g = kw.get("gamma", 0.5)
    A, b, xt, x0 = oracle.dq_generate(12345, m, n, **kw)
    u = np.zeros(m)
    ctx = Ctx(m, n, A.ctypes.data_as(dp), b.ctypes.data_as(dp), g, 0, u.ctypes.data_as(dp))
    og = _lib.default_options()
    og.max_evals = 500
    og.factor = factor
    x, f, ib = x0.copy(), np.zeros(m), _lib.IterationBehavior()
    rc = ds.lib.nlh_lm_solve(ds.h.ptr, C.byref(og), m, n, C.cast(cb.dq_user_fcn, _lib.VECFCN), C.cast(None, _lib.JACFCN),
                             C.byref(ctx), x.ctypes.data_as(dp), f.ctypes.data_as(dp), C.byref(ib))
    calls = int(ctx.ncalls)
    oo = oracle.default_options(max_evals=500, factor=factor)
    xo, fo, ibo = x0.copy(), np.zeros(m), oracle.IterationBehavior()
    ctx.ncalls = 0
    rco = oracle.lib().nlo_lm_solve(C.byref(oo), C.cast(cb.dq_user_fcn, oracle.VECFCN), C.cast(None, oracle.JACFCN),
                                    C.byref(ctx), m, n, xo.ctypes.data_as(dp), fo.ctypes.data_as(dp), C.byref(ibo))
    assert rc == rco == 0
    assert ib.as_dict() == ibo.as_dict()
    assert calls == int(ctx.ncalls) == ib.fcn_count + n * ib.jacobian_count      # n + 1 evaluations per Jacobian, none extra
    assert np.array_equal(x, xo) and np.array_equal(f, fo)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 1024])
def test_newton_solve_with_compiled_jacobian_callback_bitwise(ds, oracle, n):
    """BASELINE config 3 taken literally: newton_solver through nlh_newton_solve with a COMPILED vecfcn AND a compiled
    analytic jacobianfcn (n = 1024: the 8 MB Jacobian crosses PCIe every iteration), line search on.  The oracle drives the
    same two callbacks; x, fvec, counts and flags must be the same bits, and the callbacks must have been called exactly
    as often (fcn_count evaluations; one Jacobian per iteration plus the reference's uncounted one at :535)."""
    from nonlin_amd import _lib
    cb = _cb()
    dp = C.POINTER(C.c_double)
    A, b, xt, x0 = oracle.dq_generate(12345, n, n, sigma=0.0, square_shift=True)
    u = np.zeros(n)
    ctx = Ctx(n, n, A.ctypes.data_as(dp), b.ctypes.data_as(dp), 0.5, 0, u.ctypes.data_as(dp))
    og = _lib.default_options()
    og.max_evals = 500
    x, f, ib = x0.copy(), np.zeros(n), _lib.IterationBehavior()
    rc = ds.lib.nlh_newton_solve(ds.h.ptr, C.byref(og), n, C.cast(cb.dq_user_fcn, _lib.VECFCN), C.cast(cb.dq_user_jac, _lib.JACFCN),
                                 C.byref(ctx), x.ctypes.data_as(dp), f.ctypes.data_as(dp), C.byref(ib))
    calls = int(ctx.ncalls)
    oo = oracle.default_options(max_evals=500)
    xo, fo, ibo = x0.copy(), np.zeros(n), oracle.IterationBehavior()
    ctx.ncalls = 0
    rco = oracle.lib().nlo_newton_solve(C.byref(oo), C.cast(cb.dq_user_fcn, oracle.VECFCN), C.cast(cb.dq_user_jac, oracle.JACFCN),
                                        C.byref(ctx), n, xo.ctypes.data_as(dp), fo.ctypes.data_as(dp), C.byref(ibo))
    assert rc == rco == 0
    assert ib.as_dict() == ibo.as_dict()
    assert calls == int(ctx.ncalls) == ib.fcn_count
    assert np.array_equal(x, xo) and np.array_equal(f, fo)
    assert ib.jacobian_count >= 2
