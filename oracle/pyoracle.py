"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FCNNVAR = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int32, C.POINTER(C.c_double))
GRADFCN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double))
VECFCN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double))
JACFCN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double))


class IterationBehavior(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "iter_count", "fcn_count", "jacobian_count", "gradient_count",
        "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Options(C.Structure):
    _fields_ = [("max_evals", C.c_int32), ("ftol", C.c_double), ("xtol", C.c_double),
                ("gtol", C.c_double), ("print_status", C.c_int32), ("factor", C.c_double),
                ("use_line_search", C.c_int32), ("ls_max_evals", C.c_int32),
                ("ls_alpha", C.c_double), ("ls_factor", C.c_double)]


class Trace(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("count", C.c_int32), ("xs", C.POINTER(C.c_double))]


class DqProblem(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("A", C.POINTER(C.c_double)),
                ("b", C.POINTER(C.c_double)), ("gamma", C.c_double), ("ncalls", C.c_int64),
                ("trace", C.POINTER(Trace))]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "nonlin_oracle.c")
    hdr = os.path.join(_HERE, "nonlin_oracle.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.nlo_default_options.argtypes = [C.POINTER(Options)]
        L.nlo_set_norm2_mode.argtypes = [C.c_int]
        L.nlo_set_norm2_mode.restype = None
        L.nlo_set_lmpar_minpack.argtypes = [C.c_int]
        L.nlo_set_lmpar_minpack.restype = None
        L.nlo_lmpar_loop_entries.argtypes = [C.c_int]
        L.nlo_lmpar_loop_entries.restype = C.c_long
        L.nlo_norm2.restype = C.c_double
        L.nlo_norm2.argtypes = [C.c_int32, dp]
        L.nlo_fd_jacobian.argtypes = [VECFCN, JACFCN, C.c_void_p, C.c_int32, C.c_int32, dp, dp, dp]
        L.nlo_lmfactor.argtypes = [C.c_int32, C.c_int32, dp, C.c_int32, C.c_int32, ip, dp, dp, dp]
        L.nlo_lmfactor.restype = None
        L.nlo_lmsolve.argtypes = [C.c_int32, dp, C.c_int32, ip, dp, dp, dp, dp, dp]
        L.nlo_lmsolve.restype = None
        L.nlo_lmpar.argtypes = [C.c_int32, C.c_int32, dp, C.c_int32, ip, dp, dp, C.c_double, dp, dp, dp, dp, dp]
        L.nlo_lmpar.restype = None
        L.nlo_lm_solve.argtypes = [C.POINTER(Options), VECFCN, JACFCN, C.c_void_p, C.c_int32, C.c_int32,
                                   dp, dp, C.POINTER(IterationBehavior)]
        L.nlo_newton_solve.argtypes = [C.POINTER(Options), VECFCN, JACFCN, C.c_void_p, C.c_int32,
                                       dp, dp, C.POINTER(IterationBehavior)]
        L.nlo_lu_factor.argtypes = [C.c_int32, dp, C.c_int32, ip]
        L.nlo_lu_solve.argtypes = [C.c_int32, dp, C.c_int32, ip, dp]
        L.nlo_lu_solve.restype = None
        L.nlo_min_backtrack_search.restype = C.c_double
        L.nlo_min_backtrack_search.argtypes = [C.c_int32] + [C.c_double] * 6
        L.nlo_dq_generate.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                                      C.c_int32, dp, dp, dp, dp]
        L.nlo_dq_generate.restype = None
        L.nlo_dq_fcn.argtypes = [C.c_void_p, C.c_int32, dp, C.c_int32, dp]
        L.nlo_dq_fcn.restype = None
        L.nlo_dq_jac.argtypes = [C.c_void_p, C.c_int32, dp, C.c_int32, dp]
        L.nlo_dq_jac.restype = None
        L.nlo_dq_lm_solve.argtypes = [C.POINTER(Options), C.POINTER(DqProblem), dp, dp,
                                      C.POINTER(IterationBehavior)]
        L.nlo_dq_newton_solve.argtypes = [C.POINTER(Options), C.POINTER(DqProblem), C.c_int32, dp, dp,
                                          C.POINTER(IterationBehavior)]
        L.nlo_dq_quasi_newton_solve.argtypes = [C.POINTER(Options), C.c_int32, C.POINTER(DqProblem), C.c_int32, dp, dp,
                                                C.POINTER(IterationBehavior)]
        L.nlo_quasi_newton_solve.argtypes = [C.POINTER(Options), C.c_int32, VECFCN, JACFCN, C.c_void_p, C.c_int32, dp, dp,
                                             C.POINTER(IterationBehavior)]
        L.nlo_cls_solve.argtypes = [C.POINTER(Options), C.c_double, C.c_double, dp, dp, VECFCN, JACFCN, C.c_void_p,
                                    C.c_int32, C.c_int32, dp, dp, C.POINTER(IterationBehavior)]
        L.nlo_dq_cls_solve.argtypes = [C.POINTER(Options), C.c_double, C.c_double, dp, dp, C.POINTER(DqProblem), dp, dp,
                                       C.POINTER(IterationBehavior)]
        L.nlo_qr_factor_rhs.argtypes = [C.c_int32, C.c_int32, dp, dp]
        L.nlo_bfgs_solve.argtypes = [C.POINTER(Options), FCNNVAR, GRADFCN, C.c_void_p, C.c_int32, dp, dp,
                                     C.POINTER(IterationBehavior)]
        L.nlo_dq_bfgs_solve.argtypes = [C.POINTER(Options), C.POINTER(DqProblem), dp, dp, C.POINTER(IterationBehavior)]
        L.nlo_fd_gradient.argtypes = [FCNNVAR, GRADFCN, C.c_void_p, C.c_int32, dp, dp, dp]
        L.nlo_rtr.argtypes = [C.c_int32, dp, dp]
        L.nlo_symv.argtypes = [C.c_int32, dp, dp, dp]
        L.nlo_chol_update.argtypes = [C.c_int32, dp, dp]
        L.nlo_chol_downdate.argtypes = [C.c_int32, dp, dp]
        L.nlo_chol_factor_upper.argtypes = [C.c_int32, dp, dp]
        L.nlo_solve_cholesky_upper.argtypes = [C.c_int32, dp, dp]
        L.nlo_poly_fit.argtypes = [C.c_int32, C.c_int32, dp, dp, C.c_int32, dp]
        L.nlo_poly_eval.argtypes = [C.c_int32, dp, C.c_double]
        L.nlo_poly_eval.restype = C.c_double
        L.nlo_qr_factor_full.argtypes = [C.c_int32, dp, dp, dp]
        L.nlo_qr_rank1_update.argtypes = [C.c_int32, dp, dp, dp, dp]
        L.nlo_solve_upper.argtypes = [C.c_int32, dp, dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def set_norm2_mode(mode):
    """0 = flang algorithm (default), 1 = sqrt(sum of squares)."""
    lib().nlo_set_norm2_mode(int(mode))


def set_lmpar_minpack(on):
    """Test-only: 1 = MINPACK's own lines at src/nonlin_least_squares.f90:531 / :552 (norm over n, rows j+1..n)."""
    lib().nlo_set_lmpar_minpack(int(bool(on)))


def lmpar_loop_entries(reset=False):
    """How often lmpar's iteration (:522-563) has been entered since the last reset."""
    return int(lib().nlo_lmpar_loop_entries(int(bool(reset))))


def norm2(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return float(lib().nlo_norm2(x.size, _dp(x)))


def default_options(**kw):
    o = Options()
    lib().nlo_default_options(C.byref(o))
    for k, v in kw.items():
        if k == "factor":  # lss_set_factor clamp, src/nonlin_least_squares.f90:108-114
            v = min(max(v, 0.1), 100.0)
        setattr(o, k, v)
    return o


def _wrap_fcn(fcn, record=None):
    def _f(ctx, n, xp, m, fp):
        x = np.ctypeslib.as_array(xp, shape=(n,))
        f = np.ctypeslib.as_array(fp, shape=(m,))
        if record is not None:
            record.append(x.copy())
        fcn(x, f)
    return VECFCN(_f)


def _wrap_jac(jac):
    if jac is None:
        return C.cast(None, JACFCN)

    def _j(ctx, n, xp, m, jp):
        x = np.ctypeslib.as_array(xp, shape=(n,))
        J = np.ctypeslib.as_array(jp, shape=(n, m)).T  # column-major m x n view
        jac(x, J)
    return JACFCN(_j)


def fd_jacobian(fcn, m, n, x, fv=None, jac=None):
    """vecfcn_helper%jacobian.  Returns a column-major (Fortran-order) m x n array."""
    x = np.array(x, dtype=np.float64)
    J = np.zeros((m, n), order="F")
    cf, cj = _wrap_fcn(fcn), _wrap_jac(jac)
    fvp = _dp(np.ascontiguousarray(fv, dtype=np.float64)) if fv is not None else None
    rc = lib().nlo_fd_jacobian(cf, cj, None, m, n, _dp(x), fvp, _dp(J))
    assert rc == 0
    return J


def lm_solve(fcn, m, n, x0, jac=None, opts=None, record=None):
    """least_squares_solver%solve.  Returns (rc, x, fvec, ib_dict)."""
    x = np.array(x0, dtype=np.float64)
    fvec = np.zeros(m)
    ib = IterationBehavior()
    o = opts or default_options()
    cf, cj = _wrap_fcn(fcn, record), _wrap_jac(jac)
    rc = lib().nlo_lm_solve(C.byref(o), cf, cj, None, m, n, _dp(x), _dp(fvec), C.byref(ib))
    return rc, x, fvec, ib.as_dict()


def newton_solve(fcn, n, x0, jac=None, opts=None, record=None):
    """newton_solver%solve.  Returns (rc, x, fvec, ib_dict)."""
    x = np.array(x0, dtype=np.float64)
    fvec = np.zeros(n)
    ib = IterationBehavior()
    o = opts or default_options()
    cf, cj = _wrap_fcn(fcn, record), _wrap_jac(jac)
    rc = lib().nlo_newton_solve(C.byref(o), cf, cj, None, n, _dp(x), _dp(fvec), C.byref(ib))
    return rc, x, fvec, ib.as_dict()


def quasi_newton_solve(fcn, n, x0, jac=None, opts=None, jdelta=5, record=None):
    """quasi_newton_solver%solve.  Returns (rc, x, fvec, ib_dict)."""
    x = np.array(x0, dtype=np.float64)
    fvec = np.zeros(n)
    ib = IterationBehavior()
    o = opts or default_options()
    cf, cj = _wrap_fcn(fcn, record), _wrap_jac(jac)
    rc = lib().nlo_quasi_newton_solve(C.byref(o), int(jdelta), cf, cj, None, n, _dp(x), _dp(fvec), C.byref(ib))
    return rc, x, fvec, ib.as_dict()


def _bounds(v, n):
    if v is None:
        return None, None
    a = np.ascontiguousarray(v, dtype=np.float64)
    assert a.shape == (n,)
    return a, _dp(a)


def cls_solve(fcn, m, n, x0, jac=None, opts=None, lower=None, upper=None, delta=1.0, stepscale=1.0, record=None):
    """constrained_least_squares_solver%solve.  Returns (rc, x, fvec, ib_dict)."""
    x = np.array(x0, dtype=np.float64)
    fvec = np.zeros(m)
    ib = IterationBehavior()
    o = opts or default_options()
    cf, cj = _wrap_fcn(fcn, record), _wrap_jac(jac)
    lo, plo = _bounds(lower, n)
    hi, phi = _bounds(upper, n)
    rc = lib().nlo_cls_solve(C.byref(o), float(delta), float(stepscale), plo, phi, cf, cj, None, m, n, _dp(x), _dp(fvec),
                             C.byref(ib))
    return rc, x, fvec, ib.as_dict()


def _wrap_scalar(fcn):
    def _f(ctx, n, xp):
        return float(fcn(np.ctypeslib.as_array(xp, shape=(n,))))
    return FCNNVAR(_f)


def _wrap_grad(grad):
    if grad is None:
        return C.cast(None, GRADFCN)

    def _g(ctx, n, xp, gp):
        grad(np.ctypeslib.as_array(xp, shape=(n,)), np.ctypeslib.as_array(gp, shape=(n,)))
    return GRADFCN(_g)


def bfgs_solve(fcn, n, x0, grad=None, opts=None):
    """bfgs%solve; fcn(x) -> float, grad(x, g) fills g.  opts: max_evals (500), gtol = get_tolerance (1e-12),
    xtol = get_var_tolerance (1e-12).  Returns (rc, x, fout, ib_dict)."""
    x = np.array(x0, dtype=np.float64)
    fout = C.c_double(0.0)
    ib = IterationBehavior()
    o = opts or default_options(max_evals=500)
    cf, cg = _wrap_scalar(fcn), _wrap_grad(grad)
    rc = lib().nlo_bfgs_solve(C.byref(o), cf, cg, None, n, _dp(x), C.cast(C.byref(fout), C.POINTER(C.c_double)), C.byref(ib))
    return rc, x, fout.value, ib.as_dict()


def fd_gradient(fcn, x, fv=None):
    x = np.array(x, dtype=np.float64)
    g = np.zeros(x.size)
    cf = _wrap_scalar(fcn)
    pf = None
    if fv is not None:
        v = C.c_double(fv)
        pf = C.cast(C.byref(v), C.POINTER(C.c_double))
    lib().nlo_fd_gradient(cf, C.cast(None, GRADFCN), None, x.size, _dp(x), pf, _dp(g))
    return g


def rtr(r):
    r = np.array(r, dtype=np.float64, order="F")
    b = np.zeros_like(r, order="F")
    lib().nlo_rtr(r.shape[0], _dp(r), _dp(b))
    return b


def chol_update(r, u):
    r = np.array(r, dtype=np.float64, order="F")
    u = np.array(u, dtype=np.float64)
    lib().nlo_chol_update(r.shape[0], _dp(r), _dp(u))
    return r


def chol_downdate(r, u):
    r = np.array(r, dtype=np.float64, order="F")
    u = np.array(u, dtype=np.float64)
    rc = lib().nlo_chol_downdate(r.shape[0], _dp(r), _dp(u))
    return rc, r


def chol_factor_upper(b):
    b = np.array(b, dtype=np.float64, order="F")
    r = np.zeros_like(b, order="F")
    rc = lib().nlo_chol_factor_upper(b.shape[0], _dp(b), _dp(r))
    return rc, r


def solve_cholesky_upper(r, x):
    r = np.array(r, dtype=np.float64, order="F")
    x = np.array(x, dtype=np.float64)
    lib().nlo_solve_cholesky_upper(r.shape[0], _dp(r), _dp(x))
    return x


def poly_fit(x, y, order, thru_zero=False):
    """polynomial%fit / fit_thru_zero.  Returns (rc, coefficients c0..c_order)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    coef = np.zeros(order + 1)
    rc = lib().nlo_poly_fit(x.size, int(order), _dp(x), _dp(y), int(thru_zero), _dp(coef))
    return rc, coef


def poly_eval(coef, x):
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    return np.array([lib().nlo_poly_eval(coef.size - 1, _dp(coef), float(v)) for v in np.atleast_1d(x)])


def qr_factor_rhs(a, f):
    """Householder QR of a tall Fortran-order matrix with the reflectors applied to f.  Returns (r_full, qtf)."""
    a = np.array(a, dtype=np.float64, order="F")
    f = np.array(f, dtype=np.float64)
    lib().nlo_qr_factor_rhs(a.shape[0], a.shape[1], _dp(a), _dp(f))
    return a, f


def qr_factor_full(a):
    """Householder QR of a square Fortran-order matrix.  Returns (q, r)."""
    a = np.array(a, dtype=np.float64, order="F")
    n = a.shape[0]
    q = np.zeros((n, n), order="F")
    r = np.zeros((n, n), order="F")
    lib().nlo_qr_factor_full(n, _dp(a), _dp(q), _dp(r))
    return q, r


def qr_rank1_update(q, r, u, v):
    """Q1 R1 = Q R + u v^T.  Returns (q1, r1)."""
    q = np.array(q, dtype=np.float64, order="F")
    r = np.array(r, dtype=np.float64, order="F")
    u = np.array(u, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    lib().nlo_qr_rank1_update(q.shape[0], _dp(q), _dp(r), _dp(u), _dp(v))
    return q, r


def solve_upper(r, b):
    r = np.array(r, dtype=np.float64, order="F")
    x = np.array(b, dtype=np.float64)
    lib().nlo_solve_upper(r.shape[0], _dp(r), _dp(x))
    return x


def dq_generate(seed, m, n, gamma=0.5, sigma=1e-3, spread=0.3, square_shift=False):
    """Synthetic dense-quadratic problem.  A is Fortran-order m x n."""
    A = np.zeros((m, n), order="F")
    b = np.zeros(m)
    xt = np.zeros(n)
    x0 = np.zeros(n)
    lib().nlo_dq_generate(seed, m, n, gamma, sigma, spread, int(square_shift), _dp(A), _dp(b), _dp(xt), _dp(x0))
    return A, b, xt, x0


def _dq_problem(A, b, gamma, trace=None):
    m, n = A.shape
    assert A.flags.f_contiguous
    p = DqProblem(m, n, _dp(A), _dp(b), gamma, 0, None)
    if trace is not None:
        p.trace = C.pointer(trace)
    return p


def dq_residual(A, b, gamma, x):
    m, n = A.shape
    p = _dq_problem(A, b, gamma)
    f = np.zeros(m)
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().nlo_dq_fcn(C.byref(p), n, _dp(x), m, _dp(f))
    return f


def dq_jacobian(A, b, gamma, x):
    m, n = A.shape
    p = _dq_problem(A, b, gamma)
    J = np.zeros((m, n), order="F")
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().nlo_dq_jac(C.byref(p), n, _dp(x), m, _dp(J))
    return J


def dq_fd_jacobian(A, b, gamma, x, fv=None):
    """Forward-difference Jacobian of the dense-quadratic model through vfh_jac_fcn."""
    m, n = A.shape
    p = _dq_problem(A, b, gamma)
    J = np.zeros((m, n), order="F")
    x = np.array(x, dtype=np.float64)
    L = lib()
    fcn = C.cast(L.nlo_dq_fcn, VECFCN)
    fvp = _dp(np.ascontiguousarray(fv, dtype=np.float64)) if fv is not None else None
    rc = L.nlo_fd_jacobian(fcn, C.cast(None, JACFCN), C.byref(p), m, n, _dp(x), fvp, _dp(J))
    assert rc == 0
    return J


def dq_lm_solve(A, b, gamma, x0, opts=None, trace_capacity=0):
    """Returns (rc, x, fvec, ib_dict, ncalls, trace_xs or None)."""
    m, n = A.shape
    tr = None
    buf = None
    if trace_capacity:
        buf = np.zeros((trace_capacity, n))
        tr = Trace(trace_capacity, 0, _dp(buf))
    p = _dq_problem(A, b, gamma, tr)
    x = np.array(x0, dtype=np.float64)
    fvec = np.zeros(m)
    ib = IterationBehavior()
    o = opts or default_options()
    rc = lib().nlo_dq_lm_solve(C.byref(o), C.byref(p), _dp(x), _dp(fvec), C.byref(ib))
    xs = buf[:min(tr.count, trace_capacity)] if tr is not None else None
    return rc, x, fvec, ib.as_dict(), int(p.ncalls), xs


def dq_newton_solve(A, b, gamma, x0, analytic=True, opts=None):
    m, n = A.shape
    assert m == n
    p = _dq_problem(A, b, gamma)
    x = np.array(x0, dtype=np.float64)
    fvec = np.zeros(m)
    ib = IterationBehavior()
    o = opts or default_options()
    rc = lib().nlo_dq_newton_solve(C.byref(o), C.byref(p), int(analytic), _dp(x), _dp(fvec), C.byref(ib))
    return rc, x, fvec, ib.as_dict(), int(p.ncalls)


def dq_quasi_newton_solve(A, b, gamma, x0, analytic=True, opts=None, jdelta=5):
    m, n = A.shape
    assert m == n
    p = _dq_problem(A, b, gamma)
    x = np.array(x0, dtype=np.float64)
    fvec = np.zeros(m)
    ib = IterationBehavior()
    o = opts or default_options()
    rc = lib().nlo_dq_quasi_newton_solve(C.byref(o), int(jdelta), C.byref(p), int(analytic), _dp(x), _dp(fvec),
                                         C.byref(ib))
    return rc, x, fvec, ib.as_dict(), int(p.ncalls)


def dq_cls_solve(A, b, gamma, x0, opts=None, lower=None, upper=None, delta=1.0, stepscale=1.0):
    m, n = A.shape
    p = _dq_problem(A, b, gamma)
    x = np.array(x0, dtype=np.float64)
    fvec = np.zeros(m)
    ib = IterationBehavior()
    o = opts or default_options()
    lo, plo = _bounds(lower, n)
    hi, phi = _bounds(upper, n)
    rc = lib().nlo_dq_cls_solve(C.byref(o), float(delta), float(stepscale), plo, phi, C.byref(p), _dp(x), _dp(fvec),
                                C.byref(ib))
    return rc, x, fvec, ib.as_dict(), int(p.ncalls)


def dq_bfgs_solve(A, b, gamma, x0, opts=None):
    """bfgs on 0.5 * ||r(x)||^2 of the dense-quadratic model, FD gradient.  Returns (rc, x, fout, ib, ncalls)."""
    p = _dq_problem(A, b, gamma)
    x = np.array(x0, dtype=np.float64)
    fout = np.zeros(1)
    ib = IterationBehavior()
    o = opts or default_options(max_evals=500)
    rc = lib().nlo_dq_bfgs_solve(C.byref(o), C.byref(p), _dp(x), _dp(fout), C.byref(ib))
    return rc, x, float(fout[0]), ib.as_dict(), int(p.ncalls)


def lmfactor(a):
    """a: Fortran-order m x n.  Returns (a_out, ipvt0, rdiag, acnorm)."""
    a = np.array(a, dtype=np.float64, order="F")
    m, n = a.shape
    ipvt = np.zeros(n, dtype=np.int32)
    rdiag = np.zeros(n)
    acnorm = np.zeros(n)
    wa = np.zeros(n)
    lib().nlo_lmfactor(m, n, _dp(a), m, 1, _ip(ipvt), _dp(rdiag), _dp(acnorm), _dp(wa))
    return a, ipvt, rdiag, acnorm


def lmpar(r, ipvt, diag, qtb, delta, par, wa4):
    """r: Fortran-order m x n holding R in its top n x n.  Returns (par, x, sdiag, r_out)."""
    r = np.array(r, dtype=np.float64, order="F")
    m, n = r.shape
    x = np.zeros(n)
    sdiag = np.zeros(n)
    wa1 = np.zeros(n)
    wa2 = np.array(wa4, dtype=np.float64)
    assert wa2.shape == (m,)
    parv = (C.c_double * 1)(par)
    lib().nlo_lmpar(m, n, _dp(r), m, _ip(np.ascontiguousarray(ipvt, dtype=np.int32)),
                    _dp(np.ascontiguousarray(diag, dtype=np.float64)),
                    _dp(np.ascontiguousarray(qtb, dtype=np.float64)), delta,
                    C.cast(parv, C.POINTER(C.c_double)), _dp(x), _dp(sdiag), _dp(wa1), _dp(wa2))
    return parv[0], x, sdiag, r
