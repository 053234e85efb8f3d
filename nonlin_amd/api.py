"""Host-side mirror of nonlin's plugin / solver interface for the hot path.

Same names, argument meaning and error behaviour as the reference types
(src/nonlin_multi_eqn_mult_var.f90, src/nonlin_least_squares.f90, src/nonlin_solve.f90,
src/nonlin_linesearch.f90, src/nonlin_types.f90), bound to the C ABI of
include/nonlin_hip.h.  Where the reference `error stop`s with an NL_* code this module
raises NonlinError(code).  All numerical work happens in libnonlin_hip.so on the GPU.

User callbacks follow the Fortran subroutines:
    fcn(x, f, args)      -- fill f (length m) in place             (vecfcn,      :14-25)
    jac(x, jac, args)    -- fill jac (m x n, Fortran order)        (jacobianfcn, :27-38)
"""
import ctypes as C

import numpy as np

from . import _lib

# src/nonlin_error_handling.f90:10-38
NL_NO_ERROR = 0
NL_INVALID_INPUT_ERROR = 201
NL_ARRAY_SIZE_ERROR = 202
NL_OUT_OF_MEMORY_ERROR = 105
NL_INVALID_OPERATION_ERROR = 104
NL_CONVERGENCE_ERROR = 106
NL_DIVERGENT_BEHAVIOR_ERROR = 206
NL_SPURIOUS_CONVERGENCE_ERROR = 207
NL_TOLERANCE_TOO_SMALL_ERROR = 208
NL_INDEX_OUT_OF_RANGE_ERROR = 209
NL_DIVIDE_BY_ZERO_ERROR = 210
NL_UNDEFINED_FUNCTION_ERROR = 211
NL_UNDERDEFINED_PROBLEM_ERROR = 212


class NonlinError(RuntimeError):
    """Raised where the reference executes `error stop <code>`."""

    def __init__(self, code):
        super().__init__(f"nonlin error stop {code}")
        self.code = code


class iteration_behavior:
    """src/nonlin_types.f90:8-29."""

    def __init__(self):
        self.iter_count = 0
        self.fcn_count = 0
        self.jacobian_count = 0
        self.gradient_count = 0
        self.converge_on_fcn = False
        self.converge_on_chng = False
        self.converge_on_zero_diff = False

    def _fill(self, c):
        self.iter_count = int(c.iter_count)
        self.fcn_count = int(c.fcn_count)
        self.jacobian_count = int(c.jacobian_count)
        self.gradient_count = int(c.gradient_count)
        self.converge_on_fcn = bool(c.converge_on_fcn)
        self.converge_on_chng = bool(c.converge_on_chng)
        self.converge_on_zero_diff = bool(c.converge_on_zero_diff)

    def as_dict(self):
        return dict(self.__dict__)


_default_handle = None


def default_handle():
    global _default_handle
    if _default_handle is None:
        _default_handle = _lib.Handle(0)
    return _default_handle


def _dp(a):
    return a.ctypes.data_as(_lib.c_double_p)


class vecfcn_helper:
    """src/nonlin_multi_eqn_mult_var.f90:41-65."""

    def __init__(self):
        self._fcn = None
        self._jac = None
        self._nfcn = 0
        self._nvar = 0

    def set_fcn(self, fcn, nfcn, nvar):          # :126-140
        self._fcn = fcn
        self._nfcn = int(nfcn)
        self._nvar = int(nvar)

    def set_jacobian(self, jac):                 # :143-153
        self._jac = jac

    def is_fcn_defined(self):                    # :156-164
        return self._fcn is not None

    def is_jacobian_defined(self):               # :167-175
        return self._jac is not None

    def get_equation_count(self):                # :280-287
        return self._nfcn

    def get_variable_count(self):                # :290-297
        return self._nvar

    def fcn(self, x, f, args=None):              # :178-195 (silently does nothing if unset)
        if self._fcn is not None:
            self._fcn(x, f, args)

    # -- C trampolines -----------------------------------------------------
    def _c_fcn(self, args):
        user = self._fcn
        if user is None:
            return C.cast(None, _lib.VECFCN)

        def tramp(ctx, n, xp, m, fp):
            x = np.ctypeslib.as_array(xp, shape=(n,))
            f = np.ctypeslib.as_array(fp, shape=(m,))
            user(x, f, args)
        return _lib.VECFCN(tramp)

    def _c_jac(self, args):
        user = self._jac
        if user is None:
            return C.cast(None, _lib.JACFCN)

        def tramp(ctx, n, xp, m, jp):
            x = np.ctypeslib.as_array(xp, shape=(n,))
            J = np.ctypeslib.as_array(jp, shape=(n, m)).T   # Fortran-order m x n view
            user(x, J, args)
        return _lib.JACFCN(tramp)

    def jacobian(self, x, jac, fv=None, args=None, handle=None):
        """vfh_jac_fcn (:198-277): analytic dispatch or forward differences (GPU column write).
        x is perturbed in place and restored; jac must be an m x n Fortran-order float64 array."""
        m, n = self._nfcn, self._nvar
        if x.shape != (n,):
            raise NonlinError(2)                 # :232-233
        if jac.shape != (m, n):
            raise NonlinError(3)                 # :234-235
        if not self.is_fcn_defined():
            raise NonlinError(NL_UNDEFINED_FUNCTION_ERROR)   # :240
        if not (jac.flags.f_contiguous and jac.dtype == np.float64):
            raise ValueError("jac must be a Fortran-order float64 array")
        h = handle or default_handle()
        cf, cj = self._c_fcn(args), self._c_jac(args)
        fvp = _dp(np.ascontiguousarray(fv, dtype=np.float64)) if fv is not None else None
        rc = h.lib.nlh_fd_jacobian(h.ptr, m, n, cf, cj, None, _dp(x), fvp, _dp(jac))
        h.check(rc, "nlh_fd_jacobian")
        if rc:
            raise NonlinError(rc)


class equation_solver:
    """src/nonlin_multi_eqn_mult_var.f90:67-91; defaults :69-77."""

    def __init__(self):
        self._max_eval = 100
        self._fcn_tol = 1.0e-8
        self._xtol = 1.0e-12
        self._gtol = 1.0e-12
        self._print_status = False
        self.handle = None
        # extension: 2 = NLH_FACTOR_EXACT (default: reference operation order, bit-identical results),
        # 0 = NLH_FACTOR_AUTO (J^T J + Cholesky), 1 = NLH_FACTOR_QR
        self.factor_policy = 2

    def get_max_fcn_evals(self): return self._max_eval
    def set_max_fcn_evals(self, n): self._max_eval = int(n)
    def get_fcn_tolerance(self): return self._fcn_tol
    def set_fcn_tolerance(self, x): self._fcn_tol = float(x)
    def get_var_tolerance(self): return self._xtol
    def set_var_tolerance(self, x): self._xtol = float(x)
    def get_gradient_tolerance(self): return self._gtol
    def set_gradient_tolerance(self, x): self._gtol = float(x)
    def get_print_status(self): return self._print_status
    def set_print_status(self, x): self._print_status = bool(x)

    def _options(self):
        o = _lib.default_options()
        o.max_evals = self._max_eval
        o.ftol = self._fcn_tol
        o.xtol = self._xtol
        o.gtol = self._gtol
        o.print_status = 1 if self._print_status else 0
        o.factor_policy = int(self.factor_policy)
        return o

    def _handle(self):
        return self.handle or default_handle()


def _check_xf(fcn, x, fvec):
    if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous):
        raise ValueError("x must be a contiguous float64 numpy array (it is updated in place)")
    if not (isinstance(fvec, np.ndarray) and fvec.dtype == np.float64 and fvec.flags.c_contiguous):
        raise ValueError("fvec must be a contiguous float64 numpy array (it is filled in place)")


class least_squares_solver(equation_solver):
    """src/nonlin_least_squares.f90:20-31."""

    def __init__(self):
        super().__init__()
        self._factor = 100.0                      # :25

    def get_step_scaling_factor(self):           # :80-93
        return self._factor

    def set_step_scaling_factor(self, x):        # :96-115
        x = float(x)
        self._factor = 0.1 if x < 0.1 else (100.0 if x > 100.0 else x)

    def solve(self, fcn, x, fvec, ib=None, args=None):
        """lss_solve (:118-391).  x: initial estimate -> solution; fvec: F at the solution."""
        _check_xf(fcn, x, fvec)
        if not fcn.is_fcn_defined():
            raise NonlinError(NL_UNDEFINED_FUNCTION_ERROR)   # :188
        m, n = fcn.get_equation_count(), fcn.get_variable_count()
        if n > m:
            raise NonlinError(NL_UNDERDEFINED_PROBLEM_ERROR)  # :189
        if x.shape != (n,):
            raise NonlinError(3)                  # :191-192
        if fvec.shape != (m,):
            raise NonlinError(4)                  # :193-194
        o = self._options()
        o.factor = self._factor
        h = self._handle()
        cib = _lib.IterationBehavior()
        cf, cj = fcn._c_fcn(args), fcn._c_jac(args)
        rc = h.lib.nlh_lm_solve(h.ptr, C.byref(o), m, n, cf, cj, None, _dp(x), _dp(fvec), C.byref(cib))
        h.check(rc, "nlh_lm_solve")
        if ib is not None:
            ib._fill(cib)
        if rc:
            raise NonlinError(rc)                 # :388-390


class constrained_equation_solver(least_squares_solver):
    """src/nonlin_least_squares.f90:33-53 (bounds holder)."""

    def __init__(self):
        super().__init__()
        self._upper = np.zeros(0)
        self._lower = np.zeros(0)

    def get_upper_limits(self): return self._upper.copy()                                  # :796-808
    def set_upper_limits(self, x): self._upper = np.array(x, dtype=np.float64).ravel()      # :811-824
    def get_lower_limits(self): return self._lower.copy()                                  # :827-839
    def set_lower_limits(self, x): self._lower = np.array(x, dtype=np.float64).ravel()      # :842-855

    def apply_limits(self, x):                                                             # :858-883
        nl, nu = min(x.size, self._lower.size), min(x.size, self._upper.size)
        for i in range(nl):
            if x[i] < self._lower[i]:
                x[i] = self._lower[i]
        for i in range(nu):
            if x[i] > self._upper[i]:
                x[i] = self._upper[i]


class constrained_least_squares_solver(constrained_equation_solver):
    """src/nonlin_least_squares.f90:55-74."""

    def __init__(self):
        super().__init__()
        self._delta = 1.0                         # :60
        self._scaling = 1.0                       # :61

    def get_trust_region_radius(self): return self._delta                    # :888-895
    def set_trust_region_radius(self, x): self._delta = 1.0 if x <= 0.0 else float(x)       # :898-910
    def get_step_scaling_factor(self): return self._scaling                  # :913-920 (overrides the LM factor)
    def set_step_scaling_factor(self, x): self._scaling = 1.0 if x <= 0.0 else float(x)     # :923-935

    def solve(self, fcn, x, fvec, ib=None, args=None):
        """cls_solve (:938-1176)."""
        _check_xf(fcn, x, fvec)
        if not fcn.is_fcn_defined():
            raise NonlinError(NL_UNDEFINED_FUNCTION_ERROR)   # :988
        m, n = fcn.get_equation_count(), fcn.get_variable_count()
        if n > m:
            raise NonlinError(NL_UNDERDEFINED_PROBLEM_ERROR)  # :989
        if x.shape != (n,):
            raise NonlinError(3)
        if fvec.shape != (m,):
            raise NonlinError(4)
        big = float(np.finfo(np.float64).max)
        if self._lower.size != n:                 # :999-1009: wrong-sized limits are replaced (and stored)
            self.set_lower_limits(np.full(n, -big))
        if self._upper.size != n:
            self.set_upper_limits(np.full(n, big))
        lo = np.ascontiguousarray(self._lower)
        hi = np.ascontiguousarray(self._upper)
        o = self._options()
        h = self._handle()
        cib = _lib.IterationBehavior()
        cf, cj = fcn._c_fcn(args), fcn._c_jac(args)
        rc = h.lib.nlh_cls_solve(h.ptr, C.byref(o), self._delta, self._scaling, _dp(lo), _dp(hi), m, n, cf, cj, None,
                                 _dp(x), _dp(fvec), C.byref(cib))
        h.check(rc, "nlh_cls_solve")
        if ib is not None:
            ib._fill(cib)
        if rc:
            raise NonlinError(rc)                 # :1173-1175


class line_search:
    """src/nonlin_linesearch.f90:18-65 (configuration; the search runs inside newton_solver)."""

    def __init__(self):
        self._max_eval = 100                      # :35
        self._alpha = 1.0e-4                      # :38
        self._factor = 0.1                        # :46

    def get_max_fcn_evals(self): return self._max_eval
    def set_max_fcn_evals(self, x): self._max_eval = int(x)
    def get_scaling_factor(self): return self._alpha
    def set_scaling_factor(self, x): self._alpha = float(x)
    def get_distance_factor(self): return self._factor

    def set_distance_factor(self, x):            # :133-149
        x = float(x)
        self._factor = 0.1 if x <= 0.0 else (0.99 if x >= 1.0 else x)


class line_search_solver(equation_solver):
    """src/nonlin_solve.f90:20-41, 92-151."""

    def __init__(self):
        super().__init__()
        self._line_search = None
        self._use_line_search = True              # :30

    def get_line_search(self):                   # :92-100 (returns a copy)
        if self._line_search is None:
            return None
        ls = line_search()
        ls.__dict__.update(self._line_search.__dict__)
        return ls

    def set_line_search(self, ls):               # :103-111
        c = line_search()
        c.__dict__.update(ls.__dict__)
        self._line_search = c

    def set_default_line_search(self):           # :114-121
        self.set_line_search(line_search())

    def is_line_search_defined(self):            # :124-131
        return self._line_search is not None

    def get_use_line_search(self): return self._use_line_search
    def set_use_line_search(self, x): self._use_line_search = bool(x)


class quasi_newton_solver(line_search_solver):
    """src/nonlin_solve.f90:43-58."""

    def __init__(self):
        super().__init__()
        self._jdelta = 5                                      # m_jDelta, :51

    def get_jacobian_interval(self): return self._jdelta      # :429-436
    def set_jacobian_interval(self, n): self._jdelta = int(n)  # :439-447

    def solve(self, fcn, x, fvec, ib=None, args=None):
        """qns_solve (:156-427)."""
        _check_xf(fcn, x, fvec)
        if self.get_use_line_search() and not self.is_line_search_defined():
            self.set_default_line_search()        # :233-237
        if not fcn.is_fcn_defined():
            raise NonlinError(NL_UNDEFINED_FUNCTION_ERROR)   # :240
        m, n = fcn.get_equation_count(), fcn.get_variable_count()
        if n != m:
            raise NonlinError(NL_INVALID_INPUT_ERROR)        # :241
        if x.shape != (n,):
            raise NonlinError(3)
        if fvec.shape != (m,):
            raise NonlinError(4)
        o = self._options()
        o.use_line_search = 1 if self._use_line_search else 0
        if self._line_search is not None:
            o.ls_max_evals = self._line_search._max_eval
            o.ls_alpha = self._line_search._alpha
            o.ls_factor = self._line_search._factor
        h = self._handle()
        cib = _lib.IterationBehavior()
        cf, cj = fcn._c_fcn(args), fcn._c_jac(args)
        rc = h.lib.nlh_quasi_newton_solve(h.ptr, C.byref(o), self._jdelta, n, cf, cj, None, _dp(x), _dp(fvec),
                                          C.byref(cib))
        h.check(rc, "nlh_quasi_newton_solve")
        if ib is not None:
            ib._fill(cib)
        if rc:
            raise NonlinError(rc)


class newton_solver(line_search_solver):
    """src/nonlin_solve.f90:60-67."""

    def solve(self, fcn, x, fvec, ib=None, args=None):
        """ns_solve (:452-638)."""
        _check_xf(fcn, x, fvec)
        if self.get_use_line_search() and not self.is_line_search_defined():
            self.set_default_line_search()        # :511-515
        if not fcn.is_fcn_defined():
            raise NonlinError(NL_UNDEFINED_FUNCTION_ERROR)   # :518
        m, n = fcn.get_equation_count(), fcn.get_variable_count()
        if n != m:
            raise NonlinError(NL_INVALID_INPUT_ERROR)        # :519
        if x.shape != (n,):
            raise NonlinError(3)
        if fvec.shape != (m,):
            raise NonlinError(4)
        o = self._options()
        o.use_line_search = 1 if self._use_line_search else 0
        if self._line_search is not None:
            o.ls_max_evals = self._line_search._max_eval
            o.ls_alpha = self._line_search._alpha
            o.ls_factor = self._line_search._factor
        h = self._handle()
        cib = _lib.IterationBehavior()
        cf, cj = fcn._c_fcn(args), fcn._c_jac(args)
        rc = h.lib.nlh_newton_solve(h.ptr, C.byref(o), n, cf, cj, None, _dp(x), _dp(fvec), C.byref(cib))
        h.check(rc, "nlh_newton_solve")
        if ib is not None:
            ib._fill(cib)
        if rc:
            raise NonlinError(rc)


class polynomial:
    """src/nonlin_polynomials.f90:39-62 -- the fitting front end only (initialize, order, fit, fit_thru_zero,
    evaluate, get, get_all, set); roots / arithmetic are outside the hot path."""

    def __init__(self, order=None):
        self._c = None
        if order is not None:
            self.initialize(order)

    def initialize(self, order_or_coeffs):                    # :69-109
        if np.ndim(order_or_coeffs) == 0:
            if order_or_coeffs < 0:
                raise NonlinError(2)
            self._c = np.zeros(int(order_or_coeffs) + 1)
        else:
            self._c = np.array(order_or_coeffs, dtype=np.float64).ravel().copy()

    def order(self):                                          # :112-143
        return -1 if self._c is None else self._c.size - 1

    def _fit(self, x, y, order, thru_zero):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.size != x.size:
            raise NonlinError(3)                              # :159-162
        if order >= x.size or order < 1:
            raise NonlinError(4)                              # :163-166
        if self.order() != order:
            self.initialize(order)
        h = default_handle()
        rc = h.lib.nlh_poly_fit(h.ptr, x.size, int(order), int(thru_zero), _dp(x), _dp(y), _dp(self._c))
        h.check(rc, "nlh_poly_fit")
        if rc:
            raise NonlinError(rc)

    def fit(self, x, y, order): self._fit(x, y, order, False)                 # :146-190
    def fit_thru_zero(self, x, y, order): self._fit(x, y, order, True)        # :193-238

    def evaluate(self, x):                                    # :241-268, Horner from the top
        x = np.asarray(x, dtype=np.float64)
        order = self.order()
        if order == -1:
            return np.zeros_like(x)
        if order == 0:
            return np.full_like(x, self._c[0])
        y = self._c[order] * x + self._c[order - 1]
        for j in range(order - 2, -1, -1):
            y = y * x + self._c[j]
        return y

    def get(self, i): return float(self._c[i - 1])            # 1-based like the reference
    def get_all(self): return self._c.copy()
    def set(self, i, v): self._c[i - 1] = float(v)


class fcnnvar_helper:
    """src/nonlin_multi_var.f90:29-43: holder of a scalar objective fcn(x, args) -> float and an optional
    gradient routine grad(x, g, args)."""

    def __init__(self):
        self._fcn = None
        self._grad = None
        self._nvar = 0

    def set_fcn(self, fcn, nvar):                             # :99-106
        self._fcn = fcn
        self._nvar = int(nvar)

    def set_gradient_fcn(self, grad): self._grad = grad       # :116-120
    def is_fcn_defined(self): return self._fcn is not None
    def is_gradient_defined(self): return self._grad is not None
    def get_variable_count(self): return self._nvar

    def fcn(self, x, args=None):                              # :81-89
        return float(self._fcn(x, args)) if self._fcn is not None else 0.0

    def _c_fcn(self, args):
        f = self._fcn

        def _cb(ctx, n, xp):
            return float(f(np.ctypeslib.as_array(xp, shape=(n,)), args))
        return _lib.FCNNVAR(_cb)

    def _c_grad(self, args):
        if self._grad is None:
            return C.cast(None, _lib.GRADFCN)
        g = self._grad

        def _cb(ctx, n, xp, gp):
            g(np.ctypeslib.as_array(xp, shape=(n,)), np.ctypeslib.as_array(gp, shape=(n,)), args)
        return _lib.GRADFCN(_cb)


class equation_optimizer:
    """src/nonlin_multi_var.f90:45-57."""

    def __init__(self):
        self._max_eval = 500                      # :46
        self._tol = 1.0e-12                       # :47
        self._print = False
        self.handle = None

    def get_max_fcn_evals(self): return self._max_eval
    def set_max_fcn_evals(self, n): self._max_eval = int(n)
    def get_tolerance(self): return self._tol
    def set_tolerance(self, x): self._tol = float(x)
    def get_print_status(self): return self._print
    def set_print_status(self, x): self._print = bool(x)


class line_search_optimizer(equation_optimizer):
    """src/nonlin_optimize.f90:44-61."""

    def __init__(self):
        super().__init__()
        self._line_search = None
        self._use_line_search = True              # :46
        self._xtol = 1.0e-12                      # :47

    def get_line_search(self):
        if self._line_search is None:
            return None
        ls = line_search()
        ls.__dict__.update(self._line_search.__dict__)
        return ls

    def set_line_search(self, ls):
        c = line_search()
        c.__dict__.update(ls.__dict__)
        self._line_search = c

    def set_default_line_search(self): self._line_search = line_search()
    def is_line_search_defined(self): return self._line_search is not None
    def get_use_line_search(self): return self._use_line_search
    def set_use_line_search(self, x): self._use_line_search = bool(x)
    def get_var_tolerance(self): return self._xtol
    def set_var_tolerance(self, x): self._xtol = float(x)


class bfgs(line_search_optimizer):
    """src/nonlin_optimize.f90:69-72."""

    def solve(self, fcn, x, ib=None, args=None):
        """bfgs_solve (:557-770).  x: initial estimate -> minimiser.  Returns fout."""
        if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous):
            raise ValueError("x must be a contiguous float64 numpy array (it is updated in place)")
        if self.get_use_line_search() and not self.is_line_search_defined():
            self.set_default_line_search()        # :604-608
        if not fcn.is_fcn_defined():
            raise NonlinError(NL_UNDEFINED_FUNCTION_ERROR)   # :614
        n = fcn.get_variable_count()
        if x.shape != (n,):
            raise NonlinError(NL_INVALID_INPUT_ERROR)        # :615
        o = _lib.default_options()
        o.max_evals = self._max_eval
        o.gtol = self._tol
        o.xtol = self._xtol
        o.print_status = 1 if self._print else 0
        o.use_line_search = 1 if self._use_line_search else 0
        if self._line_search is not None:
            o.ls_max_evals = self._line_search._max_eval
            o.ls_alpha = self._line_search._alpha
            o.ls_factor = self._line_search._factor
        h = self.handle or default_handle()
        cib = _lib.IterationBehavior()
        fout = C.c_double(0.0)
        cf, cg = fcn._c_fcn(args), fcn._c_grad(args)
        rc = h.lib.nlh_bfgs_solve(h.ptr, C.byref(o), n, cf, cg, None, _dp(x), C.cast(C.byref(fout), _lib.c_double_p),
                                  C.byref(cib))
        h.check(rc, "nlh_bfgs_solve")
        if ib is not None:
            ib._fill(cib)
        if rc:
            raise NonlinError(rc)                 # :765-767
        return fout.value
