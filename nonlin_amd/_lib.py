"""ctypes loader for libnonlin_hip.so (the C ABI declared in include/nonlin_hip.h).

The product path has no CPU fallback: if the shared library is missing, or no GPU is
visible when a compute entry point is called, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnonlin_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)

VECFCN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, c_double_p, C.c_int32, c_double_p)
JACFCN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, c_double_p, C.c_int32, c_double_p)
FCNNVAR = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int32, c_double_p)
GRADFCN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, c_double_p, c_double_p)
# nlh_device_vecfcn / nlh_device_jacfcn: launchers (ctx, hip_stream, npoints, dprob, n, dX, m, dF | dJ) -> int
DEVFCN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p)


class IterationBehavior(C.Structure):
    """nlh_iteration_behavior == iteration_behavior (src/nonlin_types.f90:8-29)."""
    _fields_ = [(k, C.c_int32) for k in (
        "iter_count", "fcn_count", "jacobian_count", "gradient_count",
        "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class DqDeviceCtx(C.Structure):
    """nlh_dq_device_ctx: the dense-quadratic family behind the launchers nlh_dq_device_fcn / nlh_dq_device_jac."""
    _fields_ = [("dA", C.c_void_p), ("db", C.c_void_p), ("gamma", C.c_double)]


class Options(C.Structure):
    """nlh_options."""
    _fields_ = [("max_evals", C.c_int32), ("ftol", C.c_double), ("xtol", C.c_double),
                ("gtol", C.c_double), ("print_status", C.c_int32), ("factor", C.c_double),
                ("use_line_search", C.c_int32), ("ls_max_evals", C.c_int32),
                ("ls_alpha", C.c_double), ("ls_factor", C.c_double),
                ("factor_policy", C.c_int32), ("ne_pivot_tol", C.c_double), ("fuse_fd", C.c_int32),
                ("sub_batches", C.c_int32)]


# every symbol include/nonlin_hip.h declares: name -> (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "nlh_default_options": (None, [C.POINTER(Options)]),
    "nlh_format_status": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_char_p, C.c_int32]),
    "nlh_create": (C.c_int, [C.POINTER(_H), C.c_int32, C.c_void_p]),
    "nlh_destroy": (None, [_H]),
    "nlh_device_count": (C.c_int, []),
    "nlh_last_error": (C.c_char_p, [_H]),
    "nlh_version": (C.c_char_p, []),
    "nlh_fd_jacobian": (C.c_int, [_H, C.c_int32, C.c_int32, VECFCN, JACFCN, C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "nlh_lm_solve": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, VECFCN, JACFCN, C.c_void_p,
                               c_double_p, c_double_p, C.POINTER(IterationBehavior)]),
    "nlh_newton_solve": (C.c_int, [_H, C.POINTER(Options), C.c_int32, VECFCN, JACFCN, C.c_void_p,
                                   c_double_p, c_double_p, C.POINTER(IterationBehavior)]),
    "nlh_quasi_newton_solve": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, VECFCN, JACFCN, C.c_void_p,
                                         c_double_p, c_double_p, C.POINTER(IterationBehavior)]),
    "nlh_cls_solve": (C.c_int, [_H, C.POINTER(Options), C.c_double, C.c_double, c_double_p, c_double_p, C.c_int32, C.c_int32,
                                VECFCN, JACFCN, C.c_void_p, c_double_p, c_double_p, C.POINTER(IterationBehavior)]),
    "nlh_fd_gradient": (C.c_int, [C.c_int32, FCNNVAR, GRADFCN, C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "nlh_bfgs_solve": (C.c_int, [_H, C.POINTER(Options), C.c_int32, FCNNVAR, GRADFCN, C.c_void_p, c_double_p, c_double_p,
                                 C.POINTER(IterationBehavior)]),
    "nlh_dq_lm_solve_batch": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                        C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_dq_newton_solve_batch": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.c_double, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_dq_quasi_newton_solve_batch": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                                  C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p,
                                                  C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_dq_cls_solve_batch": (C.c_int, [_H, C.POINTER(Options), C.c_double, C.c_double, c_double_p, c_double_p, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                         C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_dq_bfgs_solve_batch": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_double, C.c_void_p, c_double_p, C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_dq_generate": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_double, C.c_double,
                                  C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nlh_dq_residual": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double,
                                  C.c_void_p, C.c_void_p]),
    "nlh_dq_fd_panel": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double,
                                  C.c_void_p, C.c_void_p]),
    "nlh_fd_jacobian_panel": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "nlh_dq_jacobian": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_double, C.c_void_p,
                                  C.c_void_p]),
    "nlh_gram": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nlh_chol_factor": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "nlh_qr_factor": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nlh_lmfactor_exact": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nlh_lmpar": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nlh_lu_factor": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nlh_lu_solve": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nlh_qr_factor_full": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nlh_qr_rank1_update": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nlh_solve_upper": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "nlh_chol_rank1": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, c_int32_p]),
    "nlh_poly_fit": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p]),
    "nlh_poly_fit_batch": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nlh_dq_model_create": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, c_double_p, c_double_p, C.c_double, C.POINTER(C.c_void_p)]),
    "nlh_device_set_create": (C.c_int, [C.POINTER(C.c_void_p), c_int32_p, C.c_int32]),
    "nlh_device_set_destroy": (None, [C.c_void_p]),
    "nlh_device_set_size": (C.c_int32, [C.c_void_p]),
    "nlh_device_set_handle": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "nlh_device_set_last_error": (C.c_char_p, [C.c_void_p]),
    "nlh_dq_model_create_on": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_double_p, c_double_p, C.c_double,
                                         C.POINTER(C.c_void_p)]),
    "nlh_dq_model_device_count": (C.c_int32, [C.c_void_p]),
    "nlh_dq_model_destroy": (None, [C.c_void_p]),
    "nlh_dq_model_shape": (None, [C.c_void_p, c_int32_p, c_int32_p, c_int32_p]),
    "nlh_dq_model_eval": (C.c_int, [_H, C.c_void_p, c_double_p, c_double_p]),
    "nlh_dq_model_lm_solve": (C.c_int, [_H, C.POINTER(Options), C.c_void_p, c_double_p, c_double_p,
                                        C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_dq_model_newton_solve": (C.c_int, [_H, C.POINTER(Options), C.c_void_p, C.c_int32, c_double_p, c_double_p,
                                            C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_dq_model_quasi_newton_solve": (C.c_int, [_H, C.POINTER(Options), C.c_void_p, C.c_int32, C.c_int32, c_double_p, c_double_p,
                                                  C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_dq_model_cls_solve": (C.c_int, [_H, C.POINTER(Options), C.c_void_p, C.c_double, C.c_double, c_double_p, c_double_p,
                                         c_double_p, c_double_p, C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_dq_model_bfgs_solve": (C.c_int, [_H, C.POINTER(Options), C.c_void_p, c_double_p, c_double_p, c_double_p,
                                          C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_fd_jacobian_device": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, DEVFCN, DEVFCN, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "nlh_lm_solve_batch_device": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, C.c_int32, DEVFCN, DEVFCN, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_newton_solve_batch_device": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, DEVFCN, DEVFCN, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_quasi_newton_solve_batch_device": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, C.c_int32, DEVFCN, DEVFCN,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_cls_solve_batch_device": (C.c_int, [_H, C.POINTER(Options), C.c_double, C.c_double, c_double_p, c_double_p, C.c_int32, C.c_int32,
                                             C.c_int32, DEVFCN, DEVFCN, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_cls_solve_batch_device_h": (C.c_int, [_H, C.POINTER(Options), C.c_double, C.c_double, c_double_p, c_double_p, C.c_int32, C.c_int32,
                                               C.c_int32, DEVFCN, DEVFCN, C.c_void_p, c_double_p, c_double_p,
                                               C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_bfgs_solve_batch_device": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, DEVFCN, DEVFCN, C.c_void_p, C.c_void_p, c_double_p,
                                              C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_bfgs_solve_batch_device_h": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, DEVFCN, DEVFCN, C.c_void_p, c_double_p, c_double_p,
                                                C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_lm_solve_batch_device_h": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, C.c_int32, DEVFCN, DEVFCN, C.c_void_p,
                                              c_double_p, c_double_p, C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_newton_solve_batch_device_h": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, DEVFCN, DEVFCN, C.c_void_p,
                                                  c_double_p, c_double_p, C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_quasi_newton_solve_batch_device_h": (C.c_int, [_H, C.POINTER(Options), C.c_int32, C.c_int32, C.c_int32, DEVFCN, DEVFCN,
                                                        C.c_void_p, c_double_p, c_double_p, C.POINTER(IterationBehavior), c_int32_p]),
    "nlh_device_fcn_model_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, DEVFCN, DEVFCN, C.c_void_p, C.POINTER(C.c_void_p)]),
    "nlh_dq_device_fcn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "nlh_dq_device_jac": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "nlh_timing_enable": (None, [_H, C.c_int32]),
    "nlh_timing_reset": (None, [_H]),
    "nlh_timing_get": (C.c_int, [_H, C.c_int32, c_double_p, C.POINTER(C.c_int64)]),
    "nlh_timing_samples": (C.c_int64, [_H, C.c_int32, C.POINTER(C.c_float), C.c_int64]),
    "nlh_kernel_name": (C.c_char_p, [C.c_int32]),
}

KERNEL_IDS = {
    "dq_residual": 0, "dq_panel": 1, "fd_jacobian": 2, "gram": 3, "gram_reduce": 4, "jtf": 5,
    "chol": 6, "lmpar": 7, "qr": 8, "update": 9, "lu": 10, "dq_jacobian": 11, "qrx_pass": 12, "qrx_pivot": 13,
}

_lib = None


class NonlinHipUnavailable(RuntimeError):
    pass


def load():
    """Load libnonlin_hip.so and bind every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NonlinHipUnavailable(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nonlin_amd/csrc` (there is no CPU fallback)")
    # PyTorch ships its own HIP runtime.  Two runtimes in one process coexist only if torch's is loaded first (the other way
    # round torch.cuda reports no device, or this library does): import torch before the library whenever it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the ABI and this table drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def default_options():
    o = Options()
    load().nlh_default_options(C.byref(o))
    return o


class Handle:
    """Owns an nlh_handle bound to a device and (optionally) an existing HIP stream."""

    def __init__(self, device=0, stream=None):
        lib = load()
        if lib.nlh_device_count() <= 0:
            raise NonlinHipUnavailable("no HIP device visible: the nonlin_amd compute path needs a GPU "
                                       "(there is no CPU fallback)")
        self._h = C.c_void_p()
        rc = lib.nlh_create(C.byref(self._h), int(device), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise NonlinHipUnavailable(f"nlh_create failed with {rc}")
        self.lib = lib

    @property
    def ptr(self):
        return self._h

    def check(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what}: library error {rc}: {self.lib.nlh_last_error(self._h).decode()}")
        return rc

    def timing_enable(self, on=True, kernels=None):
        """on: every kernel group; kernels=[names]: only those groups (two event records per timed launch)."""
        if kernels:
            mask = 0
            for k in kernels:
                mask |= 1 << (KERNEL_IDS[k] + 1)
            self.lib.nlh_timing_enable(self._h, mask)
        else:
            self.lib.nlh_timing_enable(self._h, 1 if on else 0)

    def timing_reset(self):
        self.lib.nlh_timing_reset(self._h)

    def timing(self, kernel):
        """Returns (total_ms, launches) measured with HIP events on the handle's stream."""
        ms = C.c_double(0.0)
        cnt = C.c_int64(0)
        self.lib.nlh_timing_get(self._h, KERNEL_IDS[kernel], C.byref(ms), C.byref(cnt))
        return ms.value, cnt.value

    def timing_samples(self, kernel, select_only=False):
        """Per-launch durations (ms, launch order) of one kernel group since the last timing_reset.  The group must have
        been selected before the launches (select_only=True) and be enabled in timing_enable."""
        kid = KERNEL_IDS[kernel]
        n = self.lib.nlh_timing_samples(self._h, kid, None, 0)
        if select_only or n <= 0:
            return []
        buf = (C.c_float * n)()
        n = self.lib.nlh_timing_samples(self._h, kid, buf, n)
        return [float(buf[i]) for i in range(n)]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.nlh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
