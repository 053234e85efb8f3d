"""Loader and problem generators for the user-side residual families of tests/device_model/user_models.hip: what a user of
the open device-residual path (include/nonlin_hip.h: nlh_device_vecfcn) would write -- a kernel, a launcher, and here also
the same arithmetic as a host vecfcn, which the CPU oracle drives.  Test / bench infrastructure, not part of the product."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "device_model", "libuser_models.so")
dp = C.POINTER(C.c_double)


class LorentzHost(C.Structure):
    _fields_ = [("m", C.c_int32), ("t", dp), ("y", dp), ("ncalls", C.c_int64)]


class BtriHost(C.Structure):
    _fields_ = [("c", C.c_double), ("ncalls", C.c_int64), ("njcalls", C.c_int64)]


_so = None


def lib():
    global _so
    if _so is None:
        if not os.path.exists(SO):
            subprocess.check_call(["make", "-C", os.path.dirname(SO), "-s"])
        try:
            import torch  # noqa: F401  (its HIP runtime first, as for libnonlin_hip.so)
        except ImportError:
            pass
        so = C.CDLL(SO)
        so.lorentz_create.restype = C.c_void_p
        so.lorentz_create.argtypes = [C.c_int32, C.c_int32, dp, dp]
        so.lorentz_destroy.argtypes = [C.c_void_p]
        so.btri_create.restype = C.c_void_p
        so.btri_create.argtypes = [C.c_int32, dp]
        so.btri_destroy.argtypes = [C.c_void_p]
        so.crosen_host_f.restype = C.c_double
        so.crosen_host_f.argtypes = [C.c_double, C.c_int32, dp]
        so.crosen_host_grad.argtypes = [C.c_double, C.c_int32, dp, dp]
        _so = so
    return _so


def lorentz_row_numpy(x, t, y):
    """The residual in numpy, term by term in the same order (IEEE operations: the same bits)."""
    s = np.zeros_like(t)
    for k in range(0, len(x), 3):
        d = (t - x[k + 1]) / x[k + 2]
        q = 1.0 + d * d
        s = s + x[k] / q
    return s - y


def lorentz_problems(nprob, m, K, seed=2024, sigma=1e-3, spread=0.05, hard_every=0):
    """nprob spectra of K peaks on m abscissae in [0, 1]: returns t, y [nprob, m], x_true, x0 [nprob, 3K].
    hard_every > 0: every hard_every-th problem starts four times further away (different iteration counts in one batch)."""
    rng = np.random.default_rng(seed)
    n = 3 * K
    t = np.tile(np.linspace(0.0, 1.0, m), (nprob, 1)) + rng.uniform(-0.2, 0.2, (nprob, m)) / m
    xt = np.empty((nprob, n))
    xt[:, 0::3] = rng.uniform(0.5, 1.5, (nprob, K))
    xt[:, 1::3] = (np.arange(K) + 0.5) / K + rng.uniform(-0.15, 0.15, (nprob, K)) / K
    xt[:, 2::3] = rng.uniform(0.15, 0.35, (nprob, K)) / K
    y = np.empty((nprob, m))
    for p in range(nprob):
        y[p] = lorentz_row_numpy(xt[p], t[p], np.zeros(m)) + sigma * rng.uniform(-1, 1, m)
    sp = np.full((nprob, 1), spread)
    if hard_every:
        sp[::hard_every] *= 4.0
    x0 = xt * (1.0 + sp * rng.uniform(-1, 1, (nprob, n)))
    return np.ascontiguousarray(t), np.ascontiguousarray(y), xt, np.ascontiguousarray(x0)


class LorentzBatch:
    """Device context (every problem) + per-problem host contexts of one batch of spectra."""

    def __init__(self, t, y):
        so = lib()
        self.t, self.y = np.ascontiguousarray(t), np.ascontiguousarray(y)
        self.nprob, self.m = self.t.shape
        self.ctx = so.lorentz_create(self.nprob, self.m, self.t.ctypes.data_as(dp), self.y.ctypes.data_as(dp))
        if not self.ctx:
            raise RuntimeError("lorentz_create failed (no GPU?)")
        self.launch = so.lorentz_launch
        self.host_fcn = so.lorentz_host_fcn

    def host_ctx(self, p):
        return LorentzHost(self.m, self.t[p].ctypes.data_as(dp), self.y[p].ctypes.data_as(dp), 0)

    def close(self):
        if self.ctx:
            lib().lorentz_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def btri_problems(nprob, n, seed=7, spread=0.3):
    rng = np.random.default_rng(seed)
    c = 1.0 + rng.uniform(-0.5, 0.5, nprob)
    x0 = -1.0 + spread * rng.uniform(-1, 1, (nprob, n))       # the standard start is x = -1
    return np.ascontiguousarray(c), np.ascontiguousarray(x0)


def crosen_problems(nprob, n, seed=11, spread=0.4):
    """Chained-Rosenbrock objectives (family 3: a scalar fcnnvar for bfgs): per-problem target c, starts around -0.5."""
    rng = np.random.default_rng(seed)
    c = 1.0 + rng.uniform(-0.3, 0.3, nprob)
    x0 = -0.5 + spread * rng.uniform(-1, 1, (nprob, n))
    return np.ascontiguousarray(c), np.ascontiguousarray(x0)


class BtriBatch:
    def __init__(self, c):
        so = lib()
        self.c = np.ascontiguousarray(c, dtype=np.float64)
        self.nprob = len(self.c)
        self.ctx = so.btri_create(self.nprob, self.c.ctypes.data_as(dp))
        if not self.ctx:
            raise RuntimeError("btri_create failed (no GPU?)")
        self.launch, self.launch_jac = so.btri_launch, so.btri_launch_jac
        self.host_fcn, self.host_jac = so.btri_host_fcn, so.btri_host_jac
        self.crosen_launch, self.crosen_launch_grad = so.crosen_launch, so.crosen_launch_grad   # family 3 shares the context (one c per problem)

    def host_ctx(self, p):
        return BtriHost(float(self.c[p]), 0, 0)

    def close(self):
        if self.ctx:
            lib().btri_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
