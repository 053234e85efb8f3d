"""lu_factor / solve_lu on the device: bitwise check against the oracle and host-side timing (min of 5), n from the command
line (default: sizes around every panel switch incl. a zero column, a NaN, integer ties).  profiles/r04_lu_n1024_kernel_stats.csv is
rocprofv3 --kernel-trace --stats of `python profiles/lu_time.py 1024` (six factorisations, ten solves)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nonlin_amd.device import DeviceSolver
from oracle import pyoracle as O
ds = DeviceSolver(0)
L = O.lib()
print("NLH_LU_PANEL", os.environ.get("NLH_LU_PANEL"), "NLH_LU_FAST", os.environ.get("NLH_LU_FAST"))
for n in [int(a) for a in sys.argv[1:]] or [128, 200, 257, 300, 513, 1024, 1500]:
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    if n == 200: a[:, 7] = 0.0
    if n == 257: a[5, 5] = np.nan
    if n == 300: a = rng.integers(-3, 4, size=(n, n)).astype(float)
    lu = np.array(a, order="F"); ipo = np.zeros(n, dtype=np.int32)
    rc = L.nlo_lu_factor(n, lu.ctypes.data_as(C.POINTER(C.c_double)), n, ipo.ctypes.data_as(C.POINTER(C.c_int32)))
    Ad = torch.tensor(np.ascontiguousarray(a.T), device="cuda").reshape(1, n, n)   # column-major problem
    A0 = Ad.clone()
    ipvt, info = ds.lu_factor(Ad)
    torch.cuda.synchronize()
    got = Ad[0].cpu().numpy().T
    same = np.array_equal(got, lu, equal_nan=True)
    t = []
    for _ in range(5):
        Ad.copy_(A0); torch.cuda.synchronize()
        t0 = time.perf_counter(); ds.lu_factor(Ad); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    err = np.nanmax(np.abs(got - lu)) if not same else 0.0
    print(f"n={n}: bitwise={same} pivots={np.array_equal(ipvt[0].cpu().numpy(), ipo)} info={int(info[0])}/{rc} maxdiff={err:.3e} time={1e3*min(t):.3f} ms", flush=True)
# solve timing
for n in [256, 1024]:
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n)); bvec = rng.standard_normal(n)
    lu = np.array(a, order="F"); ipo = np.zeros(n, dtype=np.int32)
    L.nlo_lu_factor(n, lu.ctypes.data_as(C.POINTER(C.c_double)), n, ipo.ctypes.data_as(C.POINTER(C.c_int32)))
    xo = bvec.copy()
    L.nlo_lu_solve(n, lu.ctypes.data_as(C.POINTER(C.c_double)), n, ipo.ctypes.data_as(C.POINTER(C.c_int32)), xo.ctypes.data_as(C.POINTER(C.c_double)))
    Ad = torch.tensor(np.ascontiguousarray(a.T), device="cuda").reshape(1, n, n)
    ipvt, info = ds.lu_factor(Ad)
    b0 = torch.tensor(bvec, device="cuda").reshape(1, n)
    t = []
    for _ in range(5):
        bd = b0.clone(); torch.cuda.synchronize()
        t0 = time.perf_counter(); ds.lu_solve(Ad, ipvt, bd); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    print(f"solve n={n}: bitwise={np.array_equal(bd[0].cpu().numpy(), xo)} time={1e3*min(t):.3f} ms", flush=True)
