"""Random sweep of lu_factor / solve_lu on the device against the oracle (sizes around every panel switch, ties, zero
columns, equal rows, scaled rows, scattered NaNs): factors, interchanges, info and solution bit for bit.  Not collected by
pytest; run on a GPU box: python tests/soak_lu.py"""
import sys, ctypes as C
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nonlin_amd.device import DeviceSolver
from oracle import pyoracle as O
ds = DeviceSolver(0); L = O.lib()
rng = np.random.default_rng(2026)
bad = 0; ncase = 0
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)); ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
for it in range(260):
    n = int(rng.choice([128, 129, 143, 160, 200, 255, 256, 257, 300, 383, 400, 512, 513, 600, 777, 1000, 1024, 1025, 1100]))
    kind = int(rng.integers(0, 6))
    a = rng.standard_normal((n, n))
    if kind == 1: a = rng.integers(-2, 3, size=(n, n)).astype(float)            # ties, singular
    if kind == 2: a[:, rng.integers(0, n)] = 0.0
    if kind == 3: a *= 10.0 ** rng.uniform(-8, 8, size=(n, 1))
    if kind == 4: a[rng.integers(0, n), :] = a[rng.integers(0, n), :]           # equal rows
    if kind == 5: a[rng.random((n, n)) < 0.001] = np.nan
    b = rng.standard_normal(n)
    lu = np.array(a, order="F"); ipo = np.zeros(n, dtype=np.int32)
    rc = L.nlo_lu_factor(n, dp(lu), n, ip(ipo))
    xo = b.copy(); L.nlo_lu_solve(n, dp(lu), n, ip(ipo), dp(xo))
    Ad = torch.tensor(np.ascontiguousarray(a.T), device="cuda").reshape(1, n, n)
    bd = torch.tensor(b, device="cuda").reshape(1, n)
    ipvt, info = ds.lu_factor(Ad); ds.lu_solve(Ad, ipvt, bd); torch.cuda.synchronize()
    ok = np.array_equal(Ad[0].cpu().numpy().T, lu, equal_nan=True) and np.array_equal(ipvt[0].cpu().numpy(), ipo) and int(info[0]) == rc
    if kind != 5: ok = ok and np.array_equal(bd[0].cpu().numpy(), xo, equal_nan=True)
    ncase += 1
    if not ok:
        bad += 1; print("MISMATCH n", n, "kind", kind, "info", int(info[0]), rc, flush=True)
print(f"lu soak: {ncase} cases, {bad} mismatches", flush=True)
