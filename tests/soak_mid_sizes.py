"""Mid-size random sweep of the lock-step solvers (sizes that take the blocked LU panels, the blocked Cholesky, the blocked
triangular solves and the Householder step's big-tile forms): every problem must carry the oracle's bits."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nonlin_amd.device import DeviceSolver
from oracle import pyoracle as O
ds = DeviceSolver(0)
rng = np.random.default_rng(555)
CK = ("iter_count", "fcn_count", "jacobian_count", "gradient_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")
bad = 0; tot = 0
def cmp(tag, xg, ibg, r):
    global bad, tot
    tot += 1
    xo, ibo = r[1], r[3]
    ok = np.array_equal(xg, xo) and all(ibg[k] == ibo[k] for k in CK)
    if not ok:
        bad += 1; print("MISMATCH", tag, [(k, ibg[k], ibo[k]) for k in CK if ibg[k] != ibo[k]], flush=True)
for case in range(36):
    kind = case % 4
    if kind == 0:      # Newton
        n = int(rng.choice([130, 200, 257, 300, 520])); nb = int(rng.integers(1, 4))
        A, b, xt, x0 = ds.generate(nb, n, n, seed0=int(rng.integers(1, 10**6)), square_shift=True)
        x = x0.clone(); analytic = bool(rng.integers(0, 2))
        fv, ibs, st = ds.newton_solve_batch(A, b, 0.5, x, analytic=analytic)
        for p in range(nb):
            r = O.dq_newton_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(), analytic=analytic)
            cmp(("newton", n, p), x[p].cpu().numpy(), ibs[p], r)
    elif kind == 1:    # quasi-Newton
        n = int(rng.choice([130, 200, 300])); nb = int(rng.integers(1, 3))
        A, b, xt, x0 = ds.generate(nb, n, n, seed0=int(rng.integers(1, 10**6)), square_shift=True)
        x = x0.clone()
        fv, ibs, st = ds.quasi_newton_solve_batch(A, b, 0.5, x, analytic=True)
        for p in range(nb):
            r = O.dq_quasi_newton_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(), analytic=True)
            cmp(("qn", n, p), x[p].cpu().numpy(), ibs[p], r)
    elif kind == 2:    # bounded least squares
        n = int(rng.choice([60, 128, 150])); m = int(rng.choice([700, 1500, 3000])); nb = int(rng.integers(1, 3))
        A, b, xt, x0 = ds.generate(nb, m, n, seed0=int(rng.integers(1, 10**6)))
        x = x0.clone()
        fv, ibs, st = ds.cls_solve_batch(A, b, 0.5, x)
        for p in range(nb):
            r = O.dq_cls_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy())
            cmp(("cls", m, n, p), x[p].cpu().numpy(), ibs[p], r)
    else:              # BFGS
        n = int(rng.choice([70, 130, 200, 300])); m = 2 * n + int(rng.integers(0, 500)); nb = int(rng.integers(1, 3))
        A, b, xt, x0 = ds.generate(nb, m, n, seed0=int(rng.integers(1, 10**6)))
        x = x0.clone()
        fo, ibs, st = ds.bfgs_solve_batch(A, b, 0.5, x, ds.options(max_evals=300))
        for p in range(nb):
            r = O.dq_bfgs_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(), O.default_options(max_evals=300))
            cmp(("bfgs", m, n, p), x[p].cpu().numpy(), ibs[p], r)
print(f"mid soak: {tot} problems, {bad} mismatches", flush=True)
