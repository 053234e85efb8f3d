#!/usr/bin/env python3
"""Go / no-go study for a parity-grade FAST factorisation (round 4 review, item N1): can ANY factorisation other than the
reference's own operation order keep the LM iterates within north_star's 1e-10 of the reference on the benchmark family?

The rest of the iteration is the CPU oracle's lss_solve, bit for bit (FD Jacobian, lmpar with both deviations, ratio test,
convergence tests); only lmfactor + Q^T f (src/nonlin_least_squares.f90:225, 241-253) are replaced, through the oracle's
test-only hook, by fp64 numpy versions of

  (0) control   lmfactor itself (the oracle's C routine) + the Q^T f sweep: must reproduce the oracle exactly (hook plumbing);
  (i) hh_tree   the SAME pivoted Householder algorithm (MINPACK qrfac) with numpy / BLAS reductions -- i.e. nothing but the
                summation ORDER of the dot products and norms changes (what a tree-reduced GPU QR does);
  (ii) chol     pivoted Cholesky of J^T J (MINPACK's pivot rule on the Schur diagonals), qtf = R^-T P^T J^T f;
  (iii) cholqr2 CholeskyQR2: Q1 = J P R1^-1, second Gram Q1^T Q1 = R2^T R2, R = R2 R1, qtf = R2^-T Q1^T f.

Families: SURVEY 8(d) dense-quadratic, 4096 x 256, seeds 12345.., sigma = 1e-3 (the bench workload) and sigma = 0
(zero residual).  Writes tests/golden/fast_policy_study.json; tests/test_oracle.py checks the file's conclusion stands.
Run from the repo root:  python tests/golden/make_fast_policy_study.py [nproblems]   (about 10 minutes for 32)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O   # noqa: E402

EPS = np.finfo(float).eps
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, dp, C.c_int32, dp, ip, dp, dp, dp, dp)


def hh_tree(J, f):
    """MINPACK qrfac, statement for statement, with numpy reductions; then Q^T f by the stored reflectors."""
    a = np.array(J, order="F", copy=True)
    m, n = a.shape
    acnorm = np.sqrt(np.einsum("ij,ij->j", a, a))
    rdiag, wa, ipvt = acnorm.copy(), acnorm.copy(), np.arange(n)
    for j in range(n):
        kmax = j + int(np.argmax(rdiag[j:]))                       # first maximum, as the strict '>' scan
        if kmax != j:
            a[:, [j, kmax]] = a[:, [kmax, j]]
            rdiag[kmax], wa[kmax] = rdiag[j], wa[j]
            ipvt[[j, kmax]] = ipvt[[kmax, j]]
        ajnorm = np.sqrt(a[j:, j] @ a[j:, j])
        if ajnorm == 0.0:
            rdiag[j] = -ajnorm
            continue
        if a[j, j] < 0.0:
            ajnorm = -ajnorm
        a[j:, j] /= ajnorm
        a[j, j] += 1.0
        if j + 1 < n:
            t = (a[j:, j] @ a[j:, j + 1:]) / a[j, j]
            a[j:, j + 1:] -= np.outer(a[j:, j], t)
            r = rdiag[j + 1:]
            nz = r != 0.0
            temp = np.where(nz, a[j, j + 1:] / np.where(nz, r, 1.0), 0.0)
            r[nz] = (r * np.sqrt(np.maximum(0.0, 1.0 - temp * temp)))[nz]
            redo = nz & (0.05 * (r / wa[j + 1:]) ** 2 <= EPS)
            if redo.any() and j + 1 < m:
                cols = np.nonzero(redo)[0] + j + 1
                nr = np.sqrt(np.einsum("ij,ij->j", a[j + 1:, cols], a[j + 1:, cols]))
                rdiag[cols], wa[cols] = nr, nr
        rdiag[j] = -ajnorm
    w = np.array(f, copy=True)
    for j in range(n):
        if a[j, j] != 0.0:
            w[j:] += a[j:, j] * (-(a[j:, j] @ w[j:]) / a[j, j])
        a[j, j] = rdiag[j]
    return np.triu(a[:n, :n]), ipvt, rdiag, acnorm, w[:n].copy(), w


def _pivoted_cholesky(G):
    n = G.shape[0]
    A = G.copy()
    perm = np.arange(n)
    R = np.zeros((n, n))
    for j in range(n):
        k = j + int(np.argmax(np.diag(A)[j:]))
        if k != j:
            A[[j, k], :] = A[[k, j], :]
            A[:, [j, k]] = A[:, [k, j]]
            R[:, [j, k]] = R[:, [k, j]]
            perm[[j, k]] = perm[[k, j]]
        d = np.sqrt(A[j, j])
        R[j, j] = d
        R[j, j + 1:] = A[j, j + 1:] / d
        A[j + 1:, j + 1:] -= np.outer(R[j, j + 1:], R[j, j + 1:])
    return R, perm


def _tail(f, qtf, m):
    w = np.zeros(m)
    n = len(qtf)
    w[:n] = qtf
    if m > n:
        w[n] = np.sqrt(max(0.0, f @ f - qtf @ qtf))
    return w


def chol(J, f):
    G = J.T @ J
    R, perm = _pivoted_cholesky(G)
    qtf = sl.solve_triangular(R, (J.T @ f)[perm], trans="T", lower=False)
    return R, perm, np.diag(R).copy(), np.sqrt(np.diag(G)), qtf, _tail(f, qtf, J.shape[0])


def cholqr2(J, f):
    G = J.T @ J
    R1, perm = _pivoted_cholesky(G)
    Q1 = sl.solve_triangular(R1, J[:, perm].T, trans="T", lower=False).T          # J P R1^-1
    R2 = np.linalg.cholesky(Q1.T @ Q1).T
    R = R2 @ R1
    qtf = sl.solve_triangular(R2, Q1.T @ f, trans="T", lower=False)
    return np.triu(R), perm, np.diag(R).copy(), np.sqrt(np.diag(G)), qtf, _tail(f, qtf, J.shape[0])


def control(J, f):
    """lmfactor itself (C) + the Q^T f sweep with sequential sums: the oracle's own bits through the hook."""
    L = O.lib()
    a = np.array(J, order="F", copy=True)
    m, n = a.shape
    ipvt = np.zeros(n, dtype=np.int32)
    rdiag, acnorm, wa = np.zeros(n), np.zeros(n), np.zeros(n)
    L.nlo_lmfactor(m, n, a.ctypes.data_as(dp), m, 1, ipvt.ctypes.data_as(ip), rdiag.ctypes.data_as(dp), acnorm.ctypes.data_as(dp),
                   wa.ctypes.data_as(dp))
    w = np.array(f, copy=True)
    for j in range(n):
        if a[j, j] != 0.0:
            sm = 0.0
            col = a[j:, j]
            wj = w[j:]
            for i in range(len(col)):
                sm = sm + col[i] * wj[i]
            temp = -sm / a[j, j]
            w[j:] = wj + col * temp
        a[j, j] = rdiag[j]
    return np.triu(a[:n, :n]), ipvt, rdiag, acnorm, w[:n].copy(), w


def solve_with(method, A, b, gamma, x0, max_evals=500):
    m, n = A.shape
    prob = O._dq_problem(A, b, gamma)

    def hook(_, m_, n_, jac, lda, fvec, jpvt, rdiag, acnorm, qtf, wa4):
        J = np.ctypeslib.as_array(jac, shape=(n_, lda)).T[:m_, :]              # column-major m x n view
        f = np.ctypeslib.as_array(fvec, shape=(m_,))
        R, perm, rd, ac, q, w = method(np.array(J), np.array(f))
        J[:n_, :n_] = np.triu(R) + np.tril(J[:n_, :n_], -1)
        np.ctypeslib.as_array(jpvt, shape=(n_,))[:] = perm
        np.ctypeslib.as_array(rdiag, shape=(n_,))[:] = rd
        np.ctypeslib.as_array(acnorm, shape=(n_,))[:] = ac
        np.ctypeslib.as_array(qtf, shape=(n_,))[:] = q
        np.ctypeslib.as_array(wa4, shape=(m_,))[:] = w
    L = O.lib()
    L.nlo_set_factor_hook.argtypes = [HOOK, C.c_void_p]
    cb = HOOK(hook)
    oo = O.default_options(max_evals=max_evals)
    x, fv, ib = x0.copy(), np.zeros(m), O.IterationBehavior()
    L.nlo_set_factor_hook(cb, None)
    try:
        rc = L.nlo_lm_solve(C.byref(oo), C.cast(L.nlo_dq_fcn, O.VECFCN), C.cast(None, O.JACFCN), C.byref(prob), m, n,
                            x.ctypes.data_as(dp), fv.ctypes.data_as(dp), C.byref(ib))
    finally:
        L.nlo_set_factor_hook(C.cast(None, HOOK), None)
    return rc, x, ib.as_dict()


def main():
    nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    m, n, gamma = 4096, 256, 0.5
    methods = {"hh_tree": hh_tree, "chol": chol, "cholqr2": cholqr2}
    out = {"generated_by": "tests/golden/make_fast_policy_study.py", "shape": [m, n], "problems_per_family": nprob, "seed0": 12345,
           "bar": 1e-10, "families": {}}
    t00 = time.time()
    for fam, sigma in (("sigma_1e-3 (bench workload)", 1e-3), ("sigma_0 (zero residual)", 0.0)):
        res = {k: {"rel_dev_x": [], "count_mismatches": 0} for k in methods}
        ctrl = None
        for k in range(nprob):
            A, b, xt, x0 = O.dq_generate(12345 + k, m, n, gamma=gamma, sigma=sigma, spread=0.3)
            rc0, x_ref, f_ref, ib_ref = O.dq_lm_solve(A, b, gamma, x0, opts=O.default_options(max_evals=500))[:4]
            if k == 0:                                           # the plumbing check: the oracle's own factorisation through the hook
                rc, x, ib = solve_with(control, A, b, gamma, x0)
                ctrl = bool(np.array_equal(x, x_ref) and all(ib[q] == ib_ref[q] for q in ("iter_count", "fcn_count", "jacobian_count")))
            for name, fn in methods.items():
                rc, x, ib = solve_with(fn, A, b, gamma, x0)
                res[name]["rel_dev_x"].append(float(np.abs(x - x_ref).max() / np.abs(x_ref).max()))
                res[name]["count_mismatches"] += int(any(ib[q] != ib_ref[q] for q in ("iter_count", "fcn_count", "jacobian_count")))
            print(fam, k, {nm: "%.2e" % res[nm]["rel_dev_x"][-1] for nm in methods}, "%.0f s" % (time.time() - t00), flush=True)
        fo = {"control_reproduces_oracle_bitwise": ctrl}
        for name in methods:
            d = np.array(res[name]["rel_dev_x"])
            fo[name] = {"max_rel_dev_x": float(d.max()), "median_rel_dev_x": float(np.median(d)), "min_rel_dev_x": float(d.min()),
                        "problems_within_1e-10": int((d <= 1e-10).sum()), "count_mismatches": res[name]["count_mismatches"],
                        "problems": nprob}
        out["families"][fam] = fo
    worst = max(out["families"][f][mth]["max_rel_dev_x"] for f in out["families"] for mth in methods)
    best_on_bench = min(out["families"]["sigma_1e-3 (bench workload)"][mth]["max_rel_dev_x"] for mth in methods)
    out["conclusion"] = {
        "any_method_meets_1e-10_on_the_bench_family": bool(best_on_bench <= 1e-10),
        "best_max_rel_dev_x_on_the_bench_family": best_on_bench, "worst_max_rel_dev_x": worst,
        "reading": "hh_tree changes nothing but the summation order of lmfactor's reductions; if even that leaves the iterates "
                   "outside 1e-10, no factorisation that is not operation-order-identical to the reference can carry parity "
                   "on this family, and the MFMA normal-equations policy stays an opt-in with its measured deviation"}
    with open(os.path.join(ROOT, "tests", "golden", "fast_policy_study.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["conclusion"], indent=1))


if __name__ == "__main__":
    main()
