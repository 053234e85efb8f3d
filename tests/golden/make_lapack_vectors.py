#!/usr/bin/env python3
"""Fixtures that pin the oracle's restatements of the un-vendored `linalg` pieces to LAPACK itself.

The reference's dense factorisations live in jchristopherson/linalg (fpm.toml:13-15), which is a thin layer over LAPACK:
lu_factor -> DGETRF, solve_lu -> DGETRS (call sites src/nonlin_solve.f90:570,577), qr_factor with Q formed -> DGEQRF /
DORGQR (src/nonlin_solve.f90:286; src/nonlin_least_squares.f90:1061 keeps the reflectors and applies them to the residual:
DORMQR), qr_rank1_update (src/nonlin_solve.f90:303), the Cholesky pieces of BFGS (src/nonlin_optimize.f90:713-724;
DPOTRF here pins the factor the normal-equations policy restates).  Neither linalg nor a Fortran LAPACK is in /root/reference, but scipy in the build container carries LAPACK
(OpenBLAS): this script runs those routines on seeded matrices and records inputs and outputs.  LAPACK's blocked routines do
not fix the order of their sums (DGETRF's bits differ between reference LAPACK, OpenBLAS and MKL), so what is pinned is what
every LAPACK agrees on: the PIVOT SEQUENCE exactly, the factors and solutions to a few ulp of the matrix norm.

Run in the build container only (needs scipy):  python tests/golden/make_lapack_vectors.py
Writes tests/golden/lapack_vectors.npz (inputs and LAPACK outputs, float64 / int32 arrays)."""
import os

import numpy as np
import scipy
from scipy.linalg import lapack, qr_update

HERE = os.path.dirname(os.path.abspath(__file__))


def lu_cases(rng):
    out = {}
    for n in (2, 5, 17, 64, 96, 160):
        a = rng.standard_normal((n, n))
        if n == 17:                                   # magnitudes over twelve decades: interchanges at every step
            a *= 10.0 ** rng.uniform(-6, 6, size=(n, 1))
        out[f"n{n}"] = a
    a = rng.standard_normal((12, 12))                 # exactly singular: columns 4 and 9 equal (info > 0 or a tiny pivot)
    a[:, 9] = a[:, 4]
    out["singular12"] = a
    a = np.zeros((6, 6))                              # a zero column at step 3: LAPACK reports info = 3 and goes on
    a[:] = rng.standard_normal((6, 6))
    a[:, 2] = 0.0
    out["zerocol6"] = a
    return out


def main():
    rng = np.random.default_rng(20261004)
    store = {"scipy_version": np.array(scipy.__version__)}
    names = []
    for name, a in lu_cases(rng).items():
        lu, piv, info = lapack.dgetrf(np.asfortranarray(a))
        b = rng.standard_normal(a.shape[0])
        store[f"lu_{name}_a"] = a
        store[f"lu_{name}_lu"] = lu
        store[f"lu_{name}_piv"] = piv.astype(np.int32)          # 0-based row interchanged with row i
        store[f"lu_{name}_info"] = np.array(info, dtype=np.int32)
        store[f"lu_{name}_b"] = b
        if info == 0:
            x, i2 = lapack.dgetrs(lu, piv, b)
            assert i2 == 0
            store[f"lu_{name}_x"] = x
        names.append(name)
    store["lu_names"] = np.array(names)

    qnames = []
    for n in (3, 10, 40):                               # QR with Q formed (quasi-Newton start) + a rank-one update
        a = rng.standard_normal((n, n))
        qr, tau, work, info = lapack.dgeqrf(np.asfortranarray(a))
        assert info == 0
        r = np.triu(qr)
        q, work, info = lapack.dorgqr(qr, tau)
        assert info == 0
        u, v = rng.standard_normal(n), rng.standard_normal(n)
        q1, r1 = qr_update(q, r, u, v)                  # scipy's own Givens implementation, independent of linalg's
        store[f"qr_n{n}_a"], store[f"qr_n{n}_q"], store[f"qr_n{n}_r"] = a, q, r
        store[f"qr_n{n}_u"], store[f"qr_n{n}_v"], store[f"qr_n{n}_q1"], store[f"qr_n{n}_r1"] = u, v, q1, r1
        qnames.append(f"n{n}")
    store["qr_names"] = np.array(qnames)

    tnames = []
    for (m, n) in ((7, 3), (50, 12), (300, 33)):        # tall QR applied to a right-hand side (bounded LSQ, polynomial fit)
        a = rng.standard_normal((m, n))
        f = rng.standard_normal(m)
        qr, tau, work, info = lapack.dgeqrf(np.asfortranarray(a))
        qtf, work, info = lapack.dormqr("L", "T", qr, tau, f.reshape(m, 1).copy(order="F"), lwork=64 * m)
        assert info == 0
        store[f"tall_{m}x{n}_a"], store[f"tall_{m}x{n}_f"] = a, f
        store[f"tall_{m}x{n}_r"], store[f"tall_{m}x{n}_qtf"] = np.triu(qr[:n, :]), qtf[:, 0]
        tnames.append(f"{m}x{n}")
    store["tall_names"] = np.array(tnames)

    cnames = []
    for n in (2, 9, 48):                                # Cholesky (upper) of an SPD matrix + triangular solves
        g = rng.standard_normal((n + 5, n))
        b = g.T @ g
        c, info = lapack.dpotrf(np.asfortranarray(b), lower=0)
        assert info == 0
        rhs = rng.standard_normal(n)
        x, info = lapack.dpotrs(c, rhs, lower=0)
        assert info == 0
        store[f"chol_n{n}_b"], store[f"chol_n{n}_r"] = b, np.triu(c)
        store[f"chol_n{n}_rhs"], store[f"chol_n{n}_x"] = rhs, x
        cnames.append(f"n{n}")
    store["chol_names"] = np.array(cnames)

    np.savez(os.path.join(HERE, "lapack_vectors.npz"), **store)
    print("wrote lapack_vectors.npz:", len(store), "arrays")


if __name__ == "__main__":
    main()
