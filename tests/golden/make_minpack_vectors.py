#!/usr/bin/env python3
"""Fixtures that pin the oracle's lmpar iteration / lmsolve / reject path against MINPACK ITSELF (build container only).

src/nonlin_least_squares.f90:400 and :674 state that lmpar / lmsolve are MINPACK's lmpar / qrsolv, and lss_solve is
lmder/lmdif with mode-1 scaling.  The reference departs from MINPACK in exactly two lines of
lmpar (:531 norm over m instead of n entries, :552 whole-vector update instead of rows j+1..n).  With those two lines
switched back (oracle.pyoracle.set_lmpar_minpack(1), a test-only switch) the oracle must walk MINPACK's trajectory: this
script runs scipy.optimize.leastsq (= MINPACK lmder; scipy's translation of the Fortran) with ANALYTIC Jacobians on
problems whose trust region binds (factor = 0.1, the "hard" generator of SURVEY 8(d), rank-deficient Jacobians, the
classic MINPACK test functions from far starts) and records x, nfev, njev, ier per solve.  Both solvers are handed the
SAME residual / Jacobian callbacks, so every function value is bit-identical between them; what remains different is
MINPACK's enorm against flang's NORM2 (a rounding-level difference), hence: counts must agree exactly, x to 1e-10 on the
zero-residual cases (looser where the minimiser is flat or not unique -- the fixture records which).

    python tests/golden/make_minpack_vectors.py        # writes tests/golden/minpack_lmder.json

tests/test_oracle.py::test_oracle_with_minpack_lines_follows_minpack re-runs the ORACLE side only (scipy is not needed
there) and compares with the recorded MINPACK results."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O        # noqa: E402

TOL = dict(ftol=1e-8, xtol=1e-12, gtol=1e-12)       # equation_solver defaults, src/nonlin_multi_eqn_mult_var.f90:71-75
MAXFEV = 500


def dq_case(name, seed, m, n, gamma, sigma, spread, factor, dup=0):
    return dict(kind="dq", name=name, seed=seed, m=m, n=n, gamma=gamma, sigma=sigma, spread=spread, factor=factor, dup=dup)


def classic_case(name, fn, x0, factor):
    return dict(kind="classic", name=name, fn=fn, x0=list(x0), factor=factor)


CASES = [
    # SURVEY 8(d) "hard" variant (gamma 2, sigma 0.1, spread 5) with set_step_scaling_factor(0.1): lmpar iterates
    dq_case("dq_hard_64x16", 12345, 64, 16, 2.0, 0.1, 5.0, 0.1),
    dq_case("dq_hard_256x32", 12346, 256, 32, 2.0, 0.1, 5.0, 0.1),
    dq_case("dq_hard_512x64", 12347, 512, 64, 2.0, 0.1, 5.0, 0.1),
    dq_case("dq_hard_f100_256x32", 12348, 256, 32, 2.0, 0.1, 5.0, 100.0),
    dq_case("dq_wild_256x32", 12349, 256, 32, 10.0, 1.0, 50.0, 0.1),
    # zero residual (sigma = 0): the solution is a point, x must agree to 1e-10
    dq_case("dq_zero_hard_64x16", 22345, 64, 16, 2.0, 0.0, 5.0, 0.1),
    dq_case("dq_zero_hard_256x32", 22346, 256, 32, 2.0, 0.0, 5.0, 0.1),
    dq_case("dq_zero_hard_300x37", 22347, 300, 37, 2.0, 0.0, 3.0, 0.1),
    dq_case("dq_zero_mild_512x64", 22348, 512, 64, 0.5, 0.0, 1.0, 0.1),
    dq_case("dq_zero_square_48x48", 22349, 48, 48, 1.0, 0.0, 1.0, 0.1),
    # rank-deficient Jacobians: the last `dup` columns of A repeat the first ones (zero pivots / nsing < n in lmpar)
    dq_case("dq_rankdef_128x24", 32345, 128, 24, 2.0, 0.1, 5.0, 0.1, dup=4),
    dq_case("dq_rankdef_zero_128x24", 32346, 128, 24, 1.0, 0.0, 2.0, 0.1, dup=3),
    # classic MINPACK test functions, far starts, small factor
    classic_case("rosenbrock_f0.1", "rosenbrock", (-1.2, 1.0), 0.1),
    classic_case("rosenbrock_x10_f0.1", "rosenbrock", (-12.0, 10.0), 0.1),
    classic_case("helical_valley_f0.1", "helical", (-1.0, 0.0, 0.0), 0.1),
    classic_case("powell_singular_f0.1", "powell_singular", (3.0, -1.0, 0.0, 1.0), 0.1),
    classic_case("freudenstein_roth_f0.1", "freudenstein", (0.5, -2.0), 0.1),
    classic_case("wood_f1", "wood", (-3.0, -1.0, -3.0, -1.0), 1.0),
    classic_case("readme_example_2_f0.1", "readme2", (1.0, 1.0, 1.0, 1.0), 0.1),
]

# README.md:145-153 (the reference's least-squares example data)
_XP = np.array([0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1, 1.2, 1.3, 1.4, 1.5, 1.6, 1.7, 1.8, 1.9, 2.0])
_YP = np.array([1.216737514, 1.250032542, 1.305579195, 1.040182335, 1.751867738, 1.109716707, 2.018141531,
                1.992418729, 1.807916923, 2.078806005, 2.698801324, 2.644662712, 3.412756702, 4.406137221,
                4.567156645, 4.999550779, 5.652854194, 6.784320119, 8.307936836, 8.395126494, 10.30252404])


def classic(fn):
    """(m, n, f(x, out), jac(x, J column-major view m x n)) -- plain Python floats, one operation per line of the formula."""
    if fn == "rosenbrock":
        def f(x, o):
            o[0] = 10.0 * (x[1] - x[0] * x[0]); o[1] = 1.0 - x[0]
        def j(x, J):
            J[0, 0] = -20.0 * x[0]; J[0, 1] = 10.0; J[1, 0] = -1.0; J[1, 1] = 0.0
        return 2, 2, f, j
    if fn == "helical":
        def f(x, o):
            th = np.arctan2(x[1], x[0]) / (2.0 * np.pi)
            o[0] = 10.0 * (x[2] - 10.0 * th); o[1] = 10.0 * (np.sqrt(x[0] * x[0] + x[1] * x[1]) - 1.0); o[2] = x[2]
        def j(x, J):
            r2 = x[0] * x[0] + x[1] * x[1]; r = np.sqrt(r2)
            J[0, 0] = 100.0 * x[1] / (2.0 * np.pi * r2); J[0, 1] = -100.0 * x[0] / (2.0 * np.pi * r2); J[0, 2] = 10.0
            J[1, 0] = 10.0 * x[0] / r; J[1, 1] = 10.0 * x[1] / r; J[1, 2] = 0.0
            J[2, 0] = 0.0; J[2, 1] = 0.0; J[2, 2] = 1.0
        return 3, 3, f, j
    if fn == "powell_singular":
        def f(x, o):
            o[0] = x[0] + 10.0 * x[1]; o[1] = np.sqrt(5.0) * (x[2] - x[3])
            o[2] = (x[1] - 2.0 * x[2]) ** 2; o[3] = np.sqrt(10.0) * (x[0] - x[3]) ** 2
        def j(x, J):
            J[:, :] = 0.0
            J[0, 0] = 1.0; J[0, 1] = 10.0; J[1, 2] = np.sqrt(5.0); J[1, 3] = -np.sqrt(5.0)
            J[2, 1] = 2.0 * (x[1] - 2.0 * x[2]); J[2, 2] = -4.0 * (x[1] - 2.0 * x[2])
            J[3, 0] = 2.0 * np.sqrt(10.0) * (x[0] - x[3]); J[3, 3] = -2.0 * np.sqrt(10.0) * (x[0] - x[3])
        return 4, 4, f, j
    if fn == "freudenstein":
        def f(x, o):
            o[0] = -13.0 + x[0] + ((5.0 - x[1]) * x[1] - 2.0) * x[1]; o[1] = -29.0 + x[0] + ((1.0 + x[1]) * x[1] - 14.0) * x[1]
        def j(x, J):
            J[0, 0] = 1.0; J[0, 1] = x[1] * (10.0 - 3.0 * x[1]) - 2.0; J[1, 0] = 1.0; J[1, 1] = x[1] * (2.0 + 3.0 * x[1]) - 14.0
        return 2, 2, f, j
    if fn == "wood":
        def f(x, o):
            o[0] = 10.0 * (x[1] - x[0] ** 2); o[1] = 1.0 - x[0]; o[2] = np.sqrt(90.0) * (x[3] - x[2] ** 2); o[3] = 1.0 - x[2]
            o[4] = np.sqrt(10.0) * (x[1] + x[3] - 2.0); o[5] = (x[1] - x[3]) / np.sqrt(10.0)
        def j(x, J):
            J[:, :] = 0.0
            J[0, 0] = -20.0 * x[0]; J[0, 1] = 10.0; J[1, 0] = -1.0; J[2, 2] = -2.0 * np.sqrt(90.0) * x[2]; J[2, 3] = np.sqrt(90.0)
            J[3, 2] = -1.0; J[4, 1] = np.sqrt(10.0); J[4, 3] = np.sqrt(10.0); J[5, 1] = 1.0 / np.sqrt(10.0); J[5, 3] = -1.0 / np.sqrt(10.0)
        return 6, 4, f, j
    if fn == "readme2":          # README.md:157: f = c3 x^3 + c2 x^2 + c1 x + c0 - y, x = (c3? ...) the example's own ordering
        def f(x, o):
            o[:] = x[0] * _XP ** 3 + x[1] * _XP ** 2 + x[2] * _XP + x[3] - _YP
        def j(x, J):
            J[:, 0] = _XP ** 3; J[:, 1] = _XP ** 2; J[:, 2] = _XP; J[:, 3] = 1.0
        return 21, 4, f, j
    raise ValueError(fn)


def problem(case):
    """m, n, x0, f(x, out), jac(x, Jview)."""
    if case["kind"] == "classic":
        m, n, f, j = classic(case["fn"])
        return m, n, np.array(case["x0"], dtype=np.float64), f, j
    A, b, xt, x0 = O.dq_generate(case["seed"], case["m"], case["n"], gamma=case["gamma"], sigma=case["sigma"],
                                 spread=case["spread"])
    if case["dup"]:
        A[:, -case["dup"]:] = A[:, :case["dup"]]
        if case["sigma"] == 0.0:                      # keep the residual zero at x_true
            b = O.dq_residual(A, np.zeros_like(b), case["gamma"], xt)
    g = case["gamma"]

    def f(x, o):
        o[:] = O.dq_residual(A, b, g, x)               # the C residual of the oracle: the same bits for both solvers

    def j(x, J):
        J[:, :] = O.dq_jacobian(A, b, g, x)
    return case["m"], case["n"], x0, f, j


def run_oracle(case, minpack_lines):
    m, n, x0, f, j = problem(case)
    O.set_lmpar_minpack(minpack_lines)
    O.lmpar_loop_entries(reset=True)
    try:
        rc, x, fvec, ib = O.lm_solve(f, m, n, x0, jac=j, opts=O.default_options(max_evals=MAXFEV, factor=case["factor"], **TOL))
    finally:
        O.set_lmpar_minpack(0)
    return rc, x, fvec, ib, O.lmpar_loop_entries(reset=True)


def main():
    from scipy.optimize import leastsq
    import scipy
    out = {"generator": "tests/golden/make_minpack_vectors.py",
           "minpack": f"scipy {scipy.__version__} scipy.optimize.leastsq with Dfun (MINPACK lmder), col_deriv=0, diag=None (mode 1)",
           "tolerances": TOL, "maxfev": MAXFEV, "cases": []}
    for case in CASES:
        m, n, x0, f, j = problem(case)

        def fs(x):
            o = np.zeros(m); f(x, o); return o

        def js(x):
            J = np.zeros((m, n)); j(x, J); return J
        x, cov, info, msg, ier = leastsq(fs, x0.copy(), Dfun=js, full_output=True, col_deriv=False, maxfev=MAXFEV,
                                         factor=case["factor"], diag=None, **TOL)
        rc, xo, fo, ib, loops = run_oracle(case, 1)
        rel = float(np.abs(xo - x).max() / max(np.abs(x).max(), 1e-300))
        frel = float(abs(np.linalg.norm(fo) - np.linalg.norm(info["fvec"])) / max(np.linalg.norm(info["fvec"]), 1e-300))
        rec = dict(case)
        rec.update({"minpack_x": [float(v).hex() for v in x], "minpack_nfev": int(info["nfev"]), "minpack_njev": int(info["njev"]),
                    "minpack_ier": int(ier), "minpack_fnorm": float(np.linalg.norm(info["fvec"])),
                    # what the switched oracle did when the fixture was made (information; the test recomputes it)
                    "oracle_lmpar_loop_entries": loops, "oracle_rel_dev_x": rel, "oracle_rel_dev_fnorm": frel,
                    "oracle_counts": [ib["fcn_count"], ib["jacobian_count"]],
                    # what the test asserts for this case.  Full-rank problems: counts and exit reason exactly, x to 1e-10
                    # (analytic Jacobians: no difference noise).  Duplicated columns: the minimiser is not unique, so only
                    # the residual norm is compared; and with a zero residual on top the zero pivots themselves are
                    # rounding coin tosses (enorm against NORM2 decides whether r(j,j) is exactly zero): recorded, not asserted.
                    "assert_counts": not (case.get("dup") and case.get("sigma") == 0.0),
                    "x_tol": None if case.get("dup") else 1e-10,
                    "fnorm_tol": 1e-10 if (case.get("dup") and case.get("sigma") != 0.0) else None})
        # unswitched oracle (the reference's lines): recorded to show that the two lines matter on these problems
        rc2, xr, fr, ibr, loops2 = run_oracle(case, 0)
        rec["reference_lines_counts"] = [ibr["fcn_count"], ibr["jacobian_count"]]
        rec["reference_lines_rel_dev_x"] = float(np.abs(xr - x).max() / max(np.abs(x).max(), 1e-300))
        out["cases"].append(rec)
        print(f"{case['name']:28s} minpack nfev/njev/ier {info['nfev']:3d}/{info['njev']:3d}/{ier}  switched oracle "
              f"{ib['fcn_count']:3d}/{ib['jacobian_count']:3d} flags {ib['converge_on_fcn']}{ib['converge_on_chng']}{ib['converge_on_zero_diff']} rc {rc} "
              f"loops {loops:3d} dx {rel:.2e} dfnorm {frel:.1e} | reference lines {ibr['fcn_count']:3d}/{ibr['jacobian_count']:3d} dx {rec['reference_lines_rel_dev_x']:.1e}")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "minpack_lmder.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
