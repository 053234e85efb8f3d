"""GPU test of the Fortran drop-in layer: a program written against nonlin's own API
(`use nonlin`, vecfcn_helper, least_squares_solver, newton_solver, iteration_behavior) linked with
nonlin_amd/fortran (ISO_C_BINDING shim) + libnonlin_hip.so, compared with the CPU oracle.
The shim's default factor policy is NLH_FACTOR_EXACT, so x is compared bit for bit."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import problems_ref as P

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "fortran", "build", "dropin_suite")


def _unhex(h):
    return struct.unpack(">d", bytes.fromhex(h))[0]


@pytest.fixture(scope="module")
def results():
    if not os.path.exists(EXE):
        if not (shutil.which("amdflang") or os.path.exists("/opt/rocm/bin/amdflang")):
            pytest.skip("no Fortran compiler and no prebuilt tests/fortran/build/dropin_suite")
        root = os.path.dirname(HERE)
        subprocess.check_call(["make", "-C", os.path.join(root, "nonlin_amd", "fortran"), "-s"])
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fortran"), "-s"])
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    res = {}
    for line in out.stdout.splitlines():
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        res.setdefault(t[0], []).append({
            "counts": (int(t[1]), int(t[2]), int(t[3])), "flags": tuple(t[4:7]),
            "x": np.array([_unhex(h) for h in t[7:]])})
    res["_stdout"] = out.stdout
    return res


def _cmp(r, rc, xo, ibo):
    assert rc == 0
    assert r["counts"] == (ibo["iter_count"], ibo["fcn_count"], ibo["jacobian_count"]), (r, ibo)
    assert r["flags"] == tuple("T" if ibo[k] else "F" for k in ("converge_on_fcn", "converge_on_chng", "converge_on_zero_diff"))
    assert np.array_equal(r["x"], xo), (r["x"], xo)


def test_polynomial_error_stops_are_the_references(results):
    """src/nonlin_polynomials.f90:399: `get` on a polynomial that was never initialised stops with
    NL_INVALID_OPERATION_ERROR (104 here, see nonlin_error_handling.f90); :402-405: an index out of range stops with
    NL_INDEX_OUT_OF_RANGE_ERROR (209) -- `set` on an initialised polynomial (on an uninitialised one it returns, :436)."""
    out = subprocess.run([EXE, "errstop_poly_get"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 104, (out.returncode, out.stderr)
    out = subprocess.run([EXE, "errstop_poly_index"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 209, (out.returncode, out.stderr)
    assert out.stdout.split()[:3] == ["2", "0.", "3"]           # order, value of the zero polynomial, coefficient count


def test_readme_example_2(results, oracle):
    rc, xo, fo, ibo = oracle.lm_solve(lambda x, f: P.lsfcn1(x, f, None), 21, 4, [1.0] * 4)
    _cmp(results["lm_readme"][0], rc, xo, ibo)
    assert "# c0: 1.1866142244" in results["_stdout"]            # README.md:165-171
    assert "# Max Residual: 0.50636" in results["_stdout"]


def test_least_squares_problems(results, oracle):
    for k, ic in enumerate([(0.5, 0.5), (1.0, 1.0)]):
        rc, xo, fo, ibo = oracle.lm_solve(lambda x, f: P.fcn1(x, f, None), 2, 2, ic)
        _cmp(results["lm_fcn1_fd"][k], rc, xo, ibo)
        rc, xo, fo, ibo = oracle.lm_solve(lambda x, f: P.fcn1(x, f, None), 2, 2, ic, jac=lambda x, J: P.jac1(x, J, None))
        _cmp(results["lm_fcn1_an"][k], rc, xo, ibo)
        rc, xo, fo, ibo = oracle.lm_solve(lambda x, f: P.fcn2(x, f, None), 2, 2, ic, opts=oracle.default_options(max_evals=1000))
        _cmp(results["lm_fcn2"][k], rc, xo, ibo)
        assert abs(abs(xo[0]) - 5.0e3) <= 1e-6 and abs(abs(xo[1]) - 10.0) <= 1e-6


def test_newton_problems(results, oracle):
    rc, xo, fo, ibo = oracle.newton_solve(lambda x, f: P.fcn1a(x, f, 2.0), 2, [1.0, 1.0])
    _cmp(results["newton_fcn1a_fd"][0], rc, xo, ibo)                 # class(*) args reached the callback
    rc, xo, fo, ibo = oracle.newton_solve(lambda x, f: P.fcn1(x, f, None), 2, [1.0, 1.0], jac=lambda x, J: P.jac1(x, J, None))
    _cmp(results["newton_fcn1_an"][0], rc, xo, ibo)
    assert results["newton_fcn1_an"][0]["counts"] == (6, 9, 6)       # recorded reference counts
    rc, xo, fo, ibo = oracle.newton_solve(lambda x, f: P.fcn2(x, f, None), 2, [0.5, 0.5],
                                          opts=oracle.default_options(use_line_search=0))
    _cmp(results["newton_fcn2_nols"][0], rc, xo, ibo)


def test_quasi_newton_problems(results, oracle):
    """quasi_newton_solver through the Fortran shim (test_quasinewton_1 / 2 / 3a): bit-identical to the CPU path."""
    rc, xo, fo, ibo = oracle.quasi_newton_solve(lambda x, f: P.fcn1(x, f, None), 2, [1.0, 1.0],
                                                jac=lambda x, J: P.jac1(x, J, None))
    _cmp(results["qn_fcn1_an"][0], rc, xo, ibo)
    assert abs(abs(xo[0]) - 5.0) <= 1e-6 and abs(abs(xo[1]) - 3.0) <= 1e-6
    rc, xo, fo, ibo = oracle.quasi_newton_solve(lambda x, f: P.fcn2(x, f, None), 2, [0.5, 0.5],
                                                opts=oracle.default_options(use_line_search=0))
    _cmp(results["qn_fcn2_nols"][0], rc, xo, ibo)
    rc, xo, fo, ibo = oracle.quasi_newton_solve(lambda x, f: P.fcn1a(x, f, 2.0), 2, [0.5, 0.5], jdelta=3)
    _cmp(results["qn_fcn1a_fd_j3"][0], rc, xo, ibo)


def test_constrained_least_squares_problems(results, oracle):
    """constrained_least_squares_solver through the Fortran shim (test_constrained_least_squares_1, _bounds)."""
    big = float(np.finfo(np.float64).max)
    rc, xo, fo, ibo = oracle.cls_solve(lambda x, f: P.fcn1(x, f, None), 2, 2, [0.5, 0.5], jac=lambda x, J: P.jac1(x, J, None),
                                       lower=[-big, -big], upper=[big, big])
    _cmp(results["cls_fcn1_an"][0], rc, xo, ibo)
    rc, xo, fo, ibo = oracle.cls_solve(lambda x, f: P.fcn1(x, f, None), 2, 2, [1.0, 1.0], lower=[4.0, 2.0], upper=[5.6, 3.6])
    _cmp(results["cls_fcn1_box"][0], rc, xo, ibo)
    assert 4.0 <= xo[0] <= 5.6 and 2.0 <= xo[1] <= 3.6


def test_polynomial_fit_readme_example_3(results, oracle):
    """README.md:175-226 through the Fortran shim: printed coefficients and residual, coefficients bitwise."""
    out = results["_stdout"]
    for line in ("# poly c0 = 1.1866141861", "# poly c1 = 0.4466136311", "# poly c2 = -.1223204989",
                 "# poly c3 = 1.0647628218", "# poly Max Residual: 0.50636"):
        assert line in out, line
    rc, co = oracle.poly_fit(P.XP, P.YP, 3)
    assert np.array_equal(results["poly_readme"][0]["x"], co)


def test_bfgs_problems(results, oracle):
    """bfgs through the Fortran shim (test_bfgs_1 / 2 / 3): minimisers within 1e-5 and bit-identical to the CPU path.
    The "counts" printed for bfgs are iter / fcn / gradient."""
    def rosen(v, a=1.0e2):
        t = v[1] - v[0] * v[0]
        return a * (t * t) + (v[0] - 1.0) * (v[0] - 1.0)

    def beale(v):
        a = 1.5 - v[0] + v[0] * v[1]
        b = 2.25 - v[0] + v[0] * (v[1] * v[1])
        c = 2.625 - v[0] + v[0] * (v[1] * v[1] * v[1])
        return a * a + b * b + c * c

    for key, fcn, x0, ans in (("bfgs_rosen", rosen, [0.0, 0.0], [1.0, 1.0]), ("bfgs_beale", beale, [1.0, 1.0], [3.0, 0.5]),
                              ("bfgs_rosen_args", rosen, [0.0, 0.0], [1.0, 1.0])):
        rc, xo, fo, ibo = oracle.bfgs_solve(fcn, 2, x0)
        r = results[key][0]
        assert rc == 0
        assert r["counts"] == (ibo["iter_count"], ibo["fcn_count"], ibo["gradient_count"]), (key, r, ibo)
        assert np.array_equal(r["x"], xo), (key, r["x"], xo)
        assert np.abs(xo - np.array(ans)).max() <= 1e-5


def test_fd_jacobian(results):
    J = results["jac_polar"][0]["x"].reshape(2, 2).T                  # printed column by column
    E = np.zeros((2, 2), order="F")
    P.polar_jac(np.array([0.5, -0.5]), E, None)
    assert np.abs(J - E).max() <= 1e-4                                # tests/nonlin_test_jacobian.f90 tolerance


# ---- device-model extension: set_device_model / device_model_batch / solve_batch -------------------------------------
EXE_DM = os.path.join(HERE, "fortran", "build", "device_model_suite")


def _write_problem_file(path, oracle, seeds, m, n, **gen):
    gamma = gen.pop("gamma", 0.5)
    data = [oracle.dq_generate(s, m, n, gamma=gamma, **gen) for s in seeds]          # (A col-major m x n, b, xt, x0)
    with open(path, "wb") as fh:
        fh.write(struct.pack("<iiid", len(seeds), m, n, gamma))
        for A, b, xt, x0 in data:
            fh.write(np.asfortranarray(A).tobytes(order="F"))
        for A, b, xt, x0 in data:
            fh.write(np.ascontiguousarray(b).tobytes())
        for A, b, xt, x0 in data:
            fh.write(np.ascontiguousarray(x0).tobytes())
    return gamma, data


@pytest.fixture(scope="module")
def dm_results(oracle, tmp_path_factory):
    if not os.path.exists(EXE_DM):
        if not (shutil.which("amdflang") or os.path.exists("/opt/rocm/bin/amdflang")):
            pytest.skip("no Fortran compiler and no prebuilt tests/fortran/build/device_model_suite")
        root = os.path.dirname(HERE)
        subprocess.check_call(["make", "-C", os.path.join(root, "nonlin_amd", "fortran"), "-s"])
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fortran"), "-s"])
    d = tmp_path_factory.mktemp("dm")
    lm_in = _write_problem_file(str(d / "lm.bin"), oracle, [12345 + k for k in range(6)], 512, 64)
    nt_in = _write_problem_file(str(d / "nt.bin"), oracle, [777 + k for k in range(4)], 48, 48, sigma=0.0, square_shift=True)
    zr_in = _write_problem_file(str(d / "zr.bin"), oracle, [12345 + k for k in range(4)], 2048, 128, sigma=0.0)
    out = subprocess.run([EXE_DM, str(d / "lm.bin"), str(d / "nt.bin"), str(d / "zr.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    res = {}
    for line in out.stdout.splitlines():
        t = line.split()
        res.setdefault(t[0], []).append({"status": int(t[1]), "counts": (int(t[2]), int(t[3]), int(t[4])),
                                         "flags": tuple(t[5:8]), "x": np.array([_unhex(h) for h in t[8:]])})
    res["_zr_in"] = zr_in
    return res, lm_in, nt_in


def _cmp_dm(r, rc, xo, ibo):
    assert r["status"] == rc == 0
    assert r["counts"] == (ibo["iter_count"], ibo["fcn_count"], ibo["jacobian_count"]), (r, ibo)
    assert r["flags"] == tuple("T" if ibo[k] else "F" for k in ("converge_on_fcn", "converge_on_chng", "converge_on_zero_diff"))
    assert np.array_equal(r["x"], xo)


def test_device_model_least_squares_through_fortran(dm_results, oracle):
    """vecfcn_helper%set_device_model + least_squares_solver%solve (the reference's own call, the whole iteration on the
    GPU) and least_squares_solver%solve_batch on six 512 x 64 problems: bit-identical to the CPU oracle."""
    res, (gamma, data), _ = dm_results
    sols = [oracle.dq_lm_solve(A, b, gamma, x0, opts=oracle.default_options(max_evals=500)) for A, b, xt, x0 in data]
    rc, xo, fo, ibo = sols[0][:4]
    _cmp_dm(res["dm_lm_single"][0], rc, xo, ibo)
    assert np.array_equal(res["dm_lm_single_fvec"][0]["x"], np.array([fo[0], fo[-1]]))
    A, b, xt, x0 = data[0]
    f0 = oracle.dq_residual(A, b, gamma, x0)
    assert np.array_equal(res["dm_eval"][0]["x"], np.array([f0[0], f0[-1]]))       # obj%fcn of a device model
    assert len(res["dm_lm_batch"]) == len(data)
    for k, (rc, xo, fo, ibo, _, _) in enumerate(sols):
        _cmp_dm(res["dm_lm_batch"][k], rc, xo, ibo)
    # nlh_use_devices([0, 0]): the batch dealt over a device set inside the Fortran process (two shares on GPU 0)
    assert len(res["dm_lm_batch_set"]) == len(data)
    for k, (rc, xo, fo, ibo, _, _) in enumerate(sols):
        _cmp_dm(res["dm_lm_batch_set"][k], rc, xo, ibo)


def test_device_model_newton_through_fortran(dm_results, oracle):
    """newton_solver%solve on a device model (the model's own Jacobian, then forward differences) and solve_batch."""
    res, _, (gamma, data) = dm_results
    A, b, xt, x0 = data[0]
    rc, xo, fo, ibo, _ = oracle.dq_newton_solve(A, b, gamma, x0, analytic=True, opts=oracle.default_options(max_evals=500))
    _cmp_dm(res["dm_newton_an"][0], rc, xo, ibo)
    rc, xo, fo, ibo, _ = oracle.dq_newton_solve(A, b, gamma, x0, analytic=False, opts=oracle.default_options(max_evals=500))
    _cmp_dm(res["dm_newton_fd"][0], rc, xo, ibo)
    for k, (A, b, xt, x0) in enumerate(data):
        rc, xo, fo, ibo, _ = oracle.dq_newton_solve(A, b, gamma, x0, analytic=True, opts=oracle.default_options(max_evals=500))
        _cmp_dm(res["dm_newton_batch"][k], rc, xo, ibo)


def test_device_model_other_solver_batches_through_fortran(dm_results, oracle):
    """quasi_newton_solver%solve_batch, constrained_least_squares_solver%solve_batch (box -0.3 .. 0.25) and bfgs%solve_batch
    (objective 0.5 ||F||^2, forward-difference gradient) on device model batches -- the lock-step state machines behind the
    reference's solver types: every problem bit-identical to the CPU oracle."""
    res, (gamma, lm_data), (gamma_n, nt_data) = dm_results
    assert len(res["dm_broyden_batch"]) == len(nt_data)
    for k, (A, b, xt, x0) in enumerate(nt_data):
        rc, xo, fo, ibo, _ = oracle.dq_quasi_newton_solve(A, b, gamma_n, x0, analytic=True, opts=oracle.default_options(max_evals=500))
        _cmp_dm(res["dm_broyden_batch"][k], rc, xo, ibo)
    n = lm_data[0][0].shape[1]
    lo, hi = np.full(n, -0.3), np.full(n, 0.25)
    assert len(res["dm_cls_batch"]) == len(lm_data)
    for k, (A, b, xt, x0) in enumerate(lm_data):
        rc, xo, fo, ibo, _ = oracle.dq_cls_solve(A, b, gamma, x0, opts=oracle.default_options(max_evals=500), lower=lo, upper=hi)
        _cmp_dm(res["dm_cls_batch"][k], rc, xo, ibo)
    assert len(res["dm_bfgs_batch"]) == len(lm_data)
    for k, (A, b, xt, x0) in enumerate(lm_data):
        rc, xo, fo, ibo, _ = oracle.dq_bfgs_solve(A, b, gamma, x0, opts=oracle.default_options(max_evals=300, gtol=1e-8, xtol=1e-12))
        r = res["dm_bfgs_batch"][k]
        assert r["status"] == rc
        assert r["counts"] == (ibo["iter_count"], ibo["fcn_count"], ibo["gradient_count"]), (r, ibo)
        assert r["flags"] == ("F", "T" if ibo["converge_on_chng"] else "F", "T" if ibo["converge_on_zero_diff"] else "F")
        assert np.array_equal(r["x"][:-1], xo) and r["x"][-1] == fo


def test_device_model_auto_policy_on_zero_residual_problems_through_fortran(dm_results, oracle):
    """N1 from the Fortran side: `lm%factor_policy = NLH_FACTOR_AUTO` (the MFMA J^T J + Cholesky formulation) on four
    zero-residual 2048 x 128 problems (BASELINE config 4's size): x within 1e-10 of the CPU oracle with every count and
    flag equal; the default policy on the same problems: the oracle's bits."""
    res, _, _ = dm_results
    gamma, data = res["_zr_in"]
    sols = [oracle.dq_lm_solve(A, b, gamma, x0, opts=oracle.default_options(max_evals=500)) for A, b, xt, x0 in data]
    assert len(res["dm_lm_auto_zero_residual"]) == len(data) == len(res["dm_lm_exact_zero_residual"])
    for k, (rc, xo, fo, ibo, _, _) in enumerate(sols):
        _cmp_dm(res["dm_lm_exact_zero_residual"][k], rc, xo, ibo)
        r = res["dm_lm_auto_zero_residual"][k]
        assert r["status"] == rc == 0
        assert r["counts"] == (ibo["iter_count"], ibo["fcn_count"], ibo["jacobian_count"]), (r, ibo)
        assert r["flags"] == tuple("T" if ibo[q] else "F" for q in ("converge_on_fcn", "converge_on_chng", "converge_on_zero_diff"))
        assert np.abs(r["x"] - xo).max() <= 1e-10 * np.abs(xo).max()
