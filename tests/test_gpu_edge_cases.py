"""Edge cases of the hot path on the GPU, each compared with the CPU oracle: degenerate sizes, an empty batch,
square least-squares systems, start points with exact zeros, a start point that is already a solution, bounds that
pin every variable, and the error codes of the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import problems_ref as P

pytestmark = pytest.mark.gpu
COUNT_KEYS = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")


def _oracle_lm(oracle, A, b, x0, **o):
    Ah = np.asfortranarray(A.cpu().numpy().T)
    return oracle.dq_lm_solve(Ah, b.cpu().numpy(), 0.5, x0.cpu().numpy(), opts=oracle.default_options(**o))


@pytest.mark.parametrize("m,n", [(1, 1), (5, 1), (2, 2), (3, 3), (64, 64), (17, 16)])
def test_lm_degenerate_and_square_shapes_exact_policy(ds, oracle, m, n):
    """n = 1, m = n (lmpar's deviation A has no tail then) and m = n + 1, bit-identical under the exact policy."""
    A, b, xt, x0 = ds.generate(2, m, n, seed0=31, square_shift=(m == n))
    x = x0.clone()
    fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=300, factor_policy=2))
    for p in range(2):
        rc, xo, fo, ibo = _oracle_lm(oracle, A[p], b[p], x0[p], max_evals=300)[:4]
        assert status[p] == rc
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (m, n, k, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo) and np.array_equal(fvec[p].cpu().numpy(), fo)


@pytest.mark.parametrize("m,n", [(128, 12), (600, 70), (1500, 130)])
def test_lm_rank_deficient_jacobian_exact_policy(ds, oracle, m, n):
    """A zero column (lmfactor meets ajnorm == 0, :646) and a duplicated column (the downdated column norm collapses
    and is recomputed, :656-661): bit-identical to the CPU path, whatever the status of the solve is."""
    A, b, xt, x0 = ds.generate(3, m, n, seed0=77)
    A[0, 3, :] = 0.0                                 # A is [nprob][n][m]: column 3 of problem 0
    A[1, 5, :] = A[1, 2, :]
    x0[1, 5] = x0[1, 2]                              # same variable value: the two Jacobian columns coincide too
    A[2, 3, :] = 0.0
    A[2, n - 1, :] = A[2, 0, :]
    x0[2, n - 1] = x0[2, 0]
    x = x0.clone()
    fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=200, factor_policy=2))
    for p in range(3):
        rc, xo, fo, ibo = _oracle_lm(oracle, A[p], b[p], x0[p], max_evals=200)[:4]
        assert status[p] == rc
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (m, n, p, k, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo, equal_nan=True)
        assert np.array_equal(fvec[p].cpu().numpy(), fo, equal_nan=True)


@pytest.mark.parametrize("m,n", [(70, 63), (64, 64), (131, 65), (200, 95), (99, 96), (300, 127), (260, 128), (259, 129),
                                 (258, 255), (515, 257), (520, 511), (513, 512)])
def test_lm_exact_policy_at_kernel_boundaries(ds, oracle, m, n):
    """Sizes on either side of every switch inside the exact-policy QR (64 / 128 / 256 / 512 column slots, the
    256- and 512-thread launches, the n = 512 hand-over to the plain kernel) with m barely above n, so that the
    last tiles are short or absent: bit-identical to the CPU path."""
    A, b, xt, x0 = ds.generate(2, m, n, seed0=1000 + n, square_shift=(m == n))
    x = x0.clone()
    fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=60 * (n + 1), factor_policy=2))
    for p in range(2):
        rc, xo, fo, ibo = _oracle_lm(oracle, A[p], b[p], x0[p], max_evals=60 * (n + 1))[:4]
        assert status[p] == rc
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (m, n, p, k, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo) and np.array_equal(fvec[p].cpu().numpy(), fo)


def test_empty_batch_is_a_no_op(ds):
    A = torch.empty((0, 4, 21), dtype=torch.float64, device=ds.device)
    b = torch.empty((0, 21), dtype=torch.float64, device=ds.device)
    x = torch.empty((0, 4), dtype=torch.float64, device=ds.device)
    fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options())
    assert ibs == [] and status == [] and fvec.shape == (0, 21)


@pytest.mark.parametrize("policy", [0, 2])
def test_start_point_with_exact_zeros_and_exact_solution(ds, oracle, policy):
    """x_j == 0 takes the h = sqrt(eps) branch of the FD step (:268-269); a start point that already solves the
    system stops at once on the gradient test with ||f|| == 0 (:256-273, Appendix A item 9)."""
    m, n = 96, 12
    A, b, xt, x0 = ds.generate(2, m, n, seed0=5, sigma=0.0)
    x0[0, ::3] = 0.0
    x0[1] = xt[1]                                   # zero residual at the start (sigma = 0)
    x = x0.clone()
    fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=300, factor_policy=policy))
    for p in range(2):
        rc, xo, fo, ibo = _oracle_lm(oracle, A[p], b[p], x0[p], max_evals=300)[:4]
        assert status[p] == rc == 0
        if policy == 2 or p == 1:
            for k in COUNT_KEYS:
                assert ibs[p][k] == ibo[k], (p, k, ibs[p], ibo)
            assert np.array_equal(x[p].cpu().numpy(), xo)
        else:
            assert np.abs(x[p].cpu().numpy() - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max())
    assert ibs[1]["jacobian_count"] == 1 and ibs[1]["converge_on_zero_diff"] == 1


def test_cls_all_variables_pinned_by_bounds(ds, oracle):
    """lower == upper: apply_limits puts x on the box, alpha_box leaves no step, the solver reports convergence
    on the change in x (:1130-1134) after a single Jacobian."""
    m, n = 64, 8
    A, b, xt, x0 = ds.generate(1, m, n, seed0=9)
    pin = np.full(n, 0.25)
    x = x0.clone()
    fvec, ibs, status = ds.cls_solve_batch(A, b, 0.5, x, opts=ds.options(max_evals=100), lower=pin, upper=pin)
    rc, xo, fo, ibo, _ = oracle.dq_cls_solve(np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy(), 0.5,
                                             x0[0].cpu().numpy(), opts=oracle.default_options(max_evals=100),
                                             lower=pin, upper=pin)
    assert status[0] == rc == 0
    assert np.array_equal(x[0].cpu().numpy(), pin) and np.array_equal(xo, pin)
    assert ibs[0]["jacobian_count"] == ibo["jacobian_count"] == 1 and ibs[0]["converge_on_chng"] == 1


def test_c_abi_error_codes(ds):
    """The status codes the Fortran shim turns into `error stop`: 212 (n > m), 201 (n /= m for the square solvers),
    4 (polynomial order >= number of points), and a null handle."""
    lib, h = ds.lib, ds.h.ptr
    o = ds.options()
    z = torch.zeros(64, dtype=torch.float64, device=ds.device)
    rc = lib.nlh_dq_lm_solve_batch(h, C.byref(o), 1, 2, 3, z.data_ptr(), z.data_ptr(), 0.5, z.data_ptr(), z.data_ptr(), None, None)
    assert rc == 212
    rc = lib.nlh_poly_fit_batch(h, 1, 3, 3, 0, z.data_ptr(), z.data_ptr(), z.data_ptr())
    assert rc == 4
    rc = lib.nlh_poly_fit_batch(None, 1, 8, 3, 0, z.data_ptr(), z.data_ptr(), z.data_ptr())
    assert rc == -3
    import nonlin_amd as nl
    obj = nl.vecfcn_helper()
    obj.set_fcn(P.lsfcn1, 21, 4)
    with pytest.raises(nl.NonlinError) as e:        # newton on a non-square system, src/nonlin_solve.f90:519
        nl.newton_solver().solve(obj, np.ones(4), np.zeros(21))
    assert e.value.code == nl.NL_INVALID_INPUT_ERROR
    with pytest.raises(nl.NonlinError) as e:        # quasi-Newton likewise, :241
        nl.quasi_newton_solver().solve(obj, np.ones(4), np.zeros(21))
    assert e.value.code == nl.NL_INVALID_INPUT_ERROR


def test_full_size_c5_fd_jacobian_columns_bit_exact(ds, oracle):
    """BASELINE config 5 size (65536 x 512): a sample of columns of the device FD Jacobian against the oracle's
    forward differences, bit for bit (the oracle evaluates only the sampled columns)."""
    m, n = 65536, 512
    A, b, xt, x0 = ds.generate(1, m, n, seed0=12345)
    f0 = ds.residual(A, b, 0.5, x0)
    J = ds.fd_jacobian_panel(ds.fd_panel(A, b, 0.5, x0), f0, x0)
    Ah = np.asfortranarray(A[0].cpu().numpy().T)
    bh, xh, f0h = b[0].cpu().numpy(), x0[0].cpu().numpy(), f0[0].cpu().numpy()
    assert np.array_equal(oracle.dq_residual(Ah, bh, 0.5, xh), f0h)
    eps = np.sqrt(np.finfo(np.float64).eps)
    for j in (0, 1, 255, 511):
        xp = xh.copy()
        hj = eps * abs(xp[j]) if xp[j] != 0.0 else eps
        xp[j] = xp[j] + hj
        col = (oracle.dq_residual(Ah, bh, 0.5, xp) - f0h) / hj
        assert np.array_equal(J[0, j].cpu().numpy(), col), j


# ---- sizes beyond what fits LDS / four columns per thread (round 5: the reference allocates for any n,
# src/nonlin_least_squares.f90:199-208) -----------------------------------------------------------------------------------
def _run_py(code, env):
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=e,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


_LM_GLOBAL_VECS = '''
import numpy as np, sys
sys.path.insert(0, "tests")
from nonlin_amd.device import DeviceSolver
from oracle import pyoracle as O
ds = DeviceSolver(0)
for m, n, gen, opt in ((300, 37, dict(gamma=2.0, sigma=0.1, spread=5.0), dict(factor=0.1)), (512, 64, {}, {}), (256, 32, dict(gamma=10.0, sigma=1.0, spread=50.0), dict(factor=0.1))):
    A, b, xt, x0 = ds.generate(3, m, n, seed0=12345, **gen)
    x = x0.clone()
    f, ibs, st = ds.lm_solve_batch(A, b, gen.get("gamma", 0.5), x, ds.options(max_evals=500, **opt))
    for p in range(3):
        rc, xo, fo, ibo = O.dq_lm_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), gen.get("gamma", 0.5), x0[p].cpu().numpy(),
                                        opts=O.default_options(max_evals=500, **opt))[:4]
        assert st[p] == rc and all(ibs[p][k] == ibo[k] for k in ("iter_count", "fcn_count", "jacobian_count")), (p, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo) and np.array_equal(f[p].cpu().numpy(), fo)
print("ok")
'''


@pytest.mark.gpu
def test_lmpar_with_its_vectors_in_global_memory_is_bitwise():
    """The form k_lmpar takes beyond n = 3000 (lmpar's n-vectors in global memory instead of LDS), forced at sizes the oracle
    finishes in seconds (NLH_LM_LDS_MAX_N=16), trust-region-binding families included: x, fvec, counts bit for bit."""
    assert "ok" in _run_py(_LM_GLOBAL_VECS, {"NLH_LM_LDS_MAX_N": "16"})


@pytest.mark.gpu
def test_least_squares_beyond_3000_columns(ds):
    """n = 3008 > the LDS bound, m = 3040: a linear zero-residual problem (gamma = 0, sigma = 0): one Jacobian, one
    (nearly square: badly conditioned) factorisations of 3008 Householder steps each, the solve must end converged with a
    residual at rounding level, and that residual must be what an independent evaluation at the returned x gives.  (The oracle needs over a minute for this factorisation; the bits of the
    global-memory form of lmpar are held to it at small sizes by test_lmpar_with_its_vectors_in_global_memory_is_bitwise.)"""
    m, n = 3040, 3008
    A, b, xt, x0 = ds.generate(1, m, n, seed0=99, gamma=0.0, sigma=0.0, spread=0.1)
    x = x0.clone()
    f, ibs, st = ds.lm_solve_batch(A, b, 0.0, x, ds.options(max_evals=50))
    assert st[0] == 0
    assert ibs[0]["jacobian_count"] <= 8 and (ibs[0]["converge_on_fcn"] or ibs[0]["converge_on_chng"] or ibs[0]["converge_on_zero_diff"])
    assert float(f.abs().max()) < 1e-6
    assert torch.equal(ds.residual(A, b, 0.0, x), f)


_QN_NC8 = '''
import numpy as np, sys
from nonlin_amd.device import DeviceSolver
from oracle import pyoracle as O
ds = DeviceSolver(0)
n = 1030                                                          # > 1024: the several-columns-per-thread instances
A, b, xt, x0 = ds.generate(1, n, n, seed0=12345, sigma=0.0, spread=0.03, square_shift=True)
x = x0.clone()
f, ibs, st = ds.quasi_newton_solve_batch(A, b, 0.5, x, analytic=True, opts=ds.options(max_evals=500))
for p in range(1):
    rc, xo, fo, ibo = O.dq_quasi_newton_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(), analytic=True,
                                              opts=O.default_options(max_evals=500))[:4]
    assert st[p] == rc and all(ibs[p][k] == ibo[k] for k in ("iter_count", "fcn_count", "jacobian_count")), (ibs[p], ibo)
    assert np.array_equal(x[p].cpu().numpy(), xo)
m, nb = 1200, 1030
A, b, xt, x0 = ds.generate(1, m, nb, seed0=77, spread=0.1)
x = x0.clone()
fo_g, ibs, st = ds.bfgs_solve_batch(A, b, 0.5, x, ds.options(max_evals=12, gtol=1e-8, xtol=1e-12))
rc, xo, fo, ibo = O.dq_bfgs_solve(np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy(), 0.5, x0[0].cpu().numpy(),
                                  opts=O.default_options(max_evals=12, gtol=1e-8, xtol=1e-12))[:4]
assert st[0] == rc and ibs[0]["iter_count"] == ibo["iter_count"] and ibs[0]["fcn_count"] == ibo["fcn_count"], (ibs[0], ibo)
assert np.array_equal(x[0].cpu().numpy(), xo)
print("ok")
'''


@pytest.mark.gpu
def test_eight_columns_per_thread_instances_are_bitwise():
    """The rotation / Cholesky-update kernels' instance for n in (4096, 8192] (eight columns per thread), forced at
    n = 1030 (NLH_QN_FORCE_NC8=1) where the oracle is affordable: quasi-Newton and BFGS bit for bit."""
    assert "ok" in _run_py(_QN_NC8, {"NLH_QN_FORCE_NC8": "1"})


@pytest.mark.gpu
def test_quasi_newton_beyond_4096_unknowns(ds):
    """n = 5000 Broyden: no oracle at this size (its O(n^3) QR with Q formed takes minutes); the solve must converge on the
    function values and the residual evaluated independently at the returned x must be what the solver reports.  (Which root:
    every equation is quadratic in u_i = (A x)_i, so the system has many; the iteration need not return the one the problem
    was generated from, and with 5000 equations it does not.)"""
    n = 5000
    A, b, xt, x0 = ds.generate(1, n, n, seed0=5, sigma=0.0, spread=0.02, square_shift=True)
    x = x0.clone()
    f, ibs, st = ds.quasi_newton_solve_batch(A, b, 0.5, x, analytic=True, opts=ds.options(max_evals=500))
    assert st[0] == 0 and ibs[0]["converge_on_fcn"] == 1
    assert float(f.abs().max()) < 1e-8
    assert torch.equal(ds.residual(A, b, 0.5, x), f)
