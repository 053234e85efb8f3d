"""Edge cases of the hot path on the GPU, each compared with the CPU oracle: degenerate sizes, an empty batch,
square least-squares systems, start points with exact zeros, a start point that is already a solution, bounds that
pin every variable, and the error codes of the C ABI."""
import ctypes as C

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
