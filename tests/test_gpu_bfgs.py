"""GPU parity tests for bfgs%solve (src/nonlin_optimize.f90:557-770), fcnnvar_helper%gradient
(src/nonlin_multi_var.f90:182-246) and the Cholesky rank-one update / downdate kernels.  The CPU restatement sums
in ascending index order and the kernels do the same operations per element: comparisons are bitwise.
Problems: test_bfgs_1..3 (tests/nonlin_test_optimize.f90:184-300) and the dense-quadratic device model."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
COUNT_KEYS = ("iter_count", "fcn_count", "gradient_count", "converge_on_chng", "converge_on_zero_diff")


def rosenbrock(x, args=None):
    a = 1.0e2 if args is None else float(args)
    t = x[1] - x[0] * x[0]
    return a * (t * t) + (x[0] - 1.0) * (x[0] - 1.0)


def beale(x, args=None):
    a = 1.5 - x[0] + x[0] * x[1]
    b = 2.25 - x[0] + x[0] * (x[1] * x[1])
    c = 2.625 - x[0] + x[0] * (x[1] * x[1] * x[1])
    return a * a + b * b + c * c


def rosenbrock_grad(x, g, args=None):
    g[0] = -4.0e2 * x[0] * (x[1] - x[0] * x[0]) + 2.0 * (x[0] - 1.0)
    g[1] = 2.0e2 * (x[1] - x[0] * x[0])


def _solve(fcn, n, x0, grad=None, args=None, use_ls=True):
    import nonlin_amd as nl
    obj = nl.fcnnvar_helper()
    obj.set_fcn(fcn, n)
    if grad is not None:
        obj.set_gradient_fcn(grad)
    s = nl.bfgs()
    s.set_use_line_search(use_ls)
    x = np.array(x0, dtype=np.float64)
    ib = nl.iteration_behavior()
    fout = s.solve(obj, x, ib, args=args)
    return x, fout, ib


def _same(ib, ibo):
    return all(getattr(ib, k) == ibo[k] for k in COUNT_KEYS)


@pytest.mark.parametrize("fcn,x0,ans,args", [(rosenbrock, [0.0, 0.0], [1.0, 1.0], None),
                                              (beale, [1.0, 1.0], [3.0, 0.5], None),
                                              (rosenbrock, [0.0, 0.0], [1.0, 1.0], 1.0e2)])
def test_bfgs_1_2_3(oracle, fcn, x0, ans, args):
    """test_bfgs_1 (Rosenbrock), _2 (Beale), _3 (Rosenbrock with args): minimiser within 1e-5, and the run is
    bit-identical to the CPU path (FD gradient, line search, Cholesky update / downdate)."""
    x, fout, ib = _solve(fcn, 2, x0, args=args)
    assert np.abs(x - np.array(ans)).max() <= 1e-5
    rc, xo, fo, ibo = oracle.bfgs_solve(lambda v: fcn(v, args), 2, x0)
    assert rc == 0 and _same(ib, ibo), (ib.as_dict(), ibo)
    assert np.array_equal(x, xo) and fout == fo


def test_bfgs_analytic_gradient(oracle):
    x, fout, ib = _solve(rosenbrock, 2, [-1.2, 1.0], grad=rosenbrock_grad)
    assert np.abs(x - 1.0).max() <= 1e-6
    rc, xo, fo, ibo = oracle.bfgs_solve(lambda v: rosenbrock(v), 2, [-1.2, 1.0], grad=lambda v, g: rosenbrock_grad(v, g))
    assert rc == 0 and _same(ib, ibo) and np.array_equal(x, xo)


@pytest.mark.parametrize("n", [1, 2, 9, 64, 300, 1100])
def test_chol_rank1_update_and_downdate_bitwise(ds, oracle, n):
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    B = M.T @ M + n * np.eye(n)
    rc, R = oracle.chol_factor_upper(B)
    assert rc == 0
    u = rng.standard_normal(n)
    Rt = torch.from_numpy(np.ascontiguousarray(R)).to(ds.device)
    assert ds.chol_rank1(Rt, torch.from_numpy(u.copy()).to(ds.device), downdate=False) == 0
    R1 = oracle.chol_update(R, u)
    assert np.array_equal(Rt.cpu().numpy(), R1)
    assert np.abs(R1.T @ R1 - (B + np.outer(u, u))).max() <= 1e-10 * n * n
    info = ds.chol_rank1(Rt, torch.from_numpy(u.copy()).to(ds.device), downdate=True)
    rc, R2 = oracle.chol_downdate(R1, u)
    assert info == rc == 0
    assert np.array_equal(Rt.cpu().numpy(), R2)
    # a downdate that would lose positive definiteness is reported, not applied
    big = 100.0 * np.sqrt(np.abs(B).max()) * np.ones(n)
    assert ds.chol_rank1(Rt, torch.from_numpy(big).to(ds.device), downdate=True) == 1
    assert oracle.chol_downdate(R2, big)[0] == 1


@pytest.mark.parametrize("m,n", [(64, 8), (512, 64), (300, 37)])
def test_dq_bfgs_batch_bitwise(ds, oracle, m, n):
    """bfgs on 0.5 ||r(x)||^2 of the device model: FD gradient from the residual panel kernel + k_bf_fd_gradient,
    Hessian factor on the device; x, f and all counts bit-identical to the CPU path."""
    nprob = 2
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=77, spread=0.1)
    x = x0.clone()
    fout, ibs, status = ds.bfgs_solve_batch(A, b, 0.5, x, opts=ds.options(max_evals=300, gtol=1e-8, xtol=1e-12))
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        rc, xo, fo, ibo, _ = oracle.dq_bfgs_solve(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(),
                                                  opts=oracle.default_options(max_evals=300, gtol=1e-8, xtol=1e-12))
        assert status[p] == rc, (status[p], rc, ibs[p], ibo)
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (k, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo)
        assert fout[p] == fo


@pytest.mark.parametrize("m,n,opts", [(256, 24, dict(max_evals=300, gtol=1e-8, xtol=1e-12)),
                                      (400, 48, dict(max_evals=40, gtol=1e-10, xtol=1e-14)),
                                      (300, 37, dict(max_evals=300, gtol=1e-8, xtol=1e-12, use_line_search=0)),
                                      (128, 16, dict(max_evals=300, gtol=1e-3, xtol=1e-12))])
def test_dq_bfgs_lockstep_batch_bitwise(ds, oracle, m, n, opts):
    """The lock-step state machine (nlh_kernels_bfgs_batch.h) on a batch whose problems take different numbers of
    iterations and backtracking steps, converge on the gradient at different times (one at its start point), run out of
    evaluations, or step without a line search: every problem bit-identical to its CPU solve, counts, status and
    objective value too."""
    nprob = 18
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=909, spread=0.15)
    x0 = xt + (x0 - xt) * torch.linspace(0.0, 2.5, nprob, dtype=torch.float64, device=x0.device)[:, None]
    x = x0.clone()
    fout, ibs, status = ds.bfgs_solve_batch(A, b, 0.5, x, opts=ds.options(**opts))
    seen = set()
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        rc, xo, fo, ibo, _ = oracle.dq_bfgs_solve(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(), opts=oracle.default_options(**opts))
        assert status[p] == rc, (p, status[p], rc, ibs[p], ibo)
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (p, k, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo), p
        assert fout[p] == fo, p
        seen.add((ibo["iter_count"], ibo["fcn_count"]))
    assert len(seen) > 1                                             # the batch is not in step
