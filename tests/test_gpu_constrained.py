"""GPU parity tests for constrained_least_squares_solver (cls_solve, src/nonlin_least_squares.f90:938-1176):
Householder QR of the tall Jacobian with the reflectors applied to f, the dog-leg step and the bounded
trust-region loop.  The CPU restatement sums in ascending index order and the kernels do the same operations on
every element, so x, fvec and all counts are compared bit for bit.  Problems: the reference's
test_constrained_least_squares_1..4 and _bounds (tests/nonlin_test_solve.f90:975-1228)."""
import numpy as np
import pytest
import torch

import problems_ref as P

pytestmark = pytest.mark.gpu
BIG = float(np.finfo(np.float64).max)
COUNT_KEYS = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng",
              "converge_on_zero_diff")


def _solve_host(fcn, m, n, x0, jac=None, lower=None, upper=None, args=None, max_evals=None):
    import nonlin_amd as nl
    obj = nl.vecfcn_helper()
    obj.set_fcn(fcn, m, n)
    if jac is not None:
        obj.set_jacobian(jac)
    s = nl.constrained_least_squares_solver()
    if lower is not None:
        s.set_lower_limits(lower)
    if upper is not None:
        s.set_upper_limits(upper)
    if max_evals:
        s.set_max_fcn_evals(max_evals)
    x = np.array(x0, dtype=np.float64)
    f = np.zeros(m)
    ib = nl.iteration_behavior()
    s.solve(obj, x, f, ib, args=args)
    return x, f, ib, s


def _same(ib, ibo):
    return all(getattr(ib, k) == ibo[k] for k in COUNT_KEYS)


@pytest.mark.parametrize("ic", [(0.5, 0.5), (1.0, 1.0)])
def test_constrained_least_squares_1(oracle, ic):
    """:975-1027: analytic Jacobian, infinite limits given explicitly."""
    x, f, ib, s = _solve_host(P.fcn1, 2, 2, ic, jac=P.jac1, lower=[-BIG, -BIG], upper=[BIG, BIG])
    rc, xo, fo, ibo = oracle.cls_solve(lambda a, b: P.fcn1(a, b, None), 2, 2, ic, jac=lambda a, b: P.jac1(a, b, None),
                                       lower=[-BIG, -BIG], upper=[BIG, BIG])
    assert rc == 0
    assert abs(abs(x[0]) - 5.0) <= 1e-6 and abs(abs(x[1]) - 3.0) <= 1e-6
    assert _same(ib, ibo), (ib.as_dict(), ibo)
    assert np.array_equal(x, xo) and np.array_equal(f, fo)


@pytest.mark.parametrize("ic", [(0.5, 0.5), (1.0, 1.0)])
def test_constrained_least_squares_2(oracle, ic):
    """:1030-1077: poorly scaled system, FD Jacobian, 5000 evaluations allowed, no limits set
    (the solver installs +-huge and keeps them, :999-1009)."""
    x, f, ib, s = _solve_host(P.fcn2, 2, 2, ic, max_evals=5000)
    rc, xo, fo, ibo = oracle.cls_solve(lambda a, b: P.fcn2(a, b, None), 2, 2, ic, opts=oracle.default_options(max_evals=5000))
    assert rc == 0
    assert abs(x[0] - 5.0e3) <= 1e-6 and abs(x[1] - 10.0) <= 1e-6
    assert _same(ib, ibo), (ib.as_dict(), ibo)
    assert np.array_equal(x, xo)
    assert s.get_lower_limits().tolist() == [-BIG, -BIG] and s.get_upper_limits().tolist() == [BIG, BIG]


def test_constrained_least_squares_3_matches_lm(oracle):
    """:1080-1123: the README cubic fit; the constrained and the plain LM solver agree within 1e-5."""
    import nonlin_amd as nl
    xc, f, ib, s = _solve_host(P.lsfcn1, 21, 4, [1.0] * 4)
    obj = nl.vecfcn_helper()
    obj.set_fcn(P.lsfcn1, 21, 4)
    x = np.ones(4)
    nl.least_squares_solver().solve(obj, x, np.zeros(21))
    assert np.abs(x - xc).max() <= 1e-5
    rc, xo, fo, ibo = oracle.cls_solve(lambda a, b: P.lsfcn1(a, b, None), 21, 4, [1.0] * 4)
    assert _same(ib, ibo) and np.array_equal(xc, xo)


def test_constrained_least_squares_4_args(oracle):
    """:1126-1184: class(*) args reaches the callbacks, FD then analytic Jacobian."""
    for jac in (None, P.jac1a):
        x, f, ib, s = _solve_host(P.fcn1a, 2, 2, [1.0, 1.0], jac=jac, args=2.0)
        assert abs(abs(x[0]) - 5.0) <= 1e-6 and abs(abs(x[1]) - 3.0) <= 1e-6
        rc, xo, fo, ibo = oracle.cls_solve(lambda a, b: P.fcn1a(a, b, 2.0), 2, 2, [1.0, 1.0],
                                           jac=(lambda a, b: P.jac1a(a, b, 2.0)) if jac else None)
        assert _same(ib, ibo) and np.array_equal(x, xo)


def test_constrained_least_squares_bounds(oracle):
    """:1187-1228: the solution respects [4, 5.6] x [2, 3.6] from a start outside the box."""
    low, high = [4.0, 2.0], [5.6, 3.6]
    x, f, ib, s = _solve_host(P.fcn1, 2, 2, [1.0, 1.0], lower=low, upper=high)
    assert np.all(x >= np.array(low)) and np.all(x <= np.array(high))
    rc, xo, fo, ibo = oracle.cls_solve(lambda a, b: P.fcn1(a, b, None), 2, 2, [1.0, 1.0], lower=low, upper=high)
    assert rc == 0 and _same(ib, ibo) and np.array_equal(x, xo)


def test_active_bound(oracle):
    """A bound that cuts the unconstrained minimiser off: the iterate stops on the face."""
    low, high = [-BIG, -BIG], [4.5, BIG]
    rc, xo, fo, ibo = oracle.cls_solve(lambda a, b: P.fcn1(a, b, None), 2, 2, [1.0, 1.0], lower=low, upper=high,
                                       opts=oracle.default_options(max_evals=200))
    import nonlin_amd as nl
    try:
        x, f, ib, s = _solve_host(P.fcn1, 2, 2, [1.0, 1.0], lower=low, upper=high, max_evals=200)
        got_rc = 0
    except nl.NonlinError as e:
        got_rc, x = e.code, None
    assert got_rc == rc
    if x is not None:
        assert x[0] <= 4.5 and np.array_equal(x, xo)


@pytest.mark.parametrize("m,n,bounded", [(64, 8, False), (512, 64, False), (512, 64, True), (300, 37, True), (20000, 12, True)])
def test_dq_cls_batch_bitwise(ds, oracle, m, n, bounded):
    """Device-model problems (FD Jacobian fused into the panel kernel): bitwise equal to the CPU path."""
    nprob = 2
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=2024)
    lower = upper = None
    if bounded:
        lower = np.full(n, -0.5)
        upper = np.full(n, 0.6)
    x = x0.clone()
    fvec, ibs, status = ds.cls_solve_batch(A, b, 0.5, x, opts=ds.options(max_evals=500), lower=lower, upper=upper)
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        rc, xo, fo, ibo, _ = oracle.dq_cls_solve(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(),
                                                 opts=oracle.default_options(max_evals=500), lower=lower, upper=upper)
        assert status[p] == rc
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (k, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo)
        assert np.array_equal(fvec[p].cpu().numpy(), fo)
        if bounded:
            assert np.all(xo >= lower) and np.all(xo <= upper)


@pytest.mark.parametrize("m,n,lo,hi,max_evals", [(256, 24, -0.3, 0.25, 500), (700, 96, -2.0, 2.0, 500), (384, 40, -0.05, 0.04, 500),
                                                 (256, 24, None, None, 6)])
def test_dq_cls_lockstep_batch_bitwise(ds, oracle, m, n, lo, hi, max_evals):
    """The lock-step state machine (nlh_kernels_cls.h) on a batch whose problems take different numbers of iterations,
    hit their bounds, leave the trust region (steepest-descent leg), go through the projected backtracking, run out of
    evaluations, or start from a non-finite point: every problem bit-identical to its CPU solve, counts and status too."""
    nprob = 20
    gen = ds.generate
    A, b, xt, x0 = gen(nprob, m, n, seed0=4242, spread=0.2)
    x0 = x0 * torch.linspace(0.2, 3.0, nprob, dtype=torch.float64, device=x0.device)[:, None]
    x0[7, 3] = float("nan")                                          # the reference returns silently (:1028-1031)
    lower = None if lo is None else np.full(n, lo)
    upper = None if hi is None else np.full(n, hi)
    x = x0.clone()
    fvec, ibs, status = ds.cls_solve_batch(A, b, 0.5, x, opts=ds.options(max_evals=max_evals), lower=lower, upper=upper)
    iters = set()
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        rc, xo, fo, ibo, _ = oracle.dq_cls_solve(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(),
                                                 opts=oracle.default_options(max_evals=max_evals), lower=lower, upper=upper)
        assert status[p] == rc, (p, status[p], rc)
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (p, k, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo, equal_nan=True), p
        if p != 7:
            assert np.array_equal(fvec[p].cpu().numpy(), fo, equal_nan=True), p
        iters.add((ibo["iter_count"], ibo["fcn_count"]))
    assert len(iters) > 1                                            # the batch is not in step
