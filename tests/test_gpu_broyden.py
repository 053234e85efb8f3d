"""GPU parity tests for quasi_newton_solver (qns_solve, src/nonlin_solve.f90:156-427) and the dense
kernels behind it: Householder QR with Q formed, the rank-one QR update, the triangular solve.

The CPU restatement performs every sum in ascending index order and the kernels do exactly the same
operations on every element, so the comparisons are bitwise.  Problems follow the reference's
test_quasinewton_1..4 (tests/nonlin_test_solve.f90:187-360, 851-896)."""
import numpy as np
import pytest
import torch

import problems_ref as P

pytestmark = pytest.mark.gpu

COUNT_KEYS = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng",
              "converge_on_zero_diff")


def _dev(M, dev):
    """numpy matrix -> device tensor holding it column-major (tensor[j][i] = M[i, j])."""
    return torch.from_numpy(np.ascontiguousarray(M.T)).to(dev)


@pytest.mark.parametrize("n", [1, 2, 7, 64, 130, 300])
def test_qr_factor_full_bitwise(ds, oracle, n):
    rng = np.random.default_rng(n)
    Ms = [rng.standard_normal((n, n)) for _ in range(2)]
    if n >= 7:
        Ms[1][3:, 2] = 0.0                         # an H = I step (column already upper triangular)
    B = torch.stack([_dev(M, ds.device) for M in Ms])
    Q, Rt = ds.qr_factor_full(B)
    for p, M in enumerate(Ms):
        qo, ro = oracle.qr_factor_full(M)
        assert np.array_equal(Rt[p].cpu().numpy(), ro)
        assert np.array_equal(Q[p].cpu().numpy().T, qo)
        assert np.abs(qo @ ro - M).max() <= 1e-12 * max(1.0, np.abs(M).max()) * n


@pytest.mark.parametrize("n", [1, 2, 7, 64, 300, 1100])
def test_qr_rank1_update_bitwise(ds, oracle, n):
    rng = np.random.default_rng(100 + n)
    nprob = 2
    Ms = [rng.standard_normal((n, n)) for _ in range(nprob)]
    us = [rng.standard_normal(n) for _ in range(nprob)]
    vs = [rng.standard_normal(n) for _ in range(nprob)]
    qr = [oracle.qr_factor_full(M) for M in Ms]
    Q = torch.stack([_dev(q, ds.device) for q, r in qr])
    Rt = torch.stack([torch.from_numpy(np.ascontiguousarray(r)).to(ds.device) for q, r in qr])
    u = torch.from_numpy(np.stack(us)).to(ds.device)
    v = torch.from_numpy(np.stack(vs)).to(ds.device)
    ds.qr_rank1_update(Q, Rt, u, v)
    for p in range(nprob):
        q1, r1 = oracle.qr_rank1_update(qr[p][0], qr[p][1], us[p], vs[p])
        assert np.array_equal(Rt[p].cpu().numpy(), r1)
        assert np.array_equal(Q[p].cpu().numpy().T, q1)
        assert np.abs(q1 @ r1 - (Ms[p] + np.outer(us[p], vs[p]))).max() <= 1e-11 * n


@pytest.mark.parametrize("n", [1, 5, 64, 700])
def test_solve_upper_bitwise(ds, oracle, n):
    rng = np.random.default_rng(7 * n)
    R = np.triu(rng.standard_normal((n, n))) + 3.0 * np.eye(n)
    b = rng.standard_normal(n)
    b[n // 2] = 0.0
    Rt = torch.from_numpy(np.ascontiguousarray(R)).to(ds.device)[None]
    x = torch.from_numpy(b.copy()).to(ds.device)[None]
    ds.solve_upper(Rt, x)
    assert np.array_equal(x[0].cpu().numpy(), oracle.solve_upper(R, b))


def _solve_host(fcn, n, x0, jac=None, use_ls=True, args=None, jdelta=None):
    import nonlin_amd as nl
    obj = nl.vecfcn_helper()
    obj.set_fcn(fcn, n, n)
    if jac is not None:
        obj.set_jacobian(jac)
    s = nl.quasi_newton_solver()
    s.set_use_line_search(use_ls)
    if jdelta is not None:
        s.set_jacobian_interval(jdelta)
    x = np.array(x0, dtype=np.float64)
    f = np.zeros(n)
    ib = nl.iteration_behavior()
    s.solve(obj, x, f, ib, args=args)
    return x, f, ib


def _same(ib, ibo):
    return all(getattr(ib, k) == ibo[k] for k in COUNT_KEYS)


@pytest.mark.parametrize("ic", [(0.5, 0.5), (1.0, 1.0)])
@pytest.mark.parametrize("with_jac", [True, False])
def test_host_quasinewton_1(oracle, ic, with_jac):
    """test_quasinewton_1 / 3 (:187-234, :290-360): x -> (+-5, +-3) within 1e-6."""
    x, f, ib = _solve_host(P.fcn1, 2, ic, jac=P.jac1 if with_jac else None)
    rc, xo, fo, ibo = oracle.quasi_newton_solve(lambda a, b: P.fcn1(a, b, None), 2, ic,
                                                jac=(lambda a, b: P.jac1(a, b, None)) if with_jac else None)
    assert rc == 0
    assert abs(abs(x[0]) - 5.0) <= 1e-6 and abs(abs(x[1]) - 3.0) <= 1e-6
    assert _same(ib, ibo), (ib.as_dict(), ibo)
    assert np.array_equal(x, xo) and np.array_equal(f, fo)


@pytest.mark.parametrize("ic", [(0.5, 0.5), (1.0, 1.0)])
def test_host_quasinewton_2_no_linesearch(oracle, ic):
    """test_quasinewton_2 (:237-287): badly scaled, FD Jacobian, line search off."""
    x, f, ib = _solve_host(P.fcn2, 2, ic, use_ls=False)
    rc, xo, fo, ibo = oracle.quasi_newton_solve(lambda a, b: P.fcn2(a, b, None), 2, ic,
                                                opts=oracle.default_options(use_line_search=0))
    assert rc == 0
    assert abs(abs(x[0]) - 5.0e3) <= 1e-6 and abs(abs(x[1]) - 10.0) <= 1e-6
    assert _same(ib, ibo)
    assert np.array_equal(x, xo)


@pytest.mark.parametrize("with_jac", [False, True])
def test_host_quasinewton_3_args(oracle, with_jac):
    """test_quasinewton_3 (:290-360): the optional args reaches the user's routines."""
    x, f, ib = _solve_host(P.fcn1a, 2, [1.0, 1.0], jac=P.jac1a if with_jac else None, args=2.0)
    assert abs(abs(x[0]) - 5.0) <= 1e-6 and abs(abs(x[1]) - 3.0) <= 1e-6
    rc, xo, fo, ibo = oracle.quasi_newton_solve(lambda a, b: P.fcn1a(a, b, 2.0), 2, [1.0, 1.0],
                                                jac=(lambda a, b: P.jac1a(a, b, 2.0)) if with_jac else None)
    assert _same(ib, ibo)
    assert np.array_equal(x, xo)


def test_host_quasinewton_4_powell(oracle):
    """test_quasinewton_4 (:851-896): Powell badly scaled, analytic Jacobian, line search off, tol 1e-5."""
    x, f, ib = _solve_host(P.powell, 2, [0.0, 1.0], jac=P.powell_jac, use_ls=False)
    assert abs(x[0] - 1.098159e-5) <= 1e-5 and abs(x[1] - 9.106146) <= 1e-5
    rc, xo, fo, ibo = oracle.quasi_newton_solve(lambda a, b: P.powell(a, b, None), 2, [0.0, 1.0],
                                                jac=lambda a, b: P.powell_jac(a, b, None),
                                                opts=oracle.default_options(use_line_search=0))
    assert _same(ib, ibo)
    assert np.array_equal(x, xo)


def test_host_quasinewton_jacobian_interval(oracle):
    """set_jacobian_interval (:439-447) changes when the Jacobian is recomputed."""
    x1, f1, ib1 = _solve_host(P.fcn1, 2, [1.0, 1.0], jac=P.jac1, jdelta=1)
    rc, xo, fo, ibo = oracle.quasi_newton_solve(lambda a, b: P.fcn1(a, b, None), 2, [1.0, 1.0],
                                                jac=lambda a, b: P.jac1(a, b, None), jdelta=1)
    assert _same(ib1, ibo) and np.array_equal(x1, xo)
    x5, f5, ib5 = _solve_host(P.fcn1, 2, [1.0, 1.0], jac=P.jac1)
    assert ib1.jacobian_count > ib5.jacobian_count


@pytest.mark.parametrize("n,analytic,spread", [(16, True, 0.1), (64, True, 0.1), (64, False, 0.03), (256, True, 0.03),
                                               (256, True, 0.3)])
def test_dq_quasi_newton_batch_bitwise(ds, oracle, n, analytic, spread):
    """Square dense-quadratic systems (A <- 2I + A): x, fvec and every count equal the CPU path's bit for
    bit, including runs with Jacobian restarts."""
    nprob = 2
    A, b, xt, x0 = ds.generate(nprob, n, n, seed0=12345, sigma=0.0, spread=spread, square_shift=True)
    x = x0.clone()
    fvec, ibs, status = ds.quasi_newton_solve_batch(A, b, 0.5, x, analytic=analytic, opts=ds.options(max_evals=500))
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        rc, xo, fo, ibo, _ = oracle.dq_quasi_newton_solve(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(),
                                                          analytic=analytic, opts=oracle.default_options(max_evals=500))
        assert status[p] == rc
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (k, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo)
        assert np.array_equal(fvec[p].cpu().numpy(), fo)


def test_dq_quasi_newton_many_problems_concurrently(ds, oracle):
    """The batch entry points deal problems to worker threads with private streams (NLH_WORKERS, default 8): every
    problem must still come out bit-identical to the CPU path, whichever thread solved it and whatever ran beside it."""
    nprob, n = 24, 48
    A, b, xt, x0 = ds.generate(nprob, n, n, seed0=4321, sigma=0.0, spread=0.05, square_shift=True)
    x = x0.clone()
    fvec, ibs, status = ds.quasi_newton_solve_batch(A, b, 0.5, x, analytic=False, opts=ds.options(max_evals=500))
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        rc, xo, fo, ibo, _ = oracle.dq_quasi_newton_solve(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(), analytic=False,
                                                          opts=oracle.default_options(max_evals=500))
        assert status[p] == rc
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (p, k, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo), p
        assert np.array_equal(fvec[p].cpu().numpy(), fo), p


def test_host_readme_example_1_counts(oracle):
    """README.md:34-99 through the drop-in API on the GPU: 11 iterations, 15 function evaluations, 1 Jacobian evaluation
    (the counts the reference prints), x and f bit-identical to the CPU oracle."""
    x, f, ib = _solve_host(P.fcn1, 2, (1.0, 1.0), jdelta=20)
    assert (ib.iter_count, ib.fcn_count, ib.jacobian_count) == (11, 15, 1)
    rc, xo, fo, ibo = oracle.quasi_newton_solve(lambda a, b: P.fcn1(a, b, None), 2, [1.0, 1.0], jdelta=20)
    assert rc == 0 and _same(ib, ibo)
    assert np.array_equal(x, xo) and np.array_equal(f, fo)
    assert ("%.2e" % f[0], "%.2e" % f[1]) == ("3.23e-12", "7.05e-12")                         # README.md:94


@pytest.mark.parametrize("n,nprob,analytic,spread,jdelta", [(64, 40, True, 0.03, 5), (96, 24, False, 0.1, 5), (129, 32, True, 0.3, 2),
                                                            (200, 16, True, 1.0, 5)])
def test_quasi_newton_lockstep_batch_bitwise(ds, oracle, n, nprob, analytic, spread, jdelta):
    """quasi_newton_solver on a batch through the lock-step state machine (nlh_kernels_newton.h, broyden mode): problems
    that restart, update and back-track at different times share every launch.  Each must carry the bits, counts, flags
    and status the CPU path gives it alone."""
    A, b, xt, x0 = ds.generate(nprob, n, n, seed0=911, sigma=0.0, spread=spread, square_shift=True)
    x = x0.clone()
    fvec, ibs, status = ds.quasi_newton_solve_batch(A, b, 0.5, x, analytic=analytic, opts=ds.options(max_evals=80), jdelta=jdelta)
    xs, fs = x.cpu().numpy(), fvec.cpu().numpy()
    seen = set()
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        rc, xo, fo, ibo, _ = oracle.dq_quasi_newton_solve(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(), analytic=analytic,
                                                          opts=oracle.default_options(max_evals=80), jdelta=jdelta)
        assert status[p] == rc, (p, status[p], rc)
        for k in ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff"):
            assert ibs[p][k] == ibo[k], (p, k, ibs[p], ibo)
        assert np.array_equal(xs[p], xo, equal_nan=True), p
        assert np.array_equal(fs[p], fo, equal_nan=True), p
        seen.add((ibo["iter_count"], ibo["fcn_count"], ibo["jacobian_count"]))
    assert len(seen) > 1
