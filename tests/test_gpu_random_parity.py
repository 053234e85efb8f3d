"""Randomised parity sweep of the exact factor policy: sizes around every kernel switch, the four generator regimes,
three trust-region scalings, duplicated and zero columns.  x, fvec, the status and all counts must be bit-identical to
the CPU oracle.  (A 7-minute run of the same generator, 869 cases, found no mismatch; this is its first 40 cases.)"""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")
SIZES = [1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 63, 64, 65, 95, 96, 97, 100, 127, 128, 129, 150, 200, 255, 256, 257, 300]


def test_exact_policy_random_sweep(ds, oracle):
    rng = random.Random(7)
    for case in range(40):
        n = rng.choice(SIZES)
        m = n + rng.choice([0, 1, 2, 7, 31, 64, 100, 500, 1500])
        nprob = rng.choice([1, 2, 3])
        gen = rng.choice([{}, dict(sigma=0.0), dict(gamma=2.0, sigma=0.1, spread=5.0), dict(gamma=10.0, sigma=1.0, spread=50.0)])
        opt = rng.choice([{}, dict(factor=0.1), dict(factor=1.0)])
        seed = rng.randrange(1, 100000)
        A, b, xt, x0 = ds.generate(nprob, m, n, seed0=seed, square_shift=(m == n), **gen)
        if rng.random() < 0.2 and n >= 4:
            A[0, 1, :] = A[0, 0, :]
            x0[0, 1] = x0[0, 0]
        if rng.random() < 0.1 and n >= 4:
            A[0, 2, :] = 0.0
        x = x0.clone()
        me = 40 * (n + 1)
        fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=me, factor_policy=2, **opt))
        for p in range(nprob):
            Ah = np.asfortranarray(A[p].cpu().numpy().T)
            rc, xo, fo, ibo = oracle.dq_lm_solve(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(),
                                                 opts=oracle.default_options(max_evals=me, **opt))[:4]
            where = dict(case=case, m=m, n=n, p=p, gen=gen, opt=opt, seed=seed)
            assert status[p] == rc, where
            for k in KEYS:
                assert ibs[p][k] == ibo[k], (where, k, ibs[p], ibo)
            assert np.array_equal(x[p].cpu().numpy(), xo, equal_nan=True), where
            assert np.array_equal(fvec[p].cpu().numpy(), fo, equal_nan=True), where


def test_exact_policy_soak_slice(ds, oracle):
    """The first 200 cases (seed 11) of the replicated-batch soak (tests/soak_cases.py, tests/soak_exact_policy.py): random
    base problems alone or replicated into batches of up to 1500 copies, so that launches have more workgroups than the
    chip holds and every form of the trailing pass and the sub-batch driver are exercised; every copy must carry the
    oracle's bits.  This is the sweep that found the slot-map race of round 2."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import soak_cases
    rng = random.Random(11)
    for case in range(200):
        what, miss = soak_cases.run_case(ds, oracle, rng)
        assert miss is None, dict(miss, case=case)


def test_lockstep_soak_slice(ds, oracle):
    """Forty random batches (seed 1) of the lock-step soak (tests/soak_lockstep.py: bounded least squares, BFGS, Newton,
    quasi-Newton in turn): random sizes, batch sizes, bounds, budgets, tolerances, analytic / FD Jacobians, refresh
    intervals; every problem of every batch must carry the oracle's bits, counts and status.  (Several seeds, ~3000
    problems, ran clean during round 3.)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import soak_lockstep
    total, misses = soak_lockstep.run(ds, oracle, 1, 40)
    assert total > 200 and not misses, misses[:3]


def test_exact_long_column_soak_slice(ds, oracle):
    """Forty random matrices with columns of 4097 .. 26000 rows, 1 .. 69 columns, alone or in batches of up to 20 (the
    workgroup-per-column sweep with the update one step behind and the pipelined NORM2 for a handful of problems, the
    lane-per-column passes under the long-column pivot kernel for more), every adversarial kind of
    test_gpu_lmfactor_exact.py: bit-identical to the oracle's lmfactor and Q^T f."""
    import test_gpu_lmfactor_exact as T
    rng = np.random.default_rng(1)
    for case in range(40):
        m = int(rng.integers(4097, 26000)); n = int(rng.integers(1, 70)); copies = int(rng.choice([1, 1, 2, 3, 8, 9, 20]))
        if rng.random() < 0.2:
            m = int(rng.integers(4097, 4200))
        kind = T.KINDS[int(rng.integers(len(T.KINDS)))]
        a = T._matrix(kind, m, n, rng)
        f = rng.standard_normal(m)
        try:
            T._check(ds, oracle, a, f, copies=copies)
        except AssertionError as e:
            raise AssertionError(dict(case=case, kind=kind, m=m, n=n, copies=copies)) from e


@pytest.mark.parametrize("m,n,base,copies,gen,sub_batches", [
    (319, 255, 3, 500, dict(sigma=0.0), 0),
    (319, 255, 3, 500, dict(sigma=0.0), 1),
    (301, 300, 3, 150, dict(gamma=10.0, sigma=1.0, spread=50.0), 0),
    (152, 150, 3, 500, dict(gamma=10.0, sigma=1.0, spread=50.0), 0),
])
def test_exact_policy_more_workgroups_than_the_chip_holds(ds, oracle, m, n, base, copies, gen, sub_batches):
    """A few short problems replicated into a batch whose launches have far more workgroups than the chip holds at once
    and rows so few that a workgroup can end before a later one of the same launch starts: every copy must carry the
    bits of its original, and the originals the oracle's.  (Found by a soak run: the flushing pass used to update the
    slot maps in place, and a late workgroup of the same launch could read the updated entry -- different bits in a few
    copies per thousand, hundreds with several sub-batches in flight.)"""
    A, b, xt, x0 = ds.generate(base, m, n, seed0=5927, square_shift=(m == n), **gen)
    A = A.repeat(copies, 1, 1).contiguous()
    b = b.repeat(copies, 1).contiguous()
    x0 = x0.repeat(copies, 1).contiguous()
    nprob = base * copies
    me = 40 * (n + 1)
    for trial in range(2):
        x = x0.clone()
        fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=me, factor_policy=2, sub_batches=sub_batches))
        xs, fs = x.cpu().numpy(), fvec.cpu().numpy()
        for p0 in range(base):
            assert bool((xs[p0::base] == xs[p0]).all() | np.isnan(xs[p0]).any()), (trial, p0)
            assert bool((fs[p0::base] == fs[p0]).all() | np.isnan(fs[p0]).any()), (trial, p0)
            assert all(status[p] == status[p0] and all(ibs[p][k] == ibs[p0][k] for k in KEYS) for p in range(p0, nprob, base))
            if trial == 0:
                Ah = np.asfortranarray(A[p0].cpu().numpy().T)
                rc, xo, fo, ibo = oracle.dq_lm_solve(Ah, b[p0].cpu().numpy(), 0.5, x0[p0].cpu().numpy(),
                                                     opts=oracle.default_options(max_evals=me))[:4]
                assert status[p0] == rc
                assert all(ibs[p0][k] == ibo[k] for k in KEYS)
                assert np.array_equal(xs[p0], xo, equal_nan=True) and np.array_equal(fs[p0], fo, equal_nan=True)
