"""The replicated-batch random sweep of the exact factor policy, one case at a time: shared by the collected slice
(tests/test_gpu_random_parity.py::test_exact_policy_soak_slice) and the open-ended script (tests/soak_exact_policy.py).

A case = one to three random base problems (sizes around every kernel switch, four generator regimes, three trust-region
scalings, duplicated and zero columns), solved alone or REPLICATED into a batch of up to 1500 copies -- launches with more
workgroups than the chip holds, every form of the trailing pass, several sub-batches in flight.  Every copy must carry the
bits of the CPU oracle's solution of its original (x, fvec, status, all counts)."""
import numpy as np

KEYS = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")
SIZES = [1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 63, 64, 65, 95, 96, 97, 100, 127, 128, 129, 150, 200, 255, 256, 257, 300]


def run_case(ds, oracle, rng, sizes=SIZES):
    """Returns (description dict, first mismatch or None)."""
    n = rng.choice(sizes)
    m = n + rng.choice([0, 1, 2, 7, 31, 64, 100, 500, 1500, 4200])
    base = rng.choice([1, 2, 3])
    reps = rng.choice([1, 1, 1, 20, 150, 500]) if m * n < 200000 else 1
    gen = rng.choice([{}, dict(sigma=0.0), dict(gamma=2.0, sigma=0.1, spread=5.0), dict(gamma=10.0, sigma=1.0, spread=50.0)])
    opt = rng.choice([{}, dict(factor=0.1), dict(factor=1.0)])
    seed = rng.randrange(1, 100000)
    what = dict(m=m, n=n, base=base, reps=reps, gen=gen, opt=opt, seed=seed)
    A, b, xt, x0 = ds.generate(base, m, n, seed0=seed, square_shift=(m == n), **gen)
    if rng.random() < 0.2 and n >= 4:
        A[0, 1, :] = A[0, 0, :]
        x0[0, 1] = x0[0, 0]
    if rng.random() < 0.1 and n >= 4:
        A[0, 2, :] = 0.0
    if reps > 1:
        A = A.repeat(reps, 1, 1).contiguous()
        b = b.repeat(reps, 1).contiguous()
        x0 = x0.repeat(reps, 1).contiguous()
    nprob = base * reps
    x = x0.clone()
    me = 40 * (n + 1)
    fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=me, factor_policy=2, **opt))
    xs, fs = x.cpu().numpy(), fvec.cpu().numpy()
    for p0 in range(base):
        Ah = np.asfortranarray(A[p0].cpu().numpy().T)
        rc, xo, fo, ibo = oracle.dq_lm_solve(Ah, b[p0].cpu().numpy(), 0.5, x0[p0].cpu().numpy(),
                                             opts=oracle.default_options(max_evals=me, **opt))[:4]
        for p in range(p0, nprob, base):
            ok = status[p] == rc and all(ibs[p][k] == ibo[k] for k in KEYS) and \
                np.array_equal(xs[p], xo, equal_nan=True) and np.array_equal(fs[p], fo, equal_nan=True)
            if not ok:
                return what, dict(what, p=p)
    return what, None
