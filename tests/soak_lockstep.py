"""Random-size soak of the lock-step bounded least-squares, BFGS, Newton and quasi-Newton batches (nlh_kernels_cls.h,
nlh_kernels_bfgs_batch.h, nlh_kernels_newton.h) against the CPU oracle: every problem of every batch must carry the oracle's bits, counts and status.  Test
infrastructure (run from tests/test_gpu_random_parity.py, or stand-alone on a GPU box:
python tests/soak_lockstep.py SEED NCASES)."""
import os
import sys
import time

import numpy as np

CK = ("iter_count", "fcn_count", "jacobian_count", "gradient_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")


def run(ds, O, seed, ncase, verbose=False):
    """Returns (problems checked, list of mismatch descriptions)."""
    import torch
    rng = np.random.default_rng(seed)
    misses, total = [], 0
    for case in range(ncase):
        n = int(rng.integers(2, 90)); m = n + int(rng.integers(0, 600)); nb = int(rng.integers(1, 24))
        spread = float(rng.choice([0.0, 0.05, 0.2])); s0 = int(rng.integers(1, 10**6))
        A, b, xt, x0 = ds.generate(nb, m, n, seed0=s0, spread=spread)
        x0 = x0 * torch.tensor(rng.uniform(0.3, 2.5, nb), dtype=torch.float64, device=x0.device)[:, None]
        if case % 4 >= 2:                                            # square systems: Newton (case % 4 == 2) / quasi-Newton (3)
            m = n
            A, b, xt, x0 = ds.generate(nb, n, n, seed0=s0, sigma=0.0, square_shift=True, spread=spread)
            x0 = xt + (x0 - xt) * torch.tensor(rng.uniform(0.2, 1.5, nb), dtype=torch.float64, device=x0.device)[:, None]
            an = bool(rng.integers(2)); me = int(rng.choice([8, 60, 500])); ls = int(rng.choice([1, 1, 0]))
            oo = dict(max_evals=me, use_line_search=ls)
            x = x0.clone()
            if case % 4 == 2:
                fv, ibs, st = ds.newton_solve_batch(A, b, 0.5, x, analytic=an, opts=ds.options(**oo))
                ref = lambda p: O.dq_newton_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(),
                                                  analytic=an, opts=O.default_options(**oo))
                name = "newton"
            else:
                jd = int(rng.choice([1, 3, 5]))
                fv, ibs, st = ds.quasi_newton_solve_batch(A, b, 0.5, x, analytic=an, opts=ds.options(**oo), jdelta=jd)
                ref = lambda p: O.dq_quasi_newton_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(),
                                                        analytic=an, opts=O.default_options(**oo), jdelta=jd)
                name = "quasi-newton"
        elif case % 2 == 0:
            w = float(rng.choice([0.02, 0.3, 2.0, 50.0])); lo, hi = np.full(n, -w), np.full(n, 0.8 * w)
            me = int(rng.choice([5, 60, 500]))
            x = x0.clone()
            fv, ibs, st = ds.cls_solve_batch(A, b, 0.5, x, opts=ds.options(max_evals=me), lower=lo, upper=hi)
            ref = lambda p: O.dq_cls_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(),
                                           opts=O.default_options(max_evals=me), lower=lo, upper=hi)
            name = "cls"
        else:
            ob = dict(max_evals=int(rng.choice([20, 120, 400])), gtol=float(rng.choice([1e-8, 1e-4])), xtol=1e-12,
                      use_line_search=int(rng.choice([1, 1, 0])))
            x = x0.clone()
            fv, ibs, st = ds.bfgs_solve_batch(A, b, 0.5, x, opts=ds.options(**ob))
            ref = lambda p: O.dq_bfgs_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(),
                                            opts=O.default_options(**ob))
            name = "bfgs"
        for p in range(nb):
            r = ref(p); total += 1
            ok = (st[p] == r[0]) and np.array_equal(x[p].cpu().numpy(), r[1], equal_nan=True) and \
                all(ibs[p][k] == r[3][k] for k in CK if k in r[3])
            if not ok:
                misses.append(dict(solver=name, case=case, p=p, m=m, n=n, nb=nb, status=int(st[p]), rc=r[0], got=ibs[p], want=r[3]))
                if verbose:
                    print("MISMATCH", misses[-1], flush=True)
    return total, misses


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nonlin_amd.device import DeviceSolver
    from oracle import pyoracle
    t0 = time.time()
    total, misses = run(DeviceSolver(0), pyoracle, int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 30, True)
    print("problems checked", total, "mismatches", len(misses), "in %.0f s" % (time.time() - t0))
