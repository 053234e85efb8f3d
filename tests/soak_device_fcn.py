#!/usr/bin/env python3
"""Soak of the open device-residual path (not collected by pytest; `python tests/soak_device_fcn.py [cases] [seed]` on a GPU
box): random shapes, batch sizes, problem difficulties, factor policies and sub-batch counts; the dense-quadratic family
through its launchers (nlh_dq_device_fcn / _jac + nlh_*_solve_batch_device) must reproduce, bit for bit, what the family's
own entry points give -- which the rest of the suite holds to the CPU oracle.  Covers LM (FD and analytic Jacobian
launcher), Newton, quasi-Newton and bounded least squares, odd sizes, partial tiles of k_fd_jacobian_qrx, stragglers."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonlin_amd.device import DeviceSolver  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20251004)
    ds = DeviceSolver(0)
    bad = 0
    only = [int(v) for v in os.environ.get("SOAK_ONLY", "").split(",") if v]
    for c in range(cases):
        kind = rng.choice(["lm", "lm", "lm", "newton", "broyden", "cls"])
        n = int(rng.integers(1, 200))
        if kind in ("lm", "cls"):
            m = n + int(rng.integers(0, 600))
        else:
            m = n
        nprob = int(rng.choice([1, 2, 3, 7, 20, 64, 150]))
        if nprob * m * n > 40e6:
            nprob = max(1, int(40e6 // (m * n)))
        hard = rng.random() < 0.4
        gen = dict(gamma=2.0, sigma=0.1, spread=5.0) if hard else dict(gamma=0.5, sigma=1e-3, spread=0.3)
        if kind in ("newton", "broyden"):
            gen = dict(gamma=0.5, sigma=0.0, spread=float(rng.choice([0.03, 0.3])), square_shift=True)
        g = gen["gamma"]
        A, b, xt, x0 = ds.generate(nprob, m, n, seed0=int(rng.integers(1, 1 << 30)), **gen)
        fcn, jac, ctx = ds.dq_launchers(A, b, g)
        pol = int(rng.choice([2, 2, 2, 0, 1])) if kind == "lm" else 2
        if pol != 2 and n > 150:
            pol = 2
        o = ds.options(max_evals=int(rng.choice([60, 500])), factor_policy=pol, sub_batches=int(rng.choice([0, 1, 2, 3])),
                       factor=float(rng.choice([100.0, 0.1])) if kind == "lm" else 100.0)
        if pol != 2:
            o.fuse_fd = 0                                      # the fused FD->Gram kernel sums in another order than FD -> J -> Gram
        x1, x2 = x0.clone(), x0.clone()
        use_jac = bool(rng.random() < 0.3)
        if only and c not in only:
            continue
        try:
            if kind == "lm":
                r1 = ds.lm_solve_batch(A, b, g, x1, o) if not use_jac else None
                if only and not use_jac:                       # replay mode: is either side run-to-run deterministic?
                    x3, x4 = x0.clone(), x0.clone()
                    ds.lm_solve_batch(A, b, g, x3, o)
                    ds.lm_solve_batch_device(fcn, ctx, m, x4, jac=None, opts=o)
                    ds.lm_solve_batch_device(fcn, ctx, m, x2, jac=None, opts=o)
                    print("  built-in twice equal:", bool((x3 == x1).all()), " launcher twice equal:", bool((x4 == x2).all()))
                    x2 = x0.clone()
                r2 = ds.lm_solve_batch_device(fcn, ctx, m, x2, jac=jac if use_jac else None, opts=o)
                if use_jac:                                    # analytic launcher: no built-in twin; check against itself with one sub-batch
                    x1 = x0.clone()
                    o.sub_batches = 1
                    r1 = ds.lm_solve_batch_device(fcn, ctx, m, x1, jac=jac, opts=o)
            elif kind == "cls":
                lo, hi = np.full(n, -0.6), np.full(n, 0.7)
                r1 = ds.cls_solve_batch(A, b, g, x1, o, lower=lo, upper=hi)
                r2 = ds.cls_solve_batch_device(fcn, ctx, m, x2, opts=o, lower=lo, upper=hi)
            else:
                br = kind == "broyden"
                r1 = (ds.quasi_newton_solve_batch if br else ds.newton_solve_batch)(A, b, g, x1, analytic=use_jac, opts=o)
                r2 = ds.square_solve_batch_device(fcn, ctx, x2, jac=jac if use_jac else None, opts=o, broyden=br)
            ok = (r1[1] == r2[1] and r1[2] == r2[2] and np.array_equal(x1.cpu().numpy(), x2.cpu().numpy(), equal_nan=True)
                  and np.array_equal(r1[0].cpu().numpy(), r2[0].cpu().numpy(), equal_nan=True))
        except Exception as e:                                 # noqa: BLE001
            ok = False
            print("EXC", repr(e))
        if not ok:
            bad += 1
            if not isinstance(r1, type(None)) and 'r2' in dir():
                d = np.abs(x1.cpu().numpy() - x2.cpu().numpy()).max(axis=1)
                print("  max|dx| per problem", d[:8], "iters", [(u["iter_count"], v["iter_count"]) for u, v in zip(r1[1], r2[1])][:8],
                      "status", r1[2][:8], r2[2][:8])
            print(f"MISMATCH case {c}: {kind} nprob={nprob} m={m} n={n} hard={hard} policy={pol} sub={o.sub_batches} jac={use_jac}", flush=True)
        del A, b, xt, x0, x1, x2
    print(f"device-fcn soak: {cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
