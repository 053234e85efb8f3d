#!/usr/bin/env python3
"""Where the time goes when the Levenberg-Marquardt parameter is NOT zero: the 'hard' generator regime (gamma = 2,
sigma = 0.1, spread = 5: rejected trials, lmpar's iteration on most steps) instead of the bench family's Gauss-Newton-like
steps.  python profiles/scripts/hard_family.py NPROB M N [reps]   (run under rocprofv3 --kernel-trace --stats for the shares)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nonlin_amd.device import DeviceSolver  # noqa: E402

nprob, m, n = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ds = DeviceSolver(0)
A, b, xt, x0 = ds.generate(nprob, m, n, seed0=12345, gamma=2.0, sigma=0.1, spread=5.0)
o = ds.options(max_evals=500)
for r in range(reps + 1):
    x = x0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f, ib, st = ds.lm_solve_batch(A, b, 2.0, x, o)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    its = sum(i["iter_count"] for i in ib)
    print(f"hard {nprob} x {m}x{n}: {dt * 1e3:9.2f} ms per solve, {its} LM iterations ({its / dt:9.1f} it/s), max iters {max(i['iter_count'] for i in ib)}, "
          f"fcn evals {sum(i['fcn_count'] for i in ib)}, failed {sum(1 for s in st if s)}", flush=True)
