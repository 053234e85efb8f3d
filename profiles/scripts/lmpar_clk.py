#!/usr/bin/env python3
"""One 4096x256 problem whose trust region collapses (seed 12391 = problem 46 of the bench family: two slow lmpars), solved
alone: with a -DNLH_DEBUG_LMPAR_CLK build of nlh_lm.hip the kernel prints lmpar's phases.  python profiles/scripts/lmpar_clk.py [seed m n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nonlin_amd.device import DeviceSolver  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12391
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ds = DeviceSolver(0)
A, b, xt, x0 = ds.generate(1, m, n, seed0=seed)
o = ds.options(max_evals=500)
for r in range(2):
    x = x0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f, ibs, st = ds.lm_solve_batch(A, b, 0.5, x, o)
    torch.cuda.synchronize()
    print(f"seed {seed} {m}x{n}: {(time.perf_counter() - t0) * 1e3:8.2f} ms  {ibs[0]}", flush=True)
