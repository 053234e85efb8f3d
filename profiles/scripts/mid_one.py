#!/usr/bin/env python3
"""One exact-policy LM batch in the mid regime, for traces: python profiles/scripts/mid_one.py NPROB M N SUB [solves]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nonlin_amd.device import DeviceSolver  # noqa: E402

nprob, m, n, sub = (int(v) for v in sys.argv[1:5])
solves = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ds = DeviceSolver(0)
A, b, xt, x0 = ds.generate(nprob, m, n, seed0=12345)
o = ds.options(max_evals=500, sub_batches=sub)
for r in range(solves):
    x = x0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ds.lm_solve_batch(A, b, 0.5, x, o)
    torch.cuda.synchronize()
    print(f"mid {nprob} x {m}x{n} sub={sub}: {(time.perf_counter() - t0) * 1e3:8.2f} ms", flush=True)
