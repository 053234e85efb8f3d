#!/usr/bin/env python3
"""The headline batch (2048 x 4096x256, bench family) solved with the library's default options: one warm solve, then
`solves` timed ones.  For kernel traces (profiles/scripts/overlap_defaults.py).  python profiles/scripts/defaults_one.py [nprob m n solves sub]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nonlin_amd.device import DeviceSolver  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
nprob, m, n, solves, sub = (a + [2048, 4096, 256, 1, 0][len(a):])[:5]
ds = DeviceSolver(0)
A, b, xt, x0 = ds.generate(nprob, m, n, seed0=12345)
o = ds.options(max_evals=500, sub_batches=sub)
x = x0.clone()
ds.lm_solve_batch(A, b, 0.5, x, o)
torch.cuda.synchronize()
time.sleep(0.05)
for r in range(solves):
    x.copy_(x0)
    torch.cuda.synchronize()
    time.sleep(0.02)
    t0 = time.perf_counter()
    f, ibs, st = ds.lm_solve_batch(A, b, 0.5, x, o)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nj = sum(i["jacobian_count"] for i in ibs)
    print(f"defaults {nprob} x {m}x{n} sub={sub}: {dt * 1e3:8.2f} ms  {nj / dt:8.1f} LM it/s", flush=True)
