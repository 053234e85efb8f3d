#!/usr/bin/env python3
"""lmpar on synthetic factors with a binding trust region (the stage entry point nlh_lmpar, one problem): with a
-DNLH_DEBUG_LMPAR_CLK build of nlh_lm.hip the kernel prints the time of lmsolve's phases and the number of lmsolve calls.
python profiles/scripts/lmsolve_time.py [n ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nonlin_amd.device import DeviceSolver  # noqa: E402

ds = DeviceSolver(0)
f64 = dict(dtype=torch.float64, device="cuda")
for n in [int(a) for a in sys.argv[1:]] or [64, 96, 128, 192, 256]:
    rng = np.random.default_rng(1000 + n)
    R = np.triu(rng.standard_normal((n, n)))
    R[np.arange(n), np.arange(n)] += np.sign(R[np.arange(n), np.arange(n)]) * 2.0
    ip = rng.permutation(n).astype(np.int32)
    diag = np.abs(rng.standard_normal(n)) + 0.5
    qtf = rng.standard_normal(n)
    xgn = np.empty(n)
    xgn[ip] = np.linalg.solve(R, qtf)
    delta = 0.2 * np.linalg.norm(diag * xgn)
    for rep in range(2):
        Rd = torch.tensor(np.ascontiguousarray(R.T), device="cuda").unsqueeze(0)
        par, x, sdiag = ds.lmpar(Rd, torch.tensor(ip, dtype=torch.int32, device="cuda").unsqueeze(0), torch.tensor(diag, **f64).unsqueeze(0),
                                 torch.tensor(qtf, **f64).unsqueeze(0), torch.tensor([float(delta)], **f64), torch.tensor([0.0], **f64),
                                 torch.tensor([0.0], **f64))
        torch.cuda.synchronize()
    print(f"n = {n}: par = {float(par[0]):.6g}", flush=True)
