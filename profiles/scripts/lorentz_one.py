#!/usr/bin/env python3
"""The Lorentzian user family alone (for kernel stats / traces): python profiles/scripts/lorentz_one.py [nprob m K sub]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from nonlin_amd.device import DeviceSolver  # noqa: E402
import user_models as UM  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
nprob, m, K, sub = (a + [2048, 4096, 32, 0][len(a):])[:4]
ds = DeviceSolver(0)
t, y, xt, x0 = UM.lorentz_problems(nprob, m, K)
batch = UM.LorentzBatch(t, y)
xd = torch.tensor(x0, device=ds.device)
o = ds.options(max_evals=500, sub_batches=sub)
t0 = time.perf_counter()
fv, ibs, st = ds.lm_solve_batch_device(batch.launch, batch.ctx, m, xd.clone(), opts=o)
torch.cuda.synchronize()
print(f"lorentz {nprob} x {m} x {3 * K}, sub_batches={sub}: {time.perf_counter() - t0:.3f} s, non-converged {sum(1 for v in st if v)}", flush=True)
batch.close()
