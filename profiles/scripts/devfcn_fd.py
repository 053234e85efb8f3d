import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from nonlin_amd.device import DeviceSolver
ds = DeviceSolver(0)
def fd_bytes(m, n): return 8 * (2 * m * n + m + 2 * n)
nprob, m, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A, b, xt, x0 = ds.generate(nprob, m, n, seed0=12345)
fcn, jac, ctx = ds.dq_launchers(A, b, 0.5)
o = ds.options(max_evals=500, sub_batches=1)
f = lambda: ds.lm_solve_batch_device(fcn, ctx, m, x0.clone(), opts=o)
f(); torch.cuda.synchronize()
ds.h.timing_enable(kernels=["fd_jacobian", "dq_panel"])
ds.h.timing_reset()
t0 = time.perf_counter(); fv, ibs, st = f(); torch.cuda.synchronize(); t = time.perf_counter() - t0
nj = sum(i["jacobian_count"] for i in ibs)
ms, cnt = ds.h.timing("fd_jacobian"); pm, pc = ds.h.timing("dq_panel")
gb = fd_bytes(m, n) * nj / (ms * 1e-3) / 1e9
print(f"chunk={os.environ.get('NLH_FD_CHUNK_MB','64')} nt={os.environ.get('NLH_FDQ_NT','1')}: {nprob}x{m}x{n} solve {t*1e3:.1f} ms {nj/t:.0f} it/s; fd {ms:.2f} ms / {cnt} launches -> {gb:.0f} GB/s ({gb/8000:.3f}); user fcn {pm:.1f} ms / {pc}", flush=True)
