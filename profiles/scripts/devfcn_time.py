import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from nonlin_amd.device import DeviceSolver
import user_models as UM
ds = DeviceSolver(0)

def fd_bytes(m, n): return 8 * (2 * m * n + m + 2 * n)

def run(name, f, nprob, m, n):
    f(); torch.cuda.synchronize()
    ds.h.timing_enable(kernels=["fd_jacobian", "dq_panel", "dq_residual", "qrx_pass", "qrx_pivot"])
    ds.h.timing_reset()
    t0 = time.perf_counter(); fv, ibs, st = f(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    nj = sum(i["jacobian_count"] for i in ibs)
    out = {k: ds.h.timing(k) for k in ("fd_jacobian", "dq_panel", "dq_residual", "qrx_pass", "qrx_pivot")}
    ds.h.timing_enable(False)
    ms, cnt = out["fd_jacobian"]
    gb = fd_bytes(m, n) * nj / (ms * 1e-3) / 1e9 if ms > 0 else 0
    print(f"{name}: {nprob}x{m}x{n} solve {t*1e3:.1f} ms, {nj/t:.0f} LM it/s, fd {ms:.2f} ms/{cnt} launches -> {gb:.0f} GB/s ({gb/8000:.3f}); "
          f"panel(user fcn) {out['dq_panel'][0]:.1f} ms, resid {out['dq_residual'][0]:.1f} ms, pass {out['qrx_pass'][0]:.1f}, pivot {out['qrx_pivot'][0]:.1f}", flush=True)

for nprob, m, n in ((256, 4096, 256), (1024, 2048, 128)):
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=12345)
    fcn, jac, ctx = ds.dq_launchers(A, b, 0.5)
    o = ds.options(max_evals=500)
    run("dq builtin", lambda: ds.lm_solve_batch(A, b, 0.5, x0.clone(), o), nprob, m, n)
    run("dq launcher", lambda: ds.lm_solve_batch_device(fcn, ctx, m, x0.clone(), opts=o), nprob, m, n)
    del A, b, xt, x0
    torch.cuda.empty_cache()
for nprob, m, K in ((2048, 4096, 32), (4096, 2048, 8)):
    t, y, xt, x0 = UM.lorentz_problems(nprob, m, K)
    batch = UM.LorentzBatch(t, y)
    xd = torch.tensor(x0, device=ds.device)
    o = ds.options(max_evals=500)
    run("lorentz", lambda: ds.lm_solve_batch_device(batch.launch, batch.ctx, m, xd.clone(), opts=o), nprob, m, 3 * K)
    batch.close()
