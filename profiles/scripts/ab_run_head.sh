#!/bin/bash
# same box, alternating: the round-4 library against the current one on the headline (old lib lacks the new symbols: use the old ctypes table via env)
for rep in 1 2; do for tag in r04 r05; do
  cp scratch/ab/lib_$tag.so nonlin_amd/libnonlin_hip.so
  for sb in 1 0; do
  python - <<PY
import sys, time, torch
sys.path.insert(0, ".")
from nonlin_amd import _lib
import ctypes as C
# bind only what this script needs (the round-4 library has no launcher entry points)
keep = {k: v for k, v in _lib.SYMBOLS.items() if k in ("nlh_default_options","nlh_create","nlh_destroy","nlh_device_count","nlh_last_error","nlh_dq_generate","nlh_dq_lm_solve_batch","nlh_timing_enable","nlh_timing_reset","nlh_timing_get","nlh_timing_samples","nlh_kernel_name")}
_lib.SYMBOLS.clear(); _lib.SYMBOLS.update(keep)
from nonlin_amd.device import DeviceSolver
ds = DeviceSolver(0)
A, b, xt, x0 = ds.generate(2048, 4096, 256, seed0=12345)
o = ds.options(max_evals=500, sub_batches=$sb)
x = x0.clone(); ds.lm_solve_batch(A, b, 0.5, x, o); torch.cuda.synchronize()
ts = []
for _ in range(3):
    x.copy_(x0); torch.cuda.synchronize(); t0 = time.perf_counter()
    f, ibs, st = ds.lm_solve_batch(A, b, 0.5, x, o); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
nj = sum(i["jacobian_count"] for i in ibs)
print("$tag sub_batches=$sb: %.1f ms  %.0f it/s" % (1e3 * min(ts), nj / min(ts)), flush=True)
PY
  done
done; done
cp scratch/ab/lib_r05.so nonlin_amd/libnonlin_hip.so
