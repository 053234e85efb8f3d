import sys, os, collections, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from nonlin_amd.device import DeviceSolver
ds = DeviceSolver(0)
nb, m, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A, b, xt, x0 = ds.generate(nb, m, n, seed0=12345)
x = x0.clone()
fv, ibs, st = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=500, sub_batches=1))
