"""Random sweep of the exact factorisation on LONG columns (several NORM2 chunks: the chain-free running sum of
ordered_possum_wave_int with its fall-backs) against the oracle, bit for bit.  Not collected by pytest; run on a GPU box:
python tests/soak_long_columns.py [SEED] [NCASES]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nonlin_amd.device import DeviceSolver
from oracle import pyoracle as O
ds = DeviceSolver(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
bad = 0
for case in range(ncase):
    m = int(rng.integers(6200, 90000)); n = int(rng.integers(2, 6)); kind = case % 8
    a = rng.standard_normal((m, n))
    if kind == 1: a = np.round(a * 2.0 ** int(rng.integers(2, 9))) / 2.0 ** int(rng.integers(2, 9))     # ties
    elif kind == 2: a = np.sign(a)
    elif kind == 3: a = a * (1.0 + 1e5 * (rng.random((m, n)) < 1e-4))                                   # late maxima
    elif kind == 4: a[rng.random((m, n)) < 0.8] = 0.0
    elif kind == 5: a = a * 10.0 ** rng.uniform(-150, 150)                                               # extreme scale
    elif kind == 6: a = a * (1.0 + 2e-4) ** (-np.arange(m))[:, None]                                     # decaying: tiny ratios late
    elif kind == 7: a = np.abs(a) ** 4
    a = np.asfortranarray(a)
    J = torch.tensor(np.ascontiguousarray(a.T)[None], device="cuda:0")
    F = torch.tensor(rng.standard_normal(m)[None], device="cuda:0")
    R, ipvt, rdiag, acnorm, qtf, wa4 = ds.lmfactor_exact(J, F)
    ao, ip, rd, acn = O.lmfactor(a)
    ok = np.array_equal(ipvt[0].cpu().numpy(), ip) and np.array_equal(rdiag[0].cpu().numpy(), rd) and np.array_equal(acnorm[0].cpu().numpy(), acn)
    if not ok:
        bad += 1; print("MISMATCH case", case, "m", m, "n", n, "kind", kind, flush=True)
print(f"long-column soak: {ncase} cases, {bad} mismatches", flush=True)
