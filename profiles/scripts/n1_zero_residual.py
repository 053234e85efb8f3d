"""N1 probe: the opt-in normal-equations policy (NLH_FACTOR_AUTO: J^T J on the fp64 MFMA + Cholesky) on the ZERO-RESIDUAL
variant of SURVEY 8(d)'s family (sigma = 0) at BASELINE sizes, against the CPU oracle: max relative deviation of x, counts
and flags.  Usage: python profiles/scripts/n1_zero_residual.py [--c5 1]"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
KEYS = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")


def cpu_one(arg):
    seed, m, n, gamma, sigma, spread = arg
    from oracle import pyoracle as O
    A, b, xt, x0 = O.dq_generate(seed, m, n, gamma=gamma, sigma=sigma, spread=spread)
    rc, x, f, ib, nc, _ = O.dq_lm_solve(A, b, gamma, x0, opts=O.default_options(max_evals=500))
    return rc, x, f, ib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c5", type=int, default=1)
    ap.add_argument("--sigma", type=float, default=0.0)
    ap.add_argument("--live-oracle", type=int, default=0,
                    help="1 = solve every problem with the CPU oracle now (a process pool; minutes); 0 = compare with "
                         "tests/golden/zero_residual_oracle.npz, the oracle's stored outputs for these seeds (sigma = 0 only)")
    args = ap.parse_args()
    import torch
    from nonlin_amd.device import DeviceSolver
    ds = DeviceSolver(0)
    gamma, spread = 0.5, 0.3
    cases = [(4096, 256, 32, 12345), (2048, 128, 12, 12345)]
    if args.c5:
        cases.append((65536, 512, 1, 12345))
    out = []
    live = bool(args.live_oracle) or args.sigma != 0.0
    gold = None if live else np.load(os.path.join(ROOT, "tests", "golden", "zero_residual_oracle.npz"))
    tags = {(4096, 256): "c2", (2048, 128): "c4", (65536, 512): "c5"}
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1) if live else 1) as pool:
        for m, n, nprob, seed0 in cases:
            if live:
                fut = pool.map(cpu_one, [(seed0 + k, m, n, gamma, args.sigma, spread) for k in range(nprob)])
            A, b, xt, x0 = ds.generate(nprob, m, n, seed0=seed0, gamma=gamma, sigma=args.sigma, spread=spread)
            row = {"m": m, "n": n, "problems": nprob, "sigma": args.sigma, "oracle": "live" if live else "golden fixture"}
            if live:
                ref = list(fut)
            else:
                t = tags[(m, n)]
                ref = [(int(gold[f"{t}_status"][k]), gold[f"{t}_x"][k], None, dict(zip(KEYS, (int(v) for v in gold[f"{t}_counts"][k]))))
                       for k in range(nprob)]
            for pol in (0, 2):
                x = x0.clone()
                ds.lm_solve_batch(A, b, gamma, x, ds.options(max_evals=500, factor_policy=pol))
                x.copy_(x0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fvec, ibs, status = ds.lm_solve_batch(A, b, gamma, x, ds.options(max_evals=500, factor_policy=pol))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                xg = x.cpu().numpy()
                fg = fvec.cpu().numpy()
                dev = [float(np.abs(xg[k] - ref[k][1]).max() / np.abs(ref[k][1]).max()) for k in range(nprob)]
                fdev = [float(np.abs(fg[k] - ref[k][2]).max()) if ref[k][2] is not None else float(np.abs(fg[k]).max()) for k in range(nprob)]
                mism = [k for k in range(nprob) if any(ibs[k][q] != ref[k][3][q] for q in KEYS) or status[k] != ref[k][0]]
                row[f"policy{pol}"] = {"ms": 1e3 * dt, "max_rel_dev_x": max(dev), "max_abs_dev_f": max(fdev),
                                       "mismatch": mism, "counts0_gpu": {q: ibs[0][q] for q in KEYS},
                                       "counts0_cpu": {q: ref[0][3][q] for q in KEYS},
                                       "mism_detail": [({q: ibs[k][q] for q in KEYS}, {q: ref[k][3][q] for q in KEYS}) for k in mism[:4]]}
            out.append(row)
            print(json.dumps(row), flush=True)
            del A, b, xt, x0
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
