"""Golden outputs of the CPU oracle (oracle/nonlin_oracle.c, the restatement of lss_solve) on the ZERO-RESIDUAL variant of
SURVEY 8(d)'s family (sigma = 0) at the BASELINE sizes: x, fvec norm, counts and flags per problem.  The N1 tests
(tests/test_gpu_auto_policy.py) hold the MFMA / Cholesky policy to them at north_star's 1e-10 without paying the oracle's
minute per 65536 x 512 solve on the GPU box; tests/test_oracle.py re-derives a sample live so that the file cannot go stale.

    python tests/golden/make_zero_residual_oracle.py        # ~2 minutes on 8 cores -> tests/golden/zero_residual_oracle.npz
"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
KEYS = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")
GAMMA, SIGMA, SPREAD, SEED0, MAX_EVALS = 0.5, 0.0, 0.3, 12345, 500
CASES = (("c2", 4096, 256, 32), ("c4", 2048, 128, 12), ("c5", 65536, 512, 1))


def solve(arg):
    seed, m, n = arg
    from oracle import pyoracle as O
    A, b, xt, x0 = O.dq_generate(seed, m, n, gamma=GAMMA, sigma=SIGMA, spread=SPREAD)
    rc, x, f, ib, _, _ = O.dq_lm_solve(A, b, GAMMA, x0, opts=O.default_options(max_evals=MAX_EVALS))
    return rc, x, float(np.sqrt(np.sum(f * f))), [int(ib[k]) for k in KEYS]


def main():
    out = {"keys": np.array(KEYS), "params": np.array([GAMMA, SIGMA, SPREAD, SEED0, MAX_EVALS])}
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        for tag, m, n, nprob in CASES:
            res = list(pool.map(solve, [(SEED0 + k, m, n) for k in range(nprob)]))
            out[f"{tag}_shape"] = np.array([m, n, nprob])
            out[f"{tag}_status"] = np.array([r[0] for r in res], dtype=np.int32)
            out[f"{tag}_x"] = np.stack([r[1] for r in res])
            out[f"{tag}_fnorm"] = np.array([r[2] for r in res])
            out[f"{tag}_counts"] = np.array([r[3] for r in res], dtype=np.int32)
            print(tag, m, n, nprob, out[f"{tag}_counts"][0], out[f"{tag}_fnorm"].max(), flush=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "zero_residual_oracle.npz"), **out)


if __name__ == "__main__":
    main()
