#!/usr/bin/env python3
"""Solve time of exact-policy LM batches between "a handful" and "a chipful" (the straggler rounds of a big batch, every
per-rank share of a sharded batch).  One process per kernel configuration: the NLH_QRX_* knobs are read once per process.

  python profiles/sweep_mid.py 4096x256:1,8,16,32,47,64,128,256 2048x128:8,32,128,256,512,1024 [--check]

Prints one line per (shape, batch): mean of three warm solves in ms, LM iterations/s, and -- with --check -- whether x of
the first, middle and last problem equals the x of the same problem solved alone (bit for bit)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nonlin_amd.device import DeviceSolver
    check = "--check" in sys.argv
    sub = 1                                                        # --sub=N: sub-batches in flight (0 = the library's automatic choice)
    for a in sys.argv[1:]:
        if a.startswith("--sub="):
            sub = int(a.split("=")[1])
    specs = [a for a in sys.argv[1:] if not a.startswith("--")]
    ds = DeviceSolver(0)
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("NLH_QRX"))
    print(f"# {tag or 'defaults'} sub_batches={sub}", flush=True)
    for spec in specs:
        shape, _, counts = spec.partition(":")
        m, n = (int(v) for v in shape.split("x"))
        for nb in (int(c) for c in counts.split(",")):
            A, b, xt, x0 = ds.generate(nb, m, n, seed0=12345)
            opts = ds.options(max_evals=500, sub_batches=sub)
            x = x0.clone()
            ds.lm_solve_batch(A, b, 0.5, x, opts)
            torch.cuda.synchronize()
            t, nj = 0.0, 0
            fv = None
            for _ in range(3):
                x.copy_(x0)
                fv = None               # (round 6) the previous result goes before the call: with it alive the second call allocated a
                #                         fresh fvec tensor inside the timed region -- 35 ms for 1024 x 2048 doubles, once per size,
                #                         a third of which ended up in every mean of three printed here (rounds 4 and 5 included)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fv, ibs, st = ds.lm_solve_batch(A, b, 0.5, x, opts)
                torch.cuda.synchronize()
                t += time.perf_counter() - t0
                nj += sum(i["jacobian_count"] for i in ibs)
            line = f"{m}x{n} batch {nb:5d}: {1e3 * t / 3:9.2f} ms per solve  {nj / t:10.1f} LM it/s"
            if check:
                ok = True
                for p in sorted({0, nb // 2, nb - 1}):
                    x1 = x0[p:p + 1].clone()
                    ds.lm_solve_batch(A[p:p + 1], b[p:p + 1], 0.5, x1, opts)
                    ok = ok and bool(torch.equal(x1[0], x[p]))
                line += f"  same bits as alone: {ok}"
            print(line, flush=True)
            del A, b, xt, x0, x


if __name__ == "__main__":
    main()
