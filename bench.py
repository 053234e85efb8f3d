#!/usr/bin/env python3
"""bench.py -- LM iterations/s on batched 4096 x 256 fp64 LM problems with the finite-difference
Jacobian (BASELINE.json metric; workload = configs[1] instantiated per SURVEY.md 8(d), batched
as north_star's "batched 4096x256 LM problems at 1 GPU").

A "step" is one pass of the hot path over one batch: every problem of the batch is solved from
its start point by least_squares_solver (nlh_dq_lm_solve_batch): FD Jacobian (n perturbed
evaluations + column write) -> J^T J / J^T f (fp64 MFMA) -> pivoted Cholesky -> lmpar -> trial
evaluation -> trust-region update, until every problem has converged.  Inputs are generated on
the device before the timed region and stay resident in HBM.

value = sum of jacobian_count (= LM outer iterations) over all problems, steps and ranks,
divided by the max-over-ranks wall time of the K timed steps.  N > 1: one process per GPU
(torchrun), independent problems sharded block-cyclically, weak scaling (fixed batch per GPU),
no data-path collective; RCCL is used for the config broadcast and the result gather.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M, N_VAR = 4096, 256
GAMMA, SIGMA, SPREAD = 0.5, 1e-3, 0.3
SEED0 = 12345
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def fd_bytes(m, n):
    # read one m x n panel + f0 (m) + x,h (2n), write J (m x n)
    return 8.0 * (2.0 * m * n + m + 2.0 * n)


def _cpu_solve_one(arg):
    k, m, n = arg
    from oracle import pyoracle as O
    A, b, xt, x0 = O.dq_generate(SEED0 + k, m, n, gamma=GAMMA, sigma=SIGMA, spread=SPREAD)
    t0 = time.perf_counter()
    rc, x, f, ib, nc, _ = O.dq_lm_solve(A, b, GAMMA, x0, opts=O.default_options(max_evals=500))
    return ib["jacobian_count"], time.perf_counter() - t0


def cpu_baseline(sample, m, n):
    """Oracle (C restatement of the reference path) on the host: one core (the reference is single-threaded),
    then the same problems farmed over every host core (the CPU analogue of sharding).  Must run before the
    GPU is initialised: the all-cores leg forks workers."""
    njac = 0
    t = 0.0
    for k in range(sample):
        nj, dt = _cpu_solve_one((k, m, n))
        njac += nj
        t += dt
    out = {"value": njac / t, "unit": "LM iterations/s", "cores": 1, "kind": "port",
           "sample": f"{sample} problems {m}x{n} (seeds {SEED0}..{SEED0 + sample - 1}), single thread, "
                     f"oracle/nonlin_oracle.c -O2 -ffp-contract=off, {t:.1f} s"}
    try:
        import multiprocessing as mp
        cores = len(os.sched_getaffinity(0))
        nprob = max(cores, sample)
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_cpu_solve_one, [(k, m, n) for k in range(nprob)], chunksize=1)
        wall = time.perf_counter() - t0
        out["all_cores"] = {"value": sum(r[0] for r in res) / wall, "unit": "LM iterations/s", "cores": cores,
                            "sample": f"{nprob} problems over {cores} worker processes, {wall:.1f} s wall "
                                      f"(includes problem generation)"}
    except Exception as e:                                   # the single-core figure is the contract
        out["all_cores"] = {"error": repr(e)}
    return out



def other_paths(ds):
    """The remaining rows of SURVEY section 8 (a16-a24 Newton, f1 quasi-Newton, f2 bounded least squares, f3 BFGS,
    f4 polynomial fit), one problem each (4096 fits for the polynomial), second (warm) run timed on the GPU;
    the CPU oracle runs the same problem on one host core and the solutions are compared bit for bit."""
    import numpy as np
    import torch
    from oracle import pyoracle as O

    def timed(f):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    def cpu(f):
        t0 = time.perf_counter()
        r = f()
        return r, time.perf_counter() - t0

    rows = []
    n = 1024
    xg = [None]
    for name, gsolve, csolve, gen in (
            ("newton_solver (LU), analytic Jacobian, n=1024", ds.newton_solve_batch, O.dq_newton_solve, {}),
            ("quasi_newton_solver (Broyden, QR + rank-1 update), analytic Jacobian, n=1024", ds.quasi_newton_solve_batch,
             O.dq_quasi_newton_solve, dict(spread=0.03))):
        A, b, xt, x0 = ds.generate(1, n, n, seed0=12345, sigma=0.0, square_shift=True, **gen)
        Ah, bh, xh = np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy(), x0[0].cpu().numpy()

        def run_square():
            xg[0] = x0.clone()
            return gsolve(A, b, 0.5, xg[0], analytic=True, opts=ds.options(max_evals=500))
        (_, ibs, st), tg = timed(run_square)
        ro, tc = cpu(lambda: csolve(Ah, bh, 0.5, xh, opts=O.default_options(max_evals=500)))
        rows.append({"path": name, "gpu_ms": 1e3 * tg, "cpu_oracle_ms": 1e3 * tc, "iterations": ibs[0]["iter_count"],
                     "bitwise_equal": bool(np.array_equal(ro[1], xg[0][0].cpu().numpy()))})
    m, n = 4096, 256
    A, b, xt, x0 = ds.generate(1, m, n, seed0=12345)
    Ah, bh, xh = np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy(), x0[0].cpu().numpy()
    lo, up = np.full(n, -2.0), np.full(n, 2.0)

    def run_cls():
        xg[0] = x0.clone()
        return ds.cls_solve_batch(A, b, 0.5, xg[0], opts=ds.options(max_evals=500), lower=lo, upper=up)
    (_, ibs, st), tg = timed(run_cls)
    ro, tc = cpu(lambda: O.dq_cls_solve(Ah, bh, 0.5, xh, opts=O.default_options(max_evals=500), lower=lo, upper=up))
    rows.append({"path": "constrained_least_squares_solver (bounded dog-leg), FD Jacobian, 4096x256", "gpu_ms": 1e3 * tg,
                 "cpu_oracle_ms": 1e3 * tc, "iterations": ibs[0]["iter_count"],
                 "bitwise_equal": bool(np.array_equal(ro[1], xg[0][0].cpu().numpy()))})
    m, n = 2048, 256
    A, b, xt, x0 = ds.generate(1, m, n, seed0=77, spread=0.1)
    Ah, bh, xh = np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy(), x0[0].cpu().numpy()
    ob = dict(max_evals=400, gtol=1e-8, xtol=1e-12)

    def run_bfgs():
        xg[0] = x0.clone()
        return ds.bfgs_solve_batch(A, b, 0.5, xg[0], opts=ds.options(**ob))
    (_, ibs, st), tg = timed(run_bfgs)
    ro, tc = cpu(lambda: O.dq_bfgs_solve(Ah, bh, 0.5, xh, opts=O.default_options(**ob)))
    rows.append({"path": "bfgs (FD gradient, Cholesky rank-1 updates), 2048x256", "gpu_ms": 1e3 * tg, "cpu_oracle_ms": 1e3 * tc,
                 "iterations": ibs[0]["iter_count"], "bitwise_equal": bool(np.array_equal(ro[1], xg[0][0].cpu().numpy()))})
    nfit, npts, order = 4096, 4096, 7
    g = torch.Generator(device=ds.device).manual_seed(1)
    px = torch.rand((nfit, npts), dtype=torch.float64, device=ds.device, generator=g) * 2 - 1
    py = torch.cos(3 * px) + 0.01 * torch.randn((nfit, npts), dtype=torch.float64, device=ds.device, generator=g)
    c, tg = timed(lambda: ds.poly_fit_batch(px, py, order))
    xs, ys = px[:16].cpu().numpy(), py[:16].cpu().numpy()
    co, tc = cpu(lambda: [O.poly_fit(xs[q], ys[q], order)[1] for q in range(16)])
    rows.append({"path": "polynomial%fit, 4096 fits x 4096 points, order 7", "gpu_ms": 1e3 * tg,
                 "cpu_oracle_ms": 1e3 * tc * nfit / 16, "cpu_sample": "16 fits, scaled to 4096",
                 "bitwise_equal": bool(all(np.array_equal(c[q].cpu().numpy(), co[q]) for q in range(16)))})
    return rows

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("NLH_BENCH_BATCH", "512")),
                    help="problems per GPU per step")
    ap.add_argument("--m", type=int, default=M)
    ap.add_argument("--n", type=int, default=N_VAR)
    ap.add_argument("--cpu-sample", type=int, default=24, help="problems timed on the CPU oracle (0 = skip)")
    ap.add_argument("--exact-sample", type=int, default=256, help="problems for the exact-policy figure (0 = skip)")
    ap.add_argument("--extras", type=int, default=1,
                    help="0 = skip the single_problem and fused_fd legs (profiles/capture.sh: every launch rocprofv3 sees "
                         "then belongs to a warm-up or timed step, so its per-kernel averages are the ones printed here)")
    ap.add_argument("--other-paths", type=int, default=1,
                    help="1 = also time the Newton / quasi-Newton / bounded LSQ / BFGS / polynomial rows (SURVEY 8 a16-a24, f1-f4) "
                         "on the GPU and on the CPU oracle and check the results bit for bit")
    ap.add_argument("--policy", type=int, default=0, help="0 auto (J^T J + Cholesky), 1 QR, 2 exact (reference order)")
    ap.add_argument("--fuse-fd", type=int, default=0,
                    help="1: form the Jacobian column in the panel kernel's epilogue (k_fd_jacobian is then not launched)")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = cpu_baseline(args.cpu_sample, args.m, args.n) if (world_env == 1 and args.cpu_sample > 0) else None

    import torch
    import torch.distributed as dist
    from nonlin_amd.device import DeviceSolver
    from nonlin_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    m, n, B = args.m, args.n, args.batch

    ds = DeviceSolver(local_rank)
    # options / seed come from rank 0 (broadcast over RCCL when N > 1)
    cfg = sharding.broadcast_config([500, SEED0, GAMMA, SIGMA, SPREAD], dev)
    max_evals, seed0 = int(cfg[0]), int(cfg[1])
    gamma, sigma, spread = cfg[2], cfg[3], cfg[4]
    nprob_total = B * world
    # block-cyclic: local problem i is global problem rank + i*world, seed = seed0 + global index
    A, b, xt, x0 = ds.generate(B, m, n, seed0=seed0 + rank, gamma=gamma, sigma=sigma, spread=spread,
                               seed_stride=world)
    opts = ds.options(max_evals=max_evals, factor_policy=args.policy, fuse_fd=args.fuse_fd)
    x = x0.clone()

    def step():
        x.copy_(x0)
        fvec, ibs, status = ds.lm_solve_batch(A, b, gamma, x, opts)
        return ibs, status

    for _ in range(args.warmup):
        step()
    # Timed region: only the roofline kernel is bracketed by HIP events (two event records per launch on the launch
    # stream cost ~3 % of a step when every one of the ~100 launches is bracketed); the per-kernel breakdown comes from
    # a second, untimed pass over the same steps below.
    ds.h.timing_enable(kernels=["fd_jacobian"])
    ds.h.timing_reset()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    gc.collect()
    gc.disable()        # a generation-2 collection of the harness (one dict per problem and step) costs tens of ms
    sync()
    t0 = time.perf_counter()
    njac = 0
    naccept = 0
    bad = 0
    last_ibs = None
    for _ in range(args.steps):
        ibs, status = step()
        njac += sum(ib["jacobian_count"] for ib in ibs)
        naccept += sum(ib["iter_count"] - 1 for ib in ibs)
        bad += sum(1 for s in status if s != 0)
        last_ibs = ibs
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()

    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    cnt = torch.tensor([float(njac), float(naccept), float(bad)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(tt[0])
    njac_all, naccept_all, bad_all = (float(v) for v in cnt.tolist())

    # gather per-problem results (iteration counts + an x checksum) on every rank: the "gather" end
    rows = torch.tensor([[ib["iter_count"], ib["fcn_count"], ib["jacobian_count"]] for ib in last_ibs],
                        dtype=torch.float64, device=dev)
    rows = torch.cat([rows, x.sum(dim=1, keepdim=True)], dim=1)
    allrows = sharding.gather_results(rows, nprob_total, rank, world)

    fd_ms, fd_launches = ds.h.timing("fd_jacobian")
    ds.h.timing_enable(True)                       # breakdown pass: same steps, every kernel group timed, not part of `value`
    ds.h.timing_reset()
    for _ in range(args.steps):
        step()
    kernel_ms = {k: ds.h.timing(k)[0] for k in
                 ("dq_residual", "dq_panel", "fd_jacobian", "gram", "gram_reduce", "jtf", "chol", "lmpar", "qr", "update")}
    ds.h.timing_enable(False)

    if rank == 0:
        achieved = fd_bytes(m, n) * (njac / max(fd_ms * 1e-3, 1e-30)) / 1e9      # this rank's launches
        # HBM traffic of the same kernel from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, gfx950
        # corrections applied), committed under profiles/; scaled to the average launch of this run.
        traffic = None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
            if (prof["m"], prof["n"]) == (m, n):
                per_problem = prof["kernels"]["k_fd_jacobian<256, 8, true>"]["hbm_bytes_per_problem"]
                traffic = per_problem * njac / max(fd_launches, 1)
        except Exception:
            traffic = None
        out = {
            "metric": "LM iterations/sec on m=4096,n=256 fp64; FD-Jacobian GB/s vs HBM peak",
            "value": njac_all / elapsed,
            "unit": "LM iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"batched LM {m}x{n} fp64, FD Jacobian, dense-quadratic residual family "
                            f"(SURVEY 8(d): gamma={gamma}, sigma={sigma}, spread={spread}, seeds {seed0}+k), "
                            f"{B} problems per GPU per step",
                "problems_per_gpu": B, "m": m, "n": n, "max_fcn_evals": max_evals,
                "factor_policy": {0: "auto (J^T J MFMA + pivoted Cholesky, QR fallback)", 1: "householder-qr",
                                  2: "exact (reference operation order)"}[args.policy],
                "parallelism": f"independent problems, block-cyclic over {world} rank(s)",
                "accepted_steps_per_s": naccept_all / elapsed,
                "non_converged": int(bad_all),
                "iters_first_problem": [int(v) for v in allrows[0, :3].tolist()],
            },
            "roofline": {
                "kernel": "k_fd_jacobian", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), bytes per average launch",
                "algorithmic_bytes_per_launch": fd_bytes(m, n) * njac / max(fd_launches, 1),
                "bytes_per_unit": fd_bytes(m, n), "units": "problem-Jacobians", "launches": int(fd_launches),
                "avg_launch_ms": fd_ms / max(fd_launches, 1),
            },
            "kernel_ms_per_step": {k: v / args.steps for k, v in kernel_ms.items()},
        }
        # the other kernels of an outer iteration against the bound that applies to each
        nfev = sum(ib["fcn_count"] for ib in last_ibs) * args.steps
        gram_flops = (m * n * (n + 1) + 2 * m * n) * njac             # symmetric half + J^T f
        panel_adds = m * (n * (n + 1) // 2) * njac                    # dependent adds of n perturbed row sums
        resid_bytes = 8 * (m * n + 2 * m + n) * nfev
        gms, pms, rms = kernel_ms["gram"], kernel_ms["dq_panel"], kernel_ms["dq_residual"]
        out["kernel_rooflines"] = [
            {"kernel": "k_gram_tri / k_gram_mfma", "bound": "mfma", "achieved": gram_flops / max(gms * 1e-3, 1e-30) / 1e12,
             "peak": 78.6, "unit": "TFLOP/s", "frac": gram_flops / max(gms * 1e-3, 1e-30) / 1e12 / 78.6},
            {"kernel": "k_dq_panel", "bound": "valu-f64-add", "achieved": panel_adds / max(pms * 1e-3, 1e-30) / 1e12,
             "peak": 39.3, "unit": "Tadd/s", "frac": panel_adds / max(pms * 1e-3, 1e-30) / 1e12 / 39.3},
            {"kernel": "k_dq_residual", "bound": "hbm", "achieved": resid_bytes / max(rms * 1e-3, 1e-30) / 1e9,
             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": resid_bytes / max(rms * 1e-3, 1e-30) / 1e9 / HBM_PEAK_GBS},
        ]
        if world == 1 and args.extras:
            # latency of BASELINE config 2 taken literally: ONE 4096 x 256 problem (seed 12345), warm handle
            x1 = x0[:1].clone()
            ds.lm_solve_batch(A[:1], b[:1], gamma, x1, opts)
            x1.copy_(x0[:1])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, ib1, _ = ds.lm_solve_batch(A[:1], b[:1], gamma, x1, opts)
            torch.cuda.synchronize()
            t1 = time.perf_counter() - t1
            out["single_problem"] = {"ms": 1e3 * t1, "lm_iterations": ib1[0]["jacobian_count"],
                                     "lm_iterations_per_s": ib1[0]["jacobian_count"] / t1}
        if world == 1 and args.extras and not args.fuse_fd:
            # same batch with the FD column write fused into the panel kernel (bit-identical results, one kernel
            # and one 8mn-byte round trip less per Jacobian); the headline keeps the stand-alone FD kernel
            of = ds.options(max_evals=max_evals, factor_policy=args.policy, fuse_fd=1)
            xf = x0.clone()
            ds.lm_solve_batch(A, b, gamma, xf, of)
            tf, nj = 0.0, 0
            for _ in range(args.steps):
                xf.copy_(x0)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _, ibf, _ = ds.lm_solve_batch(A, b, gamma, xf, of)
                torch.cuda.synchronize()
                tf += time.perf_counter() - t1
                nj += sum(i["jacobian_count"] for i in ibf)
            out["fused_fd"] = {"value": nj / tf, "unit": "LM iterations/s", "identical_x": bool(torch.equal(xf, x)),
                               "note": "opts.fuse_fd = 1: Jacobian column formed in the panel kernel's epilogue"}
        if world == 1 and args.extras:
            # several batches in flight: one handle (workspace + HIP stream) per host thread, so that the
            # latency-bound stages of one batch (Cholesky, lmpar, straggler rounds, the per-round status read-back)
            # overlap the streaming kernels of another.  Same inputs, same results; per-kernel times are not
            # meaningful in this mode, which is why the headline and its roofline are measured one batch at a time.
            import threading
            nfl, ksteps = 4, 12
            streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
            solvers = []
            for st_ in streams:
                with torch.cuda.stream(st_):
                    solvers.append(DeviceSolver(local_rank))
            for label, of in (("pipelined", opts), ("pipelined_fused_fd", ds.options(max_evals=max_evals, factor_policy=args.policy,
                                                                                  fuse_fd=1))):
                xs_ = [x0.clone() for _ in range(nfl)]
                counts = [0] * nfl

                def work(i, nsteps):
                    with torch.cuda.stream(streams[i]):
                        for _ in range(nsteps):
                            xs_[i].copy_(x0)
                            _, ibw, _ = solvers[i].lm_solve_batch(A, b, gamma, xs_[i], of)
                            counts[i] += sum(q["jacobian_count"] for q in ibw)
                for timed_pass in (False, True):
                    for i in range(nfl):
                        counts[i] = 0
                    torch.cuda.synchronize()
                    tp0 = time.perf_counter()
                    th = [threading.Thread(target=work, args=(i, ksteps // nfl if timed_pass else 1)) for i in range(nfl)]
                    [t_.start() for t_ in th]
                    [t_.join() for t_ in th]
                    torch.cuda.synchronize()
                    tp = time.perf_counter() - tp0
                out[label] = {"value": sum(counts) / tp, "unit": "LM iterations/s", "batches_in_flight": nfl, "steps": ksteps,
                              "ms_per_step": 1e3 * tp / ksteps, "identical_x": bool(all(torch.equal(xq, x) for xq in xs_))}
            del solvers
        if world == 1 and args.policy == 0 and args.exact_sample > 0:
            # the same workload under the exact factor policy (reference operation order: x, fvec and all
            # counts bit-identical to the CPU path, tests/test_gpu_solvers.py), one untimed + one timed pass
            Be = min(B, args.exact_sample)
            oe = ds.options(max_evals=max_evals, factor_policy=2)
            xe = x0[:Be].clone()
            ds.lm_solve_batch(A[:Be], b[:Be], gamma, xe, oe)
            xe.copy_(x0[:Be])
            torch.cuda.synchronize()
            te = time.perf_counter()
            _, ibe, _ = ds.lm_solve_batch(A[:Be], b[:Be], gamma, xe, oe)
            torch.cuda.synchronize()
            te = time.perf_counter() - te
            out["exact_policy"] = {"value": sum(i["jacobian_count"] for i in ibe) / te, "unit": "LM iterations/s",
                                   "problems": Be, "note": "NLH_FACTOR_EXACT: bit-identical to the CPU path"}
        if world == 1 and args.other_paths:
            out["other_paths"] = other_paths(ds)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
