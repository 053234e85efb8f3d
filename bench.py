#!/usr/bin/env python3
"""bench.py -- LM iterations/s on batched m x n fp64 LM problems with the finite-difference Jacobian
(BASELINE.json metric; default workload = configs[1] instantiated per SURVEY.md 8(d), batched as
north_star's "batched 4096x256 LM problems at 1 GPU").

A "step" is one pass of the hot path over one batch: every problem of the batch is solved from its start
point by least_squares_solver (nlh_dq_lm_solve_batch) under the factor policy that carries parity with the
reference -- NLH_FACTOR_EXACT: FD Jacobian (n perturbed evaluations, column formed in the same kernel) ->
lmfactor + Q^T f in the reference's operation order (nlh_qrx.hip) -> lmpar -> trial evaluation ->
trust-region update, until every problem has converged.  x, fvec and all counts are bit-identical to the CPU
path (checked against the oracle on a sample of the same problems, `parity` in the JSON line).  Inputs are
generated on the device before the timed region and stay resident in HBM.

value = sum of jacobian_count (= LM outer iterations) over all problems, steps and ranks, divided by the
max-over-ranks wall time of the K timed steps.

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run as a
child process, before anything touches a GPU); under an external torchrun it reads RANK / WORLD_SIZE from the
environment.  Independent problems are dealt block-cyclically (problem k -> rank k mod N), no data-path collective;
RCCL carries the config broadcast and the result gather.
  --scaling weak   (default): --batch problems per GPU
  --scaling strong          : --total-problems problems in all (north_star: 8192 x 2048x128, 1024 x 2048x128)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M, N_VAR = 4096, 256
GAMMA, SIGMA, SPREAD = 0.5, 1e-3, 0.3
SEED0 = 12345
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s achievable)
POLICY_NAMES = {0: "auto (J^T J MFMA + Cholesky, QR fallback)", 1: "householder-qr (tree reductions)",
                2: "exact (lmfactor / lmpar in the reference's operation order)"}


def fd_bytes(m, n):
    # read one m x n panel + f0 (m) + x,h (2n), write J (m x n)
    return 8.0 * (2.0 * m * n + m + 2.0 * n)


def qr_pass_bytes(m, n):
    """Algorithmic bytes of the trailing passes of ONE exact factorisation: every element of the
    trailing matrix (columns j+1..n plus the residual column, rows j..m-1) is read once per Householder step."""
    return 8.0 * sum((m - j) * (n - j) for j in range(n))


def _cpu_solve_one(arg):
    k, m, n = arg
    from oracle import pyoracle as O
    A, b, xt, x0 = O.dq_generate(SEED0 + k, m, n, gamma=GAMMA, sigma=SIGMA, spread=SPREAD)
    t0 = time.perf_counter()
    rc, x, f, ib, nc, _ = O.dq_lm_solve(A, b, GAMMA, x0, opts=O.default_options(max_evals=500))
    return ib["jacobian_count"], time.perf_counter() - t0, x, ib, rc


def _physical_cores():
    """One logical CPU per physical core of this process's affinity set, and the CPU model name (/proc/cpuinfo)."""
    allowed = sorted(os.sched_getaffinity(0))
    model, seen, pick = "unknown", set(), []
    try:
        cur = {}
        blocks = []
        for line in open("/proc/cpuinfo"):
            if not line.strip():
                if cur:
                    blocks.append(cur)
                cur = {}
                continue
            k, _, v = line.partition(":")
            cur[k.strip()] = v.strip()
        if cur:
            blocks.append(cur)
        for bl in blocks:
            cpu = int(bl.get("processor", -1))
            model = bl.get("model name", model)
            key = (bl.get("physical id", "0"), bl.get("core id", str(cpu)))
            if cpu in allowed and key not in seen:
                seen.add(key)
                pick.append(cpu)
    except OSError:
        pass
    return (pick or allowed), model


def _cpu_quota():
    """CPUs this container may actually use at once (cgroup CPU quota), or None when unlimited / unknown: a box can show
    256 logical CPUs and schedule 16 of them."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]              # cgroup v2
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())           # cgroup v1
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def _cpu_worker(cpu, ks, m, n, ready, go, out):
    """Long-lived worker of the all-cores leg: pinned to one physical core, generates its problems BEFORE the clock
    starts, waits for the common start signal, solves them back to back."""
    try:
        os.sched_setaffinity(0, {cpu})
    except OSError:
        pass
    released = [False]                                             # the body releases `ready` once itself; the handler must not do it again
    try:
        _cpu_worker_body(ks, m, n, ready, go, out, released)
    except Exception as e:                                         # the parent must hear about it, not wait for ever
        if not released[0]:
            ready.release()
        out.put(RuntimeError(f"cpu worker on cpu {cpu}: {e!r}"))


def _cpu_worker_body(ks, m, n, ready, go, out, released):
    import numpy as np
    from oracle import pyoracle as O
    probs = [O.dq_generate(SEED0 + k, m, n, gamma=GAMMA, sigma=SIGMA, spread=SPREAD) for k in ks]
    O.dq_lm_solve(probs[0][0][:64, :8].copy(order="F"), probs[0][1][:64].copy(), GAMMA, probs[0][3][:8].copy(),
                  opts=O.default_options(max_evals=20))            # library loaded, code paged in
    buf = np.ones(4 << 20)                                         # 32 MiB: the bandwidth probe's source
    dst = np.zeros_like(buf)
    dst[::512] = 1.0                                               # pages touched before the clock
    ready.release()
    released[0] = True
    go.wait()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    njac = 0
    for A, b, xt, x0 in probs:
        rc, x, f, ib, nc, _ = O.dq_lm_solve(A, b, GAMMA, x0, opts=O.default_options(max_evals=500))
        njac += ib["jacobian_count"]
    t1 = time.perf_counter()
    cpu1 = time.process_time()                                     # CPU time the solves really got (quota, steal)
    # bandwidth probe (after the solves, all workers at once): numpy copies, read + write bytes
    go2 = time.perf_counter()
    reps = 0
    while time.perf_counter() - go2 < 0.5:
        np.copyto(dst, buf)
        reps += 1
    t2 = time.perf_counter()
    out.put((njac, t0, t1, 2.0 * buf.nbytes * reps / (t2 - go2), cpu1 - cpu0))


def cpu_all_cores(m, n, one_core_rate, per_worker=4):
    """The oracle farmed over the host: one pinned worker process per PHYSICAL core, `per_worker` problems each,
    problems generated before the clock starts; wall time = first start to last finish."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    cpus, model = _physical_cores()
    quota = _cpu_quota()
    if quota is not None and quota < len(cpus):                      # more workers than the quota only time-slice
        cpus = cpus[:max(1, int(quota))]
    W = len(cpus)
    ready, go, out = ctx.Semaphore(0), ctx.Event(), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(cpus[w], list(range(w * per_worker, (w + 1) * per_worker)), m, n,
                                                   ready, go, out), daemon=True) for w in range(W)]
    for p in procs:
        p.start()

    def dead():
        return [p.pid for p in procs if p.exitcode not in (None, 0)]
    # a worker that dies before it signals (import error, out of memory) must not hang the bench: bounded waits
    for _ in procs:
        if not ready.acquire(timeout=600):
            for p in procs:
                p.terminate()
            raise RuntimeError(f"cpu_all_cores: a worker never became ready (dead workers: {dead()})")
    go.set()
    res = []
    for _ in procs:
        try:
            r = out.get(timeout=1800)
        except Exception:
            for p in procs:
                p.terminate()
            raise RuntimeError(f"cpu_all_cores: a worker never reported (dead workers: {dead()})")
        if isinstance(r, Exception):
            for p in procs:
                p.terminate()
            raise r
        res.append(r)
    for p in procs:
        p.join(timeout=60)
    wall = max(r[2] for r in res) - min(r[1] for r in res)
    njac = sum(r[0] for r in res)
    rate = njac / wall
    eff = rate / (W * one_core_rate)
    cpu_s = sum(r[4] for r in res)
    d = {"value": rate, "unit": "LM iterations/s", "cores": W, "cpu_model": model,
         "logical_cpus_available": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": quota,
         "worker_cpu_seconds_over_wall_seconds": cpu_s / wall,
         "sample": f"{W * per_worker} problems: {per_worker} per worker, one pinned worker process per physical core, problems "
                   f"generated before the clock, {wall:.1f} s wall (first start to last finish)",
         "parallel_efficiency_vs_one_core": eff,
         "copy_bandwidth_GBs_all_workers": sum(r[3] for r in res) / 1e9}
    if eff < 0.5:
        # a 4096x256 solve streams an 8 MiB Jacobian (plus its 8 MiB model matrix) through every Householder step:
        # with all cores busy the working sets leave the caches and the host's memory bandwidth is shared
        d["note"] = ("below half of cores x one-core rate.  worker_cpu_seconds_over_wall_seconds says how many cores the "
                     "workers were actually given (a container CPU quota or a busy host caps it below `cores`); beyond that, "
                     "each solve streams its 8 MiB working matrix n times per factorisation and with every core active that "
                     "traffic goes to DRAM (copy_bandwidth_GBs_all_workers: read+write bytes/s of simultaneous 32 MiB numpy "
                     "copies, one per worker)")
    return d


def cpu_baseline(sample, m, n, all_cores=True):
    """Oracle (C restatement of the reference path) on the host: one core (the reference is single-threaded),
    then problems of the same family farmed over every physical host core (the CPU analogue of sharding).  Must run
    before the GPU is initialised: the all-cores leg forks workers.  Also returns the oracle's solutions (parity check)."""
    njac = 0
    t = 0.0
    sols = []
    for k in range(sample):
        nj, dt, x, ib, rc = _cpu_solve_one((k, m, n))
        njac += nj
        t += dt
        sols.append((x, ib, rc))
    out = {"value": njac / t, "unit": "LM iterations/s", "cores": 1, "kind": "port",
           "sample": f"{sample} problems {m}x{n} (seeds {SEED0}..{SEED0 + sample - 1}), single thread, "
                     f"oracle/nonlin_oracle.c -O2 -ffp-contract=off, {t:.1f} s"}
    if all_cores:
        try:
            out["all_cores"] = cpu_all_cores(m, n, njac / t)
        except Exception as e:                                   # the single-core figure is the contract
            out["all_cores"] = {"error": repr(e)}
    return out, sols


def lapack_priced(O, Ah, bh, xh, cpu_oracle_ms, n_factorisations, gpu_ms):
    """The CPU column the reference would really show for a Newton row.  The reference's lu_factor / solve_lu
    (src/nonlin_solve.f90:570, 577) are `linalg` -> LAPACK DGETRF / DGETRS (blocked, BLAS-3); the oracle restates the
    unblocked elimination (what the bit-for-bit GPU comparison needs), which is several times slower at n = 1024.  So:
    time ONE factorisation + solve of this problem's Jacobian both ways on one pinned thread -- the oracle's nlo_lu_factor /
    nlo_lu_solve and scipy's LAPACK (DGETRF / DGETRS, its bundled OpenBLAS limited to ONE thread) -- and replace the oracle's
    share in the oracle's measured solve time: cpu_lapack_ms = max(0, cpu_oracle_ms - n_fact * t_oracle_lu) + n_fact * t_lapack_lu.
    Everything else of the iteration (Jacobian callback, J^T F, line search) is the oracle's own time, unchanged."""
    import ctypes as C
    import numpy as np
    try:
        import scipy.linalg as sl
        from threadpoolctl import threadpool_limits, threadpool_info
    except ImportError as e:
        return {"cpu_lapack_ms": None, "cpu_lapack_note": f"scipy / threadpoolctl missing: {e!r}"}
    n = Ah.shape[0]
    J = np.asfortranarray(O.dq_jacobian(Ah, bh, 0.5, xh))
    L = O.lib()
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)

    def best(f, reps=3):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return min(ts)

    def oracle_lu():
        a = J.copy(order="F")
        piv = np.zeros(n, dtype=np.int32)
        rhs = bh.copy()
        L.nlo_lu_factor(n, a.ctypes.data_as(dp), n, piv.ctypes.data_as(ip))
        L.nlo_lu_solve(n, a.ctypes.data_as(dp), n, piv.ctypes.data_as(ip), rhs.ctypes.data_as(dp))

    def lapack_lu():
        lu, piv = sl.lu_factor(J, check_finite=False)
        sl.lu_solve((lu, piv), bh, check_finite=False)

    with threadpool_limits(limits=1):
        t_or = best(oracle_lu)
        t_la = best(lapack_lu)
        blas = sorted({f"{d.get('internal_api')} {d.get('version')}" for d in threadpool_info()})
    # (the oracle's LU share is timed outside the solve: when it comes out larger than the whole solve -- it is 95 % of it at
    # n = 1024 -- the rest of the iteration is taken as zero rather than negative: the LAPACK column then errs low, against the GPU)
    other_ms = max(0.0, cpu_oracle_ms - n_factorisations * 1e3 * t_or)
    ms = other_ms + n_factorisations * 1e3 * t_la
    return {"cpu_lapack_ms": ms, "gpu_over_cpu_lapack": ms / gpu_ms, "gpu_over_cpu_oracle": cpu_oracle_ms / gpu_ms,
            "cpu_lapack_parts_ms": {"lu_oracle": n_factorisations * 1e3 * t_or, "lu_lapack": n_factorisations * 1e3 * t_la,
                                    "rest_of_the_iteration": other_ms},
            "cpu_lapack_note": f"one factorisation + solve, n = {n}, one thread: oracle (unblocked, the bitwise twin) {1e3 * t_or:.1f} ms, "
                               f"LAPACK DGETRF/DGETRS via scipy ({', '.join(blas)}; 1 thread) {1e3 * t_la:.1f} ms; "
                               f"{n_factorisations} factorisations re-priced; the reference calls LAPACK, so gpu_over_cpu_lapack is the "
                               "GPU-vs-reference figure and gpu_over_cpu_oracle is not"}


def lapack_repriced(cpu_oracle_ms, gpu_ms, count, oracle_op, lapack_op, what, reps=3):
    """The same re-pricing for the rows whose `linalg` call is a QR (quasi-Newton: qr_factor + form Q,
    src/nonlin_solve.f90:286-326; bounded least squares: qr_factor / solve_qr, src/nonlin_least_squares.f90:1061,1344;
    polynomial fit: solve_least_squares, src/nonlin_polynomials.f90:198,252).  oracle_op / lapack_op: callables that do ONE
    such operation on this row's own matrix -- the oracle's unblocked restatement and scipy's LAPACK (one thread); `count`
    operations of the oracle's measured solve are replaced.  Everything else of the solve (function evaluations, rank-1
    updates -- linalg's own Givens code, not LAPACK --, dog-leg, line search) stays the oracle's time."""
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
    except ImportError as e:
        return {"cpu_lapack_ms": None, "cpu_lapack_note": f"threadpoolctl missing: {e!r}"}

    def best(f):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return min(ts)
    with threadpool_limits(limits=1):
        t_or, t_la = best(oracle_op), best(lapack_op)
        blas = sorted({f"{d.get('internal_api')} {d.get('version')}" for d in threadpool_info()})
    other_ms = max(0.0, cpu_oracle_ms - count * 1e3 * t_or)
    ms = other_ms + count * 1e3 * t_la
    return {"cpu_lapack_ms": ms, "gpu_over_cpu_lapack": ms / gpu_ms, "gpu_over_cpu_oracle": cpu_oracle_ms / gpu_ms,
            "cpu_lapack_parts_ms": {"op_oracle": count * 1e3 * t_or, "op_lapack": count * 1e3 * t_la, "rest_of_the_solve": other_ms},
            "cpu_lapack_note": f"{what}, one thread: oracle (unblocked, the bitwise twin) {1e3 * t_or:.2f} ms, LAPACK via scipy "
                               f"({', '.join(blas)}; 1 thread) {1e3 * t_la:.2f} ms; {count} operations re-priced; the reference calls "
                               "LAPACK, so gpu_over_cpu_lapack is the GPU-vs-reference figure and gpu_over_cpu_oracle is not"}


def _qr_ops(O, J, rhs=None, full_q=False):
    """(oracle_op, lapack_op) for one Householder QR of J: with Q formed (quasi-Newton) or with Q^T rhs + the triangular
    solve (bounded least squares, polynomial fit)."""
    import ctypes as C
    import numpy as np
    import scipy.linalg as sl
    from scipy.linalg import lapack as sla
    L = O.lib()
    dp = C.POINTER(C.c_double)
    m, n = J.shape
    if full_q:
        q, r = np.zeros((n, n), order="F"), np.zeros((n, n), order="F")

        def oracle_op():
            L.nlo_qr_factor_full(n, J.ctypes.data_as(dp), q.ctypes.data_as(dp), r.ctypes.data_as(dp))

        def lapack_op():
            sl.qr(J, mode="full", check_finite=False)                # DGEQRF + DORGQR
        return oracle_op, lapack_op

    def oracle_op():
        a, f = J.copy(order="F"), rhs.copy()
        L.nlo_qr_factor_rhs(m, n, a.ctypes.data_as(dp), f.ctypes.data_as(dp))

    def lapack_op():
        qr, tau, _, _ = sla.dgeqrf(J, overwrite_a=0)
        cq, _, _ = sla.dormqr("L", "T", qr, tau, rhs.reshape(-1, 1), max(1, 64 * n), overwrite_c=0)
        sla.dtrtrs(qr[:n, :n], cq[:n], lower=0)
    return oracle_op, lapack_op


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
        row = {"path": name, "gpu_ms": 1e3 * tg, "cpu_oracle_ms": 1e3 * tc, "iterations": ibs[0]["iter_count"],
               "bitwise_equal": bool(np.array_equal(ro[1], xg[0][0].cpu().numpy()))}
        if name.startswith("newton_solver"):
            row.update(lapack_priced(O, Ah, bh, xh, 1e3 * tc, ibs[0]["jacobian_count"], 1e3 * tg))
        else:
            Jh = np.asfortranarray(O.dq_jacobian(Ah, bh, 0.5, xh))
            row.update(lapack_repriced(1e3 * tc, 1e3 * tg, ibs[0]["jacobian_count"], *_qr_ops(O, Jh, full_q=True),
                                       what=f"qr_factor + form Q (DGEQRF + DORGQR), n = {n}, once per Jacobian evaluation; the "
                                            f"{ibs[0]['iter_count']} rank-1 updates are linalg's own Givens code and keep the oracle's time"))
        rows.append(row)
    # a BATCH of Newton problems: the lock-step device state machine (nlh_kernels_newton.h); CPU: the first 8 on one core
    nb, n = 256, 256
    A, b, xt, x0 = ds.generate(nb, n, n, seed0=12345, sigma=0.0, square_shift=True)

    def run_newton_batch():
        xg[0] = x0.clone()
        return ds.newton_solve_batch(A, b, 0.5, xg[0], analytic=True, opts=ds.options(max_evals=500))
    (_, ibs, st), tg = timed(run_newton_batch)
    nc = 8
    ro, tc = cpu(lambda: [O.dq_newton_solve(np.asfortranarray(A[q].cpu().numpy().T), b[q].cpu().numpy(), 0.5, x0[q].cpu().numpy(),
                                            opts=O.default_options(max_evals=500)) for q in range(nc)])
    row = {"path": f"newton_solver (LU), analytic Jacobian, batch of {nb} x n={n} (lock-step state machine)",
           "gpu_ms": 1e3 * tg, "cpu_oracle_ms": 1e3 * tc * nb / nc, "cpu_sample": f"{nc} problems, scaled to {nb}",
           "iterations": ibs[0]["iter_count"], "solves_per_s": nb / tg,
           "bitwise_equal": bool(all(np.array_equal(ro[q][1], xg[0][q].cpu().numpy()) for q in range(nc)))}
    row.update(lapack_priced(O, np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy(), x0[0].cpu().numpy(), 1e3 * tc * nb / nc,
                             sum(i["jacobian_count"] for i in ibs), 1e3 * tg))
    rows.append(row)
    m, n = 4096, 256
    A, b, xt, x0 = ds.generate(1, m, n, seed0=12345)
    Ah, bh, xh = np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy(), x0[0].cpu().numpy()
    lo, up = np.full(n, -2.0), np.full(n, 2.0)

    def run_cls():
        xg[0] = x0.clone()
        return ds.cls_solve_batch(A, b, 0.5, xg[0], opts=ds.options(max_evals=500), lower=lo, upper=up)
    (_, ibs, st), tg = timed(run_cls)
    ro, tc = cpu(lambda: O.dq_cls_solve(Ah, bh, 0.5, xh, opts=O.default_options(max_evals=500), lower=lo, upper=up))
    row = {"path": "constrained_least_squares_solver (bounded dog-leg), FD Jacobian, 4096x256", "gpu_ms": 1e3 * tg,
           "cpu_oracle_ms": 1e3 * tc, "iterations": ibs[0]["iter_count"],
           "bitwise_equal": bool(np.array_equal(ro[1], xg[0][0].cpu().numpy()))}
    Jh = np.asfortranarray(O.dq_jacobian(Ah, bh, 0.5, xh))
    row.update(lapack_repriced(1e3 * tc, 1e3 * tg, ibs[0]["jacobian_count"], *_qr_ops(O, Jh, rhs=bh),
                               what=f"qr_factor + solve_qr (DGEQRF + DORMQR + DTRTRS), {m}x{n}, once per Jacobian evaluation"))
    rows.append(row)
    # a BATCH of bounded problems: the lock-step device state machine (nlh_kernels_cls.h); CPU: the first 4 on one core
    nb, m, n = 256, 2048, 128
    A, b, xt, x0 = ds.generate(nb, m, n, seed0=12345, spread=0.2)
    lo, up = np.full(n, -2.0), np.full(n, 2.0)

    def run_cls_batch():
        xg[0] = x0.clone()
        return ds.cls_solve_batch(A, b, 0.5, xg[0], opts=ds.options(max_evals=500), lower=lo, upper=up)
    (_, ibs, st), tg = timed(run_cls_batch)
    nc = 4
    ro, tc = cpu(lambda: [O.dq_cls_solve(np.asfortranarray(A[q].cpu().numpy().T), b[q].cpu().numpy(), 0.5, x0[q].cpu().numpy(),
                                         opts=O.default_options(max_evals=500), lower=lo, upper=up) for q in range(nc)])
    row = {"path": f"constrained_least_squares_solver, FD Jacobian, batch of {nb} x {m}x{n} (lock-step state machine)",
           "gpu_ms": 1e3 * tg, "cpu_oracle_ms": 1e3 * tc * nb / nc, "cpu_sample": f"{nc} problems, scaled to {nb}",
           "iterations": ibs[0]["iter_count"], "solves_per_s": nb / tg,
           "bitwise_equal": bool(all(np.array_equal(ro[q][1], xg[0][q].cpu().numpy()) for q in range(nc)))}
    A0, b0 = np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy()
    Jh = np.asfortranarray(O.dq_jacobian(A0, b0, 0.5, x0[0].cpu().numpy()))
    row.update(lapack_repriced(1e3 * tc * nb / nc, 1e3 * tg, sum(i["jacobian_count"] for i in ibs), *_qr_ops(O, Jh, rhs=b0),
                               what=f"qr_factor + solve_qr (DGEQRF + DORMQR + DTRTRS), {m}x{n}, once per Jacobian evaluation of every problem"))
    rows.append(row)
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
    # a BATCH of BFGS problems: the lock-step device state machine (nlh_kernels_bfgs_batch.h); CPU: the first 4 on one core
    nb, m, n = 256, 1024, 64
    A, b, xt, x0 = ds.generate(nb, m, n, seed0=77, spread=0.1)

    def run_bfgs_batch():
        xg[0] = x0.clone()
        return ds.bfgs_solve_batch(A, b, 0.5, xg[0], opts=ds.options(**ob))
    (_, ibs, st), tg = timed(run_bfgs_batch)
    nc = 4
    ro, tc = cpu(lambda: [O.dq_bfgs_solve(np.asfortranarray(A[q].cpu().numpy().T), b[q].cpu().numpy(), 0.5, x0[q].cpu().numpy(),
                                          opts=O.default_options(**ob)) for q in range(nc)])
    rows.append({"path": f"bfgs (FD gradient, Cholesky rank-1 updates), batch of {nb} x {m}x{n} (lock-step state machine)",
                 "gpu_ms": 1e3 * tg, "cpu_oracle_ms": 1e3 * tc * nb / nc, "cpu_sample": f"{nc} problems, scaled to {nb}",
                 "iterations": ibs[0]["iter_count"], "solves_per_s": nb / tg,
                 "bitwise_equal": bool(all(np.array_equal(ro[q][1], xg[0][q].cpu().numpy()) for q in range(nc)))})
    nfit, npts, order = 4096, 4096, 7
    g = torch.Generator(device=ds.device).manual_seed(1)
    px = torch.rand((nfit, npts), dtype=torch.float64, device=ds.device, generator=g) * 2 - 1
    py = torch.cos(3 * px) + 0.01 * torch.randn((nfit, npts), dtype=torch.float64, device=ds.device, generator=g)
    c, tg = timed(lambda: ds.poly_fit_batch(px, py, order))
    xs, ys = px[:16].cpu().numpy(), py[:16].cpu().numpy()
    co, tc = cpu(lambda: [O.poly_fit(xs[q], ys[q], order)[1] for q in range(16)])
    row = {"path": "polynomial%fit, 4096 fits x 4096 points, order 7", "gpu_ms": 1e3 * tg,
           "cpu_oracle_ms": 1e3 * tc * nfit / 16, "cpu_sample": "16 fits, scaled to 4096",
           "bitwise_equal": bool(all(np.array_equal(c[q].cpu().numpy(), co[q]) for q in range(16)))}
    V = np.asfortranarray(np.vander(xs[0], order + 1, increasing=True))
    row.update(lapack_repriced(1e3 * tc * nfit / 16, 1e3 * tg, nfit, *_qr_ops(O, V, rhs=ys[0]),
                               what=f"solve_least_squares (DGELS class: DGEQRF + DORMQR + DTRTRS) of the {npts} x {order + 1} Vandermonde "
                                    "panel, once per fit; building the panel keeps the oracle's time"))
    rows.append(row)
    return rows


def mode_h_rows(ds):
    """The literal drop-in case (BASELINE config 2 as a user of the reference would run it): ONE problem solved
    through nlh_lm_solve with a COMPILED HOST CALLBACK (tests/host_callback/dq_callback.c, plain C) -- the solver's linear
    algebra on the GPU, the user's function on the host, called n + 1 times per Jacobian in the reference's order.  Against
    it: the CPU oracle driving the same callback on one host core.  x must be the same bits."""
    import ctypes as C
    import numpy as np
    from nonlin_amd import _lib
    from oracle import pyoracle as O
    so = os.path.join(ROOT, "tests", "host_callback", "libdq_callback.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.dirname(so), "-s"])
    cb = C.CDLL(so)

    class Ctx(C.Structure):
        _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("A", C.POINTER(C.c_double)), ("b", C.POINTER(C.c_double)),
                    ("gamma", C.c_double), ("ncalls", C.c_int64), ("u", C.POINTER(C.c_double))]
    dp = C.POINTER(C.c_double)
    rows = []
    for m, n in ((4096, 256), (512, 64)):
        A, b, xt, x0 = O.dq_generate(SEED0, m, n, gamma=GAMMA, sigma=SIGMA, spread=SPREAD)
        u = np.zeros(m)
        ctx = Ctx(m, n, A.ctypes.data_as(dp), b.ctypes.data_as(dp), GAMMA, 0, u.ctypes.data_as(dp))
        fcn_g = C.cast(cb.dq_user_fcn, _lib.VECFCN)
        nojac_g = C.cast(None, _lib.JACFCN)
        og = _lib.default_options()
        og.max_evals = 500

        def gpu():
            x = x0.copy()
            f = np.zeros(m)
            ib = _lib.IterationBehavior()
            ctx.ncalls = 0
            rc = ds.lib.nlh_lm_solve(ds.h.ptr, C.byref(og), m, n, fcn_g, nojac_g, C.byref(ctx), x.ctypes.data_as(dp),
                                     f.ctypes.data_as(dp), C.byref(ib))
            return rc, x, ib.as_dict(), int(ctx.ncalls)
        gpu()                                                     # warm: workspaces, pinned buffers
        t0 = time.perf_counter()
        rc_g, xg, ibg, calls_g = gpu()
        tg = time.perf_counter() - t0
        # the callbacks alone (what no solver can take off the host): the same number of calls, timed
        xs = x0.copy()
        f = np.zeros(m)
        t0 = time.perf_counter()
        for _ in range(calls_g):
            cb.dq_user_fcn(C.byref(ctx), n, xs.ctypes.data_as(dp), m, f.ctypes.data_as(dp))
        tcb = time.perf_counter() - t0
        # the CPU path: the oracle with the same compiled callback
        L = O.lib()
        oo = O.default_options(max_evals=500)
        xo = x0.copy()
        fo = np.zeros(m)
        ibo = O.IterationBehavior()
        ctx.ncalls = 0
        t0 = time.perf_counter()
        rc_o = L.nlo_lm_solve(C.byref(oo), C.cast(cb.dq_user_fcn, O.VECFCN), C.cast(None, O.JACFCN), C.byref(ctx), m, n,
                              xo.ctypes.data_as(dp), fo.ctypes.data_as(dp), C.byref(ibo))
        tc = time.perf_counter() - t0
        rows.append({"path": f"least_squares_solver through nlh_lm_solve, compiled host callback (mode H), one {m}x{n} problem, FD Jacobian",
                     "gpu_ms": 1e3 * tg, "cpu_oracle_ms": 1e3 * tc, "host_callback_ms_inside_gpu_ms": 1e3 * tcb,
                     "callbacks": calls_g, "lm_iterations": ibg["jacobian_count"], "status": [int(rc_g), int(rc_o)],
                     "bitwise_equal": bool(np.array_equal(xg, xo)),
                     "counts_equal": bool(all(ibg[k] == ibo.as_dict()[k] for k in ("iter_count", "fcn_count", "jacobian_count"))),
                     "note": "host-to-host, PCIe included (panel of n perturbed residuals up, nothing but x-sized vectors down); "
                             "the n + 1 host callbacks per Jacobian are serial by the reference's contract (args may be mutated) "
                             "and bound both columns: the drop-in wins what the factorisation costs on the host"})
    # BASELINE config 3 taken literally: newton_solver, n = 1024, compiled vecfcn AND compiled analytic jacobianfcn
    # (tests/host_callback/dq_callback.c): the 8 MB Jacobian crosses PCIe every iteration
    n = 1024
    A, b, xt, x0 = O.dq_generate(SEED0, n, n, gamma=GAMMA, sigma=0.0, spread=SPREAD, square_shift=True)
    u = np.zeros(n)
    ctx = Ctx(n, n, A.ctypes.data_as(dp), b.ctypes.data_as(dp), GAMMA, 0, u.ctypes.data_as(dp))
    og = _lib.default_options()
    og.max_evals = 500

    def gpu_nt():
        x = x0.copy()
        f = np.zeros(n)
        ib = _lib.IterationBehavior()
        rc = ds.lib.nlh_newton_solve(ds.h.ptr, C.byref(og), n, C.cast(cb.dq_user_fcn, _lib.VECFCN), C.cast(cb.dq_user_jac, _lib.JACFCN),
                                     C.byref(ctx), x.ctypes.data_as(dp), f.ctypes.data_as(dp), C.byref(ib))
        return rc, x, ib.as_dict()
    gpu_nt()
    t0 = time.perf_counter()
    rc_g, xg, ibg = gpu_nt()
    tg = time.perf_counter() - t0
    xs, jb = x0.copy(), np.zeros((n, n))
    t0 = time.perf_counter()
    for _ in range(ibg["jacobian_count"] + 1):
        cb.dq_user_jac(C.byref(ctx), n, xs.ctypes.data_as(dp), n, jb.ctypes.data_as(dp))
    for _ in range(ibg["fcn_count"]):
        cb.dq_user_fcn(C.byref(ctx), n, xs.ctypes.data_as(dp), n, u.ctypes.data_as(dp))
    tcb = time.perf_counter() - t0
    oo = O.default_options(max_evals=500)
    xo, fo, ibo = x0.copy(), np.zeros(n), O.IterationBehavior()
    t0 = time.perf_counter()
    rc_o = O.lib().nlo_newton_solve(C.byref(oo), C.cast(cb.dq_user_fcn, O.VECFCN), C.cast(cb.dq_user_jac, O.JACFCN), C.byref(ctx), n,
                                    xo.ctypes.data_as(dp), fo.ctypes.data_as(dp), C.byref(ibo))
    tc = time.perf_counter() - t0
    row = {"path": f"newton_solver through nlh_newton_solve, compiled vecfcn + compiled analytic jacobianfcn (mode H), n = {n}",
           "gpu_ms": 1e3 * tg, "cpu_oracle_ms": 1e3 * tc, "host_callback_ms_inside_gpu_ms": 1e3 * tcb, "iterations": ibg["iter_count"],
           "jacobian_calls": ibg["jacobian_count"], "status": [int(rc_g), int(rc_o)], "bitwise_equal": bool(np.array_equal(xg, xo)),
           "counts_equal": bool(all(ibg[k] == ibo.as_dict()[k] for k in ("iter_count", "fcn_count", "jacobian_count"))),
           "note": "host-to-host; the n x n Jacobian (8 MB) is uploaded once per iteration"}
    row.update(lapack_priced(O, A, b, x0, 1e3 * tc, ibg["jacobian_count"], 1e3 * tg))
    rows.append(row)
    return rows


def _convergence_summary(ibs, status, key):
    """How a lock-step batch ended: problems that did not converge (status != 0: the batch runs until the last of them gives
    up at max_evals), the number of lock-step rounds (a round serves every problem still active: the largest count), and the
    histogram of the per-problem counts in eight bins."""
    counts = [int(i[key]) for i in ibs]
    hi = max(counts)
    edges = sorted(set([1, 2, 4, 8, 16, 32, 64, 128, 256, hi + 1]))
    hist = {}
    for lo, up in zip(edges[:-1], edges[1:]):
        c = sum(1 for v in counts if lo <= v < up)
        if c:
            hist[f"{lo}-{up - 1}"] = c
    return {"non_converged": int(sum(1 for v in status if v != 0)), "lock_step_rounds": hi,
            f"{key}_histogram": hist, f"{key}_median": sorted(counts)[len(counts) // 2]}


def device_vecfcn_rows(ds):
    """The OPEN device-residual path (include/nonlin_hip.h: nlh_device_vecfcn; reference plugin layer
    src/nonlin_multi_eqn_mult_var.f90:14-25, 126-140, 198-277): least_squares_solver on residuals the LIBRARY DOES NOT KNOW,
    handed in as launchers.  Per row: LM iterations/s of the whole solve, and the forward-difference kernel
    (k_fd_jacobian_qrx: panel of perturbed residuals -> Jacobian columns in the factorisation's working layout, :274)
    against the HBM roofline, timed with HIP events INSIDE that solve, algorithmic bytes 8 (2 m n + m + 2 n) per Jacobian
    (SURVEY 8(d)).  Rows: the dense-quadratic family re-expressed through the launcher (the headline's 4096 x 256 shape,
    bitwise the built-in entry point), once as one lock-step batch (the kernel alone on the chip: THE FD-Jacobian
    fraction of this path) and once as three sub-batches on private streams (the kernel shares the chip with other
    sub-batches' kernels: its event time is labelled as such and is no roofline figure); and a family written outside the library (tests/device_model/user_models.hip,
    Lorentzian peak fits) checked bit for bit against the CPU oracle driving the same arithmetic as a host callback."""
    import ctypes as C
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import user_models as UM
    from oracle import pyoracle as O
    rows = []

    def timed_solve(f, m, n):
        f()
        torch.cuda.synchronize()
        ds.h.timing_enable(kernels=["fd_jacobian", "dq_panel", "dq_residual"])
        ds.h.timing_reset()
        t0 = time.perf_counter()
        x, fv, ibs, st = f()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        fd_ms, fd_cnt = ds.h.timing("fd_jacobian")
        u_ms = ds.h.timing("dq_panel")[0] + ds.h.timing("dq_residual")[0]
        ds.h.timing_enable(False)
        nj = sum(i["jacobian_count"] for i in ibs)
        gbs = fd_bytes(m, n) * nj / max(fd_ms * 1e-3, 1e-30) / 1e9
        return x, fv, ibs, st, {"solve_ms": 1e3 * t, "lm_iterations": nj, "lm_iterations_per_s": nj / t,
                                "user_function_ms": u_ms,
                                "fd_jacobian": {"kernel": "k_fd_jacobian_qrx (timed inside the solve)", "bound": "hbm", "achieved": gbs,
                                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "launches": int(fd_cnt),
                                                "kernel_ms": fd_ms, "bytes_per_jacobian": fd_bytes(m, n), "jacobians": nj}}

    nb, m, n = 512, 4096, 256
    A, b, xt, x0 = ds.generate(nb, m, n, seed0=SEED0, gamma=GAMMA, sigma=SIGMA, spread=SPREAD)
    fcn, jac, ctx = ds.dq_launchers(A, b, GAMMA)
    xb = x0.clone()
    fb, ibb, stb = ds.lm_solve_batch(A, b, GAMMA, xb, ds.options(max_evals=500))
    for label, sb in (("one lock-step batch", 1), ("library defaults (three sub-batches in flight)", 0)):
        def run():
            x = x0.clone()
            fv, ibs, st = ds.lm_solve_batch_device(fcn, ctx, m, x, opts=ds.options(max_evals=500, sub_batches=sb))
            return x, fv, ibs, st
        x, fv, ibs, st, r = timed_solve(run, m, n)
        if sb != 1:
            # several sub-batches in flight: the FD kernel's event time includes what the other sub-batches' kernels take from
            # it -- not a roofline figure (the one-batch row above is THE FD-Jacobian fraction of this path)
            fdr = r.pop("fd_jacobian")
            r["fd_jacobian_under_concurrent_sub_batches"] = {"kernel_ms": fdr["kernel_ms"], "launches": fdr["launches"],
                                                             "apparent_GBs": fdr["achieved"],
                                                             "note": "timed while other sub-batches' kernels share the chip: not a roofline figure"}
        r.update({"path": f"least_squares_solver on a user device vecfcn (dense-quadratic family through nlh_dq_device_fcn), "
                          f"{nb} x {m}x{n}, {label}",
                  "bitwise_equal_builtin_entry_point": bool(torch.equal(x, xb) and torch.equal(fv, fb) and ibs == ibb and st == stb)})
        rows.append(r)
    del A, b, xt, x0, xb, fb
    torch.cuda.empty_cache()

    nb, m, K = 4096, 2048, 8
    n = 3 * K
    t, y, xt, x0 = UM.lorentz_problems(nb, m, K, seed=2024)
    batch = UM.LorentzBatch(t, y)
    xd0 = torch.tensor(x0, device=ds.device)

    def run_l():
        x = xd0.clone()
        fv, ibs, st = ds.lm_solve_batch_device(batch.launch, batch.ctx, m, x, opts=ds.options(max_evals=500))
        return x, fv, ibs, st
    x, fv, ibs, st, r = timed_solve(run_l, m, n)
    r.update(_convergence_summary(ibs, st, "jacobian_count"))
    fdr = r.pop("fd_jacobian")                     # small n, sub-batches in flight: launch-latency-bound, not a roofline figure
    r["fd_jacobian_under_concurrent_sub_batches"] = {"kernel_ms": fdr["kernel_ms"], "launches": fdr["launches"], "apparent_GBs": fdr["achieved"],
                                                     "note": f"{m}x{n} panels (n = {n}): 16 us launches beside other sub-batches' kernels, not a roofline figure"}
    xg = x.cpu().numpy()
    dp = C.POINTER(C.c_double)
    ok, tc = True, 0.0
    sample = (0, 1, nb // 2, nb - 1)
    for p in sample:
        hc = batch.host_ctx(p)
        oo = O.default_options(max_evals=500)
        xo, fo, ibo = x0[p].copy(), np.zeros(m), O.IterationBehavior()
        t0 = time.perf_counter()
        rc = O.lib().nlo_lm_solve(C.byref(oo), C.cast(batch.host_fcn, O.VECFCN), C.cast(None, O.JACFCN), C.byref(hc), m, n,
                                  xo.ctypes.data_as(dp), fo.ctypes.data_as(dp), C.byref(ibo))
        tc += time.perf_counter() - t0
        ok = ok and rc == st[p] and np.array_equal(xo, xg[p]) and all(ibs[p][k] == ibo.as_dict()[k] for k in ("iter_count", "fcn_count", "jacobian_count"))
    r.update({"path": f"least_squares_solver on a family written outside the library (Lorentzian peaks, n = 3 x {K}), {nb} x {m}x{n}, "
                      "library defaults",
              "bitwise_equal_oracle_host_callback": bool(ok), "problems_compared": len(sample),
              "cpu_oracle_ms": 1e3 * tc * nb / len(sample), "cpu_sample": f"{len(sample)} problems on one core, scaled to {nb}"})
    rows.append(r)
    batch.close()

    # bfgs on a user's device fcnnvar (a launcher called with m = 1; src/nonlin_multi_var.f90:17-44, nonlin_optimize.f90:557-770):
    # a chained Rosenbrock objective written outside the library, forward-difference gradient built on the device
    nb, nv = 4096, 40
    cvals, xs = UM.crosen_problems(nb, nv, seed=2025)
    sb = UM.BtriBatch(cvals)
    so = UM.lib()
    xs0 = torch.tensor(xs, device=ds.device)
    fcnl = ds._devfcn(sb.crosen_launch)
    xq = xs0.clone()
    ds.bfgs_solve_batch_device(fcnl, sb.ctx, xq, opts=ds.options(max_evals=500))
    torch.cuda.synchronize()
    xq = xs0.clone()
    t0 = time.perf_counter()
    fo_g, ibs, st = ds.bfgs_solve_batch_device(fcnl, sb.ctx, xq, opts=ds.options(max_evals=500))
    torch.cuda.synchronize()
    tg = time.perf_counter() - t0
    xg = xq.cpu().numpy()
    ok, tc = True, 0.0
    sample = (0, 1, nb // 2, nb - 1)
    for p in sample:
        cp = float(cvals[p])
        t0 = time.perf_counter()
        rc, xo, fo, ibo = O.bfgs_solve(lambda xx: so.crosen_host_f(cp, nv, np.ascontiguousarray(xx).ctypes.data_as(dp)), nv, xs[p],
                                       opts=O.default_options(max_evals=500))
        tc += time.perf_counter() - t0
        ok = ok and rc == st[p] and np.array_equal(xo, xg[p]) and fo == fo_g[p] and all(ibs[p][k] == ibo[k] for k in ("iter_count", "fcn_count", "gradient_count"))
    its = sum(i["iter_count"] for i in ibs)
    rows.append({**_convergence_summary(ibs, st, "iter_count"),
                 "path": f"bfgs on a scalar function written outside the library (chained Rosenbrock, n = {nv}, forward-difference gradient on "
                         f"the device), {nb} problems, one lock-step batch", "solve_ms": 1e3 * tg, "bfgs_iterations": its,
                 "bfgs_iterations_per_s": its / tg, "function_evaluations": sum(i["fcn_count"] for i in ibs),
                 "bitwise_equal_oracle_host_callback": bool(ok), "problems_compared": len(sample),
                 "cpu_oracle_ms": 1e3 * tc * nb / len(sample),
                 "cpu_sample": f"{len(sample)} problems on one core through a Python callback (interpreter overhead included), scaled to {nb}"})
    sb.close()
    return rows


def predicted_scaling(ds, m=2048, n=128):
    """What ONE GPU can say about the 1/2/4/8-GPU curves before an 8-GPU node measures them: the batch sizes each rank of
    BASELINE config 4 (1024 problems of 2048x128) and of north_star's strong-scaling workload (8192 problems) gets at
    1 / 2 / 4 / 8 GPUs, each solved here as rank 0's own share (problems 0, N, 2N, ...: seeds 12345 + i*N), exact policy,
    library defaults.  Independent problems, no data-path collective: a rank does nothing else, so the N-GPU step time is
    the slowest rank's solve plus the RCCL gather of x and fvec (tens of MB)."""
    import torch
    out = {"m": m, "n": n, "policy": POLICY_NAMES[2], "options": "nlh_default_options (sub-batches automatic)",
           "note": "predicted speed-up at N GPUs = time(total) / time(total / N) on this GPU; measured here, not on N GPUs"}
    for total, key in ((1024, "config4_1024_problems"), (8192, "strong_8192_problems")):
        rows = []
        t1 = None
        for N in (1, 2, 4, 8):
            nb = total // N
            A, b, xt, x0 = ds.generate(nb, m, n, seed0=SEED0, gamma=GAMMA, sigma=SIGMA, spread=SPREAD, seed_stride=N)
            o = ds.options(max_evals=500)
            x = x0.clone()
            ds.lm_solve_batch(A, b, GAMMA, x, o)
            ts = []
            for _ in range(3 if nb <= 2048 else 2):
                x.copy_(x0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, ibs, st = ds.lm_solve_batch(A, b, GAMMA, x, o)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t = min(ts)
            nj = sum(i["jacobian_count"] for i in ibs)
            if N == 1:
                t1 = t
            rows.append({"gpus": N, "problems_per_rank": nb, "solve_ms": 1e3 * t, "lm_iterations_per_s_per_rank": nj / t,
                         "predicted_speedup": t1 / t, "predicted_efficiency": t1 / t / N})
            del A, b, xt, x0, x
            torch.cuda.empty_cache()
        out[key] = rows
    return out


def auto_policy_zero_residual(ds, m, n, nprob, max_evals):
    """N1: north_star's own formulation (J^T J on the fp64 MFMA + Cholesky step solve, NLH_FACTOR_AUTO) on the ZERO-RESIDUAL
    variant of the bench family (sigma = 0), where north_star's 1e-10 is well-posed: throughput of the same
    batch shape as the headline, the deviation of x and the count / flag mismatches against the CPU oracle's outputs for
    the first problems (tests/golden/zero_residual_oracle.npz, made by the oracle: data, not code), the Gram kernel against
    the fp64 MFMA peak and the stand-alone FD column kernel against HBM from live HIP events, and BASELINE config 5 (one
    65536 x 512) the same way.  tests/test_gpu_auto_policy.py holds the policy to this bar in the -m gpu suite."""
    import numpy as np
    import torch
    keys = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")
    gold = np.load(os.path.join(ROOT, "tests", "golden", "zero_residual_oracle.npz"))
    gp = [float(v) for v in gold["params"]]
    out = {"factor_policy": POLICY_NAMES[0], "family": f"gamma={GAMMA}, sigma=0 (zero residual), spread={SPREAD}, seeds {SEED0}+k",
           "tolerance": "north_star: 1e-10 relative on x, counts and flags exact",
           "checked_against": "tests/golden/zero_residual_oracle.npz (outputs of oracle/nonlin_oracle.c on the same seeds)"}

    def compare(tag, xg, ibs, status):
        if (gp[0], gp[1], gp[2], int(gp[3])) != (GAMMA, 0.0, SPREAD, SEED0):
            return {}
        xo, co, so = gold[f"{tag}_x"], gold[f"{tag}_counts"], gold[f"{tag}_status"]
        ns = min(xo.shape[0], xg.shape[0])
        dev = [float(np.abs(xg[k] - xo[k]).max() / np.abs(xo[k]).max()) for k in range(ns)]
        mism = sum(1 for k in range(ns) if [ibs[k][q] for q in keys] != [int(v) for v in co[k]] or status[k] != so[k])
        return {"problems_compared": ns, "max_rel_dev_x": max(dev), "count_or_flag_mismatches": mism,
                "within_1e-10_with_exact_counts": bool(max(dev) <= 1e-10 and mism == 0)}

    def run(mm, nn, nb, reps):
        A, b, xt, x0 = ds.generate(nb, mm, nn, seed0=SEED0, gamma=GAMMA, sigma=0.0, spread=SPREAD)
        o = ds.options(max_evals=max_evals, factor_policy=0)
        o.fuse_fd = 0                                              # the stand-alone FD column kernel (the HBM-bound one)
        x = x0.clone()
        ds.lm_solve_batch(A, b, GAMMA, x, o)                       # warm
        ts, nj, ibs, st = [], 0, None, None
        for _ in range(reps):
            x.copy_(x0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, ibs, st = ds.lm_solve_batch(A, b, GAMMA, x, o)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        nj = sum(i["jacobian_count"] for i in ibs)
        xg = x.cpu().numpy()
        # per-kernel pass (not part of the throughput figure): Gram and FD kernels bracketed by HIP events
        ds.h.timing_enable(kernels=["gram", "fd_jacobian"])
        ds.h.timing_reset()
        x.copy_(x0)
        ds.lm_solve_batch(A, b, GAMMA, x, o)
        torch.cuda.synchronize()
        gms, gcnt = ds.h.timing("gram")
        fms, fcnt = ds.h.timing("fd_jacobian")
        ds.h.timing_enable(False)
        gflops = (mm * nn * (nn + 1) + 2 * mm * nn) * nj / max(gms * 1e-3, 1e-30) / 1e12
        fgbs = fd_bytes(mm, nn) * nj / max(fms * 1e-3, 1e-30) / 1e9
        row = {"m": mm, "n": nn, "problems": nb, "solve_ms": 1e3 * min(ts), "lm_iterations": nj,
               "lm_iterations_per_s": nj / min(ts), "non_converged": int(sum(1 for v in st if v != 0)),
               "gram": {"kernel": "k_gram_tri / k_gram_512 (J^T J + J^T f, fp64 MFMA 16x16x4)", "bound": "mfma",
                        "achieved": gflops, "peak": 78.6, "unit": "TFLOP/s", "frac": gflops / 78.6, "launches": int(gcnt),
                        "ms": gms},
               "fd_jacobian": {"kernel": "k_fd_jacobian (stand-alone FD column kernel inside the solve)", "bound": "hbm",
                               "achieved": fgbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fgbs / HBM_PEAK_GBS,
                               "launches": int(fcnt), "ms": fms}}
        del A, b, xt, x0, x
        torch.cuda.empty_cache()
        return row, xg, ibs, st

    row, xg, ibs, st = run(m, n, nprob, 3)
    if (m, n) == (4096, 256):
        row.update(compare("c2", xg, ibs, st))
    # the WHOLE batch against the exact policy on the same problems (whose x is the oracle's bit for bit wherever the oracle
    # was run: the golden sample above, `parity`): deviation and count / flag mismatches over all problems
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=SEED0, gamma=GAMMA, sigma=0.0, spread=SPREAD)
    xe = x0.clone()
    _, ibe, ste = ds.lm_solve_batch(A, b, GAMMA, xe, ds.options(max_evals=max_evals))
    xen = xe.cpu().numpy()
    devs = np.abs(xg - xen).max(axis=1) / np.abs(xen).max(axis=1)
    row["whole_batch_vs_exact_policy"] = {
        "problems": nprob, "max_rel_dev_x": float(devs.max()),
        "count_or_flag_mismatches": int(sum(1 for k in range(nprob) if [ibs[k][q] for q in keys] != [ibe[k][q] for q in keys] or st[k] != ste[k])),
        "exact_policy_non_converged": int(sum(1 for v in ste if v != 0))}
    del A, b, xt, x0, xe
    torch.cuda.empty_cache()
    out["batch"] = row
    out["value"] = row["lm_iterations_per_s"]
    out["unit"] = "LM iterations/s"
    row5, xg5, ibs5, st5 = run(65536, 512, 1, 3)
    row5.update(compare("c5", xg5, ibs5, st5))
    out["config5_one_65536x512"] = row5
    return out


def fd_mode_h_roofline(ds, m=65536, n=512, reps=5):
    """The stand-alone FD column kernel (what the host-callback path runs: J(:,j) = (P(:,j) - f0)/h_j over a panel of
    perturbed residuals) at BASELINE config 5's size, HIP events on the launch stream."""
    import torch
    P = torch.rand((1, n, m), dtype=torch.float64, device=ds.device)
    f0 = torch.rand((1, m), dtype=torch.float64, device=ds.device)
    x = torch.rand((1, n), dtype=torch.float64, device=ds.device) + 0.5
    J = torch.empty_like(P)
    ds.fd_jacobian_panel(P, f0, x, out=J)
    ds.h.timing_enable(kernels=["fd_jacobian"])
    ds.h.timing_reset()
    for _ in range(reps):
        ds.fd_jacobian_panel(P, f0, x, out=J)
    ms, cnt = ds.h.timing("fd_jacobian")
    ds.h.timing_enable(False)
    gbs = fd_bytes(m, n) * cnt / max(ms * 1e-3, 1e-30) / 1e9
    return {"kernel": "k_fd_jacobian (host-callback path, nlh_fd_jacobian_panel)", "bound": "hbm", "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "m": m, "n": n, "launches": int(cnt),
            "avg_launch_ms": ms / max(cnt, 1), "bytes_per_launch": fd_bytes(m, n)}


def spawn_ranks(ngpus):
    """`bench.py --gpus N` without an external launcher: start the N ranks as children of a fresh
    torch.distributed.run process.  Nothing in this process has touched a GPU yet, and it only waits."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ren = {"--m": "--mrows", "--n": "--ncols"}
    argv = []
    for a in sys.argv[1:]:
        key, eq, val = a.partition("=")
        argv.append(ren.get(key, key) + eq + val)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("NLH_BENCH_BATCH", "2048")),
                    help="problems per GPU per step (weak scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--total-problems", type=int, default=8192, help="problems in all (strong scaling)")
    # --mrows / --ncols: the spellings that survive torch.distributed.run's own option parser (it rejects --m / --n as
    # ambiguous abbreviations of its options even behind the script name)
    ap.add_argument("--m", "--mrows", dest="m", type=int, default=M)
    ap.add_argument("--n", "--ncols", dest="n", type=int, default=N_VAR)
    ap.add_argument("--cpu-sample", type=int, default=32,
                    help="problems solved by the CPU oracle: the cpu_baseline timing and the parity sample (0 = skip)")
    ap.add_argument("--extras", type=int, default=1,
                    help="0 = only the timed steps (profiles/capture_r06.sh: every launch rocprofv3 sees then belongs to a "
                         "warm-up or timed step, so its per-kernel averages are the ones printed here)")
    ap.add_argument("--other-paths", type=int, default=1,
                    help="1 = also time the Newton / quasi-Newton / bounded LSQ / BFGS / polynomial rows (SURVEY 8 a16-a24, "
                         "f1-f4) on the GPU and on the CPU oracle and check the results bit for bit")
    ap.add_argument("--policy", type=int, default=2,
                    help="factor policy of the timed steps: 2 exact (reference operation order, the parity-carrying one), "
                         "0 auto (J^T J + Cholesky), 1 QR with tree reductions")
    ap.add_argument("--sub-batches", type=int, default=1,
                    help="sub-batches in flight during the timed steps (1: per-kernel durations are those of one lock-step "
                         "batch; the library default -- several in flight -- is reported as `default_options`)")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(world_env or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python bench.py --gpus N starts them itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    m, n = args.m, args.n

    cpu, sols = (None, [])
    if world == 1 and args.cpu_sample > 0:
        cpu, sols = cpu_baseline(args.cpu_sample, m, n, all_cores=bool(args.extras))

    import torch
    import torch.distributed as dist
    from nonlin_amd.device import DeviceSolver
    from nonlin_amd import sharding

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    ds = DeviceSolver(local_rank)
    # options / seed come from rank 0 (broadcast over RCCL when N > 1); problem k of the job goes to rank k mod N
    nprob_total = args.total_problems if args.scaling == "strong" else args.batch * world
    cfg = sharding.broadcast_config([500, SEED0, GAMMA, SIGMA, SPREAD, nprob_total], dev)
    max_evals, seed0 = int(cfg[0]), int(cfg[1])
    gamma, sigma, spread, nprob_total = cfg[2], cfg[3], cfg[4], int(cfg[5])
    B = sharding.shard_count(nprob_total, rank, world)
    # local problem i is global problem rank + i*world, seed = seed0 + global index
    A, b, xt, x0 = ds.generate(B, m, n, seed0=seed0 + rank, gamma=gamma, sigma=sigma, spread=spread, seed_stride=world)
    opts = ds.options(max_evals=max_evals, factor_policy=args.policy, sub_batches=args.sub_batches)
    x = x0.clone()
    fv = [None]

    def step():
        x.copy_(x0)
        fv[0] = None        # the previous step's fvec goes first: its block serves this step (a fresh 67 MB tensor allocated while the
        #                     old one is alive cost the first timed step a hipMalloc -- tens of ms, once)
        fvec, ibs, status = ds.lm_solve_batch(A, b, gamma, x, opts)
        fv[0] = fvec
        return ibs, status

    for _ in range(args.warmup):
        step()
    # Timed region: only the roofline kernel is bracketed by HIP events (on the launch stream); the per-kernel breakdown
    # comes from a second, untimed pass over the same steps below.
    roof_kernel = "qrx_pass" if args.policy == 2 else "fd_jacobian"
    if args.policy != 2:
        opts.fuse_fd = 0                       # the stand-alone FD column kernel is the HBM-bound one of these policies
    ds.h.timing_enable(kernels=[roof_kernel])
    ds.h.timing_samples(roof_kernel, select_only=True)       # keep this group's per-launch durations
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
    gathered = None
    for _ in range(args.steps):
        ibs, status = step()
        if world > 1:
            # the gather end of the sharded batch belongs to the step: every problem's x and fvec, in global
            # problem order, on every rank (RCCL all_gather of equal-sized padded shards)
            gathered = sharding.gather_results(torch.cat([x, fv[0]], dim=1), nprob_total, rank, world)
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

    # per-problem counts in global problem order (reporting only; x and fvec were gathered inside the timed steps)
    rows = torch.tensor([[ib["iter_count"], ib["fcn_count"], ib["jacobian_count"]] for ib in last_ibs],
                        dtype=torch.float64, device=dev).reshape(B, 3)
    allrows = sharding.gather_results(rows, nprob_total, rank, world)
    gather_ok = None
    if gathered is not None:                                     # this rank's own rows came back where they belong
        mine = torch.tensor(sharding.shard_indices(nprob_total, rank, world), device=dev)
        gather_ok = bool(torch.equal(gathered[mine, :n], x)) and bool(torch.equal(gathered[mine, n:], fv[0]))

    roof_ms, roof_launches = ds.h.timing(roof_kernel)
    roof_samples = ds.h.timing_samples(roof_kernel) if args.sub_batches == 1 else []
    kernel_ms = None
    if args.extras:
        ds.h.timing_enable(True)                   # breakdown pass: same steps, every kernel group timed, not part of `value`
        ds.h.timing_reset()
        for _ in range(args.steps):
            step()
        kernel_ms = {k: ds.h.timing(k)[0] for k in
                     ("dq_residual", "dq_panel", "fd_jacobian", "gram", "gram_reduce", "chol", "lmpar", "qr", "qrx_pass",
                      "qrx_pivot", "update")}
    ds.h.timing_enable(False)

    if rank == 0:
        if args.policy == 2:
            unit_bytes, units = qr_pass_bytes(m, n), "problem-factorisations"
            kname = "k_qrx_pass (trailing pass of a Householder step, exact lmfactor)"
            if B <= 8:      # a handful of problems: the workgroup-per-column forms, bound by the serial chain of the ordered sums, not by HBM
                kname = "k_qrx_pass_col (trailing pass, workgroup per column: " \
                        "latency-bound by its ordered sums -- the HBM fraction is quoted for completeness)"
        else:
            unit_bytes, units = fd_bytes(m, n), "problem-Jacobians"
            kname = "k_fd_jacobian"
        achieved = unit_bytes * (njac / max(roof_ms * 1e-3, 1e-30)) / 1e9      # this rank's launches
        # HBM traffic of the same kernel from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, gfx950 corrections
        # applied) committed under profiles/: bytes moved per algorithmic byte, scaled to this run's average launch.
        traffic, traffic_src = None, None
        try:
            pmc_file = next(f for f in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json")
                            if os.path.exists(os.path.join(ROOT, "profiles", f)))
            prof = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            if (prof["m"], prof["n"], prof.get("policy")) == (m, n, args.policy):
                ratio = prof["roofline_kernel"]["hbm_bytes_per_algorithmic_byte"]
                traffic = ratio * unit_bytes * njac / max(roof_launches, 1)
                same = prof.get("problems") in (None, B) and args.sub_batches == 1
                traffic_src = (f"profiles/{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                               f"command at commit {prof.get('commit', '?')}, {prof.get('problems', '?')} problems): measured "
                               "bytes per algorithmic byte x this run's algorithmic bytes per average launch"
                               + ("" if same else "; EXTRAPOLATED: this run's batch / sub-batch count differs from the profiled one"))
        except Exception:
            traffic = None
        out = {
            "metric": f"LM iterations/sec on m={m},n={n} fp64; FD-Jacobian GB/s vs HBM peak",
            "value": njac_all / elapsed,
            "unit": "LM iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"batched LM {m}x{n} fp64, FD Jacobian, dense-quadratic residual family "
                            f"(SURVEY 8(d): gamma={gamma}, sigma={sigma}, spread={spread}, seeds {seed0}+k), "
                            + (f"{args.batch} problems per GPU per step" if args.scaling == "weak"
                               else f"{nprob_total} problems per step in all"),
                "problems_total": nprob_total, "problems_rank0": B, "m": m, "n": n, "max_fcn_evals": max_evals,
                "factor_policy": POLICY_NAMES[args.policy],
                "sub_batches_in_flight": args.sub_batches,
                "parallelism": f"independent problems, block-cyclic over {world} rank(s)"
                               + ("; x and fvec all-gathered over RCCL inside every timed step" if world > 1 else ""),
                "gather_check": gather_ok,
                "timed_region": "host call to host return of nlh_dq_lm_solve_batch per step (SURVEY 8(d)'s solve wall time) with "
                                "A, b, x0 resident in HBM and x, fvec LEFT in HBM: the device-to-host copy of the results "
                                f"({8e-6 * B * (m + n):.0f} MB per step, about 0.1 % of a step over PCIe) is not in the timed region; "
                                "per-problem counts and status come back in every step",

                "accepted_steps_per_s": naccept_all / elapsed,
                "non_converged": int(bad_all),
                "iters_first_problem": [int(v) for v in allrows[0, :3].tolist()],
            },
            "roofline": {
                "kernel": kname, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": unit_bytes * njac / max(roof_launches, 1),
                "bytes_per_unit": unit_bytes, "units": units, "launches": int(roof_launches),
                "avg_launch_ms": roof_ms / max(roof_launches, 1),
            },
        }
        if args.policy == 2:
            # The builder's own ceiling beside the spec peak: what the memory system delivers for THIS traffic shape, from the
            # stream microbenchmarks (profiles/ubench/rw_stream.hip, numbers in profiles/r04_ubench.txt: a read-only stream in
            # the pass's access shape 6,068 GB/s; read + write in place with contiguous stores 5,018 GB/s).  Of QRX_C = 10
            # consecutive passes nine only read the trailing matrices and one reads and rewrites them, so 10 X algorithmic
            # bytes cost at best 9 X / 6068 + 2 X / 5018 seconds.
            rd, rw, per = 6068.0, 5018.0, 10
            ach_peak = per / ((per - 1) / rd + 2.0 / rw)
            out["roofline"]["achievable_peak"] = ach_peak
            out["roofline"]["frac_of_achievable"] = achieved / ach_peak
            out["roofline"]["achievable_note"] = ("measured stream rates on this part for the pass's traffic mix (nine read-only passes at "
                                                  "6,068 GB/s, one read + write pass at 5,018 GB/s per flush period of ten): the ceiling of "
                                                  "this DESIGN, not of the chip; `frac` against the 8 TB/s spec stays the headline fraction")
        if args.policy == 2 and roof_samples and len(roof_samples) % (n * args.steps) == 0:
            # A step is a sequence of lock-step rounds of n launches each; in the first round of a step every problem of
            # the batch is active (later rounds serve ever fewer problems and are bound by the serial row recurrences of
            # the few that remain, not by HBM): the same kernel on full launches only.
            per_step = len(roof_samples) // args.steps
            full_ms = sum(sum(roof_samples[s * per_step: s * per_step + n]) for s in range(args.steps))
            full = unit_bytes * B * args.steps / max(full_ms * 1e-3, 1e-30) / 1e9
            out["roofline"]["full_launches"] = {
                "achieved": full, "frac": full / HBM_PEAK_GBS, "launches": n * args.steps,
                "avg_launch_ms": full_ms / (n * args.steps), "rounds_per_step": per_step // n,
                "note": "first round of each step: all problems of the batch active in every launch"}
        if sols:
            # parity of the timed path with the CPU oracle on the first problems of the batch (same seeds)
            import numpy as np
            ns = min(len(sols), B)
            xg = x[:ns].cpu().numpy()
            eq = sum(1 for k in range(ns) if np.array_equal(xg[k], sols[k][0]))
            keys = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")
            ceq = sum(1 for k in range(ns) if all(last_ibs[k][q] == sols[k][1][q] for q in keys))
            rel = max(float(np.abs(xg[k] - sols[k][0]).max() / np.abs(sols[k][0]).max()) for k in range(ns))
            out["parity"] = {"checked_against": "oracle/nonlin_oracle.c (CPU restatement of the reference path)",
                             "problems": ns, "x_bitwise_equal": eq, "counts_and_flags_equal": ceq,
                             "max_rel_dev_x": rel}
        if kernel_ms is not None:
            out["kernel_ms_per_step"] = {k: v / args.steps for k, v in kernel_ms.items()}
            # the other kernels of an outer iteration against the bound that applies to each
            nfev = sum(ib["fcn_count"] for ib in last_ibs) * args.steps
            panel_adds = m * (n * (n + 1) // 2) * njac                    # dependent adds of n perturbed row sums
            resid_bytes = 8 * (m * n + 2 * m + n) * nfev
            pms, rms = kernel_ms["dq_panel"], kernel_ms["dq_residual"]
            kr = [
                {"kernel": "k_dq_panel (n perturbed evaluations + FD column in the epilogue)", "bound": "valu-f64-add",
                 "achieved": panel_adds / max(pms * 1e-3, 1e-30) / 1e12, "peak": 39.3, "unit": "Tadd/s",
                 "frac": panel_adds / max(pms * 1e-3, 1e-30) / 1e12 / 39.3,
                 "hbm_GBs": 8.0 * (2 * m * n + 2 * m + 2 * n) * njac / max(pms * 1e-3, 1e-30) / 1e9},
                {"kernel": "k_dq_residual", "bound": "hbm", "achieved": resid_bytes / max(rms * 1e-3, 1e-30) / 1e9,
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": resid_bytes / max(rms * 1e-3, 1e-30) / 1e9 / HBM_PEAK_GBS},
            ]
            if kernel_ms["gram"] > 0:
                gram_flops = (m * n * (n + 1) + 2 * m * n) * njac         # symmetric half + J^T f
                kr.insert(0, {"kernel": "k_gram_tri / k_gram_512 / k_gram_mfma (whichever serves this n)", "bound": "mfma",
                              "achieved": gram_flops / max(kernel_ms["gram"] * 1e-3, 1e-30) / 1e12, "peak": 78.6,
                              "unit": "TFLOP/s", "frac": gram_flops / max(kernel_ms["gram"] * 1e-3, 1e-30) / 1e12 / 78.6})
            out["kernel_rooflines"] = kr
        if world == 1 and args.extras:
            def run_policy(o, nrep=1, xs=None, sel=slice(None)):
                xq = x0[sel].clone() if xs is None else xs
                ds.lm_solve_batch(A[sel], b[sel], gamma, xq, o)         # warm
                tq, nj, ibq = 0.0, 0, None
                for _ in range(nrep):
                    xq.copy_(x0[sel])
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    _, ibq, _ = ds.lm_solve_batch(A[sel], b[sel], gamma, xq, o)
                    torch.cuda.synchronize()
                    tq += time.perf_counter() - t1
                    nj += sum(i["jacobian_count"] for i in ibq)
                return nj / tq, xq, ibq, tq / nrep

            # what the HIP-event brackets of the timed region cost: the same lock-step batch with no kernel group timed
            vq, xq_, _, _ = run_policy(ds.options(max_evals=max_evals, factor_policy=args.policy, sub_batches=args.sub_batches),
                                       nrep=min(args.steps, 3))
            out["roofline"]["event_bracketing"] = {
                "note": "the timed steps bracket every launch of the roofline kernel with a HIP event pair on the launch stream "
                        "(its per-launch durations are measured live, as `roofline.achieved` requires); the same steps without "
                        "any bracket:", "value_without_brackets": vq, "unit": "LM iterations/s",
                "overhead_frac": max(0.0, 1.0 - out["value"] / vq)}
            # the library's default options (three sub-batches in flight at this size): same bits, the product's own throughput.
            # Against `roofline.event_bracketing.value_without_brackets` (one batch, no brackets) it is box-dependent, -3 ... +2 %
            # on the bench family in round 6 (the pivot / lmpar time it hides is paid back by HBM-bound passes sharing the chip:
            # profiles/r06_overlap_defaults.json); on families with long straggler tails it is worth 25 - 35 % (Lorentzian peaks,
            # 416 lock-step rounds: 15.0 s with three sub-batches, 20.6 s as one batch), which is why it stays the default
            v, xd, _, _ = run_policy(ds.options(max_evals=max_evals), nrep=min(args.steps, 3))
            out["default_options"] = {"value": v, "unit": "LM iterations/s", "identical_x": bool(torch.equal(xd, x)),
                                      "note": "nlh_default_options (exact policy, sub_batches = auto: three in flight at this size -- "
                                              "latency-bound stages of one sub-batch run under the streaming kernels of another)"}
            # latency of BASELINE config 2 taken literally: ONE problem (seed 12345), warm handle
            v1, _, ib1, t1 = run_policy(ds.options(max_evals=max_evals), nrep=3, sel=slice(0, 1))
            out["single_problem"] = {"ms": 1e3 * t1, "lm_iterations": ib1[0]["jacobian_count"], "lm_iterations_per_s": v1,
                                     "factor_policy": POLICY_NAMES[2]}
            # the fast non-parity policy: J^T J on the fp64 MFMA + Cholesky; deviation from the CPU path measured here
            va, xa, iba, _ = run_policy(ds.options(max_evals=max_evals, factor_policy=0), nrep=min(args.steps, 5))
            auto = {"value": va, "unit": "LM iterations/s", "factor_policy": POLICY_NAMES[0],
                    "note": "opt-in policy: a different factorisation (normal equations), not bit-identical; "
                            "its deviation from the CPU oracle on the first problems of the batch:"}
            if sols:
                import numpy as np
                ns = min(len(sols), B)
                xg = xa[:ns].cpu().numpy()
                dev_ = [float(np.abs(xg[k] - sols[k][0]).max() / np.abs(sols[k][0]).max()) for k in range(ns)]
                mism = sum(1 for k in range(ns) if any(iba[k][q] != sols[k][1][q] for q in ("iter_count", "fcn_count", "jacobian_count")))
                auto.update({"problems_compared": ns, "max_rel_dev_x": max(dev_), "median_rel_dev_x": sorted(dev_)[ns // 2],
                             "count_mismatch_rate": mism / ns})
            out["auto_policy"] = auto
            v1a, _, ib1a, t1a = run_policy(ds.options(max_evals=max_evals, factor_policy=0), nrep=1, sel=slice(0, 1))
            out["auto_policy"]["single_problem_ms"] = 1e3 * t1a
            out["fd_jacobian_mode_h"] = fd_mode_h_roofline(ds)
            del A, b, xt, x0, x                                    # the batch's 70 GB go before the scaling sweep allocates its own
            torch.cuda.empty_cache()
            out["auto_policy_zero_residual"] = auto_policy_zero_residual(ds, m, n, B, max_evals)
            out["predicted_scaling"] = predicted_scaling(ds)
        if world == 1 and args.other_paths:
            out["other_paths"] = other_paths(ds) + mode_h_rows(ds)
            out["device_vecfcn"] = device_vecfcn_rows(ds)
        # the figures a reader of a truncated line needs, LAST (the stored tail of a long line keeps its end)
        summ = {"value": out["value"], "roofline_frac": out["roofline"]["frac"]}
        if "default_options" in out:
            summ["default_options_value"] = out["default_options"]["value"]
            summ["default_options_identical_x"] = out["default_options"]["identical_x"]
        if "predicted_scaling" in out:
            ps = out["predicted_scaling"]
            summ["predicted_speedup_8gpu_config4"] = ps["config4_1024_problems"][-1]["predicted_speedup"]
            summ["predicted_speedup_8gpu_8192_problems"] = ps["strong_8192_problems"][-1]["predicted_speedup"]
        if "auto_policy_zero_residual" in out:
            az = out["auto_policy_zero_residual"]
            summ["auto_policy_zero_residual_value"] = az["value"]
            summ["auto_policy_zero_residual_parity"] = {k: az["batch"].get(k) for k in ("max_rel_dev_x", "count_or_flag_mismatches", "problems_compared")}
            summ["auto_policy_zero_residual_whole_batch_vs_exact"] = az["batch"].get("whole_batch_vs_exact_policy")
            summ["config5_auto_zero_residual_ms"] = az["config5_one_65536x512"]["solve_ms"]
        if "parity" in out:
            summ["parity_x_bitwise_equal"] = f'{out["parity"]["x_bitwise_equal"]}/{out["parity"]["problems"]}'
        if cpu is not None:
            cpu["gpu_over_one_core"] = out["value"] / cpu["value"]
            if isinstance(cpu.get("all_cores"), dict) and "value" in cpu["all_cores"]:
                cpu["all_cores"]["gpu_over_all_cores"] = out["value"] / cpu["all_cores"]["value"]
            out["cpu_baseline"] = cpu
        out["summary"] = summ
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
