"""GPU test of the bench.py contract: one JSON line with the driver's keys plus `roofline`, `cpu_baseline` and the
parity check against the oracle, directly (the timed path is the parity-carrying exact policy) and under the torchrun
launch line the driver uses for N > 1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline")


def _check(line, n_gpus, scaling="weak"):
    d = json.loads(line)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["unit"] == "LM iterations/s" and d["dtype"] == "f64"
    assert d["value"] > 0 and d["vs_baseline"] is None and d["scaling"] == scaling and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["non_converged"] == 0
    assert d["config"]["factor_policy"].startswith("exact")           # the headline is the parity-carrying policy
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and "k_qrx_pass" in r["kernel"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    return d


def test_bench_single_process():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "16",
                          "--m", "1024", "--n", "64", "--cpu-sample", "4"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _check(out.stdout.strip().splitlines()[-1], 1)
    assert "m=1024,n=64" in d["metric"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    p = d["parity"]                                            # timed path vs the CPU oracle on the same seeds
    assert p["problems"] == 4 and p["x_bitwise_equal"] == 4 and p["counts_and_flags_equal"] == 4 and p["max_rel_dev_x"] == 0.0
    assert d["default_options"]["identical_x"] and d["default_options"]["value"] > 0
    a = d["auto_policy"]                                       # the fast opt-in policy and its measured deviation
    assert a["value"] > 0 and a["problems_compared"] == 4 and 0.0 <= a["max_rel_dev_x"] < 1e-4
    assert d["fd_jacobian_mode_h"]["bound"] == "hbm" and d["fd_jacobian_mode_h"]["achieved"] > 0
    rows = d["other_paths"]        # Newton, quasi-Newton, bounded LSQ, BFGS, their three lock-step batches, polynomial,
    #                                and the two mode-H rows (nlh_lm_solve with a compiled host callback)
    #                                the two mode-H LM rows (nlh_lm_solve with a compiled host callback) and the mode-H Newton row
    #                                (compiled vecfcn + compiled analytic jacobianfcn, BASELINE config 3 taken literally)
    assert len(rows) == 11 and all(r["bitwise_equal"] and r["gpu_ms"] > 0 and r["cpu_oracle_ms"] > 0 for r in rows)
    assert sum("lock-step" in r["path"] for r in rows) == 3
    mh = [r for r in rows if "mode H" in r["path"]]
    assert len(mh) == 3 and all(r["counts_equal"] and r["status"] == [0, 0] for r in mh)
    assert sum(1 for r in mh if r.get("callbacks", 0) > 0) == 2
    nt = [r for r in rows if r["path"].startswith("newton_solver")]           # Newton rows are priced against LAPACK, not the oracle's LU
    assert len(nt) == 3 and all(r["cpu_lapack_ms"] > 0 and r["cpu_lapack_ms"] <= r["cpu_oracle_ms"] * 1.05 and r["gpu_over_cpu_lapack"] > 0
                                for r in nt)
    # ... and so is every other row whose reference code calls linalg's QR: quasi-Newton (qr_factor + form Q), bounded least
    # squares (qr_factor / solve_qr), the polynomial fit (solve_least_squares)
    qr_rows = [r for r in rows if r["path"].startswith(("quasi_newton", "constrained_least_squares", "polynomial%fit"))]
    assert len(qr_rows) == 4 and all(r["cpu_lapack_ms"] > 0 and r["cpu_lapack_ms"] <= r["cpu_oracle_ms"] * 1.05 and r["gpu_over_cpu_lapack"] > 0
                                     for r in qr_rows)
    dv = d["device_vecfcn"]        # the open device-residual path: a user launcher, k_fd_jacobian_qrx timed inside the solve
    assert len(dv) == 4 and all(r["lm_iterations_per_s"] > 0 for r in dv[:3])
    assert 0.0 < dv[0]["fd_jacobian"]["frac"] < 1.0            # ONE FD-Jacobian fraction: the one-batch solve; the rows with sub-batches in
    assert all("fd_jacobian" not in r and r["fd_jacobian_under_concurrent_sub_batches"]["kernel_ms"] > 0 for r in dv[1:3])   # flight say so
    assert all(r["non_converged"] >= 0 and r["lock_step_rounds"] >= 1 for r in dv[2:4])
    az = d["auto_policy_zero_residual"]                        # N1: the MFMA / Cholesky policy where 1e-10 is well-posed
    assert az["value"] > 0 and az["batch"]["non_converged"] == 0 and 0.0 < az["batch"]["gram"]["frac"] < 1.0
    wb = az["batch"]["whole_batch_vs_exact_policy"]             # every problem of the batch against the exact policy
    assert wb["problems"] == 16 and wb["max_rel_dev_x"] <= 1e-10 and wb["count_or_flag_mismatches"] == 0
    c5 = az["config5_one_65536x512"]
    assert c5["within_1e-10_with_exact_counts"] and c5["max_rel_dev_x"] <= 1e-10 and c5["count_or_flag_mismatches"] == 0
    assert 0.0 < c5["gram"]["frac"] < 1.0 and 0.0 < c5["fd_jacobian"]["frac"] < 1.0
    assert d["summary"]["value"] == d["value"] and d["summary"]["default_options_identical_x"]
    assert dv[0]["bitwise_equal_builtin_entry_point"] and dv[1]["bitwise_equal_builtin_entry_point"]
    assert dv[2]["bitwise_equal_oracle_host_callback"]
    assert dv[3]["path"].startswith("bfgs on a scalar function") and dv[3]["bfgs_iterations_per_s"] > 0 and dv[3]["bitwise_equal_oracle_host_callback"]
    assert 0.0 < d["roofline"]["frac_of_achievable"] < 1.2 and d["roofline"]["achievable_peak"] < d["roofline"]["peak"]
    ps = d["predicted_scaling"]    # per-rank batch sizes of config 4 / the 8192-problem run at 1, 2, 4, 8 GPUs, timed on this one
    for key, total in (("config4_1024_problems", 1024), ("strong_8192_problems", 8192)):
        assert [r["gpus"] for r in ps[key]] == [1, 2, 4, 8]
        assert all(r["problems_per_rank"] * r["gpus"] == total and r["solve_ms"] > 0 for r in ps[key])
    assert "timed_region" in d["config"]
    ac = c["all_cores"]                                        # pinned workers, problems generated before the clock
    assert ac["value"] > 0 and ac["cores"] >= 1 and "cpu_model" in ac and ac["gpu_over_all_cores"] > 0
    eb = d["roofline"]["event_bracketing"]                     # what the live HIP-event brackets of the timed region cost
    assert eb["value_without_brackets"] > 0 and 0.0 <= eb["overhead_frac"] < 0.5


def test_bench_under_torchrun_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "1", "--warmup", "1", "--batch", "16", "--mrows", "1024", "--ncols", "64",
                          "--cpu-sample", "0", "--extras", "0", "--other-paths", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    _check(line, 1)


def test_bench_strong_scaling_mode_one_rank():
    """north_star's strong-scaling workload shape (a fixed number of problems in all), here on one rank."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--scaling", "strong",
                          "--total-problems", "24", "--m", "512", "--n", "32", "--cpu-sample", "0", "--extras", "0",
                          "--other-paths", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _check(out.stdout.strip().splitlines()[-1], 1, scaling="strong")
    assert d["config"]["problems_total"] == 24 and d["config"]["problems_rank0"] == 24
