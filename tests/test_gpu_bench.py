"""GPU test of the bench.py contract: one JSON line with the driver's keys plus `roofline` and
`cpu_baseline`, directly and under the torchrun launch line the driver uses for N > 1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline")


def _check(line, n_gpus):
    d = json.loads(line)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["unit"] == "LM iterations/s" and d["dtype"] == "f64"
    assert d["value"] > 0 and d["vs_baseline"] is None and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["non_converged"] == 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    return d


def test_bench_single_process():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "16",
                          "--cpu-sample", "1", "--exact-sample", "4"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _check(out.stdout.strip().splitlines()[-1], 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert d["exact_policy"]["value"] > 0
    assert d["pipelined"]["identical_x"] and d["pipelined"]["value"] > 0 and d["pipelined_fused_fd"]["identical_x"]
    rows = d["other_paths"]                                   # Newton, quasi-Newton, bounded LSQ, BFGS, polynomial
    assert len(rows) == 5 and all(r["bitwise_equal"] and r["gpu_ms"] > 0 and r["cpu_oracle_ms"] > 0 for r in rows)


def test_bench_under_torchrun_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "1", "--warmup", "1", "--batch", "16", "--cpu-sample", "0",
                          "--exact-sample", "0", "--other-paths", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    _check(line, 1)
